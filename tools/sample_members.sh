#!/bin/bash
# Where a sample kernel's time goes: the batch-32 fit step of the fashion_mnist network with the kernel cut behind member k
# (EG_SAMPLE_STOP=k, a tuning aid), k = 0 .. last; the difference of two lines is what a member costs.  tools/sample_members.sh [last]
export EG_TUNING=1
last=${1:-25}
for k in $(seq 0 $last) -1; do
  out=gpurun_out/sample_members; rm -rf $out; mkdir -p $out
  (cd /tmp && export TMPDIR=/tmp && EG_SAMPLE_STOP=$k EG_NO_GRAPH=1 EG_NO_KERNEL_CACHE=1 FIT_BATCH=32 FIT_SAMPLES=1600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out -o t -- python $GRAFT_REPO_ROOT/tools/fit_once.py > /dev/null 2>&1)
  f=$(find $out -name "*kernel_stats.csv" | head -1)
  python - "$f" "$k" <<'PY'
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if r["Name"].startswith("eg_samples")]
for r in rows: print(f"stop {sys.argv[2]:>3}: {r['Name']} avg {float(r['AverageNs'])/1e3:6.2f} us min {float(r['MinNs'])/1e3:6.2f} calls {r['Calls']}")
PY
done
