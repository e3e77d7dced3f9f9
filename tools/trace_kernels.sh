#!/bin/bash
# Mean duration per kernel over the steady steps of the dense train step under rocprofv3 --kernel-trace.
# tools/trace_kernels.sh [ENV=VALUE ...]
out=gpurun_out/trace_k
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
env "$@" rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out -o t -- python $GRAFT_REPO_ROOT/bench.py --workload train --no-cpu-baseline --steps 30 --warmup 5 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $out -name "*kernel_trace.csv" | head -1)
python - "$f" "$*" <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
d=collections.defaultdict(list)
for r in rows: d[r["Kernel_Name"][:70]].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
print("==", sys.argv[2])
for k,v in d.items():
    v=v[len(v)//2:]
    if sum(v)/len(v) > 20: print(f"  {sum(v)/len(v):8.1f} us x{len(v):3d}  {k}")
PY
