#!/bin/bash
# Round-2 experiment B: LDS-staged wide stores, the ones-row fold (virtual row fetched by the loaders), staged host copies.
out=gpurun_out/perf_r02b.log
: > $out
run() { echo "## $*" >> $out; env "$@" 2>&1 | grep -v amdgpu.ids | tail -2 >> $out; }
step() { echo "## train step: $*" >> $out; env "$@" python bench.py --workload train --no-cpu-baseline --steps 30 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'])" >> $out; }
step A=1
step EG_GEMM_NO_WIDE_STORE=1
step EG_NO_ONES_ROW=1
step EG_NO_ONES_ROW=1 EG_GEMM_NO_WIDE_STORE=1
step EG_NO_OVERLAP=1
for ws in 0 1; do
  if [ $ws = 0 ]; then v="EG_GEMM_NO_WIDE_STORE=1"; else v="A=1"; fi
  run $v python tools/gemm_shape.py 65536 512 784 nn 30
  run $v python tools/gemm_shape.py 784 512 65536 tn 30
  run $v python tools/gemm_shape.py 4096 4096 4096 nn 30
  run $v python tools/gemm_shape.py 65536 512 10 nt 30
  run $v python tools/gemm_shape.py 4100 4100 4100 nn 20
  run $v python tools/gemm_shape.py 1000 1000 1000 nn 30
done
echo "## bench default line (end_to_end)" >> $out
python bench.py --no-extra --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['frac'], json.dumps(d['end_to_end']))" >> $out
echo "## same, EG_NO_STAGED_COPY=1" >> $out
EG_NO_STAGED_COPY=1 python bench.py --no-extra --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps(d['end_to_end']))" >> $out
echo "## kernel trace of the train step" >> $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r02b -o trace -- python $GRAFT_REPO_ROOT/bench.py --workload train --no-cpu-baseline --steps 30 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/prof_r02b -name "*kernel_stats.csv" | head -1)
python - "$f" >> $out <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]: print(r["Name"][:110], r["Calls"], round(float(r["AverageNs"])/1e3,1), r["Percentage"])
PY
cat $out
