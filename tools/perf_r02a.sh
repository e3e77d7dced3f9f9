#!/bin/bash
# Round-2 experiment A: 256x128 tiles (two blocks per CU, staggered) and the ones-row fold on the shapes of the
# dense train step (configs[4], 65536 samples per GPU).  Output: gpurun_out/perf_r02a.log
out=gpurun_out/perf_r02a.log
: > $out
run() { echo "## $*" >> $out; env "$@" 2>&1 | tail -3 >> $out; }
for tile in 256,256 256,128 128,128; do
  for st in 0 3; do
    [ "$tile" != "256,128" ] && [ "$st" != 0 ] && continue
    run EG_GEMM_FORCE_TILE=$tile EG_GEMM_STAGGER=$st python tools/gemm_shape.py 65536 512 784 nn 30
  done
done
run EG_GEMM_FORCE_TILE=256,128 EG_GEMM_STAGGER=6 python tools/gemm_shape.py 65536 512 784 nn 30
run EG_GEMM_FORCE_TILE=256,128 EG_GEMM_STAGGER=12 python tools/gemm_shape.py 65536 512 784 nn 30
for tile in 256,256 256,128 128,128; do
  run EG_GEMM_FORCE_TILE=$tile EG_GEMM_STAGGER=3 python tools/gemm_shape.py 784 512 65536 tn 30
done
run EG_GEMM_FORCE_TILE=256,128 EG_GEMM_STAGGER=0 python tools/gemm_shape.py 784 512 65536 tn 30
for tile in 256,256 256,128; do
  run EG_GEMM_FORCE_TILE=$tile EG_GEMM_STAGGER=3 python tools/gemm_shape.py 4096 4096 4096 nn 30
done
echo "## train step, default" >> $out
python bench.py --workload train --no-cpu-baseline --steps 30 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'])" >> $out
echo "## train step, EG_NO_ONES_ROW=1" >> $out
EG_NO_ONES_ROW=1 python bench.py --workload train --no-cpu-baseline --steps 30 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'])" >> $out
echo "## train step plan" >> $out
python tools/show_plan.py >> $out 2>&1
cat $out
