#!/bin/bash
# Round-2 experiment C: tail tiles (4100^3-type shapes), mid-size contractions (1000^3) under different tiles.
out=gpurun_out/perf_r02c.log
: > $out
run() { echo "## $*" >> $out; env "$@" 2>&1 | grep -v amdgpu.ids | tail -1 >> $out; }
run A=1 python tools/gemm_shape.py 4100 4100 4100 nn 20
run EG_GEMM_NO_TAIL=1 python tools/gemm_shape.py 4100 4100 4100 nn 20
run A=1 python tools/gemm_shape.py 4608 4608 4096 nn 20
run EG_GEMM_NO_TAIL=1 python tools/gemm_shape.py 4608 4608 4096 nn 20
run A=1 python tools/gemm_shape.py 8192 4400 2048 nt 20
run EG_GEMM_NO_TAIL=1 python tools/gemm_shape.py 8192 4400 2048 nt 20
for t in 64,64 128,64 128,128 256,64 256,256; do
  run EG_GEMM_FORCE_TILE=$t python tools/gemm_shape.py 1000 1000 1000 nn 50
done
run A=1 python tools/gemm_shape.py 1000 1000 1000 nn 50
for t in 64,64 128,64 128,128; do
  run EG_GEMM_FORCE_TILE=$t python tools/gemm_shape.py 2000 2000 2000 nn 30
done
run A=1 python tools/gemm_shape.py 2000 2000 2000 nn 30
run A=1 python tools/gemm_shape.py 1536 1536 1536 nn 30
cat $out
