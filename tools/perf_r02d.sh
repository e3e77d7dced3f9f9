#!/bin/bash
out=gpurun_out/perf_r02d.log
: > $out
run() { echo "## $*" >> $out; env "$@" 2>&1 | grep -v amdgpu.ids | tail -1 >> $out; }
run A=1 python tools/gemm_shape.py 4100 4100 4100 nn 20
run EG_GEMM_NO_TAIL=1 python tools/gemm_shape.py 4100 4100 4100 nn 20
run A=1 python tools/gemm_shape.py 4608 4608 4096 nn 20
run EG_GEMM_NO_TAIL=1 python tools/gemm_shape.py 4608 4608 4096 nn 20
run A=1 python tools/gemm_shape.py 8192 4400 2048 nt 20
run EG_GEMM_NO_TAIL=1 python tools/gemm_shape.py 8192 4400 2048 nt 20
run A=1 python tools/gemm_shape.py 4096 4096 4096 nn 30
run EG_GEMM_BK32=1 python tools/gemm_shape.py 4096 4096 4096 nn 30
run A=1 python tools/gemm_shape.py 65536 512 784 nn 30
run EG_GEMM_BK32=1 python tools/gemm_shape.py 65536 512 784 nn 30
run A=1 python tools/gemm_shape.py 784 512 65536 tn 30
run EG_GEMM_BK32=1 python tools/gemm_shape.py 784 512 65536 tn 30
run A=1 python tools/gemm_shape.py 8192 8192 8192 nn 10
run EG_GEMM_BK32=1 python tools/gemm_shape.py 8192 8192 8192 nn 10
cat $out
