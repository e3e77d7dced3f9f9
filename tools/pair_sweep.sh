#!/bin/bash
# wave-pair kernel against the four-wave kernels, same box: tools/pair_sweep.sh
export SPIN_MS=100
for l in 0 1 2; do
tools/bin/gemm_pipe 1024 1024 1024 $l 50 2 | grep -v "256x256\|8w\|pipelined\|loads\|mfma\|5 st\|128x128x16"
tools/bin/gemm_pipe 2048 2048 2048 $l 50 2 | grep -v "256x256\|8w\|pipelined\|loads\|mfma\|5 st\|128x128x16"
done
tools/bin/gemm_pipe 512 512 512 0 50 2 | grep -v "256x256\|8w\|pipelined\|loads\|mfma\|5 st\|128x128x16"
tools/bin/gemm_pipe 1536 1536 1536 0 50 2 | grep -v "256x256\|8w\|pipelined\|loads\|mfma\|5 st\|128x128x16"
tools/bin/gemm_pipe 3072 3072 3072 0 50 2 | grep -v "256x256\|8w\|pipelined\|loads\|mfma\|5 st\|128x128x16"
