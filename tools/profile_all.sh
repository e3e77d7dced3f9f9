#!/bin/bash
# The four profiled workloads of a round, condensed on the GPU box (the raw rocprofv3 output stays there):
#   gpurun -- tools/profile_all.sh r06     ->  gpurun_out/summ/{r06_matmul,r06_train,r06_xor,r06_conv2,r06_hbm}, traffic.json, bench line
TAG=${1:-r02}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
mkdir -p gpurun_out/summ
cp profiles/traffic.json gpurun_out/summ/traffic.json
for w in matmul train xor conv2 hbm; do
  tools/profile.sh ${TAG}_$w --workload $w > gpurun_out/summ/profile_$w.log 2>&1
done
python tools/summarize_profile.py gpurun_out/prof_${TAG}_matmul gpurun_out/summ/${TAG}_matmul --workload matmul4096 --kernel "gemm_f32_mfma_kernel<256, 256"
python tools/summarize_profile.py gpurun_out/prof_${TAG}_train gpurun_out/summ/${TAG}_train --workload train
python tools/summarize_profile.py gpurun_out/prof_${TAG}_xor gpurun_out/summ/${TAG}_xor --workload xor
python tools/summarize_profile.py gpurun_out/prof_${TAG}_conv2 gpurun_out/summ/${TAG}_conv2 --workload conv2 --kernel conv2_halo_kernel
python tools/summarize_profile.py gpurun_out/prof_${TAG}_hbm gpurun_out/summ/${TAG}_hbm --note "bandwidth-bound library kernels (bench.py --workload hbm): 65536 x 512 float32 operands, four operand sets in rotation"
rm -rf gpurun_out/prof_${TAG}_*
cp gpurun_out/summ/traffic.json profiles/traffic.json      # (the box's copy: the bench line below reads it)
python bench.py > gpurun_out/summ/${TAG}_bench_n1.json 2> gpurun_out/summ/bench_stderr.log
tail -c 600 gpurun_out/summ/${TAG}_bench_n1.json
