#!/bin/bash
# rocprofv3 evidence of the float64 kernels (eg_dgemm 4096^3; the reference's conv2 benchmark in float64): kernel-trace stats and
# PMC passes (one counter group per run, --kernel-trace only next to --pmc), condensed by tools/summarize_profile.py.
#   gpurun -- tools/profile_f64.sh r05   ->  gpurun_out/summ/r05_float64_{dgemm,conv2}
TAG=${1:-r05}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
mkdir -p gpurun_out/summ
[ -f gpurun_out/summ/traffic.json ] || cp profiles/traffic.json gpurun_out/summ/traffic.json   # (entries of the other workloads stay)
export PYTHONPATH=$REPO
run() {  # <name> <kernel substring> <command...>
  name=$1; kern=$2; shift 2
  OUT=$REPO/gpurun_out/prof_${TAG}_float64_$name
  mkdir -p $OUT
  (cd $REPO && python -c "import bench; print(bench.source_fingerprint())") > $OUT/source_fingerprint.txt
  (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- "$@" > $OUT/bench_trace.log 2>&1)
  for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64 GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS"; do
    g=$(echo $grp | tr ' ' '+' | cut -c1-40)
    (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc_$g -o pmc -- "$@" > $OUT/bench_pmc_$g.log 2>&1) || echo "pmc group failed: $grp" >> $OUT/errors.log
  done
  python tools/summarize_profile.py $OUT gpurun_out/summ/${TAG}_float64_$name --workload float64_$name --kernel "$kern"
  rm -rf $OUT
}
run dgemm dgemm_kernel python $REPO/tools/dgemm_bench.py 4096
run conv2 eg_conv_mfma64 python $REPO/tools/conv64_bench.py 8
ls gpurun_out/summ
