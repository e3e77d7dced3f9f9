#!/bin/bash
export EG_TUNING=1   # measurement aids (class `tuning` of csrc/switches.cpp) are honoured only with it
# Loop the GPU test suite (or its prefix up to the file that once failed) under the
# four execution toggles, one pytest process per run like the driver's, and log one line per run.
#   tools/stress_suite.sh <runs> [prefix|full] [logfile]
# Toggles cycle: none, EG_NO_GRAPH, EG_NO_OVERLAP, EG_NO_ROWFUSE, EG_NO_PREDICATE + EG_NO_ROW_PRODUCT, EG_POISON, and (round 4)
# the round-3 kernels: EG_GEMM_NO_SKEW + EG_GEMM_NO_BK32 + EG_GEMM_NO_PAIR + EG_CONV_NO_TINY + EG_CONV_NO_GRADF_HALO + EG_CONV_NO_WIDE_STORE + EG_NO_ROW_DIRECT + EG_NO_SMALL_PAIR (+ round 5: EG_CONV_NO_BAND + EG_GEMM_NO_T96); (round 5) EG_NO_ROW_TAIL + EG_NO_SAMPLE_FUSE; (round 6) EG_OVERLAP_SIDE_FIRST + EG_NO_DEFERRED_FOLD ride with EG_NO_GRAPH, EG_SAMPLE_KEEP_BARRIERS with EG_NO_OVERLAP.  Exit status 1 if any run failed.
runs=${1:-4}
what=${2:-prefix}
log=${3:-gpurun_out/stress_suite.log}
mkdir -p "$(dirname "$log")"
if [ "$what" = full ]; then
  files="tests"
else
  files="tests/test_adam.py tests/test_config_fixtures.py tests/test_dropout.py tests/test_gan.py tests/test_gpu_conv_grad.py tests/test_gpu_epilogue.py"
fi
toggles=("" "EG_NO_GRAPH=1 EG_OVERLAP_SIDE_FIRST=1 EG_NO_DEFERRED_FOLD=1" "EG_NO_OVERLAP=1 EG_SAMPLE_KEEP_BARRIERS=1" "EG_NO_ROWFUSE=1" "EG_NO_PREDICATE=1 EG_NO_ROW_PRODUCT=1" "EG_POISON=1" "EG_GEMM_NO_SKEW=1 EG_GEMM_NO_BK32=1 EG_GEMM_NO_PAIR=1 EG_CONV_NO_TINY=1 EG_CONV_NO_GRADF_HALO=1 EG_CONV_NO_WIDE_STORE=1 EG_NO_ROW_DIRECT=1 EG_NO_SMALL_PAIR=1 EG_CONV_NO_BAND=1 EG_GEMM_NO_T96=1 EG_NO_NARROW_K=1 EG_GEMM_NO_STREAMK=1 EG_SAMPLE_NO_MFMA=1 EG_SAMPLE_NO_STAGE=1" "EG_NO_ROW_TAIL=1 EG_NO_SAMPLE_FUSE=1 EG_NO_DPP_BUTTERFLY=1")
bad=0
for ((i = 0; i < runs; i++)); do
  t=${toggles[$((i % 8))]}
  start=$(date +%s)
  out=$(env $t python -m pytest $files -m gpu -q -x -p no:cacheprovider 2>&1 | tail -15)
  rc=${PIPESTATUS[0]}
  last=$(echo "$out" | grep -E "passed|failed|error" | tail -1)
  echo "run $i what=$what toggle='${t}' seconds=$(( $(date +%s) - start )) :: $last" >> "$log"
  if ! echo "$last" | grep -q "passed" || echo "$last" | grep -qE "failed|error"; then
    bad=1
    echo "$out" >> "$log"
  fi
done
echo "done runs=$runs bad=$bad" >> "$log"
exit $bad
