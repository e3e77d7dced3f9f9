#!/bin/bash
# Usage (on the GPU box, via gpurun): tools/profile.sh <tag> <bench args...>
# Writes rocprofv3 kernel-trace stats and PMC passes (one counter group per run, never mixed with
# tracing domains other than kernel-trace) under gpurun_out/prof_<tag>/.
set -u
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# (no device-clock probe under the profiler: a --pmc pass may serialise dispatches across queues, and the probe wave waits for
#  markers on the main stream; it would only give up after its 8 s guard)
export EG_BENCH_NO_DEVICE_CLOCK=1
CMD="python $REPO/bench.py --no-cpu-baseline --no-end-to-end $*"
# what the byte counts describe: bench.py flags roofline.traffic as stale when the sources differ from these
(cd $REPO && python -c "import bench; print(bench.source_fingerprint())") > $OUT/source_fingerprint.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $CMD > $OUT/bench_trace.log 2>&1
# counter passes: every dispatch is serialised, so a short spin-up and few timed steps (the byte counts are per launch)
export EG_BENCH_SPINUP_S=0.005
CMD="$CMD --steps 5 --warmup 1"
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $grp | tr ' ' '+' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc_$name -o pmc -- $CMD > $OUT/bench_pmc_$name.log 2>&1 || echo "pmc group failed: $grp" >> $OUT/errors.log
done
find $OUT -name "*.csv" | head -40
