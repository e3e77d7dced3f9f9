#!/bin/bash
# Everything a round's evidence needs, in ONE gpurun call on the final sources:  tools/final_round.sh r06
#   1. tools/profile_all.sh <tag>: rocprofv3 kernel stats + PMC passes of the five workloads, traffic.json, the default bench line
#   2. the parity survey: the GPU suite with EG_PARITY_RECORD, condensed by tools/parity_survey.py
#   3. the predicted 1 / 2 / 4 / 8 GPU table (tools/predict_scaling.py), labelled as a model
#   4. timelines: the dense step, the XOR step, the batch-32 fit step
TAG=${1:-r06}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
mkdir -p gpurun_out/summ
bash tools/profile_all.sh $TAG
rm -f gpurun_out/parity_record.jsonl
EG_PARITY_RECORD=$REPO/gpurun_out/parity_record.jsonl python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/summ/parity_run.log 2>&1
python tools/parity_survey.py gpurun_out/parity_record.jsonl gpurun_out/summ/parity_survey_$TAG.json > gpurun_out/summ/parity_survey.log 2>&1
python tools/predict_scaling.py gpurun_out/summ/scaling_prediction_$TAG.json > gpurun_out/summ/scaling.log 2>&1
bash tools/trace_step.sh > gpurun_out/summ/${TAG}_step_timeline.txt 2>&1
FIT_BATCH=32 bash tools/fit_timeline.sh > gpurun_out/summ/${TAG}_fit_timeline.txt 2>&1
bash tools/fit_sample_ab.sh 8 32 256 1024 2>&1 | grep -v amdgpu > gpurun_out/summ/${TAG}_fit_sample_ab.txt
python tools/streamk_ab.py nn 1024x1024x1024 1152x1152x1152 1280x1280x1280 1536x1536x1536 1792x1792x1792 2048x2048x2048 2304x2304x2304 1280x1280x4096 2>&1 | grep -v amdgpu > gpurun_out/summ/${TAG}_streamk_ab.txt
# conv2 forward: the round's kernel against the previous build when one is at hand (.ab/libold.so), cycle stamps of both kernels
{ for i in 1 2; do
    [ -f .ab/libold.so ] && EG_LIB_PATH=$REPO/.ab/libold.so python bench.py --workload conv2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('previous build:', d['ms_per_step'], d['roofline']['frac'], d['backward'])"
    python bench.py --workload conv2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('this build    :', d['ms_per_step'], d['roofline']['frac'], d['backward'])"
  done
  CASES=forward EG_HALO_TRACE=1 python tools/conv_shape.py 1 256 256 64 64 3 3 3 2>&1 | grep "halo trace" | tail -1
  CASES=grad_filter EG_GRADF_TRACE=1 python tools/conv_shape.py 1 256 256 64 64 3 3 3 2>&1 | grep "gradf trace" | tail -1
} > gpurun_out/summ/${TAG}_conv_ab.txt 2>&1
EG_SAMPLE_TRACE=1 EG_NO_GRAPH=1 EG_NO_KERNEL_CACHE=1 python tools/dump_fit.py gpurun_out/fit_code_final 2>&1 | grep "eg\]" | tail -2 > gpurun_out/summ/${TAG}_sample_trace.txt
ls gpurun_out/summ
tail -c 1500 gpurun_out/summ/${TAG}_bench_n1.json
python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/summ/gputests_final.log 2>&1; tail -3 gpurun_out/summ/gputests_final.log
