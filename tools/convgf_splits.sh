export EG_TUNING=1   # measurement aids (class `tuning` of csrc/switches.cpp) are honoured only with it
for sp in 0 57 76 85 100 113 128 142 170 226; do
  if [ $sp = 0 ]; then e=""; else e="EG_GEMM_FORCE_SPLITS=$sp"; fi
  env $e python bench.py --workload conv2 --no-cpu-baseline --steps 60 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('splits $sp', d['backward']['grad_filter_ms'])"
done
