mkdir -p gpurun_out/r06c
for rep in 1 2 3; do
for e in "X=1" "EG_BENCH_NO_DEVICE_CLOCK=1"; do
  echo "== $e"
  env $e python bench.py --workload train --no-cpu-baseline --steps 40 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('kernel_ms_avg'), d['roofline'].get('kernel_ms_min'), d['roofline'].get('effective_clock_mhz'))"
done; done 2>&1 | tee gpurun_out/r06c/ab_probe.txt
