#!/bin/bash
# usage: bench_train.sh [ENV=VAL ...] ; prints ms_per_step of two runs
for i in 1 2; do
env "$@" python bench.py --workload train --no-cpu-baseline --steps 60 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', d['ms_per_step'], d['roofline']['kernel_ms_min'])"
done
