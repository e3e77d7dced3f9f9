for rep in 1 2; do
for e in "X=1" "EG_NO_GRAPH=1" "EG_NO_OVERLAP=1" "EG_NO_GRAPH=1 EG_NO_OVERLAP=1"; do
  echo "== $e"
  env $e python bench.py --workload train --no-cpu-baseline --steps 40 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['frac'] if d.get('roofline') else None, d.get('telemetry',{}))"
done; done
