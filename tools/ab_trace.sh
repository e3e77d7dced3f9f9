mkdir -p gpurun_out/r06c
run() { env "$@" python bench.py --workload train --no-cpu-baseline --steps 40 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['ms_per_step'], d['roofline'].get('kernel_ms_avg'), d['roofline'].get('kernel_ms_min'))"; }
run X=1
out=gpurun_out/gap_trace; rm -rf $out; mkdir -p $out
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out -o t -- python $GRAFT_REPO_ROOT/bench.py --workload train --no-cpu-baseline --steps 40 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench under rocprofv3', d['ms_per_step'], d['roofline'].get('kernel_ms_avg'), d['roofline'].get('kernel_ms_min'))")
f=$(find $out -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv,sys,statistics
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
names=[r["Kernel_Name"] for r in rows]
key=next(n for n in names if n.startswith("eg_gemm_epi"))
idx=[i for i,n in enumerate(names) if n==key]
st=[(int(rows[b]["Start_Timestamp"])-int(rows[a]["Start_Timestamp"]))/1e3 for a,b in zip(idx[:-1],idx[1:])]
print("all step lengths by trace:", [round(x) for x in st])
PY
run X=1
