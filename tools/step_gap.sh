#!/bin/bash
# Where the time between two dense train steps goes: kernel timeline of two consecutive steps (rocprofv3 --kernel-trace)
# with the captured graphs, launched one by one (EG_NO_GRAPH=1), and without the side lane.  tools/step_gap.sh
mkdir -p gpurun_out/r06c
for e in "X=1" "EG_NO_GRAPH=1" "EG_NO_OVERLAP=1" "EG_NO_GRAPH=1 EG_NO_OVERLAP=1"; do
  out=gpurun_out/gap_trace; rm -rf $out; mkdir -p $out
  (cd /tmp && export TMPDIR=/tmp && env $e EG_BENCH_NO_DEVICE_CLOCK=1 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out -o t -- python $GRAFT_REPO_ROOT/bench.py --workload train --no-cpu-baseline --steps 30 --warmup 5 > /dev/null 2>&1)
  f=$(find $out -name "*kernel_trace.csv" | head -1)
  echo "=== $e"
  python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
names=[r["Kernel_Name"] for r in rows]
key=next(n for n in names if n.startswith("eg_gemm_epi"))
idx=[i for i,n in enumerate(names) if n==key]
# steady steps: the last 20 step lengths, and the gap in front of the first kernel of each step
steps=[]; gaps=[]
for a,b in zip(idx[-22:-2], idx[-21:-1]):
    steps.append((int(rows[b]["Start_Timestamp"])-int(rows[a]["Start_Timestamp"]))/1e3)
import statistics
print("step us: median %.1f min %.1f max %.1f"%(statistics.median(steps),min(steps),max(steps)))
a,b=idx[-4],idx[-3]
t0=int(rows[a]["Start_Timestamp"]); prev=t0
for r in rows[a:b+1]:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    print(f"{(s-t0)/1e3:9.1f} us  +{(e-s)/1e3:7.1f}  gap {(s-prev)/1e3:6.1f}  q{r.get('Queue_Id','?')}  {r['Kernel_Name'][:70]}")
    prev=max(prev,e)
PY
done 2>&1 | tee gpurun_out/r06c/step_gap.txt
