#!/bin/bash
# Kernel timeline of one training step (configs[4], one GPU): rocprofv3 --kernel-trace, then the dispatches of the
# last step with start / end relative to the step's first kernel.  tools/trace_step.sh [ENV=VALUE ...]
out=gpurun_out/trace_step
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
env "$@" rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out -o t -- python $GRAFT_REPO_ROOT/bench.py --workload train --no-cpu-baseline --steps 10 --warmup 2 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $out -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# last step: find the last occurrence of the map-fused update kernel, go back to the previous one
names=[r["Kernel_Name"] for r in rows]
upd=[i for i,n in enumerate(names) if n.startswith("eg_maps")]
if len(upd)<2: print("no steps found"); sys.exit()
a,b=upd[-2]+1,upd[-1]+1
t0=int(rows[a]["Start_Timestamp"])
for r in rows[a:b]:
    s=(int(r["Start_Timestamp"])-t0)/1e3; e=(int(r["End_Timestamp"])-t0)/1e3
    print(f'{s:9.1f} {e:9.1f} {e-s:8.1f}  q{r.get("Queue_Id","?")} grid {r.get("Grid_Size_X","?")} wg {r.get("Workgroup_Size_X","?")}  {r["Kernel_Name"][:90]}')
PY
