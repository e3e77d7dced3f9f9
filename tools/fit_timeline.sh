#!/bin/bash
# Kernel timeline of ONE Model.fit batch of the fashion_mnist network (batch FIT_BATCH, default 32): rocprofv3 --kernel-trace
# with graphs off, then the dispatches of the last batch (start, duration, gap to the previous kernel).  tools/fit_timeline.sh
out=gpurun_out/fit_timeline
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
EG_NO_GRAPH=1 FIT_BATCH=${FIT_BATCH:-32} FIT_SAMPLES=${FIT_SAMPLES:-3200} rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out -o t -- python $GRAFT_REPO_ROOT/tools/fit_once.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $out -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
names=[r["Kernel_Name"] for r in rows]
# a batch starts with the segment copy of its rows into the staging buffers
starts=[i for i,n in enumerate(names) if "segment" in n or "gather_rows" in n or "copy_rows" in n]
if len(starts)<3:
    from collections import Counter
    print("no batch boundaries found; kernels:", Counter(n[:50] for n in names).most_common(30)); sys.exit()
a,b=starts[-2],starts[-1]
t0=int(rows[a]["Start_Timestamp"]); prev=t0
for r in rows[a:b]:
    s=int(r["Start_Timestamp"]); e=int(r["End_Timestamp"])
    print(f'{(s-t0)/1e3:8.1f} us  +{(e-s)/1e3:6.1f}  gap {(s-prev)/1e3:5.1f}  grid {r.get("Grid_Size_X","?"):>7} wg {r.get("Workgroup_Size_X","?"):>4}  {r["Kernel_Name"][:100]}')
    prev=e
print(f'batch {(int(rows[b]["Start_Timestamp"])-t0)/1e3:.1f} us, {b-a} launches')
PY
