#!/bin/bash
# like tools/fit_timeline.sh, with the captured graphs left on (few batches: rocprofv3 does not survive thousands of graph launches)
out=gpurun_out/fit_timeline_g
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
env "$@" FIT_BATCH=${FIT_BATCH:-32} FIT_SAMPLES=${FIT_SAMPLES:-1280} rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out -o t -- python $GRAFT_REPO_ROOT/tools/fit_once.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $out -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
names=[r["Kernel_Name"] for r in rows]
starts=[i for i,n in enumerate(names) if "copy_segments" in n]
# the shortest batch of the trace: under rocprofv3 batches are 94 - 160 us with gaps of 4 - 70 us at random places (the
# profiler's own work per graph launch); the shortest one has none
spans=[(int(rows[b]["Start_Timestamp"])-int(rows[a]["Start_Timestamp"]),a,b) for a,b in zip(starts[:-1],starts[1:])]
_,a,b=min(spans)
print(f"{len(spans)} batches traced: shortest {min(spans)[0]/1e3:.1f} us, median {sorted(spans)[len(spans)//2][0]/1e3:.1f} us")
t0=int(rows[a]["Start_Timestamp"]); prev=t0
for r in rows[a:b]:
    s=int(r["Start_Timestamp"]); e=int(r["End_Timestamp"])
    print(f'{(s-t0)/1e3:8.1f} us  +{(e-s)/1e3:6.1f}  gap {(s-prev)/1e3:5.1f}  {r["Kernel_Name"][:90]}')
    prev=e
print(f'batch {(int(rows[b]["Start_Timestamp"])-t0)/1e3:.1f} us, {b-a} launches')
PY
