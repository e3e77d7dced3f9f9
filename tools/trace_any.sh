#!/bin/bash
# Mean duration and mean gap-to-predecessor per kernel over the steady half of a bench workload under rocprofv3 --kernel-trace.
# tools/trace_any.sh <workload> [ENV=VALUE ...]
wl=$1; shift
out=gpurun_out/trace_any
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
env "$@" rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out -o t -- python $GRAFT_REPO_ROOT/bench.py --workload $wl --no-cpu-baseline --no-extra --steps 30 --warmup 5 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $out -name "*kernel_trace.csv" | head -1)
python - "$f" "$wl $*" <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
rows=rows[len(rows)//2:]
d=collections.defaultdict(list); g=collections.defaultdict(list)
prev=None
for r in rows:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    k=r["Kernel_Name"][:60]
    d[k].append((e-s)/1e3)
    if prev is not None: g[k].append((s-prev)/1e3)
    prev=e
print("==", sys.argv[2])
for k,v in d.items():
    gg=sorted(g[k]) or [0]
    print(f"  {sum(v)/len(v):8.2f} us (min {min(v):6.2f}) x{len(v):4d}  gap median {gg[len(gg)//2]:6.2f}  {k}")
PY
