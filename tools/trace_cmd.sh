#!/bin/bash
# Mean duration / gap per kernel (steady half) of an arbitrary command under rocprofv3 --kernel-trace.
# tools/trace_cmd.sh [ENV=VALUE ...] -- <command...>
envs=()
while [ "$1" != "--" ] && [ $# -gt 0 ]; do envs+=("$1"); shift; done
shift
out=$GRAFT_REPO_ROOT/gpurun_out/trace_cmd
rm -rf $out; mkdir -p $out
here=$(pwd)
cd /tmp && export TMPDIR=/tmp
(cd $here && env "${envs[@]}" rocprofv3 --kernel-trace --output-format csv -d $out -o t -- "$@" > $out/cmd.log 2>&1)
cd $here
f=$(find $out -name "*kernel_trace.csv" | head -1)
python - "$f" "${envs[*]} $*" <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
rows=rows[len(rows)//2:]
d=collections.defaultdict(list); g=collections.defaultdict(list)
prev=None
for r in rows:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    k=r["Kernel_Name"][:70]
    d[k].append((e-s)/1e3)
    if prev is not None: g[k].append((s-prev)/1e3)
    prev=e
print("==", sys.argv[2])
for k,v in d.items():
    gg=sorted(g[k]) or [0]
    print(f"  {sum(v)/len(v):8.2f} us (min {min(v):6.2f}) x{len(v):5d}  gap median {gg[len(gg)//2]:6.2f}  {k}")
PY
