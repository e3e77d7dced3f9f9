#!/bin/bash
# Per-kernel means of one PMC counter group for an arbitrary command: tools/pmc_cmd.sh "<counters>" <kernel substring> [ENV=VAL ...] -- <command...>
grp=$1; kern=$2; shift 2
envs=()
while [ "$1" != "--" ] && [ $# -gt 0 ]; do envs+=("$1"); shift; done
shift
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_cmd
rm -rf $out; mkdir -p $out
here=$(pwd)
cd /tmp && export TMPDIR=/tmp
(cd $here && env "${envs[@]}" rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out -o p -- "$@" > $out/cmd.log 2>&1)
cd $here
f=$(find $out -name "*counter_collection.csv" | head -1)
python - "$f" "$kern" <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
d=collections.defaultdict(list)
for r in rows:
    if sys.argv[2] in r["Kernel_Name"]: d[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in sorted(d.items()): print(f"  {k:32s} n={len(v):4d} mean={sum(v)/len(v):.4g}")
PY
