#!/bin/bash
# One steady dense train step as a timeline (start offset, duration, gap to the end of the previous kernel) plus the
# plan's launch list.  tools/step_timeline.sh [ENV=VALUE ...]
out=gpurun_out/timeline
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
env "$@" rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out -o t -- python $GRAFT_REPO_ROOT/bench.py --workload train --no-cpu-baseline --steps 30 --warmup 5 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $out -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
names=[r["Kernel_Name"] for r in rows]
# a step = from one forward kernel (first big fused contraction) to the next
key=next(n for n in names if n.startswith("eg_gemm_epi"))
idx=[i for i,n in enumerate(names) if n==key]
a,b=idx[-3],idx[-2]
t0=int(rows[a]["Start_Timestamp"]); prev=t0
for r in rows[a:b]:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    print(f"{(s-t0)/1e3:9.1f} us  +{(e-s)/1e3:7.1f}  gap {(s-prev)/1e3:6.1f}  q{r.get('Queue_Id','?')}  {r['Kernel_Name'][:90]}")
    prev=max(prev,e)
print(f"step {(int(rows[b]['Start_Timestamp'])-t0)/1e3:.1f} us")
PY
env "$@" python - <<'PY'
import numpy as np, exprgrad_amd as eg
from exprgrad_amd import examples, model as egm
ctx=eg.newGpuContext(0)
m=egm.compile(*examples.dense_softmax_net(), gpu=ctx)
rng=np.random.default_rng(0); f=np.float32
ins={"x": rng.random((65536,784),dtype=f), "y": np.eye(10,dtype=f)[rng.integers(0,10,65536)]}
m.apply("train", ins); ctx.sync()
print(m.launch_plan("train"))
PY
