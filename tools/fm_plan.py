import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, exprgrad_amd as eg
from exprgrad_amd import examples, model as egm
ctx=eg.newGpuContext(0)
m=egm.compile(*examples.fashion_mnist_net(), gpu=ctx)
rng=np.random.default_rng(0); f=np.float32
ins={"x": rng.random((32,784),dtype=f), "y": np.eye(10,dtype=f)[rng.integers(0,10,32)]}
for t in ("fit","train"):
    try:
        m.apply(t, ins); ctx.sync(); print(t); print(m.launch_plan(t)); break
    except Exception as e: print(t, "ERR", e)
