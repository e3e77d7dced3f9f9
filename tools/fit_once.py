"""Three epochs of Model.fit on the reference's fashion_mnist network (synthetic data) — a workload for
rocprofv3 --kernel-trace --stats (EG_NO_GRAPH=1: rocprofv3 does not survive thousands of graph launches).  FIT_BATCH=4096."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np

import exprgrad_amd as eg
from exprgrad_amd import examples, model as egm

ctx = eg.newGpuContext(0)
m = egm.compile(*examples.fashion_mnist_net(), gpu=ctx)
rng = np.random.default_rng(0)
f = np.float32
n = int(os.environ.get("FIT_SAMPLES", "60000"))
x = rng.random((n, 784), dtype=f)
y = np.eye(10, dtype=f)[rng.integers(0, 10, n)]
batch = int(os.environ.get("FIT_BATCH", "4096"))
for _ in range(3):
    m.fit("fit", {"x": x, "y": y}, batch_size=batch)
ctx.sync()
