#!/usr/bin/env python
"""N training steps of the dense net (BASELINE configs[4], one GPU's shard) from fixed parameters and data;
prints a checksum of the parameters.  Run twice (e.g. with and without EG_NO_OVERLAP / EG_NO_GRAPH):
the checksums must be identical.  tools/soak_train.py [steps] [batch]"""
import hashlib
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import exprgrad_amd as eg
from exprgrad_amd import examples
from exprgrad_amd import model as egm

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
ctx = eg.newGpuContext(0)
m = egm.compile(*examples.dense_softmax_net(), gpu=ctx)
rng = np.random.default_rng(5)
for tid in m.params.ids():
    m.params[tid] = (rng.random(m.params[tid].shape, dtype=np.float32) * 0.2 - 0.1).astype(np.float32)
x = rng.random((batch, 784), dtype=np.float32)
y = np.eye(10, dtype=np.float32)[rng.integers(0, 10, size=batch)]
m.apply("train", {"x": x, "y": y})
import torch  # device-resident inputs for the loop: no host copies between the steps
xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
torch.cuda.synchronize()
for _ in range(steps):
    m.apply("train", [("x", xd), ("y", yd)])
h = hashlib.sha256()
for tid in m.params.ids():
    h.update(m.params[tid].tobytes())
print(f"{steps} steps, batch {batch}: params sha256 {h.hexdigest()[:16]}")
