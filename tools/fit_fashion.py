#!/usr/bin/env python
"""One epoch of Model.fit on the reference's fashion_mnist network with synthetic data of the data set's
shape (60000 x 784, one-hot 10): tools/fit_fashion.py [batch_size] [samples]   (GPU box)"""
import os
import sys
import time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import exprgrad_amd as eg
from exprgrad_amd import examples
from exprgrad_amd import model as egm

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 32
samples = int(sys.argv[2]) if len(sys.argv) > 2 else 60000
ctx = eg.newGpuContext(0)
m = egm.compile(*examples.fashion_mnist_net(), gpu=ctx)
rng = np.random.default_rng(0)
x = rng.random((samples, 784), dtype=np.float32)
y = np.eye(10, dtype=np.float32)[rng.integers(0, 10, samples)]
loss0 = float(m.call("loss", {"x": x[:2048], "y": y[:2048]})[0])
m.fit("fit", {"x": x[:batch * 4], "y": y[:batch * 4]}, batch_size=batch)   # builds, captures
ctx.sync() if hasattr(ctx, "sync") else None
t = time.perf_counter()
m.fit("fit", {"x": x, "y": y}, batch_size=batch)
_ = m.call("loss", {"x": x[:2048], "y": y[:2048]})      # blocks until the epoch is done
dt = time.perf_counter() - t
print(f"batch {batch}: one epoch of {samples} samples in {dt*1e3:.1f} ms ({samples/dt/1e3:.1f} K samples/s, "
      f"{dt/(samples//batch)*1e6:.1f} us per batch); loss {loss0:.4f} -> {float(_[0]):.4f}")
print(m.launch_plan("fit"))
