#!/bin/bash
export EG_TUNING=1   # measurement aids (class `tuning` of csrc/switches.cpp) are honoured only with it
# Model.fit on the fashion_mnist network: us per batch at several batch sizes, tiny convolution kernels on / off (one box)
python - <<'PY'
import os, time, subprocess, sys, json
code = r'''
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import exprgrad_amd as eg
from exprgrad_amd import examples, model as egm
ctx = eg.newGpuContext(0)
m = egm.compile(*examples.fashion_mnist_net(), gpu=ctx)
rng = np.random.default_rng(0); f = np.float32
batch = int(os.environ["FIT_BATCH"]); n = max(60000 // batch * batch, batch * 64)
n = min(n, 60000 // batch * batch) if 60000 // batch >= 64 else batch * 64
x = rng.random((n, 784), dtype=f); y = np.eye(10, dtype=f)[rng.integers(0, 10, n)]
for _ in range(2): m.fit("fit", {"x": x, "y": y}, batch_size=batch)
ctx.sync(); t0 = time.perf_counter()
for _ in range(3): m.fit("fit", {"x": x, "y": y}, batch_size=batch)
ctx.sync(); dt = (time.perf_counter() - t0) / 3
print(f"batch {batch}: {dt / (n // batch) * 1e6:.1f} us per batch, {n / dt / 1e3:.0f} K samples/s")
'''
for batch in (16, 32, 64, 128, 256, 1024):
    for env in ({}, {"EG_CONV_NO_TINY": "1"}):
        e = dict(os.environ, FIT_BATCH=str(batch), **env)
        out = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True)
        print(("tiny off: " if env else "default:  ") + (out.stdout.strip().splitlines() or [out.stderr[-300:]])[-1], flush=True)
PY
