#!/bin/bash
export EG_TUNING=1   # measurement aids (class `tuning` of csrc/switches.cpp) are honoured only with it
# Model.fit on the fashion_mnist network: us per batch at several batch sizes with and without sample groups (one box).
# tools/fit_sample_ab.sh [batch sizes...]
python - "$@" <<'PY'
import os, subprocess, sys
code = r'''
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import exprgrad_amd as eg
from exprgrad_amd import examples, model as egm
ctx = eg.newGpuContext(0)
m = egm.compile(*examples.fashion_mnist_net(), gpu=ctx)
rng = np.random.default_rng(0); f = np.float32
batch = int(os.environ["FIT_BATCH"]); n = 60000 // batch * batch
x = rng.random((n, 784), dtype=f); y = np.eye(10, dtype=f)[rng.integers(0, 10, n)]
for _ in range(2): m.fit("fit", {"x": x, "y": y}, batch_size=batch)
ctx.sync(); t0 = time.perf_counter()
for _ in range(3): m.fit("fit", {"x": x, "y": y}, batch_size=batch)
ctx.sync(); dt = (time.perf_counter() - t0) / 3
print(f"batch {batch}: {dt / (n // batch) * 1e6:.1f} us per batch, {n / dt / 1e3:.0f} K samples/s, epoch {dt * 1e3:.1f} ms")
'''
sizes = [int(a) for a in sys.argv[1:]] or [8, 16, 32, 64, 128, 256]
for batch in sizes:
    for label, env in (("sample groups, matrix-core convolution members", {"EG_SAMPLE_FUSE_MAX_BATCH": "100000"}),
                       ("sample groups, parameters not staged in LDS   ", {"EG_SAMPLE_FUSE_MAX_BATCH": "100000", "EG_SAMPLE_NO_STAGE": "1"}),
                       ("sample groups, scalar convolution members     ", {"EG_SAMPLE_FUSE_MAX_BATCH": "100000", "EG_SAMPLE_NO_MFMA": "1", "EG_SAMPLE_NO_STAGE": "1"}),
                       ("launch chain                                  ", {"EG_NO_SAMPLE_FUSE": "1"})):
        e = dict(os.environ, FIT_BATCH=str(batch), **env)
        out = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True)
        print(label + ": " + (out.stdout.strip().splitlines() or [out.stderr[-300:]])[-1], flush=True)
PY
