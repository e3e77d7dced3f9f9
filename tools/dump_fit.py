"""Dump the generated kernels of the batch-32 fit step of the fashion_mnist network: tools/dump_fit.py <dir>   (GPU box)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/fit_code"
os.makedirs(out, exist_ok=True)
os.environ["EG_DUMP_CODE"] = out
os.environ["EG_NO_KERNEL_CACHE"] = "1"
import numpy as np, exprgrad_amd as eg
from exprgrad_amd import examples, model as egm
ctx = eg.newGpuContext(0)
m = egm.compile(*examples.fashion_mnist_net(), gpu=ctx)
rng = np.random.default_rng(0); f = np.float32
x = rng.random((128, 784), dtype=f); y = np.eye(10, dtype=f)[rng.integers(0, 10, 128)]
m.fit("fit", {"x": x, "y": y}, batch_size=32)
ctx.sync()
