"""The launch plan of one batch-32 Model.fit step of the fashion_mnist network (which launches leave the main lane)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import exprgrad_amd as eg
from exprgrad_amd import examples, model as egm
ctx = eg.newGpuContext(0)
m = egm.compile(*examples.fashion_mnist_net(), gpu=ctx)
rng = np.random.default_rng(0); f = np.float32
b = int(os.environ.get("FIT_BATCH", "32"))
x = rng.random((b, 784), dtype=f); y = np.eye(10, dtype=f)[rng.integers(0, 10, b)]
m.apply("fit", {"x": x, "y": y}); ctx.sync()
print(m.launch_plan("fit"))
