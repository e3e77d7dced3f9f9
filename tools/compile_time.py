import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import exprgrad_amd as eg
from exprgrad_amd import examples, model as egm
ctx = eg.newGpuContext(0)
for name, graphs, inputs in (
    ("xor", examples.xor_from_scratch(), lambda: {"x": np.random.rand(1024, 2).astype(np.float32), "y": np.random.rand(1024, 1).astype(np.float32)}),
    ("dense", examples.dense_softmax_net(), lambda: {"x": np.random.rand(4096, 784).astype(np.float32), "y": np.eye(10, dtype=np.float32)[np.random.randint(0, 10, 4096)]}),
    ("fashion", examples.fashion_mnist_net(), lambda: {"x": np.random.rand(256, 784).astype(np.float32), "y": np.eye(10, dtype=np.float32)[np.random.randint(0, 10, 256)]}),
):
    t0 = time.perf_counter(); m = egm.compile(*graphs, gpu=ctx); t1 = time.perf_counter()
    tgt = "train" if name != "fashion" else "fit"
    m.apply(tgt, inputs()); ctx_sync = m.call("loss", inputs()); t2 = time.perf_counter()
    print(f"{name}: compile {t1-t0:.2f} s, first apply+loss {t2-t1:.2f} s")
