import os, sys
sys.path.insert(0, os.getcwd())
os.makedirs("gpurun_out/r06d/dump", exist_ok=True)
os.environ["EG_DUMP_FUSED"] = "gpurun_out/r06d/dump"
os.environ["EG_DUMP_CODE"] = "gpurun_out/r06d/dump"
os.environ["EG_NO_KERNEL_CACHE"] = "1"
import numpy as np, exprgrad_amd as eg
from exprgrad_amd import examples, model as egm
ctx = eg.newGpuContext(0)
m = egm.compile(*examples.dense_softmax_net(), gpu=ctx)
rng = np.random.default_rng(0); f = np.float32
ins = {"x": rng.random((65536, 784), dtype=f), "y": np.eye(10, dtype=f)[rng.integers(0, 10, 65536)]}
m.apply("train", ins); ctx.sync()
print(m.launch_plan("train"))
