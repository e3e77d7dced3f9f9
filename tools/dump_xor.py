import os, sys
sys.path.insert(0, os.getcwd())
os.makedirs("gpurun_out/r06x", exist_ok=True)
os.environ["EG_DUMP_CODE"] = "gpurun_out/r06x"
os.environ["EG_NO_KERNEL_CACHE"] = "1"
import numpy as np, exprgrad_amd as eg
from exprgrad_amd import examples, model as egm
ctx = eg.newGpuContext(0)
m = egm.compile(*examples.xor_from_scratch(), gpu=ctx)
rng = np.random.default_rng(0); f = np.float32
x = rng.integers(0, 2, size=(65536, 2)).astype(f); y = (x[:, :1] != x[:, 1:]).astype(f)
for _ in range(3): m.apply("train", {"x": x, "y": y})
ctx.sync()
print(m.launch_plan("train"))
