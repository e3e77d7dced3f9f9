export EG_TUNING=1   # measurement aids (class `tuning` of csrc/switches.cpp) are honoured only with it
for b in 256 512 768 1024 2048; do
EG_CONV_DIRECT_BLOCKS=$b python - <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, exprgrad_amd as eg
from exprgrad_amd import examples, model as egm
ctx = eg.newGpuContext(0)
m = egm.compile(*examples.fashion_mnist_net(), gpu=ctx)
rng = np.random.default_rng(0); f = np.float32
x = rng.random((60000, 784), dtype=f); y = np.eye(10, dtype=f)[rng.integers(0, 10, 60000)]
for _ in range(2): m.fit("fit", {"x": x, "y": y}, batch_size=4096)
ctx.sync(); t=time.perf_counter()
for _ in range(5): m.fit("fit", {"x": x, "y": y}, batch_size=4096)
ctx.sync(); dt=(time.perf_counter()-t)/5
print("blocks", os.environ["EG_CONV_DIRECT_BLOCKS"], "epoch ms %.2f" % (dt*1e3), "us/batch %.1f" % (dt*1e6/14))
PY
done
