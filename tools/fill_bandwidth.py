import os, sys
sys.path.insert(0, os.getcwd())
import torch, exprgrad_amd as eg
from exprgrad_amd import ops
ctx = eg.newGpuContext(0, stream=torch.cuda.current_stream().cuda_stream)
for mb in (16, 67, 134, 268, 1024):
    n = mb * (1 << 20) // 4
    t = torch.empty(n, device="cuda")
    s = torch.rand(n, device="cuda")
    for name, fn in (("fill", lambda: ops.fill(ctx, n, 1.0, t)), ("torch copy (read+write)", lambda: t.copy_(s))):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        moved = mb * (1 << 20) * (2 if "copy" in name else 1)
        print(f"{mb:5d} MB {name:26s} {us:8.1f} us  {moved / us * 1e-6:6.2f} TB/s")
