#!/usr/bin/env python
"""conv2 forward and its two gradients with back-to-back launches (one event pair around 20 calls, best of 5):
tools/conv_b2b.py [N H W C F] (GPU box; 3 x 3 filters; default = BASELINE configs[3])"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import exprgrad_amd as eg
from exprgrad_amd import ops

N, H, W, C, F = (int(v) for v in sys.argv[1:6]) if len(sys.argv) > 5 else (1, 256, 256, 64, 64)
stream = torch.cuda.Stream()
torch.cuda.set_stream(stream)
ctx = eg.newGpuContext(0, stream=stream.cuda_stream)
img = torch.rand((N, H, W, C), device="cuda")
flt = torch.rand((F, 3, 3, C), device="cuda") * 2 - 1
gout = torch.rand((N, H - 2, W - 2, F), device="cuda") - 0.5
out = torch.empty((N, H - 2, W - 2, F), device="cuda")
gflt, gimg = torch.empty_like(flt), torch.empty_like(img)
flops = 2.0 * N * (H - 2) * (W - 2) * F * 9 * C
cases = {"forward": lambda: ops.conv2_nhwc(ctx, N, H, W, C, F, 3, 3, img, flt, out),
         "grad_filter": lambda: ops.conv2_nhwc_grad_filter(ctx, N, H, W, C, F, 3, 3, img, gout, gflt),
         "grad_image": lambda: ops.conv2_nhwc_grad_image(ctx, N, H, W, C, F, 3, 3, flt, gout, gimg)}
for _ in range(200):          # clock ramp
    cases["forward"]()
for name, run in cases.items():
    for _ in range(5):
        run()
    best = 1e9
    for _ in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(stream)
        for _ in range(20):
            run()
        e.record(stream)
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / 20 * 1e3)
    print(f"{name:12s} {N}x{H}x{W}x{C} -> {F}: {best:6.1f} us  {flops / best / 1e6:6.1f} TFLOP/s  ({flops / best / 157.3e6:.3f} of peak)", flush=True)
