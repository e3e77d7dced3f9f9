#!/usr/bin/env python
"""Time conv2 forward and its two gradients: tools/conv_shape.py N H W C F FH FW [reps] (GPU box)."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import exprgrad_amd as eg
from exprgrad_amd import ops

N, H, W, C, F, FH, FW = (int(v) for v in sys.argv[1:8])
reps = int(sys.argv[8]) if len(sys.argv) > 8 else 30
stream = torch.cuda.Stream()
torch.cuda.set_stream(stream)
ctx = eg.newGpuContext(0, stream=stream.cuda_stream)
Ho, Wo = H - FH + 1, W - FW + 1
img = torch.rand((N, H, W, C), device="cuda")
flt = torch.rand((F, FH, FW, C), device="cuda") * 2 - 1
gout = torch.rand((N, Ho, Wo, F), device="cuda") - 0.5
out = torch.empty((N, Ho, Wo, F), device="cuda")
gflt = torch.empty_like(flt)
gimg = torch.empty_like(img)
flops = 2.0 * N * Ho * Wo * F * FH * FW * C
cases = {
    "forward": lambda: ops.conv2_nhwc(ctx, N, H, W, C, F, FH, FW, img, flt, out),
    "grad_filter": lambda: ops.conv2_nhwc_grad_filter(ctx, N, H, W, C, F, FH, FW, img, gout, gflt),
    "grad_image": lambda: ops.conv2_nhwc_grad_image(ctx, N, H, W, C, F, FH, FW, flt, gout, gimg),
}
only = os.environ.get("CASES")  # e.g. CASES=forward,grad_image
for name, run in cases.items():
    if only and name not in only.split(","):
        continue
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    s = [torch.cuda.Event(enable_timing=True) for _ in range(reps)]
    e = [torch.cuda.Event(enable_timing=True) for _ in range(reps)]
    for i in range(reps):
        s[i].record(stream); run(); e[i].record(stream)
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in zip(s, e))
    print(f"{name:12s} {N}x{H}x{W}x{C} -> {F} ({FH}x{FW}): median {t[len(t)//2]*1e3:.1f} us, {flops/t[len(t)//2]/1e9:.1f} TFLOP/s")
