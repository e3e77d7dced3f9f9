#!/usr/bin/env python
"""Where the halo convolution's time goes: forward conv2 3x3, 256x256, 64 filters with back-to-back launches for
C = 64 / 128 / 256 channels (4 / 8 / 16 chunks per patch) and N = 1 / 2 / 4 images (1 / 2 / 4 patches per persistent
block): T = fixed + items x (chunks x per-chunk + per-item).  GPU box: python tools/conv_scan.py"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import exprgrad_amd as eg
from exprgrad_amd import ops

stream = torch.cuda.Stream()
torch.cuda.set_stream(stream)
ctx = eg.newGpuContext(0, stream=stream.cuda_stream)


def timed(run, reps=20):
    for _ in range(5):
        run()
    best = 1e9
    for _ in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(stream)
        for _ in range(reps):
            run()
        e.record(stream)
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / reps * 1e3)
    return best


for N, C in ((1, 64), (1, 128), (1, 256), (2, 64), (4, 64), (8, 64), (2, 128)):
    H = W = 256
    F = 64
    img = torch.rand((N, H, W, C), device="cuda")
    flt = torch.rand((F, 3, 3, C), device="cuda") * 2 - 1
    out = torch.empty((N, H - 2, W - 2, F), device="cuda")
    us = timed(lambda: ops.conv2_nhwc(ctx, N, H, W, C, F, 3, 3, img, flt, out))
    flops = 2.0 * N * (H - 2) * (W - 2) * F * 9 * C
    print(f"N {N} C {C:3d}: {us:7.1f} us  {flops / us / 1e6:6.1f} TFLOP/s  ({flops / 157.3e6 / us:.3f} of peak)  "
          f"items per block {N}, chunks per item {C // 16}", flush=True)
