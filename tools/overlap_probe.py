#!/usr/bin/env python
"""Does an HBM-bound kernel hide under an MFMA-bound contraction on another stream?  (GPU box)
The weight-gradient contraction of the dense net (784 x 512 x 65536, TN) on one context, the bias
gradient (column sum of 65536 x 512) and the N = 10 weight gradient on a second one."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import exprgrad_amd as eg
from exprgrad_amd import ops

sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
ca = eg.newGpuContext(0, stream=sa.cuda_stream)
cb = eg.newGpuContext(0, stream=sb.cuda_stream)
B = 65536
x = torch.rand((B, 784), device="cuda")
g1 = torch.rand((B, 512), device="cuda")
a1 = torch.rand((B, 512), device="cuda")
g2 = torch.rand((B, 10), device="cuda")
gw1 = torch.empty((784, 512), device="cuda")
gw2 = torch.empty((512, 10), device="cuda")
gb1 = torch.empty((512,), device="cuda")
torch.cuda.synchronize()


def big(ctx):
    ops.sgemm(ctx, 784, 512, B, x, 784, g1, 512, gw1, 512, trans_a=True)


def small(ctx):
    ops.colsum(ctx, B, 512, g1, gb1)
    ops.sgemm(ctx, 512, 10, B, a1, 512, g2, 10, gw2, 10, trans_a=True)


def timed(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(sa)
    for _ in range(reps):
        fn()
    e.record(sa)
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


def serial():
    small(ca)
    big(ca)


ev_fork, ev_join = torch.cuda.Event(), torch.cuda.Event()


def overlapped():
    ev_fork.record(sa)
    sb.wait_event(ev_fork)
    small(cb)
    ev_join.record(sb)
    big(ca)
    sa.wait_event(ev_join)


print(f"big alone {timed(lambda: big(ca)):.1f} us, small alone {timed(lambda: small(ca)):.1f} us, "
      f"serial {timed(serial):.1f} us, overlapped {timed(overlapped):.1f} us")
