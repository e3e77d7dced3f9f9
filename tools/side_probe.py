#!/usr/bin/env python
"""What a bandwidth-bound launch on a second stream costs the long weight-gradient contraction of cfg 5 (784 x 512 x 65536, TN)
that runs next to it: the contraction alone, next to the 512 x 10 x 65536 TN product on matrix tiles (what the dense step's side lane
runs today), and next to pure streaming kernels over the same 134 MB (column sum, row sum).  tools/side_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import exprgrad_amd as eg
from exprgrad_amd import ops

s_main, s_side = torch.cuda.Stream(), torch.cuda.Stream()
ctx_main = eg.newGpuContext(0, stream=s_main.cuda_stream)
ctx_side = eg.newGpuContext(0, stream=s_side.cuda_stream)
B = 65536
x = torch.rand((B, 784), device="cuda") - 0.5
gh = torch.rand((B, 512), device="cuda") - 0.5
a = torch.rand((B, 512), device="cuda") - 0.5
gz = torch.rand((B, 10), device="cuda") - 0.5
gw1 = torch.empty((784, 512), device="cuda")
gw2 = torch.empty((512, 10), device="cuda")
col = torch.empty((512,), device="cuda")
row = torch.empty((B,), device="cuda")
big = lambda: ops.sgemm(ctx_main, 784, 512, B, x, 784, gh, 512, gw1, 512, trans_a=True)
sides = {
    "nothing": None,
    "512 x 10 x 65536 TN on matrix tiles (today's side lane)": lambda: ops.sgemm(ctx_side, 512, 10, B, a, 512, gz, 10, gw2, 10, trans_a=True),
    "column sum of the same 134 MB (streaming)": lambda: ops.colsum(ctx_side, B, 512, a, col),
    "row sum of the same 134 MB (streaming)": lambda: ops.rowsum(ctx_side, B, 512, a, row),
    "two column sums": lambda: (ops.colsum(ctx_side, B, 512, a, col), ops.colsum(ctx_side, B, 512, gh, col)),
}
def iteration(side):
    big()                      # the contraction first, so that the side launch starts inside it
    if side:
        side()
        ev = torch.cuda.Event()
        ev.record(s_side)
        s_main.wait_event(ev)  # the join of the step: the next launch of the main lane needs both


for name, side in sides.items():
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.15:     # sustained clocks
        for _ in range(10):
            iteration(side)
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s_main)
    for _ in range(30):
        iteration(side)
    e1.record(s_main)
    torch.cuda.synchronize()
    print(f"{name:60s}: {e0.elapsed_time(e1) * 1e3 / 30:7.1f} us per (contraction + side launch, joined)", flush=True)
