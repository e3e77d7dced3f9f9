#!/usr/bin/env python
"""Per-launch time series of eg_sgemm 4096^3 (clock ramp / throttling check): tools/gemm_series.py [reps]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import exprgrad_amd as eg
from exprgrad_amd import ops
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
ctx = eg.newGpuContext(0, stream=stream.cuda_stream)
n = 4096
A = torch.rand((n, n), device="cuda"); B = torch.rand((n, n), device="cuda"); C = torch.empty((n, n), device="cuda")
torch.cuda.synchronize()
s = [torch.cuda.Event(enable_timing=True) for _ in range(reps)]
e = [torch.cuda.Event(enable_timing=True) for _ in range(reps)]
for i in range(reps):
    s[i].record(stream); ops.sgemm(ctx, n, n, n, A, n, B, n, C, n); e[i].record(stream)
torch.cuda.synchronize()
t = [a.elapsed_time(b) for a, b in zip(s, e)]
for i in range(0, reps, 10):
    print(i, " ".join(f"{x*1e3:.0f}" for x in t[i:i+10]))
