#!/usr/bin/env python
"""Time eg_sgemm on one problem shape: tools/gemm_shape.py M N K [nn|nt|tn|tt] [reps]
(run on the GPU box; EG_GEMM_FORCE_TILE=bm,bn forces a tile)."""
import os
os.environ.setdefault("EG_TUNING", "1")   # measurement aids (class `tuning` of csrc/switches.cpp) are honoured only with it
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import exprgrad_amd as eg
from exprgrad_amd import ops

M, N, K = (int(v) for v in sys.argv[1:4])
mode = sys.argv[4] if len(sys.argv) > 4 else "nn"
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 20
ta, tb = mode[0] == "t", mode[1] == "t"
stream = torch.cuda.Stream()
torch.cuda.set_stream(stream)
ctx = eg.newGpuContext(0, stream=stream.cuda_stream)
A = torch.rand((K, M) if ta else (M, K), device="cuda")
B = torch.rand((N, K) if tb else (K, N), device="cuda")
C = torch.empty((M, N), device="cuda")
run = lambda: ops.sgemm(ctx, M, N, K, A, A.shape[1], B, B.shape[1], C, N, trans_a=ta, trans_b=tb)
for _ in range(3):
    run()
torch.cuda.synchronize()
s = [torch.cuda.Event(enable_timing=True) for _ in range(reps)]
e = [torch.cuda.Event(enable_timing=True) for _ in range(reps)]
for i in range(reps):
    s[i].record(stream); run(); e[i].record(stream)
torch.cuda.synchronize()
t = sorted(a.elapsed_time(b) for a, b in zip(s, e))
# the same launches back to back between ONE pair of events: what a caller that keeps the queue full sees
# (an event pair per launch adds the event packets and the gap they open: ~40 us on a three-launch call)
s2, e2 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s2.record(stream)
for _ in range(reps):
    run()
e2.record(stream)
torch.cuda.synchronize()
b2b = s2.elapsed_time(e2) / reps
ref = (A.double().T if ta else A.double()) @ (B.double().T if tb else B.double())
err = ((C.double() - ref).abs().max() / ref.abs().max()).item()
print(f"{mode} {M}x{N}x{K} tile={os.environ.get('EG_GEMM_FORCE_TILE', 'auto')}: per-launch events median {t[len(t)//2]*1e3:.1f} us "
      f"(min {t[0]*1e3:.1f}), back to back {b2b*1e3:.1f} us = {2.0*M*N*K/b2b/1e9:.1f} TFLOP/s, rel err {err:.2e}")
