#!/usr/bin/env python
"""Fixed cost vs per-k cost of eg_sgemm on an M x N output: tools/gemm_sweep_k.py M N [nn|nt|tn|tt] K1 K2 ...
Times each K with a run of back-to-back launches between two events (no per-launch event overhead) and
prints microseconds per launch; T(K) = a + b K separates prologue / epilogue / second pass from the main loop."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import exprgrad_amd as eg
from exprgrad_amd import ops

M, N = int(sys.argv[1]), int(sys.argv[2])
mode = sys.argv[3]
ks = [int(v) for v in sys.argv[4:]]
ta, tb = mode[0] == "t", mode[1] == "t"
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
ctx = eg.newGpuContext(0, stream=stream.cuda_stream)
for K in ks:
    A = torch.rand((K, M) if ta else (M, K), device="cuda")
    B = torch.rand((N, K) if tb else (K, N), device="cuda")
    C = torch.empty((M, N), device="cuda")
    run = lambda: ops.sgemm(ctx, M, N, K, A, A.shape[1], B, B.shape[1], C, N, trans_a=ta, trans_b=tb)
    for _ in range(5):
        run()
    best = 1e9
    for rep in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(stream)
        for _ in range(10):
            run()
        e.record(stream)
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / 10 * 1e3)
    print(f"{mode} {M}x{N}x{K}: {best:.1f} us  {2.0*M*N*K/best/1e6:.1f} TFLOP/s", flush=True)
