#!/usr/bin/env python
"""Random sweep of bias-sized contractions (N <= 40, every transpose / accumulate / bias combination) against float64 on the GPU box: tools/sweep_gemm.py"""
import os, sys, random
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import exprgrad_amd as eg
from exprgrad_amd import ops
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
ctx = eg.newGpuContext(0, stream=stream.cuda_stream)
rng = random.Random(7)
bad = 0
for case in range(500):
    M = rng.choice([1, 3, 64, 100, 128, 257, 1000, 4096, 20000])
    N = rng.choice([1, 2, 3, 5, 7, 8, 10, 12, 16, 17, 31, 32, 33, 40])
    K = rng.choice([4, 8, 12, 16, 20, 64, 100, 512, 516, 1000, 3000, 4097])
    ta, tb = rng.random() < 0.4, rng.random() < 0.4
    acc, bias = rng.random() < 0.3, rng.random() < 0.3
    A = torch.rand((K, M) if ta else (M, K), device="cuda") - 0.5
    B = torch.rand((N, K) if tb else (K, N), device="cuda") - 0.5
    C = torch.rand((M, N), device="cuda")
    bv = torch.rand((N,), device="cuda") if bias else None
    ref = (A.double().T if ta else A.double()) @ (B.double().T if tb else B.double())
    if acc: ref = ref + C.double()
    if bias: ref = ref + bv.double()
    ops.sgemm(ctx, M, N, K, A, A.shape[1], B, B.shape[1], C, N, trans_a=ta, trans_b=tb, accumulate=acc, bias=bv)
    torch.cuda.synchronize()
    err = ((C.double() - ref).abs().max() / ref.abs().max().clamp_min(1e-30)).item()
    if not err <= 2e-5:
        bad += 1
        print("BAD", M, N, K, ta, tb, acc, bias, err)
print("done, bad =", bad)
