#!/usr/bin/env python
"""Tile / split / k-depth sweep over square mid-size contractions in one process (tuning aid):
tools/sweep_mid.py [nn|tn] size..."""
import os, sys
os.environ.setdefault("EG_TUNING", "1")   # measurement aids (class `tuning` of csrc/switches.cpp) are honoured only with it
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import exprgrad_amd as eg
from exprgrad_amd import ops

mode = sys.argv[1]
sizes = [int(v) for v in sys.argv[2:]]
ta, tb = mode[0] == "t", mode[1] == "t"
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
ctx = eg.newGpuContext(0, stream=stream.cuda_stream)

def timed(M, N, K, A, B, C):
    run = lambda: ops.sgemm(ctx, M, N, K, A, A.shape[1], B, B.shape[1], C, N, trans_a=ta, trans_b=tb)
    for _ in range(3):
        run()
    best = 1e9
    for rep in range(4):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(stream)
        for _ in range(10):
            run()
        e.record(stream)
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / 10 * 1e3)
    return best

configs = [("auto", {})]
for t in ("64,64", "128,64", "128,128", "256,64", "256,256"):
    for sp in (1, 2, 3, 4, 6, 8, 16):
        configs.append((f"{t}/s{sp}", {"EG_GEMM_FORCE_TILE": t, "EG_GEMM_FORCE_SPLITS": str(sp)}))
        if t == "64,64":
            configs.append((f"{t}/s{sp}/k32", {"EG_GEMM_FORCE_TILE": t, "EG_GEMM_FORCE_SPLITS": str(sp), "EG_GEMM_SMALL_BK32": "1"}))
for n in sizes:
    M = N = K = n
    A = torch.rand((K, M) if ta else (M, K), device="cuda")
    B = torch.rand((N, K) if tb else (K, N), device="cuda")
    C = torch.empty((M, N), device="cuda")
    res = []
    for name, env in configs:
        for k in ("EG_GEMM_FORCE_TILE", "EG_GEMM_FORCE_SPLITS", "EG_GEMM_SMALL_BK32"):
            os.environ.pop(k, None)
        os.environ.update(env)
        res.append((timed(M, N, K, A, B, C), name))
    auto = res[0][0]
    res.sort()
    print(f"{mode} {n}^3: auto {auto:.1f} us ({2.0*n**3/auto/1e6:.1f} TF) | best " +
          ", ".join(f"{name} {t:.1f}" for t, name in res[:4]) + f" -> {2.0*n**3/res[0][0]/1e6:.1f} TF", flush=True)
