#!/usr/bin/env python
"""The library's own tile choice over a list of shapes, old cost model beside the new one is a matter of
running it twice (EG_GEMM_OLD_TILE_MODEL=1): [SPIN_MS=100] tools/sweep_auto.py [nn|tn|nt] MxNxK ..."""
import os, sys
os.environ.setdefault("EG_TUNING", "1")   # measurement aids (class `tuning` of csrc/switches.cpp) are honoured only with it
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import exprgrad_amd as eg
from exprgrad_amd import ops

mode = sys.argv[1]
ta, tb = mode[0] == "t", mode[1] == "t"
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
ctx = eg.newGpuContext(0, stream=stream.cuda_stream)
for spec in sys.argv[2:]:
    M, N, K = (int(v) for v in spec.split("x"))
    A = torch.rand((K, M) if ta else (M, K), device="cuda")
    B = torch.rand((N, K) if tb else (K, N), device="cuda")
    C = torch.empty((M, N), device="cuda")
    run = lambda: ops.sgemm(ctx, M, N, K, A, A.shape[1], B, B.shape[1], C, N, trans_a=ta, trans_b=tb)
    for _ in range(3):
        run()
    if os.environ.get("SPIN_MS"):  # sustained clocks: keep the device busy with this launch first (bench.py does the same)
        import time
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < float(os.environ["SPIN_MS"]) * 1e-3:
            for _ in range(50):
                run()
            torch.cuda.synchronize()
    best = 1e9
    for rep in range(4):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(stream)
        for _ in range(10):
            run()
        e.record(stream)
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / 10 * 1e3)
    ref = (A.double().T if ta else A.double()) @ (B.double().T if tb else B.double())
    err = ((C.double() - ref).abs().max() / ref.abs().max()).item()
    print(f"{mode} {spec}: {best:.1f} us  {2.0*M*N*K/best/1e6:.1f} TFLOP/s  rel err {err:.1e}", flush=True)
