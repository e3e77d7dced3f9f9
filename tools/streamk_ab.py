#!/usr/bin/env python
"""Stream-K on 64 x 64 tiles against one block per tile (EG_GEMM_NO_STREAMK=1): us, TFLOP/s, distance to the float64 product and
run-to-run bit equality, per shape and layout.  tools/streamk_ab.py [nn|tn|nt] MxNxK ..."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import exprgrad_amd as eg
from exprgrad_amd import ops, _lib

mode = sys.argv[1]
ta, tb = mode[0] == "t", mode[1] == "t"
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
ctx = eg.newGpuContext(0, stream=stream.cuda_stream)
for spec in sys.argv[2:]:
    M, N, K = (int(v) for v in spec.split("x"))
    A = torch.rand((K, M) if ta else (M, K), device="cuda") - 0.5
    B = torch.rand((N, K) if tb else (K, N), device="cuda") - 0.5
    ref = (A.double().T if ta else A.double()) @ (B.double().T if tb else B.double())
    line = f"{mode} {spec}:"
    for off in (False, True):
        if off:
            os.environ["EG_GEMM_NO_STREAMK"] = "1"
        else:
            os.environ.pop("EG_GEMM_NO_STREAMK", None)
        _lib.reload_switches()
        C = torch.empty((M, N), device="cuda")
        run = lambda: ops.sgemm(ctx, M, N, K, A, A.shape[1], B, B.shape[1], C, N, trans_a=ta, trans_b=tb)
        run(); torch.cuda.synchronize()
        first = C.clone()
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.05:
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
        same = bool(torch.equal(first, C))
        err = ((C.double() - ref).abs().max() / ref.abs().max()).item()
        line += f"  {'tile-per-block' if off else 'stream-K'} {best:7.1f} us {2.0*M*N*K/best/1e6:6.1f} TF err {err:.1e} {'same bits' if same else 'BITS DIFFER run to run'}"
    print(line, flush=True)
