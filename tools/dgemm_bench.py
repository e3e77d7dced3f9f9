"""eg_dgemm timing: sizes from the command line (default 4096), NN and TN; TFLOP/s against the float64 matrix peak."""
import ctypes
import sys

import numpy as np
import torch

import exprgrad_amd as eg
from exprgrad_amd import _lib

PEAK = 78.6  # TFLOP/s, v_mfma_f64_16x16x4_f64: 32 FLOP / clk / SIMD x 1024 SIMDs x 2.4 GHz


def main():
    sizes = [int(v) for v in sys.argv[1:]] or [4096]
    ctx = eg.newGpuContext()
    dev = torch.device("cuda", ctx.device)
    for n in sizes:
        a = torch.rand((n, n), dtype=torch.float64, device=dev) - 0.5
        b = torch.rand((n, n), dtype=torch.float64, device=dev) - 0.5
        c = torch.empty((n, n), dtype=torch.float64, device=dev)
        for ta, tb in ((0, 0), (1, 0), (0, 1), (0, 0)):
            def run():
                _lib.call("eg_dgemm", ctx.handle, ta, tb, n, n, n, ctypes.c_void_p(a.data_ptr()), n, ctypes.c_void_p(b.data_ptr()), n,
                          ctypes.c_void_p(c.data_ptr()), n, 0, None)
            for _ in range(3):
                run()
            ctx.sync()
            stream = torch.cuda.ExternalStream(ctx.stream)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = int(__import__('os').environ.get('REPS', '10'))
            e0.record(stream)
            for _ in range(reps):
                run()
            e1.record(stream)
            ctx.sync()
            ms = e0.elapsed_time(e1) / reps
            tf = 2.0 * n ** 3 / ms / 1e9
            print(f"dgemm {n}^3 {'T' if ta else 'N'}{'T' if tb else 'N'}: {ms * 1e3:9.1f} us  {tf:6.2f} TFLOP/s  {tf / PEAK:.3f} of {PEAK}")
        if n <= 2048:
            want = (a.cpu().numpy() @ b.cpu().numpy().T)
            print("   max rel err vs numpy:", float(np.max(np.abs(c.cpu().numpy() - want)) / np.max(np.abs(want))))


if __name__ == "__main__":
    main()
