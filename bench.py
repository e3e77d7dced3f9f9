#!/usr/bin/env python
"""Benchmark of the compiled-tensor hot path on MI355X (contract: see DESIGN.md "Measurement").

  python bench.py --gpus 1 --steps K --warmup W      # matmul M=N=K=4096 float32 (BASELINE configs[1])

One "step" is one pass of the hot path over one batch of synthetic input that is already
resident in HBM when the timed region starts.  Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: f32-input MFMA = 64 FLOP/clk/SIMD x 4 x 256 CUs x 2.4 GHz
HBM_PEAK_GBS = 8000.0         # HBM3E spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="auto", choices=["auto", "matmul"])
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


def cpu_baseline_matmul(n, budget_s=12.0):
    """The oracle's matmul loop nest (reference order y, it, x; threaded over y across all host
    cores exactly as builtinRunThreads splits it) on a bounded row-slice of the same problem."""
    import numpy as np
    from oracle import refcpu
    refcpu.build()
    cores = os.cpu_count() or 1
    rng = np.random.default_rng(2)
    b = rng.random((n, n), dtype=np.float32)
    rows = min(n, max(cores, 64))
    a = rng.random((rows, n), dtype=np.float32)
    threads = refcpu.thread_count(rows, n * (n + 1), cores)
    t0 = time.perf_counter()
    refcpu.sgemm(a, b, threads=threads)
    dt = time.perf_counter() - t0
    # scale the slice so the timed run is about budget_s of CPU wall time (bounded by the full problem)
    rows2 = int(min(n, max(rows, rows * budget_s / max(dt, 1e-6))))
    rows2 = max(cores, rows2 // cores * cores)
    a = rng.random((rows2, n), dtype=np.float32)
    threads = refcpu.thread_count(rows2, n * (n + 1), cores)
    t0 = time.perf_counter()
    refcpu.sgemm(a, b, threads=threads)
    dt = time.perf_counter() - t0
    gflops = 2.0 * rows2 * n * n / dt / 1e9
    return {"value": round(gflops, 2), "unit": "GFLOP/s", "cores": threads, "kind": "port",
            "sample": f"rows 0..{rows2} of the {n}x{n}x{n} product ({rows2}x{n}x{n}), oracle/refcpu.c ref_sgemm, "
                      f"{dt:.2f} s wall"}


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import exprgrad_amd as eg
    from exprgrad_amd import ops

    stream = torch.cuda.current_stream()
    ctx = eg.newGpuContext(local_rank, stream=stream.cuda_stream)

    n = args.size
    gen = torch.Generator(device="cuda")
    gen.manual_seed(2 + rank)
    a = torch.rand((n, n), device="cuda", dtype=torch.float32, generator=gen)  # U[0,1): matmul_gpu.nim:69-70
    b = torch.rand((n, n), device="cuda", dtype=torch.float32, generator=gen)
    c = torch.empty((n, n), device="cuda", dtype=torch.float32)

    def step():
        ops.sgemm(ctx, n, n, n, a, n, b, n, c, n)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    starts = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ends = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        starts[i].record(stream)
        step()
        ends[i].record(stream)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kernel_ms = sorted(s.elapsed_time(e) for s, e in zip(starts, ends))
    avg_kernel_ms = sum(kernel_ms) / len(kernel_ms)

    flops = 2.0 * n * n * n
    if rank == 0:
        value = flops * args.steps * world / elapsed / 1e9
        achieved = flops / (avg_kernel_ms * 1e-3) / 1e12
        line = {
            "metric": "GFLOP/s matmul 4096^3 f32 (1 GPU)" if n == 4096 else f"GFLOP/s matmul {n}^3 f32",
            "value": round(value, 1), "unit": "GFLOP/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"matmul M=N=K={n} float32, C = A*B via eg_sgemm (MFMA+LDS tiled HIP kernel), "
                                   "inputs resident in HBM", "parallelism": "replicas" if world > 1 else "single"},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": F32_MFMA_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(achieved / F32_MFMA_PEAK_TFLOPS, 4), "traffic": None,
                         "kernel": "gemm_f32_mfma_kernel<128,128,64,64,NN>",
                         "kernel_ms_avg": round(avg_kernel_ms, 4), "kernel_ms_min": round(kernel_ms[0], 4)},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline_matmul(n)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
