"""The reference's conv2 benchmark program in float64 (benchmarks/conv2/conv2.nim:330-364 shape) through compile + apply."""
import sys

import numpy as np
import torch

import exprgrad_amd as eg
from exprgrad_amd import examples
from exprgrad_amd import model as egm


def main():
    F = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    ctx = eg.newGpuContext()
    model = egm.compile(*examples.conv2_3d(), gpu=ctx, dtype=np.float64)
    H, W, C = 960, 1280, 8
    image = torch.rand((H, W, C), device="cuda", dtype=torch.float64)
    filters = torch.rand((F, 3, 3, C), device="cuda", dtype=torch.float64) * 4 - 2
    feed = {"image": image, "filters": filters}
    for _ in range(5):
        model.apply("conv2", feed)
    ctx.sync()
    stream = torch.cuda.ExternalStream(ctx.stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 50
    e0.record(stream)
    for _ in range(reps):
        model.apply("conv2", feed)
    e1.record(stream)
    ctx.sync()
    ms = e0.elapsed_time(e1) / reps
    nbytes = 8.0 * (H * W * C + (H - 2) * (W - 2) * F)
    flops = 2.0 * (H - 2) * (W - 2) * F * 72
    print(f"conv2 f64 960x1280x8 -> {F}: {ms * 1e3:.1f} us  {nbytes / ms / 1e6:.0f} GB/s  {flops / ms / 1e9:.2f} TFLOP/s")


if __name__ == "__main__":
    main()
