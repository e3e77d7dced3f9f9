"""exprgrad_amd — MI355X (gfx950) kernel backend for exprgrad's compiled tensor hot path.

The product is libexprgrad_hip.so (C ABI: include/exprgrad_hip.h; sources: exprgrad_amd/csrc).
This package is the host-side mirror of the reference's interface for that path, used by the
tests and the benchmark:

    runtime  — exprgrad/runtimes/gpu.nim   (listDevices, newGpuContext, GpuBuffer, GpuKernel, GpuTensor)
    ops      — library fast path           (sgemm, map, bias_add, colsum, conv2_nhwc, ...)

There is no CPU fallback anywhere in this package.
"""
from ._lib import GpuError, RuntimeErrorEG, ShapeError, LIB_PATH  # noqa: F401
from .runtime import (GpuBuffer, GpuContext, GpuDevice, GpuKernel, GpuKernelSource, GpuTensor,  # noqa: F401
                      listDevices, list_devices, newGpuContext, new_gpu_context)
from . import ops  # noqa: F401
