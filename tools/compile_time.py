#!/usr/bin/env python
"""Model-compile latency (what `compile[float32](graphs, gpu=ctx)` — model.nim:270-273 / newModel, model.nim:215-251 —
costs on this backend): eg_model_compile plus the first run of the train target, which is when the shape-specialised
kernels of the plan are built (fusion groups, contractions with a generated epilogue: the 1000-line MFMA template per
epilogue).  One JSON line; run twice with the same EG_KERNEL_CACHE directory for the cold / warm pair (bench.py does).
tools/compile_time.py [--batch-scale S]"""
import ctypes
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np

import exprgrad_amd as eg
from exprgrad_amd import _lib, examples
from exprgrad_amd import model as egm


def cache_stats():
    h, m, s = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_double()
    _lib.call("eg_kernel_cache_stats", ctypes.byref(h), ctypes.byref(m), ctypes.byref(s))
    return h.value, m.value, s.value


def main():
    ctx = eg.newGpuContext(0)
    rng = np.random.default_rng(0)
    f = np.float32
    cases = (
        ("xor", examples.xor_from_scratch, "train", lambda: {"x": rng.integers(0, 2, (65536, 2)).astype(f), "y": rng.random((65536, 1), dtype=f)}),
        ("dense", examples.dense_softmax_net, "train", lambda: {"x": rng.random((65536, 784), dtype=f), "y": np.eye(10, dtype=f)[rng.integers(0, 10, 65536)]}),
        ("fashion_mnist", examples.fashion_mnist_net, "fit", lambda: {"x": rng.random((256, 784), dtype=f), "y": np.eye(10, dtype=f)[rng.integers(0, 10, 256)]}),
    )
    out = {}
    for name, graphs, target, inputs in cases:
        ins = inputs()
        h0, m0, s0 = cache_stats()
        t0 = time.perf_counter()
        model = egm.compile(*graphs(), gpu=ctx)
        t1 = time.perf_counter()
        model.apply(target, ins)
        ctx.sync()
        t2 = time.perf_counter()
        model.apply(target, ins)          # the captured form (second run of a plan is the capture)
        ctx.sync()
        t3 = time.perf_counter()
        h1, m1, s1 = cache_stats()
        out[name] = {"compile_s": round(t1 - t0, 3), "first_step_s": round(t2 - t1, 3), "second_step_s": round(t3 - t2, 3),
                     "kernels_from_cache": h1 - h0, "kernels_compiled": m1 - m0, "seconds_in_the_compiler": round(s1 - s0, 3)}
        model.close()
    buf = ctypes.create_string_buffer(512)
    _lib.call("eg_compiler_info", buf, 512)
    out["compiler"] = buf.value.decode()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
