#!/usr/bin/env python
"""Writes tests/golden/config_fixtures.json: expected outputs of DOWNSIZED versions of BASELINE.json's
five configurations, computed by the oracle (oracle/refcpu.c, oracle/kd.py +
refinterp.c — the CPU restatement of the reference, itself pinned by known_answers.json).  Inputs
are regenerated from the seeds below; only shapes, seeds and expected values are stored.

The reference itself (Nim + LLVM 13) cannot run in the build image, so these are oracle outputs
frozen at the time of writing: a regression anchor for the oracle (CPU test) and a fixed target
for the HIP path (GPU test).  Re-run: python tests/golden/make_config_fixtures.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, os.path.join(HERE, "..", ".."))

import refcases  # noqa: E402
from exprgrad_amd import examples  # noqa: E402


def T(a):
    a = np.asarray(a, dtype=np.float32)
    return {"shape": list(a.shape), "data": [float(v) for v in a.ravel()]}


def inputs_of(name):
    """Seeded inputs of a downsized configuration (same distributions as SURVEY.md §8d)."""
    f = np.float32
    if name == "cfg1_matmul_256":       # configs[0]: the reference's own CPU case, full size
        rng = np.random.default_rng(1)
        return {"a": rng.random((256, 256), dtype=f), "b": rng.random((256, 256), dtype=f)}
    if name == "cfg2_matmul_small":     # configs[1] downsized: 48 x 40 x 56, ragged on purpose
        rng = np.random.default_rng(2)
        return {"a": rng.random((48, 56), dtype=f), "b": rng.random((56, 40), dtype=f)}
    if name == "cfg3_xor_step":         # configs[2] downsized: batch 64
        rng = np.random.default_rng(3)
        x = rng.integers(0, 2, (64, 2)).astype(f)
        return {"x": x, "y": (x[:, :1] != x[:, 1:]).astype(f)}
    if name == "cfg4_conv2_small":      # configs[3] downsized: 12 x 14 x 4 -> 8 filters, 3 x 3
        rng = np.random.default_rng(4)
        return {"image": rng.random((12, 14, 4), dtype=f), "filters": (rng.random((8, 3, 3, 4), dtype=f) * 4 - 2).astype(f)}
    if name == "cfg5_dense_step":       # configs[4] downsized: 48 -> 32 -> 10, batch 40
        rng = np.random.default_rng(5)
        return {"x": rng.random((40, 48), dtype=f), "y": np.eye(10, dtype=f)[rng.integers(0, 10, 40)]}
    raise KeyError(name)


def params_of(name, model_params):
    """Seeded parameters, U[-0.1, 0.1) as parser.nim:714 draws them (the reference's RNG is unpinned)."""
    rng = np.random.default_rng({"cfg3_xor_step": 30, "cfg5_dense_step": 50}[name])
    return {t: (rng.random(np.shape(model_params[t]), dtype=np.float32) * 0.2 - 0.1).astype(np.float32) for t in sorted(model_params)}


CFG1_ROWS = [0, 37, 101, 255]

GRAPHS = {
    "cfg1_matmul_256": (refcases.matmul, "c"),
    "cfg2_matmul_small": (refcases.matmul, "c"),
    "cfg3_xor_step": (examples.xor_from_scratch, "train"),
    "cfg4_conv2_small": (examples.conv2_3d, None),
    "cfg5_dense_step": (lambda: examples.dense_softmax_net(48, 32, 10, 0.01), "train"),
}


def main():
    from oracle import kd
    out = {}
    for name, (build, target) in GRAPHS.items():
        graphs = build()
        text = refcases.program_text(graphs)
        model = kd.Model(text, threads=1)      # one thread: the reference's serial summation order
        if target is None:
            target = [ln.split()[1] for ln in text.splitlines() if ln.startswith("target ")][0]
        inputs = inputs_of(name)
        entry = {"target": target}
        if model.params:
            for t, v in params_of(name, model.params).items():
                model.params[t][...] = v
            model.apply(target, inputs)
            model.apply(target, inputs)
            entry["params_after_two_steps"] = {str(t): T(model.params[t]) for t in sorted(model.params)}
            entry["predict"] = T(model.call("predict", {k: v for k, v in inputs.items() if k == "x"}))
        elif name == "cfg1_matmul_256":   # full size: a few complete rows and all column sums instead of 65 536 values
            c = model.call(target, inputs)
            entry["rows"] = CFG1_ROWS
            entry["output_rows"] = T(c[CFG1_ROWS])
            entry["column_sums_f64"] = [float(v) for v in c.astype(np.float64).sum(axis=0)]
        else:
            entry["output"] = T(model.call(target, inputs))
        out[name] = entry
    path = os.path.join(HERE, "config_fixtures.json")
    with open(path, "w") as f:
        json.dump(out, f)
    print("wrote", path, {k: list(v) for k, v in out.items()})


if __name__ == "__main__":
    main()
