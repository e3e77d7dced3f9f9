"""Loader / checker for tests/golden/handwritten.json + handwritten/*.kd (hand-written kernel
descriptions with closed-form known answers; see tests/golden/make_handwritten.py)."""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def arr(spec):
    return np.array(spec["data"], dtype=np.float64).reshape(spec["shape"])


def load():
    with open(os.path.join(HERE, "golden", "handwritten.json")) as f:
        cases = json.load(f)
    for name, case in cases.items():
        with open(os.path.join(HERE, "golden", "handwritten", name + ".kd")) as f:
            case["text"] = f.read()
    return cases


def close(got, want, tol):
    got = np.asarray(got, dtype=np.float64)
    assert list(got.shape) == list(want.shape), (got.shape, want.shape)
    if tol == 0.0:
        return bool(np.array_equal(got, want))
    return bool(np.max(np.abs(got - want)) <= tol * max(np.max(np.abs(want)), 1e-30))


def check(case, model, set_param, get_param, get_cache, set_epoch):
    """Drive `model` (oracle or backend: .call / .apply) through one case."""
    for tid, spec in case["params"].items():
        set_param(int(tid), arr(spec).astype(np.float32))
    if "epoch" in case:
        set_epoch(case["epoch"])
    inputs = {k: arr(v).astype(np.float32) for k, v in case["inputs"].items()}
    for call in case.get("calls", []):
        args = {k: inputs[k] for k in call.get("inputs", inputs)}
        got = model.call(call["target"], args)
        assert close(got, arr(call["expect"]), case["tol"]), (call["target"], got, arr(call["expect"]))
    if "apply" in case:
        model.apply(case["apply"], inputs)
        for tid, spec in case["expect_params"].items():
            assert close(get_param(int(tid)), arr(spec), case["tol"]), (tid, get_param(int(tid)), arr(spec))
        for tid, spec in case["expect_caches"].items():
            assert close(get_cache(int(tid)), arr(spec), case["tol"]), (tid, get_cache(int(tid)), arr(spec))
