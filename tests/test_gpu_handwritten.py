"""The backend against closed-form known answers (softmax / crossEntropy and its gradient, one adam
step, 4-D conv2, maxpool2 + customGrad) on kernel-description text written by hand — the DSL mirror
is not involved, so the text front-end of the C ABI is pinned by itself
(tests/golden/make_handwritten.py, tests/golden/handwritten/*.kd)."""
import numpy as np
import pytest

import handwritten
from conftest import TOL, rel_err
from exprgrad_amd import model as egm

pytestmark = pytest.mark.gpu
CASES = handwritten.load()


def build(gpu_ctx, text):
    return egm.Model(egm._LoadedProgram(text), gpu_ctx)


@pytest.mark.parametrize("name", sorted(CASES))
def test_backend_reproduces_the_closed_forms(gpu_ctx, name):
    case = CASES[name]
    gpu = build(gpu_ctx, case["text"])

    def set_param(t, v):
        gpu.params[t] = v

    def set_epoch(e):
        gpu.epoch = e

    handwritten.check(case, gpu, set_param, lambda t: gpu.params[t], lambda t: gpu.caches[t], set_epoch)
    gpu.close()


@pytest.mark.parametrize("name", sorted(CASES))
def test_backend_equals_the_oracle_on_the_handwritten_programs(gpu_ctx, name):
    """Larger seeded inputs through the same hand-written text: backend vs oracle at 1e-5."""
    from oracle import kd
    case = CASES[name]
    gpu, ref = build(gpu_ctx, case["text"]), kd.Model(case["text"])
    rng = np.random.default_rng(7)
    f = np.float32
    if name == "conv2_4d":
        ins = {"images": rng.random((3, 9, 11, 5), dtype=f), "filters": (rng.random((7, 3, 2, 5), dtype=f) - 0.5).astype(f)}
        assert rel_err(gpu.call("conv2", ins), ref.call("conv2", ins)) <= TOL
    elif name == "maxpool2_grad":
        ins = {"x": (rng.random((2, 6, 8, 3), dtype=f) - 0.5).astype(f), "w": rng.random((2, 3, 4, 3), dtype=f)}
        assert np.array_equal(gpu.call("pool", {"x": ins["x"]}), ref.call("pool", {"x": ins["x"]}))
        assert np.array_equal(gpu.call("grad", ins), ref.call("grad", ins))       # selection only: exact
    elif name == "xor_from_scratch":
        for tid in sorted(ref.params):
            v = (rng.random(ref.params[tid].shape, dtype=f) * 2 - 1).astype(f)
            gpu.params[tid] = v
            ref.params[tid][...] = v
        x = rng.integers(0, 2, size=(256, 2)).astype(f)
        ins = {"x": x, "y": (x[:, :1] != x[:, 1:]).astype(f)}
        assert rel_err(gpu.call("predict", {"x": x}), ref.call("predict", {"x": x})) <= TOL
        # (the reference sums the 256 squares one after the other in float32, the backend as a tree: n * u / 2 apart at most)
        assert rel_err(gpu.call("loss", ins), ref.call("loss", ins)) <= 256 * 6e-8
        for _ in range(3):
            gpu.apply("train", ins)
            ref.apply("train", ins)
        for tid in sorted(ref.params):
            assert rel_err(gpu.params[tid], ref.params[tid]) <= 3 * 256 * 6e-8, tid      # three steps of 256-term sums
    elif name == "pool_chain_adam":
        flt = (rng.random((2, 3, 3, 1), dtype=f) - 0.5).astype(f)
        gpu.params[1] = flt
        ref.params[1][...] = flt
        ins = {"x": rng.random((5, 36), dtype=f), "y": rng.random((5, 2, 2, 2), dtype=f)}
        assert rel_err(gpu.call("predict", {"x": ins["x"]}), ref.call("predict", {"x": ins["x"]})) <= TOL
        assert rel_err(gpu.call("loss", ins), ref.call("loss", ins)) <= TOL
        for step in range(1, 4):
            gpu.epoch = ref.epoch = step
            gpu.apply("fit", ins)
            ref.apply("fit", ins)
        # (adam's update is ill-conditioned in the gradient: three steps from identical state stay within 1e-4)
        assert rel_err(gpu.params[1], ref.params[1]) <= 1e-4
        for c in (2, 10):
            assert rel_err(gpu.caches[c], ref.caches[c]) <= 1e-4
    elif name == "softmax_xent":
        z = (rng.random((2, 3), dtype=f) * 4 - 2).astype(f)
        gpu.params[1] = z
        ref.params[1][...] = z
        y = np.eye(3, dtype=f)[[2, 0]]
        for target in ("predict", "loss", "grad"):
            args = {} if target == "predict" else {"y": y}
            assert rel_err(gpu.call(target, args), ref.call(target, args)) <= TOL, target
    else:
        p = (rng.random(3, dtype=f) * 4 - 2).astype(f)
        gpu.params[1] = p
        ref.params[1][...] = p
        t = rng.random(3, dtype=f)
        for step in range(1, 4):
            gpu.epoch = ref.epoch = step
            gpu.apply("train", {"t": t})
            ref.apply("train", {"t": t})
        assert rel_err(gpu.params[1], ref.params[1]) <= TOL
        for c in (5, 6):
            assert rel_err(gpu.caches[c], ref.caches[c]) <= TOL
    gpu.close()
