"""Random tensors (`rand`, parser.nim:732-736) and dropout (dnn.nim:96-100).  The reference draws them
from Nim's global generator on the host — nothing to pin bit for bit — so the tests check the
distribution and, for parity, hand the oracle the very numbers the GPU drew."""
import ctypes

import numpy as np
import pytest

import refcases
from conftest import TOL, rel_err
from exprgrad_amd import dsl, layers


def graphs(prob=0.25):
    x = dsl.input("x")
    d = layers.dropout(x, prob)
    it = dsl.iters("it")
    loss = dsl.Fun()
    loss[0] += d.raw[it] * d.raw[it]
    return [d.target("d"), loss.target("loss").backwards().grad(x).target("g")]


def random_tensor_id(program):
    return [i for i, t in enumerate(program.tensors, 1) if t["kind"] == "random"][0]


def test_oracle_dropout_keeps_and_rescales():
    from oracle import kd
    m = kd.Model(refcases.program_text(graphs()))
    a = np.random.default_rng(1).random((400, 50), dtype=np.float32) + 1
    out = m.call("d", {"x": a})
    keep = out != 0
    assert abs(keep.mean() - 0.75) < 0.02
    assert np.allclose(out[keep], a[keep] / np.float32(0.75), rtol=1e-6)
    out2 = m.call("d", {"x": a})
    assert not np.array_equal(out2 != 0, keep)         # a fresh mask per call (model.nim:286-294)
    m.random_override[2] = np.full(a.shape, 0.5, np.float32)
    assert np.allclose(m.call("d", {"x": a}), a / np.float32(0.75), rtol=1e-6)


@pytest.mark.gpu
def test_gpu_uniform_fill(gpu_ctx):
    from exprgrad_amd import _lib
    import exprgrad_amd as eg
    n = 1 << 20
    out = gpu_ctx.allocTensor((n,))
    state = gpu_ctx.allocTensor((4,))                   # 2 x uint64
    state.write(np.array([12345, 0, 0, 0], dtype=np.uint32).view(np.float32))

    def draw(stream):
        _lib.call("eg_fill_uniform", gpu_ctx.handle, n, -2.0, 3.0, ctypes.c_void_p(state.ptr), stream, ctypes.c_void_p(out.ptr))
        return out.read()

    a = draw(7)
    assert a.min() >= -2.0 and a.max() < 3.0
    assert abs(a.mean() - 0.5) < 0.01 and abs(a.std() - 5 / np.sqrt(12)) < 0.01
    hist, _ = np.histogram(a, bins=50, range=(-2, 3))
    assert hist.min() > 0.9 * n / 50 and hist.max() < 1.1 * n / 50
    assert abs(np.corrcoef(a[:-1], a[1:])[0, 1]) < 0.005            # neighbours are not correlated
    assert np.array_equal(draw(7), a)                               # same state, same stream: same numbers
    assert not np.array_equal(draw(8), a)                           # another stream (tensor)
    _lib.call("eg_rng_advance", gpu_ctx.handle, ctypes.c_void_p(state.ptr))
    b = draw(7)
    assert not np.array_equal(b, a) and abs(np.corrcoef(a, b)[0, 1]) < 0.005


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(64, 10), (300, 700)])
def test_gpu_dropout_matches_the_oracle_on_the_same_mask(gpu_ctx, shape):
    from oracle import kd
    from exprgrad_amd import model as egm
    gpu = egm.compile(*graphs(), gpu=gpu_ctx)
    ref = kd.Model(refcases.program_text(graphs()))
    rid = random_tensor_id(gpu.program)
    a = np.random.default_rng(2).random(shape, dtype=np.float32) + 1
    seen = []
    for target in ("d", "g", "d"):
        got = gpu.call(target, {"x": a})
        mask = gpu.read_tensor(target, rid)
        assert mask.shape == a.shape and 0.0 <= mask.min() and mask.max() < 1.0
        ref.random_override[rid] = mask
        assert rel_err(got, ref.call(target, {"x": a})) <= TOL
        seen.append(mask)
    assert not np.array_equal(seen[0], seen[2])                     # redrawn on every call, graph replay included
    assert abs((seen[0] >= 0.25).mean() - 0.75) < 0.03
    # same seed, same call sequence: same masks
    gpu.set_seed(99)
    first = (gpu.call("d", {"x": a}), gpu.call("d", {"x": a}))
    gpu.set_seed(99)
    again = (gpu.call("d", {"x": a}), gpu.call("d", {"x": a}))
    assert np.array_equal(first[0], again[0]) and np.array_equal(first[1], again[1])
    assert not np.array_equal(first[0], first[1])
    gpu.close()
