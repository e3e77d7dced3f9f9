"""adam (layers/base.nim:40-53): cache tensors (adam.m / adam.v), epoch(), pow/sqrt — SURVEY.md §8(f) row f2.

The reference's flagship example (examples/fashion_mnist/fashion_mnist.nim:39-57) trains with it
through Model.fit, which bumps Model.epoch once per call (model.nim:436).
"""
import numpy as np
import pytest

from conftest import TOL, rel_err
from exprgrad_amd import dsl, layers

X = np.array([[0, 0], [0, 1], [1, 0], [1, 1]], dtype=np.float32)
Y = np.array([[0], [1], [1], [0]], dtype=np.float32)


def xor_adam(eta=0.05):
    net = layers.dense(dsl.input("x"), 2, 4)
    net = layers.leaky_relu(net)
    net = layers.dense(net, 4, 1)
    net = layers.sigmoid(net).target("predict")
    net = layers.mse(net, dsl.input("y")).target("loss")
    return [net.backprop(layers.adam(eta=eta)).target("train")]


def init(model, seed=1):
    rng = np.random.default_rng(seed)
    vals = {}
    for tid in sorted(model.params):
        shape = model.params[tid].shape
        vals[tid] = (rng.random(shape, dtype=np.float32) * 0.2 - 0.1).astype(np.float32)
    return vals


def test_oracle_adam_converges(refcpu):
    from oracle import kd
    m = kd.Model(dsl.to_program(*xor_adam()).to_text(), fast_contractions=False)
    for tid, v in init(m).items():
        m.params[tid][...] = v
    # 4 parameters x (m, v, update) + forward/backward: caches are state, never eliminated
    assert len(m.caches) == 8
    for epoch in range(1, 401):
        m.epoch = epoch                      # what fit does (model.nim:436)
        m.apply("train", {"x": X, "y": Y})
    assert float(m.call("loss", {"x": X, "y": Y})[0]) < 1e-2
    assert all(np.any(c != 0) for c in m.caches.values())


@pytest.mark.gpu
def test_gpu_adam_matches_oracle_through_fit(gpu_ctx):
    from parity import Trio
    t = Trio(gpu_ctx, xor_adam)
    t.ref.fast = t.exact.fast = False
    gpu, ref = t.gpu, t.ref
    for tid, v in init(ref).items():
        t.set_param(tid, v)
    assert sorted(gpu.caches.ids()) == sorted(ref.caches)
    # every step from the backend's own state (parameters and both moments): gradients, moments and
    # parameters within 1e-5 of the float64 shadow at each of the 30 steps
    for step in range(1, 31):
        t.set_epoch(step)
        t.step("train", {"x": X, "y": Y}, n=4)
    # the same 30 steps through fit (bumps Model.epoch, then one batch) reproduce the apply() path bit for bit
    from exprgrad_amd import model as egm
    fitted = egm.compile(*xor_adam(), gpu=gpu_ctx)
    for tid, v in init(ref).items():
        fitted.params[tid] = v
    for step in range(1, 31):
        fitted.fit("train", {"x": X, "y": Y}, batch_size=4)
        assert fitted.epoch == step
    for tid in gpu.params.ids():
        assert np.array_equal(fitted.params[tid], gpu.params[tid]), tid
    for tid in gpu.caches.ids():
        assert np.array_equal(fitted.caches[tid], gpu.caches[tid]), tid
    for _ in range(400):
        fitted.fit("train", {"x": X, "y": Y}, batch_size=4)
    assert float(fitted.call("loss", {"x": X, "y": Y})[0]) < 1e-2
    fitted.close()
    t.close()


@pytest.mark.gpu
def test_gpu_save_load_resumes_identically(gpu_ctx, tmp_path):
    """save / loadModel (io/serialize.nim:344-379): device state is flushed,
    and a reloaded model continues bit-identically (params, adam caches and epoch all restored)."""
    from exprgrad_amd import model as egm
    a = egm.compile(*xor_adam(), gpu=gpu_ctx)
    rng = np.random.default_rng(1)
    for tid in a.params.ids():
        a.params[tid] = (rng.random(a.params[tid].shape, dtype=np.float32) * 0.2 - 0.1).astype(np.float32)
    for _ in range(10):
        a.fit("train", {"x": X, "y": Y}, batch_size=4)
    path = tmp_path / "model.npz"
    a.save(path)
    b = egm.load_model(path, gpu=gpu_ctx)
    assert b.epoch == a.epoch == 10
    for _ in range(5):
        a.fit("train", {"x": X, "y": Y}, batch_size=4)
        b.fit("train", {"x": X, "y": Y}, batch_size=4)
    for tid in a.params.ids():
        assert np.array_equal(a.params[tid], b.params[tid])
    for tid in a.caches.ids():
        assert np.array_equal(a.caches[tid], b.caches[tid])
    assert np.array_equal(a.call("predict", {"x": X}), b.call("predict", {"x": X}))
    a.close()
    b.close()
