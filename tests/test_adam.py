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
    from oracle import kd
    from exprgrad_amd import model as egm
    gpu = egm.compile(*xor_adam(), gpu=gpu_ctx)
    ref = kd.Model(dsl.to_program(*xor_adam()).to_text(), fast_contractions=False)
    for tid, v in init(ref).items():
        ref.params[tid][...] = v
        gpu.params[tid] = v
    assert sorted(gpu.caches.ids()) == sorted(ref.caches)
    for step in range(1, 31):
        gpu.fit("train", {"x": X, "y": Y}, batch_size=4)       # bumps Model.epoch, then one batch
        ref.epoch = step
        ref.apply("train", {"x": X, "y": Y})
        assert gpu.epoch == step
    for tid in sorted(ref.params):
        assert rel_err(gpu.params[tid], ref.params[tid]) <= 1e-4, tid      # 30 compounding steps
    for tid in sorted(ref.caches):
        assert rel_err(gpu.caches[tid], ref.caches[tid]) <= 1e-4, tid
    # one step from identical state: the per-step bound
    for tid in sorted(ref.params):
        gpu.params[tid] = ref.params[tid]
    for tid in sorted(ref.caches):
        gpu.caches[tid] = ref.caches[tid]
    gpu.fit("train", {"x": X, "y": Y}, batch_size=4)
    ref.epoch = 31
    ref.apply("train", {"x": X, "y": Y})
    for tid in sorted(ref.params):
        assert rel_err(gpu.params[tid], ref.params[tid]) <= TOL, tid
    for _ in range(400):
        gpu.fit("train", {"x": X, "y": Y}, batch_size=4)
    assert float(gpu.call("loss", {"x": X, "y": Y})[0]) < 1e-2
    gpu.close()


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
