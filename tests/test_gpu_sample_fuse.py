"""Sample groups (round 5, rowfuse.hpp): at a small batch the forward and backward pass of a network run as ONE generated
kernel with one block per sample, the parameter gradients as per-sample slab rows folded by one slab_sum launch — the
reference's flagship network (examples/fashion_mnist/fashion_mnist.nim:39-57: conv2 / leakyRelu / maxpool2 twice, dense,
softmax, crossEntropy, adam) at its default batch of 32 (model.nim:413) goes from 16 dependent launches to 4.

Compared with the plan without sample groups (EG_NO_SAMPLE_FUSE=1 when the plan is made: the hand-written tiny
convolutions, row group, small contractions) step by step FROM IDENTICAL STATE, and with the oracle through the
ordinary three-way check; the two plans sum in different orders (per-sample contributions, then a tree over the samples),
so 1e-5 of the gradient, not bits."""
import numpy as np
import pytest

import refcases
from conftest import debug_toggles_active
from exprgrad_amd import examples
from exprgrad_amd import model as egm
from parity import Trio

pytestmark = pytest.mark.gpu


def build(gpu_ctx, graphs, monkeypatch, fused, seed=2):
    if fused:
        monkeypatch.delenv("EG_NO_SAMPLE_FUSE", raising=False)
    else:
        monkeypatch.setenv("EG_NO_SAMPLE_FUSE", "1")
    m = egm.compile(*graphs(), gpu=gpu_ctx)
    rng = np.random.default_rng(seed)
    for tid in m.params.ids():
        m.params[tid] = (rng.random(m.params[tid].shape, dtype=np.float32) * 0.2 - 0.1).astype(np.float32)
    return m


def data(batch, seed):
    rng = np.random.default_rng(seed)
    return {"x": rng.random((batch, 784), dtype=np.float32), "y": np.eye(10, dtype=np.float32)[rng.integers(0, 10, batch)]}


@pytest.mark.parametrize("batch", [32, 5, 64])
def test_fashion_mnist_step_as_one_kernel_matches_the_launch_chain(gpu_ctx, monkeypatch, batch):
    args = data(batch, batch)
    fused = build(gpu_ctx, examples.fashion_mnist_net, monkeypatch, True)
    fused.epoch = 1
    fused.apply("fit", args)             # (the plan is made here: the switch is read when a plan is made)
    plain = build(gpu_ctx, examples.fashion_mnist_net, monkeypatch, False)
    plain.epoch = 1
    plain.apply("fit", args)
    if not debug_toggles_active():
        assert "sample-fused" in fused.launch_plan("fit"), fused.launch_plan("fit")
        assert "sample-fused" not in plain.launch_plan("fit")
        assert fused.launch_plan("fit").count("\n") <= 5, fused.launch_plan("fit")
    from oracle import kd
    pairs = kd.Model(refcases.program_text(examples.fashion_mnist_net())).param_grads("fit")
    assert pairs
    for step in range(6):
        # The gradients of the step just taken, from identical state.  (The PARAMETERS are no fair comparison under adam:
        # at step 1 an update is eta * g / (|g| + 1e-8) — a gradient element that is rounding noise around zero flips the
        # sign of a full-size update with the summation order.)
        for ptid, gtid in pairs:
            ga, gb = fused.read_tensor("fit", gtid), plain.read_tensor("fit", gtid)
            assert np.all(np.isfinite(ga)), (step, ptid)
            assert np.max(np.abs(ga - gb)) <= 1e-5 * max(np.max(np.abs(gb)), 1e-30), (step, ptid, float(np.max(np.abs(ga - gb))), float(np.max(np.abs(gb))))
        for tid in fused.params.ids():
            plain.params[tid] = fused.params[tid]
        for cid in fused.caches.ids():
            plain.caches[cid] = fused.caches[cid]
        fused.epoch = plain.epoch = step + 2
        fused.apply("fit", args)
        plain.apply("fit", args)
    fused.close()
    plain.close()


def test_fashion_mnist_small_batch_step_against_the_oracle(gpu_ctx, monkeypatch):
    """The same step through the three-way check (backend | oracle | float64 shadow), gradients first, then the optimizer
    on the backend's gradients (tests/parity.py)."""
    monkeypatch.delenv("EG_NO_SAMPLE_FUSE", raising=False)
    t = Trio(gpu_ctx, examples.fashion_mnist_net)
    t.init_params(np.random.default_rng(4), -0.1, 0.1)
    args = data(8, 1)
    for step in range(1, 4):
        t.set_epoch(step)
        t.step("fit", args, n=8 * 24 * 24)
    if not debug_toggles_active():
        assert "sample-fused" in t.gpu.launch_plan("fit")
    t.close()


def test_dense_net_small_batch(gpu_ctx, monkeypatch):
    """dense -> relu -> dense -> softmax -> crossEntropy -> gradientDescent at batch 16: contractions, bias gradients and
    the softmax chain inside one sample group."""
    monkeypatch.delenv("EG_NO_SAMPLE_FUSE", raising=False)
    graphs = lambda: refcases.dense_softmax_net(n_in=20, n_hidden=12, n_out=5, rate=0.1)
    t = Trio(gpu_ctx, graphs)
    t.init_params(np.random.default_rng(9))
    rng = np.random.default_rng(10)
    args = {"x": rng.random((16, 20), dtype=np.float32), "y": np.eye(5, dtype=np.float32)[rng.integers(0, 5, 16)]}
    for _ in range(3):
        t.step("train", args, n=16)
    t.close()


def test_backward_update_split_takes_the_slab_pass(gpu_ctx, monkeypatch):
    """What the data-parallel step does: the backward range alone (the exchange needs the batch totals in the gradient
    bucket, so the sample kernel is followed by its slab_sum launch), then the update range — against Model.apply, where
    the optimizer's map group adds the slab rows up itself.  Two orders of the same 8 terms per element: 1e-6 of the
    gradient; and the bucket must hold the totals after the backward range."""
    from oracle import kd
    monkeypatch.delenv("EG_NO_SAMPLE_FUSE", raising=False)
    args = data(8, 21)
    whole = build(gpu_ctx, examples.fashion_mnist_net, monkeypatch, True)
    split = build(gpu_ctx, examples.fashion_mnist_net, monkeypatch, True)
    pairs = kd.Model(refcases.program_text(examples.fashion_mnist_net())).param_grads("fit")
    for step in range(1, 5):
        whole.epoch = split.epoch = step
        whole.apply("fit", args)
        split.run_backward("fit", args)
        grads = {gt: split.read_tensor("fit", gt) for _, gt in pairs}          # complete BEFORE the update range runs
        split.run_update("fit")
        for _, gt in pairs:
            gw = whole.read_tensor("fit", gt)
            assert np.all(np.isfinite(gw))
            assert np.max(np.abs(gw - grads[gt])) <= 1e-6 * max(np.max(np.abs(gw)), 1e-30), (step, gt)
            assert np.array_equal(split.read_tensor("fit", gt), grads[gt])       # the update range leaves them alone
        for tid in whole.params.ids():
            split.params[tid] = whole.params[tid]
        for cid in whole.caches.ids():
            split.caches[cid] = whole.caches[cid]
    if not debug_toggles_active():
        assert "folded by launch" in whole.launch_plan("fit")
    whole.close()
    split.close()


def test_intermediates_of_a_sample_group_are_refused_unless_values_are_kept(gpu_ctx, monkeypatch):
    """A tensor that lives in the blocks' LDS was never stored: read_tensor says so instead of returning stale arena
    memory; eg_model_keep_values(model, 1) re-plans with every value in memory, and the two plans agree on it."""
    from exprgrad_amd._lib import GpuError
    from oracle import kd
    monkeypatch.delenv("EG_NO_SAMPLE_FUSE", raising=False)
    args = data(8, 2)
    m = build(gpu_ctx, examples.fashion_mnist_net, monkeypatch, True)
    ref = kd.Model(refcases.program_text(examples.fashion_mnist_net()))
    for tid in m.params.ids():
        ref.params[tid][...] = m.params[tid]
    m.epoch = ref.epoch = 1
    m.apply("fit", args)
    if debug_toggles_active():
        m.close()
        return
    conv1_out = 14                      # t14 = conv2(reshape(x), filters) of the first layer (the program text numbers it)
    with pytest.raises(GpuError, match="eg_model_keep_values"):
        m.read_tensor("fit", conv1_out)
    m.keep_values(True)
    for tid in m.params.ids():
        m.params[tid] = ref.params[tid]          # the same step again, from the same state
    for cid in m.caches.ids():
        m.caches[cid] = np.zeros_like(m.caches[cid])
    m.apply("fit", args)
    got = m.read_tensor("fit", conv1_out)
    x = args["x"].reshape(8, 28, 28, 1)
    flt = ref.params[1]
    from parity import exact_conv2
    want = exact_conv2(x, flt)
    assert got.shape == want.shape or got.size == want.size
    assert np.max(np.abs(got.reshape(want.shape) - want)) <= 1e-5 * np.max(np.abs(want))
    m.close()
