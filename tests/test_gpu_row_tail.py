"""Row tails (round 5, rowfuse.hpp / plan_groups.cpp fuse_row_tails): a row group of several blocks lets the LAST block to
arrive fold the partial rows (in the order of row_finalize_kernel) and go on with the small group that follows — the
whole XOR train step (examples/xor_from_scratch: 15 per-sample kernels, bias / weight gradient totals, four
gradientDescent kernels, xor_from_scratch.nim:19-31) is one launch instead of three dependent ones.

The terms of every sum stay what they were; with at most 64 blocks a launch is what the three-launch plan
(EG_NO_ROW_TAIL=1 when the plan is made) computes to the bit, with more samples than 64 x 256 a thread's totals span
several samples (another order: 1e-5, stated below).  One range (MODE 2) against backward | update (MODE 1 + the update
launch: what the data-parallel step does) is held to the bit always.  Many steps: the ticket counter must be back at
zero for every launch and the hand-off between blocks must never read a stale partial row — a stale row of the
PREVIOUS step is a plausible number, so the trajectory is compared step by step."""
import numpy as np
import pytest

import refcases
from conftest import debug_toggles_active
from exprgrad_amd import model as egm

pytestmark = pytest.mark.gpu


def xor_data(batch, seed):
    rng = np.random.default_rng(seed)
    x = rng.integers(0, 2, size=(batch, 2)).astype(np.float32)
    return x, (x[:, :1] != x[:, 1:]).astype(np.float32)


def build(gpu_ctx, graphs, monkeypatch, tails, seed=5):
    if tails:
        monkeypatch.delenv("EG_NO_ROW_TAIL", raising=False)
    else:
        monkeypatch.setenv("EG_NO_ROW_TAIL", "1")
    m = egm.compile(*graphs(), gpu=gpu_ctx)
    rng = np.random.default_rng(seed)
    for tid in m.params.ids():
        m.params[tid] = (rng.random(m.params[tid].shape, dtype=np.float32) * 0.2 - 0.1).astype(np.float32)
    return m


@pytest.mark.parametrize("batch", [65536, 3000, 257, 16384, 70000])
def test_xor_step_in_one_launch(gpu_ctx, monkeypatch, batch):
    """examples/xor with the layer library (dense / leakyRelu / sigmoid / mse, examples/xor/xor.nim:19-28): a batch-mean
    loss, so 40 steps stay finite at any batch (the from-scratch example's sum-of-squares loss diverges beyond one step
    at 65 536 samples: test_cfg3 runs that one)."""
    x, y = xor_data(batch, batch)
    args = {"x": x, "y": y}
    fused = build(gpu_ctx, refcases.xor_layers, monkeypatch, True)
    fused.apply("train", args)          # the plan is made here, with the switch as it is now
    plain = build(gpu_ctx, refcases.xor_layers, monkeypatch, False)
    plain.apply("train", args)
    split = build(gpu_ctx, refcases.xor_layers, monkeypatch, True)
    split.run_backward("train", args)
    split.run_update("train")
    if not debug_toggles_active():
        assert "goes on with launch" in fused.launch_plan("train"), fused.launch_plan("train")
        assert "goes on with launch" not in plain.launch_plan("train")
    same_order = batch <= 64 * 256
    for step in range(40):                   # eager, captured, replayed ... the counter returns to zero every time
        for tid in fused.params.ids():
            a, b, c = fused.params[tid], plain.params[tid], split.params[tid]
            assert np.all(np.isfinite(a)), (step, tid)
            assert np.array_equal(a, c), (step, tid, "one range vs backward | update")
            if same_order:
                assert np.array_equal(a, b), (step, tid, "one launch vs three")
            else:       # (from identical state: the plain model follows the fused one, so a step's difference is that step's)
                assert np.max(np.abs(a - b)) <= 1e-5 * max(np.max(np.abs(b)), 1e-30), (step, tid)
                plain.params[tid] = a
        fused.apply("train", args)
        plain.apply("train", args)
        split.run_backward("train", args)
        split.run_update("train")
    for m in (fused, plain, split):
        m.close()


def test_xor_from_scratch_first_step_at_the_config_batch(gpu_ctx, monkeypatch):
    """BASELINE configs[2] itself (sum-of-squares loss, batch 65 536): the one step that stays finite, one launch against
    three (four samples per thread: another summation order, 1e-5 of the update)."""
    x, y = xor_data(65536, 3)
    args = {"x": x, "y": y}
    fused = build(gpu_ctx, refcases.xor_from_scratch, monkeypatch, True)
    plain = build(gpu_ctx, refcases.xor_from_scratch, monkeypatch, False)
    before = {tid: fused.params[tid].copy() for tid in fused.params.ids()}
    fused.apply("train", args)
    plain.apply("train", args)
    for tid in before:
        da, db = fused.params[tid] - before[tid], plain.params[tid] - before[tid]
        assert np.all(np.isfinite(da))
        assert np.max(np.abs(da - db)) <= 1e-5 * np.max(np.abs(db)), tid
    fused.close()
    plain.close()


def test_dense_softmax_row_group_folds_its_partial_rows_itself(gpu_ctx, monkeypatch):
    """The softmax / cross-entropy chain of the dense net: its row group has no small group behind it (the weight
    gradients follow), so only the fold moves into the kernel (MODE 1).  Gradients and parameters to the bit."""
    rng = np.random.default_rng(8)
    x = rng.random((5000, 12), dtype=np.float32)
    y = np.eye(5, dtype=np.float32)[rng.integers(0, 5, size=5000)]
    graphs = lambda: refcases.dense_softmax_net(n_in=12, n_hidden=8, n_out=5, rate=0.5)
    a = build(gpu_ctx, graphs, monkeypatch, True)
    a.apply("train", {"x": x, "y": y})
    b = build(gpu_ctx, graphs, monkeypatch, False)
    b.apply("train", {"x": x, "y": y})
    for step in range(10):
        for tid in a.params.ids():
            assert np.array_equal(a.params[tid], b.params[tid]), (step, tid)
        a.apply("train", {"x": x, "y": y})
        b.apply("train", {"x": x, "y": y})
    a.close()
    b.close()


def test_dense_step_hands_the_fold_to_the_side_lane(gpu_ctx, monkeypatch):
    """Round 6 (plan_overlap.cpp, Overlap::deferred_row): with a long contraction behind it (the first layer's weight
    gradient: 2 * 784 * 64 * 16384 = 1.6 GFLOP is not enough, 784 x 512 x 16384 is) the row group's fold runs as the side lane's
    first launch instead of the row kernel's tail.  Same kernel order of additions (row_finalize_kernel): parameters to
    the bit against EG_NO_DEFERRED_FOLD=1, eager and replayed."""
    rng = np.random.default_rng(9)
    batch = 16384
    x = rng.random((batch, 784), dtype=np.float32)
    y = np.eye(10, dtype=np.float32)[rng.integers(0, 10, size=batch)]
    graphs = lambda: refcases.dense_softmax_net()
    monkeypatch.delenv("EG_NO_DEFERRED_FOLD", raising=False)
    a = build(gpu_ctx, graphs, monkeypatch, True)
    a.apply("train", {"x": x, "y": y})
    if not debug_toggles_active():
        assert "its fold runs beside launch" in a.launch_plan("train"), a.launch_plan("train")
    monkeypatch.setenv("EG_NO_DEFERRED_FOLD", "1")
    b = build(gpu_ctx, graphs, monkeypatch, True)
    b.apply("train", {"x": x, "y": y})
    assert "its fold runs beside launch" not in b.launch_plan("train")
    for step in range(6):
        for tid in a.params.ids():
            assert np.array_equal(a.params[tid], b.params[tid]), (step, tid)
        a.apply("train", {"x": x, "y": y})
        b.apply("train", {"x": x, "y": y})
    a.close()
    b.close()
