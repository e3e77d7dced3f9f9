"""Model.fit (model.nim:413-454) through eg_model_fit: the data set is uploaded once (piecewise, on
a second stream) and every mini-batch is a segment copy + the captured launch sequence.  Compared
with the oracle stepping through the same batches (viewFirst slices, tail dropped, epoch bumped
once per call), and with the GPU's own apply() loop bit for bit."""
import numpy as np
import pytest

import refcases
from conftest import TOL, rel_err
from exprgrad_amd import examples, layers, dsl
from exprgrad_amd import model as egm
from exprgrad_amd._lib import RuntimeErrorEG
from parity import Trio

pytestmark = pytest.mark.gpu


def oracle(graphs, threads=4):
    from oracle import kd
    return kd.Model(refcases.program_text(graphs), threads=threads)


def shadow(graphs):
    from oracle import kd
    return kd.Model(refcases.program_text(graphs), shadow=True)


def dense_graphs(optim):
    net = layers.tanh(layers.dense(dsl.input("x"), 24, 16))
    net = layers.sigmoid(layers.dense(net, 16, 5)).target("predict")
    net = layers.mse(net, dsl.input("y")).target("loss")
    opt = layers.adam(0.01) if optim == "adam" else layers.gradient_descent(0.05)
    return [net.backprop(opt).target("train")]


def same_start(models, seed):
    rng = np.random.default_rng(seed)
    ids = models[0].params.ids() if hasattr(models[0].params, "ids") else sorted(models[0].params)
    for tid in ids:
        shape = models[0].params[tid].shape
        v = (rng.random(shape, dtype=np.float32) - 0.5).astype(np.float32)
        for m in models:
            if hasattr(m.params, "ids"):
                m.params[tid] = v
            else:
                m.params[tid][...] = v


def oracle_fit(ref, target, x, y, batch):
    ref.epoch += 1
    for b in range(x.shape[0] // batch):
        ref.apply(target, {"x": x[b * batch:(b + 1) * batch], "y": y[b * batch:(b + 1) * batch]})


@pytest.mark.parametrize("optim", ["sgd", "adam"])
@pytest.mark.parametrize("rows,batch", [(64, 8), (77, 16), (10, 16)])
def test_fit_matches_the_oracle_batch_by_batch(gpu_ctx, optim, rows, batch):
    gpu = egm.compile(*dense_graphs(optim), gpu=gpu_ctx)
    looped = egm.compile(*dense_graphs(optim), gpu=gpu_ctx)
    ref = oracle(dense_graphs(optim))
    exact = shadow(dense_graphs(optim))
    same_start((gpu, looped, ref, exact), seed=rows)
    rng = np.random.default_rng(batch)
    x = rng.random((rows, 24), dtype=np.float32)
    y = rng.random((rows, 5), dtype=np.float32)
    for _ in range(3):
        gpu.fit("train", {"x": x, "y": y}, batch_size=batch)
        oracle_fit(ref, "train", x, y, batch)
        oracle_fit(exact, "train", x, y, batch)
        looped.epoch = looped.epoch + 1
        for b in range(rows // batch):
            looped.apply("train", {"x": x[b * batch:(b + 1) * batch], "y": y[b * batch:(b + 1) * batch]})
    assert gpu.epoch == ref.epoch == 3
    for tid in gpu.params.ids():
        assert np.array_equal(gpu.params[tid], looped.params[tid]), tid
        # the trajectory over all batches of three epochs, backend and oracle each against the float64 shadow
        # run through the same batches (tests/parity.py: 1e-5, or at most twice the oracle's own distance)
        Trio.check(gpu.params[tid], ref.params[tid], exact.params[tid], what=f"parameter {tid} after three epochs")
    for m in (gpu, looped):
        m.close()


def test_fit_in_small_segments_and_pieces(gpu_ctx, monkeypatch):
    """The data set does not fit the device copy at once: segments are reused, pieces overlap."""
    rows, batch = 96, 8
    rng = np.random.default_rng(2)
    x = rng.random((rows, 24), dtype=np.float32)
    y = rng.random((rows, 5), dtype=np.float32)
    whole = egm.compile(*dense_graphs("sgd"), gpu=gpu_ctx)
    pieces = egm.compile(*dense_graphs("sgd"), gpu=gpu_ctx)
    same_start((whole, pieces), seed=9)
    whole.fit("train", {"x": x, "y": y}, batch_size=batch)
    monkeypatch.setenv("EG_FIT_SEGMENT_BYTES", str(3 * batch * 29 * 4))   # three batches per segment
    monkeypatch.setenv("EG_FIT_PIECE_BYTES", str(2 * batch * 29 * 4))     # two batches per upload
    pieces.fit("train", {"x": x, "y": y}, batch_size=batch)
    for tid in whole.params.ids():
        assert np.array_equal(whole.params[tid], pieces.params[tid]), tid
    whole.close()
    pieces.close()


def test_fit_calls_with_different_data_do_not_overtake_each_other(gpu_ctx):
    """Every epoch gets another permutation of the rows (what a training loop does).  fit returns when
    its uploads are complete, not its kernels: the next call's upload must wait for the batches still
    queued, or they read rows of the following epoch.  Many small batches keep the queue
    deep; compared with the oracle stepping through the same permutations."""
    rows, batch = 4096, 16
    gpu = egm.compile(*dense_graphs("sgd"), gpu=gpu_ctx)
    ref = oracle(dense_graphs("sgd"))
    exact = shadow(dense_graphs("sgd"))
    same_start((gpu, ref, exact), seed=3)
    rng = np.random.default_rng(11)
    x = rng.random((rows, 24), dtype=np.float32)
    y = rng.random((rows, 5), dtype=np.float32)
    for epoch in range(4):
        perm = rng.permutation(rows)
        xe, ye = np.ascontiguousarray(x[perm]), np.ascontiguousarray(y[perm])
        gpu.fit("train", {"x": xe, "y": ye}, batch_size=batch)
        xe[...] = -1e6   # the caller may reuse its arrays as soon as fit returns
        ye[...] = 1e6
        oracle_fit(ref, "train", x[perm], y[perm], batch)
        oracle_fit(exact, "train", x[perm], y[perm], batch)
    for tid in gpu.params.ids():
        # 1024 sequential steps: both float32 trajectories drift from the float64 one; a batch read from the wrong
        # epoch would be orders of magnitude beyond either
        Trio.check(gpu.params[tid], ref.params[tid], exact.params[tid], what=f"parameter {tid} after four epochs")
    gpu.close()


def test_fit_reads_device_resident_data_in_place(gpu_ctx):
    import torch
    rows, batch = 40, 8
    rng = np.random.default_rng(4)
    x = rng.random((rows, 24), dtype=np.float32)
    y = rng.random((rows, 5), dtype=np.float32)
    host = egm.compile(*dense_graphs("adam"), gpu=gpu_ctx)
    dev = egm.compile(*dense_graphs("adam"), gpu=gpu_ctx)
    same_start((host, dev), seed=1)
    host.fit("train", {"x": x, "y": y}, batch_size=batch)
    dev.fit("train", {"x": torch.from_numpy(x).cuda(), "y": y}, batch_size=batch)   # mixed: one device, one host
    torch.cuda.synchronize()
    for tid in host.params.ids():
        assert np.array_equal(host.params[tid], dev.params[tid]), tid
    host.close()
    dev.close()


def test_fit_errors_follow_the_reference(gpu_ctx):
    gpu = egm.compile(*dense_graphs("sgd"), gpu=gpu_ctx)
    with pytest.raises(RuntimeErrorEG, match="requires at least one input tensor"):   # model.nim:417-421
        gpu.fit("train", {}, batch_size=4)
    x = np.zeros((8, 24), np.float32)
    with pytest.raises(RuntimeErrorEG, match="is not a target of the model"):
        gpu.fit("nope", {"x": x, "y": np.zeros((8, 5), np.float32)}, batch_size=4)
    with pytest.raises(RuntimeErrorEG, match="is not an input to the model"):
        gpu.fit("train", {"x": x, "z": x}, batch_size=4)
    with pytest.raises(Exception, match="fewer rows"):
        gpu.fit("train", {"x": x, "y": np.zeros((4, 5), np.float32)}, batch_size=4)
    assert gpu.epoch == 0
    gpu.close()


def test_fit_of_the_fashion_mnist_network(gpu_ctx):
    """examples/fashion_mnist/fashion_mnist.nim:59-66 in small: fit over several batches, then the loss fell."""
    graphs = lambda: examples.fashion_mnist_net(size=12, f1=4, f2=8, eta=0.02)
    gpu = egm.compile(*graphs(), gpu=gpu_ctx)
    ref = oracle(graphs(), threads=8)
    same_start((gpu, ref), seed=0)
    for m in (gpu, ref):
        for tid in (m.params.ids() if hasattr(m.params, "ids") else sorted(m.params)):
            v = np.asarray(m.params[tid]) * np.float32(0.4)
            if hasattr(m.params, "ids"):
                m.params[tid] = v
            else:
                m.params[tid][...] = v
    rng = np.random.default_rng(0)
    x = rng.random((40, 144), dtype=np.float32)
    y = np.eye(10, dtype=np.float32)[rng.integers(0, 10, 40)]
    first = float(gpu.call("loss", {"x": x, "y": y}).sum())
    for _ in range(4):
        gpu.fit("fit", {"x": x, "y": y}, batch_size=8)
        oracle_fit(ref, "fit", x, y, 8)
    assert float(gpu.call("loss", {"x": x, "y": y}).sum()) < first
    # 20 adam steps through max-pooling and leaky units: single-bit differences flip maxima, so the two
    # float32 trajectories separate (the per-step parity of this network is tests/test_pooling_reshape.py, fit ==
    # apply loop bit for bit is the first test of this file); what remains to say here: both learn
    assert float(ref.call("loss", {"x": x, "y": y}).sum()) < first
    gpu.close()


def test_small_batch_shortcuts_change_no_bit(gpu_ctx, monkeypatch):
    """Round 4: at a small batch two independent tiny contractions next to each other (a dense layer's weight gradient and
    input gradient) run as one launch, and a row group of one block adds its batch totals to the destinations itself
    instead of through row_finalize.  Both shortcuts compute every value exactly as the long way round: a model built
    with EG_NO_SMALL_PAIR=1 EG_NO_ROW_DIRECT=1 ends three training steps with the same bits, and both are at the
    oracle's values within the tolerance."""
    rng = np.random.default_rng(3)
    x = rng.random((32, 24), dtype=np.float32)
    y = rng.random((32, 5), dtype=np.float32)
    results = []
    for env in ({}, {"EG_NO_SMALL_PAIR": "1", "EG_NO_ROW_DIRECT": "1"}):
        for k in ("EG_NO_SMALL_PAIR", "EG_NO_ROW_DIRECT"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        m = egm.compile(*dense_graphs("adam"), gpu=gpu_ctx)
        same_start([m], 9)
        m.epoch = 1          # (adam divides by 1 - beta^epoch)
        for _ in range(3):
            m.apply("train", {"x": x, "y": y})
        results.append({tid: np.array(m.params[tid]) for tid in m.params.ids()})
    for tid in results[0]:
        assert np.isfinite(results[0][tid]).all(), tid
        assert np.array_equal(results[0][tid], results[1][tid]), tid
    ref = oracle(dense_graphs("adam"))
    same_start([ref], 9)
    ref.epoch = 1
    for _ in range(3):
        ref.apply("train", {"x": x, "y": y})
    for tid in results[0]:
        assert rel_err(results[0][tid], np.asarray(ref.params[tid])) <= TOL, tid


@pytest.mark.parametrize("optim", ["sgd", "adam"])
def test_fit_in_groups_of_batches_changes_no_bit(gpu_ctx, monkeypatch, optim):
    """Round 4: at a small batch size eg_model_fit captures 16 consecutive batches (segment copy + launch sequence each) as
    one graph and re-points the copy nodes per launch.  200 batches (two single ones, twelve groups, a tail of six) over
    two epochs against the same fit with every batch launched by itself (EG_FIT_GROUP=1) and with groups of 5."""
    rows, batch = 200 * 8, 8
    rng = np.random.default_rng(4)
    x = rng.random((rows, 24), dtype=np.float32)
    y = rng.random((rows, 5), dtype=np.float32)
    results = []
    for group in (None, "1", "5"):
        monkeypatch.delenv("EG_FIT_GROUP", raising=False)
        if group:
            monkeypatch.setenv("EG_FIT_GROUP", group)
        m = egm.compile(*dense_graphs(optim), gpu=gpu_ctx)
        same_start([m], 12)
        for _ in range(2):
            m.fit("train", {"x": x, "y": y}, batch_size=batch)
        results.append({tid: np.array(m.params[tid]) for tid in m.params.ids()})
        m.close()
    for tid in results[0]:
        assert np.isfinite(results[0][tid]).all(), tid
        assert np.array_equal(results[0][tid], results[1][tid]), tid
        assert np.array_equal(results[0][tid], results[2][tid]), tid
