"""A bias gradient that rides along with its layer's weight gradient (host/plan_epilogue.cpp
fold_bias_gradients, GemmArgs::ones_row): `gW[it,x] ++= in[y,it] * g[y,x]` and `gb[x] ++= g[y,x]`
(dense, dnn.nim:19-24, differentiated by passes.nim:519-549) become one contraction [in | 1]^T * g whose
last row is gb.  Shapes chosen for the corners of the kernel: the virtual row in a ragged edge tile, as
the first row of a tile of its own (M a multiple of the tile height), together with a ragged K tail."""
import numpy as np
import pytest

from exprgrad_amd import dsl, layers
from parity import Trio

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def no_sample_groups(monkeypatch):
    """These tests are about the plans of LARGE layers, exercised at small sizes (EG_EPILOGUE_MIN_ELEMS etc.): the
    small-batch sample groups of round 5 (tests/test_gpu_sample_fuse.py) would take the narrow layers first."""
    monkeypatch.setenv("EG_NO_SAMPLE_FUSE", "1")


def net(n_in, n_hidden, n_out=4):
    def build():
        x = layers.dense(dsl.input("x"), n_in, n_hidden)
        x = layers.tanh(x)
        x = layers.dense(x, n_hidden, n_out).target("predict")
        x = layers.mse(x, dsl.input("y")).target("loss")
        return [x.backprop(layers.gradient_descent(0.05)).target("train")]
    return build


@pytest.mark.parametrize("dims", [(784, 512, 4096), (256, 96, 1000), (255, 128, 1024), (127, 200, 777), (128, 72, 520),
                                  (512, 260, 2048), (100, 68, 16), (1024, 128, 4099),
                                  # 64-row tiles: the virtual row alone in the second tile row, 32-deep k-tiles, uneven slices
                                  (64, 512, 2048), (64, 96, 3000), (192, 512, 2050), (128, 136, 300)])
def test_bias_gradient_as_the_last_row_of_the_weight_gradient(gpu_ctx, monkeypatch, dims):
    n_in, n_hidden, batch = dims
    monkeypatch.setenv("EG_EPILOGUE_MIN_ELEMS", str(1 << 40))     # keep the contractions plain: this test is about the fold
    t = Trio(gpu_ctx, net(n_in, n_hidden), threads=8)
    rng = np.random.default_rng(n_in + batch)
    t.init_params(rng, -0.1, 0.1)
    x = (rng.random((batch, n_in), dtype=np.float32) - 0.5).astype(np.float32)
    y = rng.random((batch, 4), dtype=np.float32)
    for _ in range(3):          # eager, captured, replayed
        t.step("train", {"x": x, "y": y}, n=batch)
    plan = t.gpu.launch_plan("train")
    # the first layer's pair qualifies whenever the batch is large enough for its bias gradient to be a launch of
    # its own (at batch 16 it joins a single-block small-kernel group instead)
    if batch >= 256:
        assert "+ones-row" in plan, plan
    t.close()


def test_fold_can_be_switched_off_and_agrees(gpu_ctx):
    """EG_NO_ONES_ROW is read once per process, so the unfused path is exercised by its own fallback: a
    caller-owned input that is not 16-byte aligned disqualifies the LDS-DMA loop at run time, and the
    launch falls back to contraction + column sum."""
    torch = pytest.importorskip("torch")
    from exprgrad_amd import model as egm
    n_in, n_hidden, batch = 256, 96, 512
    a = egm.compile(*net(n_in, n_hidden)(), gpu=gpu_ctx)
    b = egm.compile(*net(n_in, n_hidden)(), gpu=gpu_ctx)
    rng = np.random.default_rng(3)
    for tid in a.params.ids():
        v = (rng.random(a.params[tid].shape, dtype=np.float32) * 0.2 - 0.1).astype(np.float32)
        a.params[tid] = v
        b.params[tid] = v
    x = (rng.random((batch, n_in), dtype=np.float32) - 0.5).astype(np.float32)
    y = rng.random((batch, 4), dtype=np.float32)
    big = torch.zeros(batch * n_in + 1, device="cuda")
    big[1:] = torch.from_numpy(x.ravel()).cuda()
    x_unaligned = big[1:].view(batch, n_in)                 # same values, 4 bytes off a 16-byte boundary
    assert x_unaligned.data_ptr() % 16 == 4
    a.apply("train", {"x": x, "y": y})
    b.apply("train", {"x": x_unaligned, "y": torch.from_numpy(y).cuda()})
    torch.cuda.synchronize()
    assert "+ones-row" in a.launch_plan("train")
    for tid in a.params.ids():
        ga, gb = a.params[tid], b.params[tid]
        assert np.max(np.abs(ga - gb)) <= 1e-5 * max(np.max(np.abs(ga)), 1e-30), tid
    a.close()
    b.close()
