"""The two gradients of conv2 as library kernels (eg_conv2_nhwc_grad_filter / _grad_image) against
the oracle's loop nests (oracle/refcpu.c), and through the model path (derive -> matcher -> the
same kernels) against the oracle's graph executor."""
import numpy as np
import pytest

import refcases
from conftest import TOL, rel_err
from parity import check_op, exact_conv2_grad_filter
from exprgrad_amd import dsl, layers, ops
from exprgrad_amd import model as egm

pytestmark = pytest.mark.gpu


def dev(ctx, arr):
    t = ctx.allocTensor(arr.shape)
    t.write(arr)
    return t


SHAPES = [  # N, H, W, C, F, FH, FW
    (1, 8, 8, 4, 4, 3, 3), (2, 12, 10, 8, 16, 3, 3), (1, 9, 7, 3, 5, 2, 3), (2, 20, 20, 32, 64, 3, 3),
    (1, 34, 34, 64, 64, 3, 3), (3, 6, 6, 4, 8, 1, 1), (1, 5, 5, 4, 4, 5, 5), (2, 7, 9, 5, 6, 3, 2),
    (1, 40, 36, 16, 32, 5, 5), (4, 16, 16, 64, 32, 3, 3),
]


@pytest.mark.parametrize("shape", SHAPES)
def test_conv2_gradients_against_the_oracle(gpu_ctx, refcpu, shape):
    N, H, W, C, F, FH, FW = shape
    rng = np.random.default_rng(sum(shape))
    img = rng.random((N, H, W, C), dtype=np.float32)
    flt = (rng.random((F, FH, FW, C), dtype=np.float32) * 2 - 1).astype(np.float32)
    gout = (rng.random((N, H - FH + 1, W - FW + 1, F), dtype=np.float32) - 0.5).astype(np.float32)
    dimg, dflt, dg = dev(gpu_ctx, img), dev(gpu_ctx, flt), dev(gpu_ctx, gout)

    gflt = gpu_ctx.allocTensor(flt.shape)
    gflt.write(np.full(flt.shape, 7.0, dtype=np.float32))            # must be overwritten
    ops.conv2_nhwc_grad_filter(gpu_ctx, N, H, W, C, F, FH, FW, dimg, dg, gflt)
    want = refcpu.conv2_nhwc_grad_filter(img, gout, flt.shape)
    assert rel_err(gflt.read(), want) <= TOL
    base = rng.random(flt.shape, dtype=np.float32)
    gflt.write(base)
    ops.conv2_nhwc_grad_filter(gpu_ctx, N, H, W, C, F, FH, FW, dimg, dg, gflt, accumulate=True)
    assert rel_err(gflt.read(), refcpu.conv2_nhwc_grad_filter(img, gout, flt.shape, out=base.copy())) <= TOL

    gimg = gpu_ctx.allocTensor(img.shape)
    gimg.write(np.full(img.shape, -3.0, dtype=np.float32))
    ops.conv2_nhwc_grad_image(gpu_ctx, N, H, W, C, F, FH, FW, dflt, dg, gimg)
    want = refcpu.conv2_nhwc_grad_image(flt, gout, img.shape)
    assert rel_err(gimg.read(), want) <= TOL
    base = rng.random(img.shape, dtype=np.float32)
    gimg.write(base)
    ops.conv2_nhwc_grad_image(gpu_ctx, N, H, W, C, F, FH, FW, dflt, dg, gimg, accumulate=True)
    assert rel_err(gimg.read(), refcpu.conv2_nhwc_grad_image(flt, gout, img.shape, out=base.copy())) <= TOL


HALO_SHAPES = [  # N, H, W, C, F: 3 x 3 filters, channels and filters multiples of 32 -> kernels/conv2_gradf_halo.hip
    (1, 34, 34, 64, 64),      # one segment per row, one pixel range
    (1, 66, 66, 64, 64),      # two whole segments per row
    (2, 40, 70, 32, 96),      # a ragged last segment (4 of 32 pixels), three filter blocks
    (3, 30, 34, 96, 32),      # three channel blocks, three images
    (1, 130, 258, 64, 64),    # 64 pixel ranges: the XCD-aware block order
    (2, 50, 100, 128, 64),    # eight quadrants
]


@pytest.mark.parametrize("shape", HALO_SHAPES)
def test_filter_gradient_halo_form_against_the_oracle_and_the_contraction(gpu_ctx, refcpu, monkeypatch, shape):
    """The halo form of the filter gradient (every image pixel staged once per row step) against the oracle's loop nest,
    against the one-contraction form it replaces (EG_CONV_NO_GRADF_HALO=1), with accumulation, and twice for identity."""
    N, H, W, C, F = shape
    rng = np.random.default_rng(sum(shape))
    img = rng.random((N, H, W, C), dtype=np.float32)
    gout = (rng.random((N, H - 2, W - 2, F), dtype=np.float32) - 0.5).astype(np.float32)
    dimg, dg = dev(gpu_ctx, img), dev(gpu_ctx, gout)
    gflt = gpu_ctx.allocTensor((F, 3, 3, C))
    gflt.write(np.full((F, 3, 3, C), 7.0, dtype=np.float32))         # must be overwritten
    monkeypatch.delenv("EG_CONV_NO_GRADF_HALO", raising=False)
    ops.conv2_nhwc_grad_filter(gpu_ctx, N, H, W, C, F, 3, 3, dimg, dg, gflt)
    halo = gflt.read()
    ops.conv2_nhwc_grad_filter(gpu_ctx, N, H, W, C, F, 3, 3, dimg, dg, gflt)
    assert np.array_equal(gflt.read(), halo)                          # fixed summation order: run-to-run identical
    want = refcpu.conv2_nhwc_grad_filter(img, gout, (F, 3, 3, C))
    assert rel_err(halo, want) <= TOL
    monkeypatch.setenv("EG_CONV_NO_GRADF_HALO", "1")
    ops.conv2_nhwc_grad_filter(gpu_ctx, N, H, W, C, F, 3, 3, dimg, dg, gflt)
    assert rel_err(halo, gflt.read()) <= TOL
    monkeypatch.delenv("EG_CONV_NO_GRADF_HALO")
    base = rng.random((F, 3, 3, C), dtype=np.float32)
    gflt.write(base)
    ops.conv2_nhwc_grad_filter(gpu_ctx, N, H, W, C, F, 3, 3, dimg, dg, gflt, accumulate=True)
    assert rel_err(gflt.read(), refcpu.conv2_nhwc_grad_filter(img, gout, (F, 3, 3, C), out=base.copy())) <= TOL


def test_gradient_kernels_are_deterministic(gpu_ctx):
    N, H, W, C, F, FH, FW = 2, 40, 40, 16, 32, 3, 3      # split-K over 2888 pixels
    rng = np.random.default_rng(0)
    dimg = dev(gpu_ctx, rng.random((N, H, W, C), dtype=np.float32))
    dg = dev(gpu_ctx, rng.random((N, H - 2, W - 2, F), dtype=np.float32))
    outs = []
    for _ in range(2):
        gflt = gpu_ctx.allocTensor((F, FH, FW, C))
        ops.conv2_nhwc_grad_filter(gpu_ctx, N, H, W, C, F, FH, FW, dimg, dg, gflt)
        outs.append(gflt.read())
    assert np.array_equal(outs[0], outs[1])


def cnn(c_in=4, f1=8, f2=16, rate=0.05):
    """conv2 -> relu -> conv2 -> mse -> gradientDescent: both gradient forms appear in `train`
    (filter gradients of both layers, image gradient of the second)."""
    x = dsl.input("x")
    net = layers.conv2(x, dsl.param([f1, 3, 3, c_in], name="k1"))
    net = layers.relu(net)
    net = layers.conv2(net, dsl.param([f2, 3, 3, f1], name="k2")).target("predict")
    net = layers.mse(net, dsl.input("y")).target("loss")
    return [net.backprop(layers.gradient_descent(rate)).target("train")]


@pytest.mark.parametrize("dims", [(2, 24, 24, 4, 8, 16), (1, 40, 36, 3, 5, 6), (2, 14, 14, 32, 64, 32)])
def test_cnn_train_step_matches_the_oracle(gpu_ctx, dims):
    from oracle import kd
    n, h, w, c_in, f1, f2 = dims
    from parity import Trio
    t = Trio(gpu_ctx, lambda: cnn(c_in, f1, f2))
    rng = np.random.default_rng(h * w)
    t.init_params(rng, -0.2, 0.2)
    x = rng.random((n, h, w, c_in), dtype=np.float32)
    y = rng.random((n, h - 4, w - 4, f2), dtype=np.float32)
    t.call("predict", {"x": x}, n=9 * max(c_in, f1))
    t.step("train", {"x": x, "y": y}, n=n * h * w)     # a filter gradient sums over every pixel of every image
    plan = t.gpu.launch_plan("train")
    assert plan.count("conv2-grad-filter") == 2 and plan.count("conv2-grad-image") == 1, plan
    t.close()


HALO_SHAPES = [  # N, H, W, C, F, FH, FW — at least 128 patches of 16x16 pixels: the LDS-halo kernel
    (8, 66, 66, 16, 64, 3, 3), (2, 130, 70, 32, 64, 3, 3), (9, 64, 64, 16, 96, 1, 1), (3, 100, 120, 48, 40, 2, 3),
    (1, 250, 131, 16, 128, 3, 1),
]


@pytest.mark.parametrize("shape", HALO_SHAPES)
def test_halo_convolution_against_the_oracle(gpu_ctx, refcpu, monkeypatch, shape):
    N, H, W, C, F, FH, FW = shape
    rng = np.random.default_rng(sum(shape))
    img = rng.random((N, H, W, C), dtype=np.float32)
    flt = (rng.random((F, FH, FW, C), dtype=np.float32) * 2 - 1).astype(np.float32)
    want = refcpu.conv2_nhwc(img, flt, threads_n=N, threads_y=max(1, 16 // N))
    dimg, dflt = dev(gpu_ctx, img), dev(gpu_ctx, flt)
    out = gpu_ctx.allocTensor(want.shape)
    out.write(np.full(want.shape, 5.0, dtype=np.float32))
    ops.conv2_nhwc(gpu_ctx, N, H, W, C, F, FH, FW, dimg, dflt, out)
    got = out.read()
    assert rel_err(got, want) <= TOL
    base = rng.random(want.shape, dtype=np.float32)
    out.write(base)
    ops.conv2_nhwc(gpu_ctx, N, H, W, C, F, FH, FW, dimg, dflt, out, accumulate=True)
    assert rel_err(out.read(), want + base) <= TOL


@pytest.mark.parametrize("seed", range(20))
def test_random_convolution_problems(gpu_ctx, refcpu, seed):
    # forward (halo kernel / implicit GEMM / 1x1 contraction chosen by shape) and both gradients
    rng = np.random.default_rng(500 + seed)
    C = int(rng.choice([1, 3, 8, 16, 32, 48]))
    F = int(rng.choice([2, 8, 24, 64, 72]))
    FH, FW = [(1, 1), (3, 3), (3, 3), (2, 2), (5, 5), (1, 3)][int(rng.integers(6))]
    big = C % 16 == 0 and FH <= 3 and FW <= 3 and rng.random() < 0.6       # enough patches for the halo kernel
    N = int(rng.choice([8, 10])) if big else int(rng.choice([1, 2, 5]))
    H = int(rng.choice([66, 70])) if big else int(rng.integers(FH + 1, 20))
    W = int(rng.choice([66, 100])) if big else int(rng.integers(FW + 1, 24))
    img = rng.random((N, H, W, C), dtype=np.float32)
    flt = (rng.random((F, FH, FW, C), dtype=np.float32) * 2 - 1).astype(np.float32)
    gout = (rng.random((N, H - FH + 1, W - FW + 1, F), dtype=np.float32) - 0.5).astype(np.float32)
    dimg, dflt, dg = dev(gpu_ctx, img), dev(gpu_ctx, flt), dev(gpu_ctx, gout)
    out = gpu_ctx.allocTensor(gout.shape)
    ops.conv2_nhwc(gpu_ctx, N, H, W, C, F, FH, FW, dimg, dflt, out)
    assert rel_err(out.read(), refcpu.conv2_nhwc(img, flt, threads_n=N, threads_y=4)) <= TOL, (N, H, W, C, F, FH, FW)
    gflt = gpu_ctx.allocTensor(flt.shape)
    ops.conv2_nhwc_grad_filter(gpu_ctx, N, H, W, C, F, FH, FW, dimg, dg, gflt)
    assert rel_err(gflt.read(), refcpu.conv2_nhwc_grad_filter(img, gout, flt.shape)) <= TOL, (N, H, W, C, F, FH, FW)
    gimg = gpu_ctx.allocTensor(img.shape)
    ops.conv2_nhwc_grad_image(gpu_ctx, N, H, W, C, F, FH, FW, dflt, dg, gimg)
    assert rel_err(gimg.read(), refcpu.conv2_nhwc_grad_image(flt, gout, img.shape)) <= TOL, (N, H, W, C, F, FH, FW)


@pytest.mark.parametrize("shape", [(256, 28, 28, 1, 8, 5, 5), (64, 50, 50, 3, 4, 3, 3), (300, 24, 24, 2, 6, 2, 4)])
def test_few_channel_convolutions_use_the_direct_kernels(gpu_ctx, refcpu, shape):
    # an image network's first layer (C <= 4, enough pixels): per-pixel kernels specialised with hiprtc
    N, H, W, C, F, FH, FW = shape
    rng = np.random.default_rng(sum(shape))
    img = rng.random((N, H, W, C), dtype=np.float32)
    flt = (rng.random((F, FH, FW, C), dtype=np.float32) * 2 - 1).astype(np.float32)
    gout = (rng.random((N, H - FH + 1, W - FW + 1, F), dtype=np.float32) - 0.5).astype(np.float32)
    dimg, dflt, dg = dev(gpu_ctx, img), dev(gpu_ctx, flt), dev(gpu_ctx, gout)
    out = gpu_ctx.allocTensor(gout.shape)
    base = rng.random(gout.shape, dtype=np.float32)
    for accumulate in (False, True):
        out.write(base)
        ops.conv2_nhwc(gpu_ctx, N, H, W, C, F, FH, FW, dimg, dflt, out, accumulate=accumulate)
        want = refcpu.conv2_nhwc(img, flt, out=base.copy() if accumulate else None, threads_n=16)
        assert rel_err(out.read(), want) <= TOL
    gflt = gpu_ctx.allocTensor(flt.shape)
    fbase = rng.random(flt.shape, dtype=np.float32)
    for accumulate in (False, True):
        gflt.write(fbase)
        ops.conv2_nhwc_grad_filter(gpu_ctx, N, H, W, C, F, FH, FW, dimg, dg, gflt, accumulate=accumulate)
        want = refcpu.conv2_nhwc_grad_filter(img, gout, flt.shape, out=fbase.copy() if accumulate else None)
        exact = exact_conv2_grad_filter(img, gout, flt.shape) + (fbase if accumulate else 0)
        # 150k-term sums: the backend at 1e-5 of the float64 value, the reference's sequential order at n * u
        check_op(gflt.read(), want, exact, n=N * (H - FH + 1) * (W - FW + 1), what="filter gradient")
    # run-to-run determinism of the block-partial reduction
    again = gpu_ctx.allocTensor(flt.shape)
    again.write(fbase)
    ops.conv2_nhwc_grad_filter(gpu_ctx, N, H, W, C, F, FH, FW, dimg, dg, again, accumulate=True)
    assert np.array_equal(again.read(), gflt.read())


PADDED_HALO_SHAPES = [  # N, H, W, C, F, FH, FW — enough patches of the IMAGE (the gradient's output) for the halo kernel
    (8, 66, 66, 16, 64, 3, 3), (2, 130, 130, 32, 64, 3, 3), (3, 100, 120, 48, 48, 2, 3), (1, 250, 131, 24, 128, 3, 1),
    (2, 129, 97, 8, 16, 3, 3),
]


@pytest.mark.parametrize("shape", PADDED_HALO_SHAPES)
def test_image_gradient_with_virtual_padding(gpu_ctx, refcpu, monkeypatch, shape):
    """gImg is a full correlation of gOut with the flipped bank: the halo kernel reads the (FH - 1, FW - 1) border from a
    block of zeros instead of a padded copy of gOut.  Against the oracle, with and without accumulate, and bit for bit
    against the padded-copy route of rounds 1 and 2 (same kernel, same operand values, same order)."""
    N, H, W, C, F, FH, FW = shape
    rng = np.random.default_rng(sum(shape))
    flt = (rng.random((F, FH, FW, C), dtype=np.float32) * 2 - 1).astype(np.float32)
    gout = (rng.random((N, H - FH + 1, W - FW + 1, F), dtype=np.float32) - 0.5).astype(np.float32)
    want = refcpu.conv2_nhwc_grad_image(flt, gout, (N, H, W, C))
    dflt, dg = dev(gpu_ctx, flt), dev(gpu_ctx, gout)
    gimg = gpu_ctx.allocTensor((N, H, W, C))
    gimg.write(np.full((N, H, W, C), 3.0, dtype=np.float32))
    ops.conv2_nhwc_grad_image(gpu_ctx, N, H, W, C, F, FH, FW, dflt, dg, gimg)
    got = gimg.read()
    assert rel_err(got, want) <= TOL
    base = rng.random((N, H, W, C), dtype=np.float32)
    gimg.write(base)
    ops.conv2_nhwc_grad_image(gpu_ctx, N, H, W, C, F, FH, FW, dflt, dg, gimg, accumulate=True)
    assert rel_err(gimg.read(), want + base) <= TOL
    monkeypatch.setenv("EG_CONV_NO_VIRTUAL_PAD", "1")
    ops.conv2_nhwc_grad_image(gpu_ctx, N, H, W, C, F, FH, FW, dflt, dg, gimg)
    assert np.array_equal(gimg.read(), got)


TINY_SHAPES = [  # N, H, W, C, F, FH, FW: a few million multiply-adds in all -> kernels/conv2_tiny.hip
    (32, 28, 28, 1, 8, 5, 5),     # fashion_mnist, first layer at the reference's batch (fashion_mnist.nim:39-57)
    (32, 12, 12, 8, 16, 3, 3),    # ... second layer
    (5, 9, 11, 3, 7, 2, 4),       # nothing a multiple of anything
    (1, 3, 3, 2, 4, 3, 3),        # one output pixel
    (2, 6, 5, 4, 12, 1, 3),       # one filter row
    (64, 12, 12, 8, 16, 3, 3),    # 7.4 M multiply-adds: past the limit, the contraction route (both settings the same)
    (4, 10, 10, 16, 4, 3, 3),     # filter rows of 48 floats, 576 filter-gradient outputs
    (3, 10, 10, 16, 96, 3, 3),    # 13 824 filter-gradient outputs: more than the tiny kernel's eight per thread -> contraction route for that one
]


@pytest.mark.parametrize("shape", TINY_SHAPES)
def test_tiny_convolutions_against_the_oracle_and_the_contraction_route(gpu_ctx, refcpu, monkeypatch, shape):
    """Round 4: conv2 and both gradients of problems of a few million multiply-adds run as one thread per output element
    (filter gradient: blocks of pixels + the fixed-order slab sum).  Against the oracle's loop nests, against the
    contraction route (EG_CONV_NO_TINY=1), onto existing values, and twice for run-to-run identity."""
    N, H, W, C, F, FH, FW = shape
    Ho, Wo = H - FH + 1, W - FW + 1
    rng = np.random.default_rng(sum(shape))
    img = rng.random((N, H, W, C), dtype=np.float32)
    flt = (rng.random((F, FH, FW, C), dtype=np.float32) * 2 - 1).astype(np.float32)
    gout = (rng.random((N, Ho, Wo, F), dtype=np.float32) - 0.5).astype(np.float32)
    dimg, dflt, dg = dev(gpu_ctx, img), dev(gpu_ctx, flt), dev(gpu_ctx, gout)
    base_out = rng.random((N, Ho, Wo, F), dtype=np.float32)
    base_flt = rng.random(flt.shape, dtype=np.float32)
    base_img = rng.random(img.shape, dtype=np.float32)
    want = {
        "out": refcpu.conv2_nhwc(img, flt), "out+": refcpu.conv2_nhwc(img, flt, out=base_out.copy()),
        "gflt": refcpu.conv2_nhwc_grad_filter(img, gout, flt.shape),
        "gflt+": refcpu.conv2_nhwc_grad_filter(img, gout, flt.shape, out=base_flt.copy()),
        "gimg": refcpu.conv2_nhwc_grad_image(flt, gout, img.shape),
        "gimg+": refcpu.conv2_nhwc_grad_image(flt, gout, img.shape, out=base_img.copy()),
    }

    def run():
        got = {}
        out = gpu_ctx.allocTensor(base_out.shape)
        out.write(np.full(base_out.shape, np.nan, dtype=np.float32))        # must be overwritten
        ops.conv2_nhwc(gpu_ctx, N, H, W, C, F, FH, FW, dimg, dflt, out)
        got["out"] = out.read()
        out.write(base_out)
        ops.conv2_nhwc(gpu_ctx, N, H, W, C, F, FH, FW, dimg, dflt, out, accumulate=True)
        got["out+"] = out.read()
        gflt = gpu_ctx.allocTensor(flt.shape)
        gflt.write(np.full(flt.shape, np.nan, dtype=np.float32))
        ops.conv2_nhwc_grad_filter(gpu_ctx, N, H, W, C, F, FH, FW, dimg, dg, gflt)
        got["gflt"] = gflt.read()
        gflt.write(base_flt)
        ops.conv2_nhwc_grad_filter(gpu_ctx, N, H, W, C, F, FH, FW, dimg, dg, gflt, accumulate=True)
        got["gflt+"] = gflt.read()
        gimg = gpu_ctx.allocTensor(img.shape)
        gimg.write(np.full(img.shape, np.nan, dtype=np.float32))
        ops.conv2_nhwc_grad_image(gpu_ctx, N, H, W, C, F, FH, FW, dflt, dg, gimg)
        got["gimg"] = gimg.read()
        gimg.write(base_img)
        ops.conv2_nhwc_grad_image(gpu_ctx, N, H, W, C, F, FH, FW, dflt, dg, gimg, accumulate=True)
        got["gimg+"] = gimg.read()
        return got

    monkeypatch.delenv("EG_CONV_NO_TINY", raising=False)
    tiny, again = run(), run()
    monkeypatch.setenv("EG_CONV_NO_TINY", "1")
    other = run()
    for key, ref in want.items():
        assert rel_err(tiny[key], ref) <= TOL, key
        assert rel_err(other[key], ref) <= TOL, key
        assert np.array_equal(tiny[key], again[key]), key
        assert rel_err(tiny[key], other[key].astype(np.float64)) <= TOL, key


BAND_SHAPES = [  # N, H, W, C, F, FH, FW: at most 16 channels and 16 filters, past the tiny limit -> kernels/conv2_band.cpp
    (256, 28, 28, 1, 8, 5, 5),     # fashion_mnist, first layer (fashion_mnist.nim:39-57): whole images per block
    (300, 12, 12, 8, 16, 3, 3),    # ... second layer; 300 images are not a whole number of seven-image bands
    (3, 70, 90, 3, 5, 3, 3),       # bands of rows of one image; 5 outputs per pixel: the element-wise store path
    (130, 16, 20, 16, 16, 3, 3),   # 16 -> 16
    (2, 100, 64, 2, 3, 7, 5),      # 70 taps of a 7 x 5 filter
]


@pytest.mark.parametrize("shape", BAND_SHAPES)
def test_band_convolutions_against_the_oracle_and_the_other_routes(gpu_ctx, refcpu, monkeypatch, shape):
    """Round 5: conv2 and both gradients with at most 16 channels and 16 filters as 16 x 16 x 4 matrix instructions with the
    small operand in registers (kernels/conv2_band.cpp).  Against the oracle's loop nests, against the routes they replace
    (EG_CONV_NO_BAND=1: per-pixel kernels, implicit-GEMM tiles), onto existing values, and twice for run-to-run identity
    (the filter gradient's per-block partial rows are folded in a fixed order)."""
    N, H, W, C, F, FH, FW = shape
    Ho, Wo = H - FH + 1, W - FW + 1
    rng = np.random.default_rng(sum(shape))
    img = rng.random((N, H, W, C), dtype=np.float32)
    flt = (rng.random((F, FH, FW, C), dtype=np.float32) * 2 - 1).astype(np.float32)
    gout = (rng.random((N, Ho, Wo, F), dtype=np.float32) - 0.5).astype(np.float32)
    dimg, dflt, dg = dev(gpu_ctx, img), dev(gpu_ctx, flt), dev(gpu_ctx, gout)
    base_out = rng.random((N, Ho, Wo, F), dtype=np.float32)
    base_flt = rng.random(flt.shape, dtype=np.float32)
    base_img = rng.random(img.shape, dtype=np.float32)
    want = {
        "out": refcpu.conv2_nhwc(img, flt), "out+": refcpu.conv2_nhwc(img, flt, out=base_out.copy()),
        "gflt": refcpu.conv2_nhwc_grad_filter(img, gout, flt.shape),
        "gflt+": refcpu.conv2_nhwc_grad_filter(img, gout, flt.shape, out=base_flt.copy()),
        "gimg": refcpu.conv2_nhwc_grad_image(flt, gout, img.shape),
        "gimg+": refcpu.conv2_nhwc_grad_image(flt, gout, img.shape, out=base_img.copy()),
    }

    def run():
        got = {}
        out = gpu_ctx.allocTensor(base_out.shape)
        out.write(np.full(base_out.shape, np.nan, dtype=np.float32))        # must be overwritten
        ops.conv2_nhwc(gpu_ctx, N, H, W, C, F, FH, FW, dimg, dflt, out)
        got["out"] = out.read()
        out.write(base_out)
        ops.conv2_nhwc(gpu_ctx, N, H, W, C, F, FH, FW, dimg, dflt, out, accumulate=True)
        got["out+"] = out.read()
        gflt = gpu_ctx.allocTensor(flt.shape)
        gflt.write(np.full(flt.shape, np.nan, dtype=np.float32))
        ops.conv2_nhwc_grad_filter(gpu_ctx, N, H, W, C, F, FH, FW, dimg, dg, gflt)
        got["gflt"] = gflt.read()
        gflt.write(base_flt)
        ops.conv2_nhwc_grad_filter(gpu_ctx, N, H, W, C, F, FH, FW, dimg, dg, gflt, accumulate=True)
        got["gflt+"] = gflt.read()
        gimg = gpu_ctx.allocTensor(img.shape)
        gimg.write(np.full(img.shape, np.nan, dtype=np.float32))
        ops.conv2_nhwc_grad_image(gpu_ctx, N, H, W, C, F, FH, FW, dflt, dg, gimg)
        got["gimg"] = gimg.read()
        gimg.write(base_img)
        ops.conv2_nhwc_grad_image(gpu_ctx, N, H, W, C, F, FH, FW, dflt, dg, gimg, accumulate=True)
        got["gimg+"] = gimg.read()
        return got

    monkeypatch.delenv("EG_CONV_NO_BAND", raising=False)
    band, again = run(), run()
    monkeypatch.setenv("EG_CONV_NO_BAND", "1")
    other = run()
    for key, ref in want.items():
        # (the filter gradient sums N * Ho * Wo terms per element: the bound is relative to the largest element as everywhere in this file)
        assert rel_err(band[key], ref) <= TOL, key
        assert rel_err(other[key], ref) <= TOL, key
        assert np.array_equal(band[key], again[key]), key
        assert rel_err(band[key], other[key].astype(np.float64)) <= TOL, key


def test_band_convolutions_on_random_shapes(gpu_ctx, monkeypatch):
    """Forty random shapes with at most 16 channels and 16 filters (image sizes, filter sizes, batches that do not divide
    into bands, outputs per pixel that do and do not make 16-byte pieces): the band kernels against the routes they replace
    (EG_CONV_NO_BAND=1), forward and both gradients, overwrite and accumulate."""
    rng = np.random.default_rng(2026)
    for case in range(40):
        C, F = int(rng.integers(1, 17)), int(rng.integers(1, 17))
        FH, FW = int(rng.integers(1, 6)), int(rng.integers(1, 6))
        if FH == 1 and FW == 1:
            FW = 2          # (1 x 1 filters are plain contractions)
        H, W = int(rng.integers(FH + 3, 48)), int(rng.integers(FW + 3, 48))
        Ho, Wo = H - FH + 1, W - FW + 1
        N = int(max(1, (4096 + 2000 * rng.random()) // (Ho * Wo)) + 1)
        while N * Ho * Wo * F * FH * FW * C < 6.5e6:     # past the tiny kernels' limit
            N += 1
        shape = (N, H, W, C, F, FH, FW)
        img = rng.random((N, H, W, C), dtype=np.float32)
        flt = (rng.random((F, FH, FW, C), dtype=np.float32) * 2 - 1).astype(np.float32)
        gout = (rng.random((N, Ho, Wo, F), dtype=np.float32) - 0.5).astype(np.float32)
        base = {"out": rng.random((N, Ho, Wo, F), dtype=np.float32), "gflt": rng.random(flt.shape, dtype=np.float32),
                "gimg": rng.random(img.shape, dtype=np.float32)}
        dimg, dflt, dg = dev(gpu_ctx, img), dev(gpu_ctx, flt), dev(gpu_ctx, gout)

        def run(acc):
            got = {}
            for key, fn, a, b in (("out", ops.conv2_nhwc, dimg, dflt), ("gflt", ops.conv2_nhwc_grad_filter, dimg, dg),
                                  ("gimg", ops.conv2_nhwc_grad_image, dflt, dg)):
                t = gpu_ctx.allocTensor(base[key].shape)
                t.write(base[key] if acc else np.full(base[key].shape, np.nan, dtype=np.float32))
                fn(gpu_ctx, N, H, W, C, F, FH, FW, a, b, t, accumulate=acc)
                got[key] = t.read()
                t.buffer.dealloc()
            return got
        for acc in (False, True):
            monkeypatch.delenv("EG_CONV_NO_BAND", raising=False)
            band = run(acc)
            monkeypatch.setenv("EG_CONV_NO_BAND", "1")
            other = run(acc)
            for key in band:
                assert rel_err(band[key], other[key].astype(np.float64)) <= TOL, (shape, key, acc)
        for t in (dimg, dflt, dg):
            t.buffer.dealloc()
    monkeypatch.delenv("EG_CONV_NO_BAND", raising=False)


@pytest.mark.parametrize("shape", SHAPES)
def test_conv2_gradients_on_the_contraction_route(gpu_ctx, refcpu, monkeypatch, shape):
    """The shapes of test_conv2_gradients_against_the_oracle are small enough for kernels/conv2_tiny.hip now; the
    contraction / halo / per-pixel routes they used to exercise stay covered with EG_CONV_NO_TINY=1."""
    monkeypatch.setenv("EG_CONV_NO_TINY", "1")
    test_conv2_gradients_against_the_oracle(gpu_ctx, refcpu, shape)
