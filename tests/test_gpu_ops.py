"""GPU parity of the library fast path (group 2 of the C ABI) against the oracle.

Every test drives the HIP kernels through the C ABI (exprgrad_amd.ops -> libexprgrad_hip.so) and
compares with oracle/refcpu.c on the same seeded inputs.  Tolerance: 1e-5 relative (float32),
the bound BASELINE.json's north_star states; the GPU sums in a different order (MFMA fmaf chains,
tree reductions) than the reference's sequential loop, so bit equality is not expected.
"""
import numpy as np
import pytest

import exprgrad_amd as eg
from exprgrad_amd import ops
from conftest import TOL, rel_err

pytestmark = pytest.mark.gpu


def dev(ctx, arr):
    t = ctx.allocTensor(arr.shape)
    t.write(arr)
    return t


def gemm_case(ctx, refcpu, M, N, K, trans_a=False, trans_b=False, accumulate=False, bias=False, seed=0,
              threads=8, return_inputs=False):
    rng = np.random.default_rng(seed)
    a = rng.random((K, M) if trans_a else (M, K), dtype=np.float32)
    b = (rng.random((N, K) if trans_b else (K, N), dtype=np.float32) * 2 - 1).astype(np.float32)
    c0 = rng.random((M, N), dtype=np.float32) if accumulate else np.zeros((M, N), dtype=np.float32)
    bv = (rng.random((N,), dtype=np.float32) - 0.5).astype(np.float32) if bias else None
    want = refcpu.sgemm(a, b, trans_a, trans_b, out=c0.copy(), threads=threads)
    if bias:
        refcpu.bias_add(bv, want)
    da, db, dc = dev(ctx, a), dev(ctx, b), dev(ctx, c0)
    dbias = dev(ctx, bv) if bias else None
    ops.sgemm(ctx, M, N, K, da, a.shape[1], db, b.shape[1], dc, N, trans_a, trans_b, accumulate, dbias)
    got = dc.read()
    if return_inputs:
        return rel_err(got, want), got, want, a, b
    return rel_err(got, want), got, want


@pytest.mark.parametrize("shape", [(256, 256, 256), (128, 128, 16), (512, 384, 272)])
def test_matmul_nn_aligned(gpu_ctx, refcpu, shape):
    # configs[0] of BASELINE.json: matmul 256^3 float32 (benchmarks/matmul/matmul_gpu.nim:28-36 measure_cpu)
    err, _, _ = gemm_case(gpu_ctx, refcpu, *shape)
    assert err <= TOL


def test_known_answer_2x3_3x2(gpu_ctx):
    # tests/test_model.nim:37-44 — exact on integer-valued data
    a = np.array([[1, 2, 3], [4, 5, 6]], dtype=np.float32)
    b = np.array([[1, 2], [3, 4], [5, 6]], dtype=np.float32)
    da, db = dev(gpu_ctx, a), dev(gpu_ctx, b)
    dc = gpu_ctx.allocTensor((2, 2))
    ops.sgemm(gpu_ctx, 2, 2, 3, da, 3, db, 2, dc, 2)
    assert np.array_equal(dc.read(), np.array([[22, 28], [49, 64]], dtype=np.float32))


@pytest.mark.parametrize("shape", [(1, 1, 1), (3, 5, 7), (130, 70, 33), (257, 129, 100), (64, 200, 19),
                                   (1000, 10, 512), (999, 4, 2), (515, 1, 4), (200, 33, 17), (127, 65, 50)])
@pytest.mark.parametrize("layout", ["nn", "nt", "tn", "tt"])
def test_ragged_shapes_all_layouts(gpu_ctx, refcpu, shape, layout):
    ta, tb = layout[0] == "t", layout[1] == "t"
    err, _, _ = gemm_case(gpu_ctx, refcpu, *shape, trans_a=ta, trans_b=tb, seed=hash((shape, layout)) % 1000)
    assert err <= TOL


@pytest.mark.parametrize("shape", [(784, 512, 8192), (300, 256, 20000), (144, 128, 40000), (272, 200, 12000),
                                   (520, 516, 4096)])
@pytest.mark.parametrize("layout", ["nn", "nt", "tn"])
def test_split_k_with_a_ragged_last_tile_row(gpu_ctx, refcpu, shape, layout):
    # weight-gradient shapes (K = batch): split-K with the ragged last tile row cut into fewer, longer
    # slices, ragged tiles on the LDS-DMA loop with clamped rows, empty 32x32 sub-blocks skipped
    ta, tb = layout[0] == "t", layout[1] == "t"
    err, _, _ = gemm_case(gpu_ctx, refcpu, *shape, trans_a=ta, trans_b=tb, seed=11)
    assert err <= TOL
    err, _, _ = gemm_case(gpu_ctx, refcpu, *shape, trans_a=ta, trans_b=tb, accumulate=True, bias=True, seed=12)
    assert err <= TOL


@pytest.mark.parametrize("layout", ["nn", "nt", "tn"])
def test_accumulate_and_bias(gpu_ctx, refcpu, layout):
    ta, tb = layout[0] == "t", layout[1] == "t"
    for M, N, K in [(256, 128, 64), (77, 50, 31)]:
        err, _, _ = gemm_case(gpu_ctx, refcpu, M, N, K, ta, tb, accumulate=True, bias=True, seed=3)
        assert err <= TOL
        err, _, _ = gemm_case(gpu_ctx, refcpu, M, N, K, ta, tb, accumulate=False, bias=True, seed=4)
        assert err <= TOL


def test_empty_extents(gpu_ctx):
    # K = 0: the result is the (zeroed or accumulated-onto) tensor plus bias; M or N = 0: no-op
    c = gpu_ctx.allocTensor((4, 4))
    c.fill(7.0)
    a = gpu_ctx.allocTensor((4, 1))
    ops.sgemm(gpu_ctx, 4, 4, 0, a, 1, a, 4, c, 4, accumulate=False)
    assert np.array_equal(c.read(), np.zeros((4, 4), dtype=np.float32))
    ops.sgemm(gpu_ctx, 0, 4, 3, a, 3, a, 4, c, 4)
    ops.sgemm(gpu_ctx, 4, 0, 3, a, 3, a, 4, c, 4)
    gpu_ctx.sync()


def test_weight_gradient_split_k(gpu_ctx, refcpu):
    # gW[it,x] += a[y,it]*g[y,x] with a long batch: exercises split-K + the deterministic second pass.
    # (cfg5's gW1 is [784,512] over a 65536 batch; here 8192 so the sequential oracle stays fast.)
    err, got, _ = gemm_case(gpu_ctx, refcpu, 784, 512, 8192, trans_a=True, seed=5)
    assert err <= TOL
    _, got2, _ = gemm_case(gpu_ctx, refcpu, 784, 512, 8192, trans_a=True, seed=5)
    assert np.array_equal(got, got2), "split-K must be run-to-run deterministic"
    for shape in [(2, 4, 65536), (4, 1, 65536), (512, 10, 16384)]:
        err, _, _ = gemm_case(gpu_ctx, refcpu, *shape, trans_a=True, seed=6)
        assert err <= TOL


def test_dense_layer_shapes(gpu_ctx, refcpu):
    # dense forward + its two gradient contractions at the layer shapes of configs 3 and 5 (reduced batch)
    B = 4096
    for (i, o) in [(2, 4), (4, 1), (784, 512), (512, 10)]:
        err, _, _ = gemm_case(gpu_ctx, refcpu, B, o, i, bias=True, seed=7)           # fwd
        assert err <= TOL
        err, _, _ = gemm_case(gpu_ctx, refcpu, B, i, o, trans_b=True, seed=8)        # grad input
        assert err <= TOL
        err, _, _ = gemm_case(gpu_ctx, refcpu, i, o, B, trans_a=True, seed=9)        # grad weights
        assert err <= TOL


def test_matmul_vs_f64_shadow(gpu_ctx, refcpu):
    # error budgeting (SURVEY.md §7 hard part 2): both the reference order and the MFMA order must sit
    # within 1e-5 of the exact product at K = 1024
    M = N = 128
    K = 1024
    err, got, want, a, b = gemm_case(gpu_ctx, refcpu, M, N, K, seed=11, return_inputs=True)
    exact = refcpu.dgemm(a, b)
    assert rel_err(want, exact) <= TOL
    assert rel_err(got, exact) <= TOL
    assert err <= TOL


def test_transpose_detection(gpu_ctx):
    # A = I with an asymmetric B: a row/column swap in the accumulator write-back would show here
    n = 96
    a = np.eye(n, dtype=np.float32)
    b = np.arange(n * n, dtype=np.float32).reshape(n, n)
    da, db = dev(gpu_ctx, a), dev(gpu_ctx, b)
    dc = gpu_ctx.allocTensor((n, n))
    ops.sgemm(gpu_ctx, n, n, n, da, n, db, n, dc, n)
    assert np.array_equal(dc.read(), b)


MAPS = [("identity", 0.0), ("relu", 0.0), ("leaky_relu", 0.01), ("sigmoid", 0.0), ("tanh", 0.0),
        ("scale", 2.5), ("sin", 0.0), ("xor_leaky", 0.1), ("exp", 0.0)]


@pytest.mark.parametrize("op,param", MAPS)
@pytest.mark.parametrize("n", [1, 6, 1023, 4096, 262144 + 3])
def test_maps_and_their_gradients(gpu_ctx, refcpu, op, param, n):
    rng = np.random.default_rng(n)
    x = ((rng.random(n, dtype=np.float32) - 0.5) * 8).astype(np.float32)
    g = (rng.random(n, dtype=np.float32) - 0.5).astype(np.float32)
    dx, dg = dev(gpu_ctx, x), dev(gpu_ctx, g)
    out = gpu_ctx.allocTensor((n,))
    ops.map_(gpu_ctx, op, n, dx, out, param)
    assert rel_err(out.read(), refcpu.map_(op, x, param)) <= TOL
    base = rng.random(n, dtype=np.float32)
    out.write(base)
    ops.map_(gpu_ctx, op, n, dx, out, param, accumulate=True)
    assert rel_err(out.read(), refcpu.map_(op, x, param, out=base.copy())) <= TOL
    gin = gpu_ctx.allocTensor((n,))
    ops.map_grad(gpu_ctx, op, n, dx, dg, gin, param)
    assert rel_err(gin.read(), refcpu.map_grad(op, x, g, param)) <= TOL


def test_known_answer_maps(gpu_ctx):
    # tests/test_model.nim:46-54 and tests/test_gpu.nim:238-246 (exact)
    x = np.array([0, -1, 10, -20, 0.1, -0.1], dtype=np.float32)
    dx, out = dev(gpu_ctx, x), gpu_ctx.allocTensor((6,))
    ops.map_(gpu_ctx, "relu", 6, dx, out)
    assert np.array_equal(out.read(), np.array([0, 0, 10, 0, 0.1, 0], dtype=np.float32))
    x = np.array([1, 2, -1, -2, 0, 3], dtype=np.float32)
    dx.write(x)
    ops.map_(gpu_ctx, "leaky_relu", 6, dx, out, 0.01)
    want = np.array([1, 2, np.float32(-1) * np.float32(0.01), np.float32(-2) * np.float32(0.01), 0, 3],
                    dtype=np.float32)
    assert np.array_equal(out.read(), want)


@pytest.mark.parametrize("rows,cols", [(1, 1), (5, 1), (1000, 4), (4096, 10), (777, 512), (300, 130), (65536, 1),
                                       (64, 33)])
def test_bias_colsum_rowsum(gpu_ctx, refcpu, rows, cols):
    rng = np.random.default_rng(rows * 131 + cols)
    x = (rng.random((rows, cols), dtype=np.float32) - 0.5).astype(np.float32)
    b = rng.random((cols,), dtype=np.float32)
    dx, db = dev(gpu_ctx, x), dev(gpu_ctx, b)
    ops.bias_add(gpu_ctx, rows, cols, db, dx, accumulate=True)
    assert rel_err(dx.read(), refcpu.bias_add(b, x.copy())) <= TOL
    dx.write(x)
    oc = gpu_ctx.allocTensor((cols,))
    ops.colsum(gpu_ctx, rows, cols, dx, oc)
    want = refcpu.colsum(x)
    # a reduction's error scales with sum|x|, not |sum x| (cancellation): normalise by it
    scale = np.abs(x).sum(axis=0).max()
    assert np.max(np.abs(oc.read().astype(np.float64) - want)) <= TOL * scale
    oc.write(b)
    ops.colsum(gpu_ctx, rows, cols, dx, oc, accumulate=True)
    assert np.max(np.abs(oc.read().astype(np.float64) - refcpu.colsum(x, out=b.copy()))) <= TOL * (scale + 1)
    orow = gpu_ctx.allocTensor((rows,))
    ops.rowsum(gpu_ctx, rows, cols, dx, orow)
    scale = np.abs(x).sum(axis=1).max()
    assert np.max(np.abs(orow.read().astype(np.float64) - refcpu.rowsum(x))) <= TOL * scale


@pytest.mark.parametrize("n", [1, 4, 1000, 65536, 1 << 20])
def test_sum_axpy_fill(gpu_ctx, refcpu, n):
    rng = np.random.default_rng(n)
    x = rng.random(n, dtype=np.float32)
    dx = dev(gpu_ctx, x)
    o = gpu_ctx.allocTensor((1,))
    ops.total(gpu_ctx, n, dx, o)
    # positive data: the sequential f32 sum the reference performs drifts ~n*eps; compare both to f64
    exact = x.astype(np.float64).sum()
    assert abs(float(o.read()[0]) - exact) <= TOL * exact
    assert abs(float(refcpu.total(x)[0]) - exact) <= 4e-4 * exact  # the reference's own drift at n = 2^20
    y = rng.random(n, dtype=np.float32)
    dy = dev(gpu_ctx, y)
    ops.axpy(gpu_ctx, n, -0.1, dx, dy)
    assert rel_err(dy.read(), refcpu.axpy(-0.1, x, y.copy())) <= TOL
    ops.fill(gpu_ctx, n, 1.0, dy)
    assert np.array_equal(dy.read(), np.ones(n, dtype=np.float32))


def test_sum_of_squares_known_answer(gpu_ctx):
    # tests/test_model.nim:56-69
    d = (np.array([1, 2, 3, 4], dtype=np.float32) - np.array([4, 3, 2, 1], dtype=np.float32)) ** 2
    dd, o = dev(gpu_ctx, d), gpu_ctx.allocTensor((1,))
    ops.total(gpu_ctx, 4, dd, o)
    assert np.array_equal(o.read(), np.array([20], dtype=np.float32))


@pytest.mark.parametrize("geom", [
    (1, 1, 7, 1, 1, 1, 3),        # conv1 of tests/test_model.nim:91-97 as a 1 x 7 image
    (1, 9, 8, 3, 4, 3, 3),
    (2, 12, 17, 8, 8, 3, 3),      # benchmarks/conv2/conv2.nim:330-338 filter bank [8,3,3,8], small image
    (1, 20, 20, 64, 64, 3, 3),    # cfg4 geometry (C = F = 64, 3x3), small image
    (3, 10, 11, 5, 7, 2, 4),      # nothing divisible by anything
    (1, 6, 6, 16, 33, 1, 1),      # 1x1 filter = plain matmul
])
def test_conv2(gpu_ctx, refcpu, geom):
    N, H, W, C, F, FH, FW = geom
    rng = np.random.default_rng(sum(geom))
    img = rng.random((N, H, W, C), dtype=np.float32)
    flt = ((rng.random((F, FH, FW, C), dtype=np.float32) - 0.5) * 4).astype(np.float32)
    want = refcpu.conv2_nhwc(img, flt)
    di, df = dev(gpu_ctx, img), dev(gpu_ctx, flt)
    out = gpu_ctx.allocTensor(want.shape)
    ops.conv2_nhwc(gpu_ctx, N, H, W, C, F, FH, FW, di, df, out)
    assert rel_err(out.read(), want) <= TOL
    base = rng.random(want.shape, dtype=np.float32)
    out.write(base)
    ops.conv2_nhwc(gpu_ctx, N, H, W, C, F, FH, FW, di, df, out, accumulate=True)
    assert rel_err(out.read(), refcpu.conv2_nhwc(img, flt, out=base.copy())) <= TOL


def test_conv1_known_answer(gpu_ctx):
    img = np.array([1, 2, 3, 2, 1, 0, -1], dtype=np.float32).reshape(1, 1, 7, 1)
    flt = np.array([1, 2, 3], dtype=np.float32).reshape(1, 1, 3, 1)
    di, df = dev(gpu_ctx, img), dev(gpu_ctx, flt)
    out = gpu_ctx.allocTensor((1, 1, 5, 1))
    ops.conv2_nhwc(gpu_ctx, 1, 1, 7, 1, 1, 1, 3, di, df, out)
    assert np.array_equal(out.read().reshape(-1), np.array([14, 14, 10, 4, -2], dtype=np.float32))


def test_runtime_mirror(gpu_ctx):
    """gpu.nim:24-52 through the C ABI: the __main__ smoke of cl.nim:209-245 (vector add) in HIP."""
    devices = eg.listDevices()
    assert devices and devices[0].isGpu and "gfx" in devices[0].version
    src = r'''
    extern "C" __global__ void add(const float* a, const float* b, float* c, long n) {
      long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
      if (i < n) c[i] = a[i] + b[i];
    }'''
    ctx = gpu_ctx
    a, b, c = (ctx.allocBuffer(4 * 32) for _ in range(3))
    data = np.arange(32, dtype=np.float32)
    a.write(data)
    b.write(data)
    kernel = ctx.compile("add", src)
    kernel.arg(0, a).arg(1, b).arg(2, c).arg(3, 32).run([1], [32])
    assert np.array_equal(c.read(), data * 2)
    with pytest.raises(eg.GpuError):       # cl.nim:112-113
        a.write(np.zeros(31, dtype=np.float32))
    with pytest.raises(eg.GpuError):       # cl.nim:163-171: build log in the error
        ctx.compile("broken", "extern \"C\" __global__ void broken() { this is not C++ }")
    with pytest.raises(eg.GpuError):       # cl.nim:191-194
        kernel.run([], [])
    c.fill(3.0)
    assert np.array_equal(c.read(), np.full(32, 3.0, dtype=np.float32))


@pytest.mark.parametrize("seed", range(60))
def test_random_contraction_shapes(gpu_ctx, seed):
    # tile choice, split-K (even and uneven), ragged M / N / K on the DMA loop, unaligned operands:
    # random problems against a float64 product (the oracle's loop nest is too slow for the long-K cases)
    rng = np.random.default_rng(1000 + seed)
    M = int(rng.choice([1, 7, 33, 64, 100, 257, 784, 1000, 2048]))
    N = int(rng.choice([1, 4, 10, 32, 65, 130, 512, 1000]))
    K = int(rng.choice([1, 3, 16, 50, 400, 784, 4096, 20000, 65536]))
    if M * N * K > 3e9:
        K = max(1, int(3e9 // (M * N)))
    ta, tb = bool(rng.integers(2)), bool(rng.integers(2))
    acc, bias = bool(rng.integers(2)), bool(rng.integers(2))
    a = (rng.random((K, M) if ta else (M, K), dtype=np.float32) - 0.5).astype(np.float32)
    b = (rng.random((N, K) if tb else (K, N), dtype=np.float32) - 0.5).astype(np.float32)
    c0 = rng.random((M, N), dtype=np.float32) if acc else np.zeros((M, N), dtype=np.float32)
    bv = (rng.random((N,), dtype=np.float32) - 0.5).astype(np.float32) if bias else None
    want = (a.T if ta else a).astype(np.float64) @ (b.T if tb else b).astype(np.float64) + c0
    if bias:
        want = want + bv
    da, db, dc = dev(gpu_ctx, a), dev(gpu_ctx, b), dev(gpu_ctx, c0)
    dbias = dev(gpu_ctx, bv) if bias else None
    ops.sgemm(gpu_ctx, M, N, K, da, a.shape[1], db, b.shape[1], dc, N, ta, tb, acc, dbias)
    got = dc.read()
    # error budget: every output is a sum of K products of magnitude <= 0.25
    scale = max(np.abs(want).max(), 0.25 * np.sqrt(K) * 0.3)
    assert np.abs(got - want).max() <= TOL * scale, (M, N, K, ta, tb, acc, bias)


@pytest.mark.parametrize("case", [(4100, 4100, 512, False, False, False, False), (4352, 4100, 257, False, True, True, True),
                                  (4100, 4200, 300, True, False, True, False), (4099, 4097, 1000, False, False, False, True)])
def test_tail_tiles_are_cut_along_k(gpu_ctx, case):
    """More tiles than block slots with a short last round (289 tiles of 256 x 256 on 256 CUs): the last
    round's tiles are cut into k-slices (GemmArgs::tail_tiles), folded by a second pass.  Ragged M / N / K,
    transposed operands, bias and accumulation against a float64 product; deterministic."""
    M, N, K, ta, tb, acc, bias = case
    rng = np.random.default_rng(M + K)
    a = (rng.random((K, M) if ta else (M, K), dtype=np.float32) - 0.5).astype(np.float32)
    b = (rng.random((N, K) if tb else (K, N), dtype=np.float32) - 0.5).astype(np.float32)
    c0 = rng.random((M, N), dtype=np.float32) if acc else np.zeros((M, N), dtype=np.float32)
    bv = (rng.random((N,), dtype=np.float32) - 0.5).astype(np.float32) if bias else None
    want = (a.T if ta else a).astype(np.float64) @ (b.T if tb else b).astype(np.float64) + c0
    if bias:
        want = want + bv
    da, db = dev(gpu_ctx, a), dev(gpu_ctx, b)
    dbias = dev(gpu_ctx, bv) if bias else None
    outs = []
    for _ in range(2):
        dc = dev(gpu_ctx, c0)
        ops.sgemm(gpu_ctx, M, N, K, da, a.shape[1], db, b.shape[1], dc, N, ta, tb, acc, dbias)
        outs.append(dc.read())
        dc.buffer.dealloc()
    scale = max(np.abs(want).max(), 0.25 * np.sqrt(K) * 0.3)
    assert np.abs(outs[0] - want).max() <= TOL * scale
    assert np.array_equal(outs[0], outs[1])


@pytest.mark.parametrize("case", [
    # 64 x 64 tiles with 32-deep k-tiles (one block per CU), whole and ragged in every direction
    (1024, 1024, 1024, False, False, False, False), (1000, 1000, 1000, False, False, True, True),
    (1000, 1000, 1000, True, False, False, False), (960, 1008, 1000, False, True, False, True),
    (300, 700, 900, False, False, True, False), (1536, 1536, 300, True, True, False, False),
    # slices planned in 16-deep k-tiles, run in 32-deep ones: no empty slice, no k counted twice
    (64, 1000, 784, False, False, True, True), (128, 128, 4000, False, False, False, False),
    # 64 x 64 tiles four per CU (16-deep k-tiles), several rounds
    (2560, 2560, 256, False, False, False, True), (3000, 1500, 200, True, False, True, False),
    # whole 256 x 256 tiles + remainder rows + remainder columns as contractions of their own, K ending inside a k-tile
    (4100, 4100, 260, False, False, True, True), (4104, 4096, 100, True, False, False, False),
    (4096, 4128, 72, False, True, True, False), (4100, 4100, 260, True, True, False, True),
    # one round of 96 x 96 tiles, four waves per 96 x 32 sub-tile (round 5): every layout, accumulate and bias, 193 .. 256 tiles
    (1536, 1536, 1536, False, False, False, False), (1536, 1536, 640, True, False, True, True), (1440, 1536, 512, False, True, False, True),
    (1536, 1248, 1024, True, True, True, False), (1344, 1440, 576, False, False, True, True),
    # skinny outputs next to the wide-tile model's boundary
    (8192, 128, 512, False, False, False, False), (100, 8192, 512, False, False, True, True), (65, 65, 5000, True, False, False, False)])
def test_mid_size_and_remainder_contractions(gpu_ctx, case):
    """The tile / slice choices of the calibrated time model (gemm_f32_mfma.hip: wide_tile_time) and the
    remainder split of large ragged outputs, against a float64 product; two runs bit-identical."""
    M, N, K, ta, tb, acc, bias = case
    rng = np.random.default_rng(7 * M + 3 * N + K)
    a = (rng.random((K, M) if ta else (M, K), dtype=np.float32) - 0.5).astype(np.float32)
    b = (rng.random((N, K) if tb else (K, N), dtype=np.float32) - 0.5).astype(np.float32)
    c0 = rng.random((M, N), dtype=np.float32) if acc else np.zeros((M, N), dtype=np.float32)
    bv = (rng.random((N,), dtype=np.float32) - 0.5).astype(np.float32) if bias else None
    want = (a.T if ta else a).astype(np.float64) @ (b.T if tb else b).astype(np.float64) + c0
    if bias:
        want = want + bv
    da, db = dev(gpu_ctx, a), dev(gpu_ctx, b)
    dbias = dev(gpu_ctx, bv) if bias else None
    outs = []
    for _ in range(2):
        dc = dev(gpu_ctx, c0)
        ops.sgemm(gpu_ctx, M, N, K, da, a.shape[1], db, b.shape[1], dc, N, ta, tb, acc, dbias)
        outs.append(dc.read())
        dc.buffer.dealloc()
    scale = max(np.abs(want).max(), 0.25 * np.sqrt(K) * 0.3)
    assert np.abs(outs[0] - want).max() <= TOL * scale, case
    assert np.array_equal(outs[0], outs[1])


@pytest.mark.parametrize("shape", [(784, 512, 16384), (785, 512, 8192), (272, 256, 32768), (260, 512, 20000 // 16 * 16), (1040, 768, 8192),
                                   (288, 256, 16384), (513, 256, 16384)])
def test_extra_rows_ride_along_with_the_last_tile_row(gpu_ctx, shape):
    """TN products (weight gradients: K = the batch) whose M is 1 .. 32 rows beyond whole 256-row tiles: the blocks of the
    last tile row carry the extra rows as a ninth accumulator block (GemmArgs::x_rows) instead of a ragged tile row.
    Against the float64 product, and bit-identical from run to run (fixed-order second pass)."""
    m, n, k = shape
    rng = np.random.default_rng(m + k)
    a = (rng.random((k, m), dtype=np.float32) - 0.5).astype(np.float32)
    b = (rng.random((k, n), dtype=np.float32) - 0.5).astype(np.float32)
    da, db = dev(gpu_ctx, a), dev(gpu_ctx, b)
    outs = []
    for _ in range(2):
        dc = gpu_ctx.allocTensor((m, n))
        ops.fill(gpu_ctx, m * n, 7.0, dc)           # must be overwritten completely
        ops.sgemm(gpu_ctx, m, n, k, da, m, db, n, dc, n, trans_a=True)
        outs.append(dc.read())
    want = a.astype(np.float64).T @ b.astype(np.float64)
    assert rel_err(outs[0], want) <= TOL
    assert np.array_equal(outs[0], outs[1])


@pytest.mark.parametrize("case", [(65536, 10, 512, True, False), (4096, 16, 64, False, False), (5003, 1, 1024, True, True),
                                  (8200, 13, 272, False, True), (70001, 10, 512, True, False)])
def test_tall_products_with_a_few_columns(gpu_ctx, case):
    """NN products with N <= 16 (the classifier's last layer: 65536 x 10 x 512): gemm_skinny_nn_kernel streams A once with
    B resident in LDS.  Ragged last row groups, accumulate, bias, lda > K; against the float64 product, and identical
    from run to run; EG_NO_SKINNY_GEMM's path (the tile kernels) agrees to rounding."""
    m, n, k, bias, accumulate = case
    rng = np.random.default_rng(m + n + k)
    lda = k + 8
    a_full = (rng.random((m, lda), dtype=np.float32) - 0.5).astype(np.float32)
    a = a_full[:, :k]
    b = (rng.random((k, n), dtype=np.float32) - 0.5).astype(np.float32)
    c0 = (rng.random((m, n), dtype=np.float32) - 0.5).astype(np.float32)
    bv = (rng.random((n,), dtype=np.float32) - 0.5).astype(np.float32)
    da, db, dbias = dev(gpu_ctx, a_full), dev(gpu_ctx, b), dev(gpu_ctx, bv)
    outs = []
    for _ in range(2):
        dc = dev(gpu_ctx, c0)
        ops.sgemm(gpu_ctx, m, n, k, da, lda, db, n, dc, n, accumulate=accumulate, bias=dbias if bias else None)
        outs.append(dc.read())
    want = a.astype(np.float64) @ b.astype(np.float64)
    if accumulate:
        want = want + c0
    if bias:
        want = want + bv
    assert rel_err(outs[0], want) <= TOL
    assert np.array_equal(outs[0], outs[1])


@pytest.mark.parametrize("mode", ["nn", "nt", "tn", "tt"])
def test_skewed_and_deep_k_loops_equal_the_in_phase_loop_bit_for_bit(gpu_ctx, monkeypatch, mode):
    """Round 4: the odd wave of every SIMD pair runs one k-group late, and long whole-tile products use 32-deep k-tiles —
    same MFMAs on the same accumulators in the same k order.  4096 x 4096 x 2048 (256 x 256 tiles) in the four layouts, with
    a bias and onto an existing C: the default, EG_GEMM_NO_SKEW=1 and EG_GEMM_NO_BK32=1 must agree to the bit, and the
    result with a float64 product on sampled rows."""
    M = N = 4096
    K = 2048
    ta, tb = mode[0] == "t", mode[1] == "t"
    rng = np.random.default_rng(11)
    a = (rng.random((K, M) if ta else (M, K), dtype=np.float32) - 0.5).astype(np.float32)
    b = (rng.random((N, K) if tb else (K, N), dtype=np.float32) - 0.5).astype(np.float32)
    bias = rng.random((N,), dtype=np.float32)
    base = rng.random((M, N), dtype=np.float32)
    da, db, dbias = dev(gpu_ctx, a), dev(gpu_ctx, b), dev(gpu_ctx, bias)
    dc = gpu_ctx.allocTensor((M, N))
    outs = []
    for env in ({}, {"EG_GEMM_NO_SKEW": "1"}, {"EG_GEMM_NO_BK32": "1"}, {"EG_GEMM_NO_SKEW": "1", "EG_GEMM_NO_BK32": "1"}):
        for k in ("EG_GEMM_NO_SKEW", "EG_GEMM_NO_BK32"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        dc.write(base)
        ops.sgemm(gpu_ctx, M, N, K, da, a.shape[1], db, b.shape[1], dc, N, trans_a=ta, trans_b=tb, accumulate=True, bias=dbias)
        outs.append(dc.read())
    for other in outs[1:]:
        assert np.array_equal(outs[0], other)
    rows = np.sort(rng.choice(M, size=16, replace=False))
    a64 = (a.T if ta else a).astype(np.float64)[rows]
    b64 = (b.T if tb else b).astype(np.float64)
    want = base[rows].astype(np.float64) + a64 @ b64 + bias.astype(np.float64)
    assert rel_err(outs[0][rows], want) <= TOL


@pytest.mark.parametrize("mode", ["nn", "nt", "tn", "tt"])
@pytest.mark.parametrize("shape", [(1024, 1024, 1024), (512, 768, 320), (64, 64, 256), (1024, 960, 128),
                                   (1000, 1000, 1000), (960, 1000, 200), (196, 260, 136), (1000, 1024, 132),
                                   (512, 512, 512), (384, 512, 300), (500, 500, 1000), (416, 448, 256),
                                   (256, 256, 512), (320, 288, 1000), (128, 160, 900)])
def test_wave_pair_kernel_against_the_exact_product_and_the_four_wave_kernel(gpu_ctx, monkeypatch, mode, shape):
    """Round 4 (kernels/gemm_f32_pair.hpp): 64 x 64 tiles with at most one block per CU run two waves per sub-tile
    that split every 64-deep k-tile; shapes five to eight are ragged in M, N and / or K (clamped loads, masked stores, a
    zero-filled last k-tile); the last four are small outputs on 32 x 32 tiles with eight waves per tile (128-deep k-tiles).  Four layouts, a bias, onto an existing C; the odd wave's skew changes nothing
    (EG_GEMM_NO_SKEW=1: bit-identical); against the four-wave kernel (EG_GEMM_NO_PAIR=1) the result differs by rounding
    only — an element is the sum of two f32 chains instead of one — and both meet the float64 product of the operands."""
    M, N, K = shape
    ta, tb = mode[0] == "t", mode[1] == "t"
    rng = np.random.default_rng(M + N + K)
    a = (rng.random((K, M) if ta else (M, K), dtype=np.float32) - 0.5).astype(np.float32)
    b = (rng.random((N, K) if tb else (K, N), dtype=np.float32) - 0.5).astype(np.float32)
    bias = rng.random((N,), dtype=np.float32)
    base = rng.random((M, N), dtype=np.float32)
    da, db, dbias = dev(gpu_ctx, a), dev(gpu_ctx, b), dev(gpu_ctx, bias)
    dc = gpu_ctx.allocTensor((M, N))
    outs = []
    for env in ({}, {"EG_GEMM_NO_SKEW": "1"}, {"EG_GEMM_NO_PAIR": "1"}):
        for k in ("EG_GEMM_NO_SKEW", "EG_GEMM_NO_PAIR"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        dc.write(base)
        ops.sgemm(gpu_ctx, M, N, K, da, a.shape[1], db, b.shape[1], dc, N, trans_a=ta, trans_b=tb, accumulate=True, bias=dbias)
        outs.append(dc.read())
    assert np.array_equal(outs[0], outs[1])
    want = base.astype(np.float64) + (a.T if ta else a).astype(np.float64) @ (b.T if tb else b).astype(np.float64) + bias
    assert rel_err(outs[0], want) <= TOL
    assert rel_err(outs[2], want) <= TOL
    assert rel_err(outs[0], outs[2].astype(np.float64)) <= 5e-6


def test_wave_pair_kernel_without_accumulation_and_bias(gpu_ctx):
    """C = A * B on the pair kernel: C starts as NaN, so a chunk the row walk skipped would show."""
    M, N, K = 512, 512, 512
    rng = np.random.default_rng(5)
    a = (rng.random((M, K), dtype=np.float32) - 0.5).astype(np.float32)
    b = (rng.random((K, N), dtype=np.float32) - 0.5).astype(np.float32)
    dc = gpu_ctx.allocTensor((M, N))
    dc.write(np.full((M, N), np.nan, dtype=np.float32))
    ops.sgemm(gpu_ctx, M, N, K, dev(gpu_ctx, a), K, dev(gpu_ctx, b), N, dc, N)
    assert rel_err(dc.read(), a.astype(np.float64) @ b.astype(np.float64)) <= TOL


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["nn", "nt", "tn", "tt"])
@pytest.mark.parametrize("shape,per_cu", [((1792, 1792, 512), 0), ((1152, 1216, 384), 0), ((1280, 1280, 1056), 4), ((1344, 1280, 288), 2),
                                          ((1664, 1728, 320), 3), ((2112, 2048, 256), 4)])
def test_stream_k_blocks_against_the_exact_product_and_one_block_per_tile(gpu_ctx, monkeypatch, mode, shape, per_cu):
    """Round 6 (gemm_f32_mfma.hpp gemm_streamk_kernel): persistent blocks share the (tile, k-tile) space of a 64 x 64 launch
    evenly; cut tiles are folded in k order by gemm_streamk_fixup_kernel.  Forced on here whatever the balance model says
    (EG_STREAMK_MIN_RATIO=0; per_cu > 0: the hybrid form with that many blocks per CU — whole tiles first, the rest shared):
    four layouts, a bias, onto an existing C and into a NaN-filled one; against the float64 product, against one block per
    tile (EG_GEMM_NO_STREAMK=1: the same k order inside a piece, pieces added in k order — rounding only), and twice in a row
    (fixed order: not one bit may differ from run to run)."""
    M, N, K = shape
    ta, tb = mode[0] == "t", mode[1] == "t"
    rng = np.random.default_rng(M + N + K)
    a = (rng.random((K, M) if ta else (M, K), dtype=np.float32) - 0.5).astype(np.float32)
    b = (rng.random((N, K) if tb else (K, N), dtype=np.float32) - 0.5).astype(np.float32)
    bias = rng.random((N,), dtype=np.float32)
    base = rng.random((M, N), dtype=np.float32)
    da, db, dbias = dev(gpu_ctx, a), dev(gpu_ctx, b), dev(gpu_ctx, bias)
    dc = gpu_ctx.allocTensor((M, N))
    monkeypatch.setenv("EG_STREAMK_MIN_RATIO", "0")
    if per_cu:
        monkeypatch.setenv("EG_STREAMK_BLOCKS_PER_CU", str(per_cu))
    outs = []
    for off in (False, False, True):
        if off:
            monkeypatch.setenv("EG_GEMM_NO_STREAMK", "1")
        dc.write(base)
        ops.sgemm(gpu_ctx, M, N, K, da, a.shape[1], db, b.shape[1], dc, N, trans_a=ta, trans_b=tb, accumulate=True, bias=dbias)
        outs.append(dc.read())
    monkeypatch.delenv("EG_GEMM_NO_STREAMK")
    exact = (a.T if ta else a).astype(np.float64) @ (b.T if tb else b).astype(np.float64)
    assert np.array_equal(outs[0], outs[1])
    assert rel_err(outs[0], base.astype(np.float64) + exact + bias) <= TOL
    assert rel_err(outs[0], outs[2].astype(np.float64), what="stream-K vs one block per tile") <= 5e-6
    dc.write(np.full((M, N), np.nan, dtype=np.float32))         # a chunk the blocks or the fix-up skipped would show
    ops.sgemm(gpu_ctx, M, N, K, da, a.shape[1], db, b.shape[1], dc, N, trans_a=ta, trans_b=tb)
    assert rel_err(dc.read(), exact) <= TOL
