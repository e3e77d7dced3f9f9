"""Pins the oracle (oracle/refcpu.c) against the known-answer vectors of the reference's own tests.

Each case cites the reference test it is transcribed from (paths relative to the exprgrad repo).
Comparison is exact `==`, as in the reference (small integer-valued float32 data).
"""
import numpy as np


def f32(x):
    return np.array(x, dtype=np.float32)


def test_matmul_2x3_3x2(refcpu):
    # tests/test_model.nim:37-44, tests/test_tensors.nim:20-23,69, examples/matmul/matmul.nim:22-30
    a = f32([[1, 2, 3], [4, 5, 6]])
    b = f32([[1, 2], [3, 4], [5, 6]])
    assert np.array_equal(refcpu.sgemm(a, b), f32([[22, 28], [49, 64]]))


def test_matmul_3x2_2x3(refcpu):
    # examples/matmul/matmul.nim:22-30 (the other product order, vs the host triple loop tensors.nim:248-256)
    a = f32([[1, 2], [3, 4], [5, 6]])
    b = f32([[1, 2, 3], [4, 5, 6]])
    assert np.array_equal(refcpu.sgemm(a, b), f32([[9, 12, 15], [19, 26, 33], [29, 40, 51]]))


def test_matmul_talks(refcpu):
    # tests/test_talks.nim:21-37
    a = f32([[1, 2], [3, 4]])
    b = f32([[1, 2, 3], [4, 5, 6]])
    assert np.array_equal(refcpu.sgemm(a, b), f32([[9, 12, 15], [19, 26, 33]]))


def test_relu(refcpu):
    # tests/test_model.nim:46-54 — select(0 < x, x, 0): same values as the layer's `>=` form on this data
    x = f32([[0, -1, 10], [-20, 0.1, -0.1]])
    assert np.array_equal(refcpu.map_("relu", x), f32([[0, 0, 10], [0, 0.1, 0]]))


def test_leaky_relu_gpu_values(refcpu):
    # tests/test_gpu.nim:238-246 — select(x > 0, x, 0.01 x)
    x = f32([1, 2, -1, -2, 0, 3])
    got = refcpu.map_("leaky_relu", x, 0.01)
    want = f32([1, 2, f32(-1) * f32(0.01), f32(-2) * f32(0.01), 0, 3])
    assert np.array_equal(got, want)


def test_sum_of_squares_loss(refcpu):
    # tests/test_model.nim:56-69 — loss[0] += sq(pred - labels)
    pred = f32([[1, 2], [3, 4]])
    labels = f32([[4, 3], [2, 1]])
    d = pred - labels
    assert np.array_equal(refcpu.total(d * d), f32([20]))
    d0 = pred - pred
    assert np.array_equal(refcpu.total(d0 * d0), f32([0]))


def test_sum_positive(refcpu):
    # tests/test_talks.nim:62-72
    x = f32([1, -2, -3, 4, 5, -6])
    assert np.array_equal(refcpu.total(refcpu.map_("relu", x)), f32([10]))


def test_linear_layer(refcpu):
    # tests/test_talks.nim:83-96 — x*W + b
    x = f32([[0, 0], [1, 0], [0, 1], [1, 1], [1, 2]])
    w = f32([[2], [3]])
    b = f32([1])
    out = refcpu.bias_add(b, refcpu.sgemm(x, w))
    assert np.array_equal(out.reshape(-1), f32([1, 3, 4, 6, 9]))


def test_multiply_and_square(refcpu):
    # tests/test_talks.nim:98-122
    a = f32([[1, 2], [3, 4]])
    b = f32([[1], [2]])
    c = refcpu.sgemm(a, b)
    assert np.array_equal(c.reshape(-1), f32([5, 11]))
    assert np.array_equal((c * c).reshape(-1), f32([25, 121]))


def test_conv1_as_conv2(refcpu):
    # tests/test_model.nim:91-97 — R[x] += I[x+dx]*F[dx]; expressed as a 1 x W x 1 image, 1 x 3 x 1 filter
    image = f32([1, 2, 3, 2, 1, 0, -1]).reshape(1, 1, 7, 1)
    flt = f32([1, 2, 3]).reshape(1, 1, 3, 1)
    out = refcpu.conv2_nhwc(image, flt)
    assert np.array_equal(out.reshape(-1), f32([14, 14, 10, 4, -2]))


def test_gradient_contractions_against_definition(refcpu):
    # passes.nim:519-549: gradA[y,it] += g[y,x]*b[it,x] ; gradB[it,x] += a[y,it]*g[y,x]
    a = f32([[1, 2, 3], [4, 5, 6]])
    b = f32([[1, 2], [3, 4], [5, 6]])
    g = f32([[1, -1], [2, 0.5]])
    ga = refcpu.sgemm(g, b, trans_b=True)
    gb = refcpu.sgemm(a, g, trans_a=True)
    assert np.array_equal(ga, g @ b.T)
    assert np.array_equal(gb, a.T @ g)


def test_map_grad_closed_forms(refcpu):
    # tests/test_model.nim:294-334: d sin = cos, d exp = exp (libm on both sides, 17 points in [-8, 8])
    x = np.linspace(-8, 8, 17).astype(np.float32)
    ones = np.ones_like(x)
    # numpy's float32 cos is its own SIMD routine, not glibc cosf: compare to 1 ulp
    assert np.allclose(refcpu.map_grad("sin", x, ones), np.cos(x.astype(np.float64)), rtol=2e-7, atol=1e-7)
    assert np.allclose(refcpu.map_grad("exp", x, ones), np.exp(x.astype(np.float64)), rtol=1e-6)
    # sigmoid'(x) = s (1 - s); tanh'(x) = 1 - tanh^2 — closed forms, loose (different formula, same function)
    s = 1 / (1 + np.exp(-x.astype(np.float64)))
    assert np.allclose(refcpu.map_grad("sigmoid", x, ones), s * (1 - s), rtol=2e-5, atol=1e-7)
    xs = np.linspace(-4, 4, 17).astype(np.float32)
    assert np.allclose(refcpu.map_grad("tanh", xs, np.ones_like(xs)), 1 - np.tanh(xs.astype(np.float64)) ** 2,
                       rtol=2e-5, atol=1e-6)


def test_accumulate_semantics(refcpu):
    # model.nim:295-300 + passes.nim:888-897: kernels accumulate into whatever the result holds
    a = f32([[1, 2], [3, 4]])
    out = np.ones((2, 2), dtype=np.float32)
    refcpu.sgemm(a, a, out=out)
    assert np.array_equal(out, a @ a + 1)


def test_thread_policy(refcpu):
    # passes.nim:2415-2437 / model.nim:116-121: 256^3 matmul row = 256*257 work units -> 1 thread;
    # 4096^3 -> all cores
    assert refcpu.thread_count(256, 256 * 257, 8) == 1
    assert refcpu.thread_count(4096, 4096 * 4097, 8) == 8


def test_threaded_equals_serial(refcpu):
    rng = np.random.default_rng(0)
    a = rng.random((37, 19), dtype=np.float32)
    b = rng.random((19, 23), dtype=np.float32)
    assert np.array_equal(refcpu.sgemm(a, b, threads=4), refcpu.sgemm(a, b, threads=1))
    img = rng.random((2, 9, 8, 3), dtype=np.float32)
    flt = rng.random((4, 3, 3, 3), dtype=np.float32)
    assert np.array_equal(refcpu.conv2_nhwc(img, flt, threads_n=2, threads_y=3), refcpu.conv2_nhwc(img, flt))
