"""compile[float64] on the GPU (model.nim:253-260: toScalarType(float64) = Scalar64, every tensor a Tensor[float64]).

The reference instantiates every kernel of a program over the model's scalar type; several of its own tests
(tests/test_model.nim:129-167: singleWrite, shape, dimensions, extern) and its conv2 benchmark
(benchmarks/conv2/conv2.nim:134-138) compile for float64.  Here: the header line of the kernel-description text says
`kd 1 f64`, contractions run on the float64 matrix cores (eg_dgemm), everything else as generated kernels over `double`.

Compared against the ORACLE in its compile[float64] form (oracle/refinterp.c "_c64": the reference's loop nests with
double registers, double constants, libm's double functions) on the same inputs.  Bound: 1e-12 relative to the largest
magnitude of the compared tensor (float64 unit roundoff 1.1e-16; the sums here have at most a few thousand terms, and the
device's exp / log / sin are 1-2 ulp implementations) — stated per test where it differs.
"""
import ctypes

import numpy as np
import pytest

import refcases
from exprgrad_amd import _lib, dsl, examples, layers
from exprgrad_amd import model as egm

pytestmark = pytest.mark.gpu
TOL64 = 1e-12
GOLDEN = refcases.load_golden()


def rel(got, want):
    want = np.asarray(want, np.float64)
    scale = max(float(np.max(np.abs(want))) if want.size else 0.0, 1e-300)
    return float(np.max(np.abs(np.asarray(got, np.float64) - want))) / scale if want.size else 0.0


def text64(graphs):
    prog = dsl.to_program(*graphs)
    prog.scalar = "f64"
    return prog.to_text()


# ---- group 2: eg_dgemm -------------------------------------------------------------------------------------------------
def dgemm(ctx, a, b, ta, tb, M, N, K, c0=None, bias=None):
    """eg_dgemm on the library's own buffers (group 1 of the C ABI: allocBuffer / write / readInto)."""
    def put(x):
        buf = ctx.allocBuffer(x.nbytes)
        buf.write(np.ascontiguousarray(x, dtype=np.float64))
        return buf
    A, B = put(a), put(b)
    C = put(c0 if c0 is not None else np.full((M, N), np.nan))
    bias_b = put(bias) if bias is not None else None
    _lib.call("eg_dgemm", ctx.handle, int(ta), int(tb), M, N, K, ctypes.c_void_p(A.ptr), a.shape[1], ctypes.c_void_p(B.ptr), b.shape[1],
              ctypes.c_void_p(C.ptr), N, 1 if c0 is not None else 0, ctypes.c_void_p(bias_b.ptr) if bias is not None else None)
    out = np.empty((M, N), dtype=np.float64)
    C.readInto(out)
    for buf in (A, B, C, bias_b):
        if buf is not None:
            buf.dealloc()
    return out


@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize("M,N,K", [(1, 1, 1), (16, 16, 4), (64, 64, 16), (65, 63, 17), (128, 128, 128), (200, 300, 77), (257, 129, 1000),
                                   (40, 24, 5000), (1024, 1024, 64), (3, 700, 33)])
def test_dgemm_against_numpy(gpu_ctx, ta, tb, M, N, K):
    rng = np.random.default_rng(M * 7 + N * 3 + K + ta * 2 + tb)
    a = rng.standard_normal((K, M) if ta else (M, K))
    b = rng.standard_normal((N, K) if tb else (K, N))
    want = (a.T if ta else a) @ (b.T if tb else b)
    mags = np.abs(a.T if ta else a) @ np.abs(b.T if tb else b)
    got = dgemm(gpu_ctx, a, b, ta, tb, M, N, K)
    # every element within a few roundoffs of the sum of the magnitudes of its terms (order of summation differs from BLAS)
    assert np.all(np.abs(got - want) <= 4e-16 * np.sqrt(K) * mags + 1e-300), float(np.max(np.abs(got - want) / mags))


def test_dgemm_accumulate_and_bias(gpu_ctx):
    rng = np.random.default_rng(5)
    M, N, K = 100, 90, 300
    a, b = rng.standard_normal((M, K)), rng.standard_normal((K, N))
    c0, bias = rng.standard_normal((M, N)), rng.standard_normal(N)
    got = dgemm(gpu_ctx, a, b, 0, 0, M, N, K, c0=c0, bias=bias)
    assert rel(got, c0 + a @ b + bias[None, :]) <= TOL64
    # sliced product (few tiles, long K): the slabs are summed in a fixed order — two runs agree to the bit
    a, b = rng.standard_normal((64, 40000)), rng.standard_normal((40000, 48))
    g1 = dgemm(gpu_ctx, a, b, 0, 0, 64, 48, 40000, c0=np.ones((64, 48)), bias=np.arange(48.0))
    g2 = dgemm(gpu_ctx, a, b, 0, 0, 64, 48, 40000, c0=np.ones((64, 48)), bias=np.arange(48.0))
    assert np.array_equal(g1, g2)
    assert rel(g1, 1.0 + a @ b + np.arange(48.0)[None, :]) <= TOL64


def test_dgemm_identity_with_asymmetric_operand(gpu_ctx):
    """A = I against an asymmetric B: a transposed or row-permuted C / D map (the float64 MFMA has its own) would show."""
    n = 96
    b = np.arange(n * n, dtype=np.float64).reshape(n, n) * 1.25 + 0.5
    for ta, tb in ((0, 0), (1, 0), (0, 1), (1, 1)):
        got = dgemm(gpu_ctx, np.eye(n), b, ta, tb, n, n, n)
        assert np.array_equal(got, b.T if tb else b), (ta, tb)


# ---- group 3: the reference's known answers, compiled for float64 ---------------------------------------------------------
@pytest.mark.parametrize("name", sorted(GOLDEN))
def test_known_answers_f64(gpu_ctx, name):
    """Every known-answer case of the reference's tests (tests/golden/known_answers.json), compile[float64]: against the
    oracle's float64 form on the same inputs, and against the transcribed answers where those are exact in both types."""
    from oracle import kd
    graphs = refcases.BUILDERS[name]()
    model = egm.compile(*graphs, gpu=gpu_ctx, dtype=np.float64)
    assert model.dtype == np.float64
    ref = kd.Model(text64(refcases.BUILDERS[name]()))
    assert ref.c64
    for tid in sorted(ref.params):      # parameters: the same explicit values on both sides
        v = np.random.default_rng(tid).random(ref.params[tid].shape) * 0.2 - 0.1
        ref.params[tid][...] = v
        model.params[tid] = v
    for c in GOLDEN[name]["calls"]:
        inputs = {k: refcases.arr(v).astype(np.float64) for k, v in c["inputs"].items()}
        got = model.call(c["target"], inputs)
        want = ref.call(c["target"], inputs)
        assert got.dtype == np.float64 and list(got.shape) == list(want.shape)
        finite = np.isfinite(want)
        assert np.array_equal(np.isfinite(got), finite)
        assert np.allclose(got[finite], want[finite], rtol=1e-12, atol=1e-13), (name, got, want)
        if not ref.params and c["mode"] != "sumsq" and not name.startswith("derive/"):
            # integer-valued answers are the same numbers in float32 and float64 (singleWrite / shape / dimensions / extern
            # are float64 tests in the reference: tests/test_model.nim:129-167)
            exp = refcases.arr(c["expected"]).astype(np.float64)
            if np.array_equal(exp, np.round(exp)):
                assert np.array_equal(got, exp), (name, got, exp)
    model.close()


def test_constants_are_doubles(gpu_ctx):
    """`x * 0.1` under compile[float64] multiplies by the double 0.1 (const_real(double type, v), llvmgen.nim:215-216),
    not by float32(0.1) widened — the difference is 1.5e-9 relative, far above the bound of this file."""
    model = egm.compile(*refcases.extern(0.1)(), gpu=gpu_ctx, dtype=np.float64)
    x = np.arange(1.0, 7.0).reshape(2, 3)
    got = model.call("y", {"x": x})
    assert np.array_equal(got, x * 0.1)
    assert not np.array_equal(got, x * np.float64(np.float32(0.1)))
    model.close()


def test_scalar_type_of_the_entry_points(gpu_ctx):
    """A Tensor[float32] does not type-check against a Model[float64] in the reference; here the float32-typed entry
    points refuse a float64 model and the _f64 ones a float32 model."""
    m64 = egm.compile(*examples.matmul_graph(), gpu=gpu_ctx, dtype=np.float64)
    m32 = egm.compile(*examples.matmul_graph(), gpu=gpu_ctx)
    a = np.ones((4, 4), dtype=np.float32)
    shape = (ctypes.c_int64 * 2)(4, 4)
    with pytest.raises(_lib.GpuError, match="float64"):
        _lib.call("eg_model_set_input_host", m64.handle, b"a", a.ctypes.data_as(ctypes.c_void_p), 2, shape)
    with pytest.raises(_lib.GpuError, match="float32"):
        _lib.call("eg_model_set_input_host_f64", m32.handle, b"a", a.astype(np.float64).ctypes.data_as(ctypes.c_void_p), 2, shape)
    with pytest.raises(ValueError):
        egm.compile(*examples.matmul_graph(), gpu=gpu_ctx, dtype=np.float16)      # model.nim:259
    assert _lib.call("eg_model_scalar_bytes", m64.handle) == 8 and _lib.call("eg_model_scalar_bytes", m32.handle) == 4
    m64.close()
    m32.close()


# ---- training steps from identical state ------------------------------------------------------------------------------------
def step_pair(gpu_ctx, graphs_fn, seed, lo=-0.3, hi=0.3):
    from oracle import kd
    gpu = egm.compile(*graphs_fn(), gpu=gpu_ctx, dtype=np.float64)
    ref = kd.Model(text64(graphs_fn()))
    rng = np.random.default_rng(seed)
    for tid in sorted(ref.params):
        v = lo + (hi - lo) * rng.random(ref.params[tid].shape)
        ref.params[tid][...] = v
        gpu.params[tid] = v
    return gpu, ref, rng


def compare_state(gpu, ref, tol=TOL64, what=""):
    for tid in sorted(ref.params):
        assert rel(gpu.params[tid], ref.params[tid]) <= tol, (what, "param", tid, rel(gpu.params[tid], ref.params[tid]))
    for tid in sorted(ref.caches):
        assert rel(gpu.caches[tid], ref.caches[tid]) <= tol, (what, "cache", tid)


def test_xor_training_f64(gpu_ctx):
    """tests/test_model.nim:169-194 (xor) compiled for float64: 50 steps next to the oracle, then both have learnt XOR."""
    gpu, ref, _ = step_pair(gpu_ctx, lambda: examples.xor_from_scratch(rate=0.5), 3, -1.0, 1.0)
    x = np.array([[0, 0], [0, 1], [1, 0], [1, 1]], dtype=np.float64)
    y = np.array([[0], [1], [1], [0]], dtype=np.float64)
    for step in range(50):
        gpu.apply("train", {"x": x, "y": y})
        ref.apply("train", {"x": x, "y": y})
        compare_state(gpu, ref, 1e-11, f"step {step}")       # (50 steps of drift: the states are NOT re-synchronised)
    assert rel(gpu.call("loss", {"x": x, "y": y}), ref.call("loss", {"x": x, "y": y})) <= 1e-10


@pytest.mark.parametrize("batch", [7, 64, 300])
def test_dense_softmax_step_f64(gpu_ctx, batch):
    """configs[4]'s network (dense -> relu -> dense -> softmax -> crossEntropy -> gradientDescent), small, float64:
    loss, every gradient and the updated parameters against the oracle."""
    gpu, ref, rng = step_pair(gpu_ctx, lambda: examples.dense_softmax_net(48, 40, 10, rate=0.05), batch)
    x = rng.standard_normal((batch, 48))
    y = np.eye(10)[rng.integers(0, 10, batch)]
    for step in range(3):
        assert rel(gpu.call("loss", {"x": x, "y": y}), ref.call("loss", {"x": x, "y": y})) <= TOL64
        gpu.apply("train", {"x": x, "y": y})
        ref.apply("train", {"x": x, "y": y})
        for _, gt in ref.prog.param_grads["train"]:
            assert rel(gpu.read_tensor("train", gt), ref.last[gt]) <= TOL64, ("gradient", gt)
        compare_state(gpu, ref, TOL64, f"step {step}")
    assert "gemm" in gpu.emit_ir()           # the contractions are library contractions (eg_dgemm), not generated loops
    gpu.close()


def test_conv_pool_adam_f64(gpu_ctx):
    """reshape -> conv2 -> leakyRelu -> maxpool2 (customGrad) -> conv2 -> ... -> dense -> softmax -> crossEntropy -> adam
    (fashion_mnist.nim:39-57, small) in float64: convolutions and their gradients as generated kernels over double,
    adam with epoch()."""
    gpu, ref, rng = step_pair(gpu_ctx, lambda: examples.fashion_mnist_net(eta=0.01, size=12, f1=3, f2=4, classes=5), 11)
    x = rng.random((6, 12 * 12))
    y = np.eye(5)[rng.integers(0, 5, 6)]
    for step in range(3):
        gpu.epoch = step + 1
        ref.epoch = step + 1
        gpu.apply("fit", {"x": x, "y": y})
        ref.apply("fit", {"x": x, "y": y})
        # adam at step t divides by sqrt(v) + 1e-8 with v ~ g^2: relative error of an update ~ that of g; 1e-10 leaves
        # room for gradients that cancel to 1e-2 of their terms
        compare_state(gpu, ref, 1e-10, f"step {step}")
    gpu.close()


def test_fit_equals_apply_per_batch_f64(gpu_ctx):
    """Model.fit (model.nim:413-454) of a float64 model = apply on every batch, to the bit."""
    rng = np.random.default_rng(8)
    x = rng.standard_normal((96, 20))
    y = np.eye(4)[rng.integers(0, 4, 96)]
    a = egm.compile(*examples.dense_softmax_net(20, 16, 4, rate=0.1), gpu=gpu_ctx, dtype=np.float64)
    b = egm.compile(*examples.dense_softmax_net(20, 16, 4, rate=0.1), gpu=gpu_ctx, dtype=np.float64)
    for tid in a.params:
        b.params[tid] = a.params[tid]
    a.fit("train", {"x": x, "y": y}, batch_size=32)
    for i in range(3):
        b.apply("train", {"x": x[32 * i:32 * i + 32], "y": y[32 * i:32 * i + 32]})
    assert a.epoch == 1
    for tid in a.params:
        assert np.array_equal(a.params[tid], b.params[tid])
    a.close()
    b.close()


def test_save_and_load_f64(gpu_ctx, tmp_path):
    """serialize.nim:35: a float64 element is 8 bytes; a saved float64 model comes back as one, bit for bit."""
    a = egm.compile(*examples.dense_softmax_net(6, 5, 3), gpu=gpu_ctx, dtype=np.float64)
    path = tmp_path / "m64.bin"
    a.save(path)
    b = egm.load_model(path, gpu=gpu_ctx)
    assert b.dtype == np.float64
    for tid in a.params:
        assert np.array_equal(a.params[tid], b.params[tid]) and a.params[tid].dtype == np.float64
    x = np.random.default_rng(0).standard_normal((4, 6))
    assert np.array_equal(a.call("predict", {"x": x}), b.call("predict", {"x": x}))
    a.close()
    b.close()


# ---- the reference's conv2 benchmark, its own scalar type -------------------------------------------------------------------
def test_conv2_benchmark_program_f64(gpu_ctx):
    """benchmarks/conv2/conv2.nim:128-138: `result[y, x, filter] ++= image[y + dy, x + dx, chan] * filters[filter, dy, dx, chan]`
    compiled for float64 (8 filters of 3 x 3 x 8 there; a crop of the 960 x 1280 image here) against the oracle."""
    from oracle import kd
    gpu = egm.compile(*examples.conv2_3d(), gpu=gpu_ctx, dtype=np.float64)
    ref = kd.Model(text64(examples.conv2_3d()))
    rng = np.random.default_rng(2)
    image, filters = rng.random((40, 52, 8)), rng.random((8, 3, 3, 8))
    got = gpu.call("conv2", {"image": image, "filters": filters})
    want = ref.call("conv2", {"image": image, "filters": filters})
    assert got.shape == (38, 50, 8) and rel(got, want) <= TOL64
    gpu.close()


def test_matmul_program_f64_full_tiles(gpu_ctx):
    """The matmul program at 512^3 in float64: backend against the oracle's sequential sums (ref_dgemm) — 1e-13 of the
    largest element (K = 512 terms of magnitude <= 1: both orders are within 512 * 1.1e-16 * |terms|)."""
    from oracle import kd
    gpu = egm.compile(*examples.matmul_graph(), gpu=gpu_ctx, dtype=np.float64)
    ref = kd.Model(text64(examples.matmul_graph()))
    rng = np.random.default_rng(4)
    a, b = rng.random((512, 512)) - 0.5, rng.random((512, 512)) - 0.5
    assert rel(gpu.call("c", {"a": a, "b": b}), ref.call("c", {"a": a, "b": b})) <= 1e-13
    gpu.close()


def conv_numpy(image, filters):
    """out[n, y, x, f] = sum_{dy, dx, c} image[n, y + dy, x + dx, c] * filters[f, dy, dx, c] as FH * FW small matrix products."""
    F, FH, FW, C = filters.shape
    Ho, Wo = image.shape[-3] - FH + 1, image.shape[-2] - FW + 1
    out = np.zeros(image.shape[:-3] + (Ho, Wo, F))
    for dy in range(FH):
        for dx in range(FW):
            out += image[..., dy:dy + Ho, dx:dx + Wo, :] @ filters[:, dy, dx, :].T
    return out


@pytest.mark.parametrize("H,W,C,F,FH", [(130, 90, 8, 8, 3), (100, 120, 8, 16, 3), (200, 200, 1, 1, 3), (96, 100, 3, 5, 5)])
def test_direct_convolution_f64(gpu_ctx, H, W, C, F, FH):
    """The shapes of the reference's conv2 benchmark targets (conv2.nim:134-138: 8 -> 8, 8 -> 16, 1 -> 1 filters of 3 x 3) with
    enough pixels for the direct per-pixel kernel over double (kernels/conv2_direct.cpp), against the oracle."""
    from oracle import kd
    gpu = egm.compile(*examples.conv2_3d(), gpu=gpu_ctx, dtype=np.float64)
    ref = kd.Model(text64(examples.conv2_3d()))
    rng = np.random.default_rng(H + C)
    image, filters = rng.random((H, W, C)), rng.random((F, FH, FH, C)) * 4 - 2
    got = gpu.call("conv2", {"image": image, "filters": filters})
    want = ref.call("conv2", {"image": image, "filters": filters})
    assert rel(got, want) <= TOL64
    assert rel(got, conv_numpy(image, filters)) <= TOL64
    gpu.close()


def test_conv2_benchmark_full_size_f64(gpu_ctx):
    """benchmarks/conv2/conv2.nim:330-364 at its own size (960 x 1280 x 8 image, 8 filters of 3 x 3 x 8, float64): 1.2 M
    pixels, against nine float64 matrix products per tap on the host; and linear in the filter bank (a size-independent
    property: conv(a, f + g) = conv(a, f) + conv(a, g) to rounding)."""
    gpu = egm.compile(*examples.conv2_3d(), gpu=gpu_ctx, dtype=np.float64)
    rng = np.random.default_rng(9)
    image = rng.random((960, 1280, 8))
    f, g = rng.random((8, 3, 3, 8)) * 4 - 2, rng.random((8, 3, 3, 8)) * 4 - 2
    out_f = gpu.call("conv2", {"image": image, "filters": f})
    assert out_f.shape == (958, 1278, 8)
    assert rel(out_f, conv_numpy(image, f)) <= 1e-13
    out_g = gpu.call("conv2", {"image": image, "filters": g})
    out_fg = gpu.call("conv2", {"image": image, "filters": f + g})
    assert rel(out_fg, out_f + out_g) <= 1e-13
    gpu.close()


def test_batched_convolution_layer_f64(gpu_ctx):
    """layers.conv2 (dnn.nim:45-49, four-dimensional images) in float64 with its two gradients (generated kernels) in a
    training step: conv2 -> mse -> gradientDescent against the oracle."""
    def net():
        out = layers.conv2(dsl.input("x"), 4, 3, 3, 6).target("predict")
        return [layers.mse(out, dsl.input("y")).target("loss").backprop(layers.gradient_descent(0.05)).target("train")]
    gpu, ref, rng = step_pair(gpu_ctx, net, 21)
    x, y = rng.random((12, 30, 34, 4)), rng.random((12, 28, 32, 6))
    assert rel(gpu.call("predict", {"x": x}), ref.call("predict", {"x": x})) <= TOL64
    for step in range(2):
        gpu.apply("train", {"x": x, "y": y})
        ref.apply("train", {"x": x, "y": y})
        compare_state(gpu, ref, 1e-11, f"step {step}")
    gpu.close()


def test_native_data_parallel_step_f64(gpu_ctx):
    """eg_model_step_dp on a float64 model: the gradient bucket goes through RCCL as float64 (eg_dp_allreduce_sum_f64;
    one rank here, a sum over one rank is the identity), so the step equals apply() bit for bit; and the bucket the C ABI
    hands out is counted in elements (doubles)."""
    from exprgrad_amd.parallel import NativeDataParallel, RcclGroup
    group = RcclGroup(gpu_ctx, RcclGroup.unique_id(), rank=0, world=1)
    whole = egm.compile(*examples.dense_softmax_net(24, 16, 4, rate=0.1), gpu=gpu_ctx, dtype=np.float64)
    split = egm.compile(*examples.dense_softmax_net(24, 16, 4, rate=0.1), gpu=gpu_ctx, dtype=np.float64)
    for tid in whole.params:
        split.params[tid] = whole.params[tid]
    rng = np.random.default_rng(6)
    x, y = rng.standard_normal((50, 24)), np.eye(4)[rng.integers(0, 4, 50)]
    dp = NativeDataParallel(split, "train", group, reduction="mean")
    for _ in range(3):
        whole.apply("train", {"x": x, "y": y})
        dp.step({"x": x, "y": y})
    for tid in whole.params:
        assert np.array_equal(whole.params[tid], split.params[tid]), tid
    _, count = split.grad_bucket("train")
    assert count >= 24 * 16 + 16 + 16 * 4 + 4 and count < 2 * (24 * 16 + 16 + 16 * 4 + 4)
    whole.close()
    split.close()
    group.close()


def test_split_step_with_a_caller_owned_bucket_f64(gpu_ctx):
    """The torch.distributed form of the data-parallel step (parallel.GpuEngine + DataParallel: run_backward | all-reduce of a
    caller-owned bucket tensor | run_update) on a float64 model: the bucket is a float64 torch tensor bound with
    eg_model_bind_grad_bucket (counts in elements), one rank, equal to apply() bit for bit."""
    torch = pytest.importorskip("torch")
    import exprgrad_amd as eg
    from exprgrad_amd.parallel import DataParallel, GpuEngine
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        ctx = eg.newGpuContext(0, stream=stream.cuda_stream)
        whole = egm.compile(*examples.dense_softmax_net(12, 8, 3, rate=0.1), gpu=ctx, dtype=np.float64)
        split = egm.compile(*examples.dense_softmax_net(12, 8, 3, rate=0.1), gpu=ctx, dtype=np.float64)
        for tid in whole.params:
            split.params[tid] = whole.params[tid]
        engine = GpuEngine(split, "train")
        assert engine.bucket.dtype == torch.float64
        dp = DataParallel(engine, reduction="mean")
        rng = np.random.default_rng(1)
        x, y = rng.standard_normal((20, 12)), np.eye(3)[rng.integers(0, 3, 20)]
        for _ in range(3):
            whole.apply("train", {"x": x, "y": y})
            dp.step({"x": x, "y": y})
        stream.synchronize()
        assert float(engine.bucket.abs().sum()) > 0.0          # the gradients really live in the caller's tensor
        for tid in whole.params:
            assert np.array_equal(whole.params[tid], split.params[tid]), tid
        whole.close()
        split.close()


@pytest.mark.parametrize("shape", [(48, 12, 12, 8, 16, 3, 3), (9, 28, 28, 1, 8, 5, 5), (3, 40, 50, 3, 5, 3, 3), (50, 12, 12, 16, 8, 3, 3)])
def test_band_convolution_training_step_f64(gpu_ctx, shape):
    """conv2 and its two gradients of a float64 model on the matrix cores (kernels/conv2_band.cpp, the float64 instantiation):
    conv2 -> mse -> gradientDescent at shapes past the kernels' pixel threshold, gradients and updated parameters against
    the oracle; the launch list names the kernels."""
    N, H, W, C, F, FH, FW = shape

    def net():
        out = layers.conv2(dsl.input("x"), C, FH, FW, F).target("predict")
        return [layers.mse(out, dsl.input("y")).target("loss").backprop(layers.gradient_descent(0.05)).target("train")]
    gpu, ref, rng = step_pair(gpu_ctx, net, sum(shape))
    x, y = rng.random((N, H, W, C)), rng.random((N, H - FH + 1, W - FW + 1, F))
    assert rel(gpu.call("predict", {"x": x}), ref.call("predict", {"x": x})) <= TOL64
    for step in range(2):
        gpu.apply("train", {"x": x, "y": y})
        ref.apply("train", {"x": x, "y": y})
        compare_state(gpu, ref, 1e-11, f"step {step}")
    assert "filter gradient" in gpu.launch_plan("train")
    gpu.close()
