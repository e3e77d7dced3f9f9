"""Special values and empty inputs (missing #3): what the reference's semantics define at
the edges of float32 must come out of the backend the same way.

  * tanh is the naive (e^x - e^-x) / (e^x + e^-x) (layers/dnn.nim:35-40): inf / inf = NaN for
    |x| >~ 88.73, where e^x overflows — restated faithfully, not "stabilised";
  * sigmoid 1 / (1 + e^-x) saturates to exactly 0 and 1 (dnn.nim:32-33);
  * comparisons are ORDERED (llvmgen.nim:212-301: fcmp ole / olt / oeq): false whenever an operand is
    NaN, so relu(NaN) = select(0 <= NaN, NaN, 0) = 0 while leakyRelu(NaN) = select(..) * NaN = NaN;
    -0.0 passes `0 <= x` and stays -0.0;
  * denormal operands and results;
  * empty tensors through every group-2 call and through a model.
The library's hand-written maps are compared with the oracle's restatement (oracle/refcpu.c), the
generated-kernel path (layers through the model compiler) with the oracle's interpreter, bit patterns
included where the value is exact.
"""
import numpy as np
import pytest

import refcases
from conftest import TOL, rel_err
from exprgrad_amd import dsl, layers, ops
from exprgrad_amd import model as egm

pytestmark = pytest.mark.gpu
f32 = np.float32

SPECIALS = np.array([0.0, -0.0, 1.0, -1.0, 87.0, -87.0, 88.0, -88.0, 88.7, -88.7, 88.8, -88.8, 89.0, -89.0, 100.0, -100.0,
                     1e-40, -1e-40, 1.4e-45, 3.0e38, -3.0e38, np.inf, -np.inf, np.nan, 16.6, -16.7, 103.9, -103.9], dtype=f32)


def dev(ctx, a):
    t = ctx.allocTensor(a.shape)
    t.write(a)
    return t


def same_special(got, want, what, tol=TOL, compare=None):
    """NaN where the reference has NaN, the same infinities, zeros where it has zeros, finite
    values within tol of each other relative to the value itself."""
    got, want = np.asarray(got), np.asarray(want)
    assert np.array_equal(np.isnan(got), np.isnan(want)), (what, "NaN positions", got, want)
    assert np.array_equal(np.isinf(got), np.isinf(want)) and np.array_equal(got[np.isinf(want)], want[np.isinf(want)]), (what, "infinities")
    fin = np.isfinite(want)
    zero = fin & (want == 0)
    # The SIGN of a zero is not pinned by the reference: a first writer stores its value (InstrOverwrite,
    # -0.0 stays -0.0) while an accumulating write computes 0 + (-0.0) = +0.0, and which of the two a
    # kernel gets depends on `hasWritten`, which inlineTensorOps shares across targets visited in hash
    # order of their names (passes.nim:927-934).  Zeros must be zeros; either sign is the reference's.
    assert np.all(got[zero] == 0), (what, "zeros")
    nz = fin & (want != 0)
    if compare is not None:      # points where the value is compared; everywhere else only its kind (NaN / inf / zero / finite)
        assert np.all(np.isfinite(got[nz & ~compare])), (what, "finite where the reference is")
        nz = nz & compare
    # denormal results (|want| < 1.18e-38) may differ in the last bits of a 23-bit-or-less significand: absolute 2^-149 steps
    err = np.abs(got[nz].astype(np.float64) - want[nz]) / np.maximum(np.abs(want[nz].astype(np.float64)), 1.2e-38)
    assert err.size == 0 or err.max() <= tol, (what, err.max(), got[nz], want[nz])


@pytest.mark.parametrize("op", ["relu", "leaky_relu", "sigmoid", "tanh", "xor_leaky", "exp", "identity", "scale"])
def test_library_maps_on_special_values(gpu_ctx, refcpu, op):
    x = np.concatenate([SPECIALS, SPECIALS[::-1]])          # 56 values: the 16-byte path and its tail
    param = 0.01 if op != "scale" else -3.0
    din, dout = dev(gpu_ctx, x), gpu_ctx.allocTensor(x.shape)
    ops.map_(gpu_ctx, op, x.size, din, dout, param=param)
    with np.errstate(all="ignore"):
        want = refcpu.map_(op, x, param=param)
    same_special(dout.read(), want, op)
    # the derived gradient kernels (passes.nim:392-505) on the same points, upstream gradient 1 and NaN / inf mixed in
    g = np.where(np.arange(x.size) % 7 == 3, f32(2.0), f32(1.0)).astype(f32)
    dg, dgin = dev(gpu_ctx, g), gpu_ctx.allocTensor(x.shape)
    ops.map_grad(gpu_ctx, op, x.size, din, dg, dgin, param=param)
    with np.errstate(all="ignore"):
        want_g = refcpu.map_grad(op, x, g, param=param)
    # tanh / sigmoid gradients divide by (e^x + e^-x)^2-type terms: beyond |x| = 20 the float32 value is the
    # quotient of an overflowing or fully cancelled pair and its last bits depend on the exp implementation's;
    # there only the kind of the result (NaN / inf / zero / finite) is compared
    compare = np.abs(x) <= 20 if op in ("tanh", "sigmoid") else None
    same_special(dgin.read(), want_g, op + " gradient", compare=compare)


def test_tanh_is_the_naive_formula_and_overflows_to_nan(gpu_ctx, refcpu):
    x = np.array([88.0, 88.7, 88.8, 89.0, 100.0, -89.0, -100.0, np.inf, -np.inf, 20.0, -20.0, 0.0], dtype=f32)
    din, dout = dev(gpu_ctx, x), gpu_ctx.allocTensor(x.shape)
    ops.map_(gpu_ctx, "tanh", x.size, din, dout)
    got = dout.read()
    # e^88.7 is finite (3.3e38), e^88.8 is not: (inf - 0) / (inf + 0) = NaN from there on, in both directions
    assert np.array_equal(np.isnan(got), np.array([0, 0, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0], dtype=bool)), got
    assert got[0] == 1.0 and got[9] == 1.0 and got[10] == -1.0 and got[11] == 0.0
    with np.errstate(all="ignore"):
        same_special(got, refcpu.map_("tanh", x), "tanh")


def layer_net(layer):
    return [layer(dsl.input("x")).target("y")]


@pytest.mark.parametrize("name", ["relu", "leaky_relu", "sigmoid", "tanh", "max0", "select_lt", "select_eq"])
def test_generated_kernels_on_special_values(gpu_ctx, name):
    """The same points through the model compiler (generated HIP from the kernel description) against the
    oracle's interpreter: ordered compares, select, max through select (dsl.nim:135-142)."""
    from oracle import kd
    it = dsl.iters("it")

    def custom(expr):
        def build(x):
            r = dsl.Fun()
            r.raw[it] += expr(x.raw[it])
            r.copy_shape(x)
            return r
        return build

    nets = {
        "relu": layers.relu, "leaky_relu": layers.leaky_relu, "sigmoid": layers.sigmoid, "tanh": layers.tanh,
        "max0": custom(lambda v: dsl.max(v, 0.0)),
        "select_lt": custom(lambda v: dsl.select(v < 1.0, v * 2.0, -v)),
        "select_eq": custom(lambda v: dsl.select(v.eq(0.0), 7.0, v)),
    }
    graphs = lambda: layer_net(nets[name])
    gpu = egm.compile(*graphs(), gpu=gpu_ctx)
    ref = kd.Model(refcases.program_text(graphs()))
    x = np.concatenate([SPECIALS, SPECIALS[::-1], SPECIALS[:4]]).astype(f32)      # 60 values: the four-per-thread path
    with np.errstate(all="ignore"):
        want = ref.call("y", {"x": x})
    same_special(gpu.call("y", {"x": x}), want, name)
    odd = x[:57]                                                                 # not a multiple of four: the scalar path
    with np.errstate(all="ignore"):
        want = ref.call("y", {"x": odd})
    same_special(gpu.call("y", {"x": odd}), want, name + " (scalar path)")
    gpu.close()


def test_nan_and_inf_propagate_through_a_contraction_like_the_reference(gpu_ctx, refcpu):
    rng = np.random.default_rng(0)
    a = rng.random((70, 40), dtype=f32)
    b = rng.random((40, 50), dtype=f32)
    a[3, 5] = np.nan
    a[9, 0] = np.inf
    b[7, 11] = -np.inf
    a[20, 7] = 0.0          # 0 * -inf = NaN in row 20, column 11
    da, db, dc = dev(gpu_ctx, a), dev(gpu_ctx, b), gpu_ctx.allocTensor((70, 50))
    ops.sgemm(gpu_ctx, 70, 50, 40, da, 40, db, 50, dc, 50)
    with np.errstate(all="ignore"):
        want = refcpu.sgemm(a, b)
    got = dc.read()
    assert np.array_equal(np.isnan(got), np.isnan(want))
    assert np.array_equal(np.isinf(got), np.isinf(want)) and np.array_equal(got[np.isinf(want)], want[np.isinf(want)])
    fin = np.isfinite(want)
    assert rel_err(got[fin], want[fin]) <= TOL


@pytest.mark.parametrize("shape", [(64, 64, 64), (300, 200, 1000), (1, 33, 5)])
def test_denormal_operands_through_the_matrix_cores(gpu_ctx, refcpu, shape):
    """Denormal inputs (1e-40 .. 1e-39) times large factors: the products are normal numbers.  MFMA
    must not flush the denormal operand to zero (the reference's CPU arithmetic does not)."""
    M, N, K = shape
    rng = np.random.default_rng(M)
    a = (rng.random((M, K), dtype=np.float64) * 9e-40 + 1e-40).astype(f32)       # every entry denormal
    assert np.all(np.abs(a) < 1.1754944e-38) and np.all(a != 0)
    b = (rng.random((K, N), dtype=f32) * f32(1e30)).astype(f32)
    da, db, dc = dev(gpu_ctx, a), dev(gpu_ctx, b), gpu_ctx.allocTensor((M, N))
    ops.sgemm(gpu_ctx, M, N, K, da, K, db, N, dc, N)
    want = refcpu.sgemm(a, b)
    assert np.all(want > 0)
    assert rel_err(dc.read(), want) <= TOL
    # denormal RESULTS: tiny times tiny-ish stays below 1.18e-38
    b2 = (rng.random((K, N), dtype=f32) * f32(1e-3)).astype(f32)
    a2 = (rng.random((M, K), dtype=f32) * f32(1e-37)).astype(f32)
    da2, db2 = dev(gpu_ctx, a2), dev(gpu_ctx, b2)
    ops.sgemm(gpu_ctx, M, N, K, da2, K, db2, N, dc, N)
    want2 = refcpu.sgemm(a2, b2)
    got2 = dc.read()
    # products of ~1e-40 each round to a multiple of 2^-149 = 1.4e-45 one by one: absolute steps, K of them
    assert np.max(np.abs(got2.astype(np.float64) - want2)) <= K * 1.5e-45 + TOL * np.max(np.abs(want2))


def test_empty_inputs_through_every_library_call(gpu_ctx):
    z = gpu_ctx.allocTensor((16,))          # a real allocation to pass where a pointer is required
    keep = np.arange(16, dtype=f32)
    z.write(keep)
    for (M, N, K) in [(0, 5, 3), (4, 0, 3), (0, 0, 0)]:
        ops.sgemm(gpu_ctx, M, N, K, z, max(K, 1), z, max(N, 1), z, max(N, 1))
    assert np.array_equal(z.read(), keep)                    # nothing written
    # K = 0: an empty sum.  C = 0 (or stays, when accumulating; + bias either way)
    c = gpu_ctx.allocTensor((3, 4))
    c.write(np.full((3, 4), 5.0, f32))
    bias = dev(gpu_ctx, np.array([1, 2, 3, 4], f32))
    ops.sgemm(gpu_ctx, 3, 4, 0, z, 1, z, 4, c, 4)
    assert np.array_equal(c.read(), np.zeros((3, 4), f32))
    c.write(np.full((3, 4), 5.0, f32))
    ops.sgemm(gpu_ctx, 3, 4, 0, z, 1, z, 4, c, 4, accumulate=True, bias=bias)
    assert np.array_equal(c.read(), np.full((3, 4), 5.0, f32) + np.array([1, 2, 3, 4], f32))
    out = gpu_ctx.allocTensor((4,))
    out.write(np.full(4, 9.0, f32))
    ops.colsum(gpu_ctx, 0, 4, z, out)                        # no rows: the sums are zero
    assert np.array_equal(out.read(), np.zeros(4, f32))
    out.write(np.full(4, 9.0, f32))
    ops.colsum(gpu_ctx, 0, 4, z, out, accumulate=True)
    assert np.array_equal(out.read(), np.full(4, 9.0, f32))
    ops.rowsum(gpu_ctx, 4, 0, z, out)                        # rows without columns: zero each
    assert np.array_equal(out.read(), np.zeros(4, f32))
    one = gpu_ctx.allocTensor((1,))
    one.write(np.array([3.0], f32))
    ops.total(gpu_ctx, 0, z, one)
    assert one.read()[0] == 0.0
    for call in (lambda: ops.bias_add(gpu_ctx, 0, 4, bias, z), lambda: ops.bias_add(gpu_ctx, 4, 0, bias, z),
                 lambda: ops.axpy(gpu_ctx, 0, 2.0, z, z), lambda: ops.fill(gpu_ctx, 0, 1.0, z),
                 lambda: ops.map_(gpu_ctx, "tanh", 0, z, z), lambda: ops.map_grad(gpu_ctx, "relu", 0, z, z, z),
                 lambda: ops.conv2_nhwc(gpu_ctx, 0, 8, 8, 4, 4, 3, 3, z, z, z),
                 lambda: ops.conv2_nhwc(gpu_ctx, 2, 8, 8, 4, 0, 3, 3, z, z, z),
                 lambda: ops.conv2_nhwc_grad_image(gpu_ctx, 0, 8, 8, 4, 4, 3, 3, z, z, z)):
        call()
    gpu_ctx.sync()
    assert np.array_equal(z.read(), keep)
    # an empty batch through a filter gradient: every filter weight sees an empty sum
    gf = gpu_ctx.allocTensor((4, 3, 3, 4))
    gf.write(np.full((4, 3, 3, 4), 2.0, f32))
    ops.conv2_nhwc_grad_filter(gpu_ctx, 0, 8, 8, 4, 4, 3, 3, z, z, gf)
    assert not gf.read().any()


def test_empty_batch_through_a_model(gpu_ctx):
    """Zero rows through dense -> leakyRelu -> dense -> sigmoid -> mse and its training step: outputs of
    shape [0, n], a loss of exactly 0 (the division by shape[0] = 0 sits inside a loop that never runs,
    base.nim:57-58), parameters unchanged (every gradient is an empty sum)."""
    from oracle import kd
    graphs = lambda: refcases.xor_layers()
    gpu = egm.compile(*graphs(), gpu=gpu_ctx)
    ref = kd.Model(refcases.program_text(graphs()))
    rng = np.random.default_rng(0)
    for tid in sorted(ref.params):
        v = (rng.random(ref.params[tid].shape, dtype=f32) - 0.5).astype(f32)
        ref.params[tid][...] = v
        gpu.params[tid] = v
    x, y = np.zeros((0, 2), f32), np.zeros((0, 1), f32)
    got = gpu.call("predict", {"x": x})
    assert got.shape == (0, 1)
    with np.errstate(all="ignore"):
        want_loss = ref.call("loss", {"x": x, "y": y})
    same_special(gpu.call("loss", {"x": x, "y": y}), want_loss, "loss over an empty batch")
    with np.errstate(all="ignore"):
        ref.apply("train", {"x": x, "y": y})
    gpu.apply("train", {"x": x, "y": y})
    for tid in sorted(ref.params):
        same_special(gpu.params[tid], ref.params[tid], f"parameter {tid}")
    gpu.close()
