"""Contractions with a generated epilogue (csrc/host/epilogue.cpp, kernels/gemm_fused.hpp): the
activation after `dense` and the activation gradient after the backward contraction run on the
accumulator registers of the matrix kernel.  Parity with the oracle, and with the unfused path.
"""
import numpy as np
import pytest

import refcases
from conftest import TOL, rel_err
from exprgrad_amd import dsl, layers
from exprgrad_amd import model as egm
from parity import Trio

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def no_sample_groups(monkeypatch):
    """These tests are about the plans of LARGE layers, exercised at small sizes (EG_EPILOGUE_MIN_ELEMS etc.): the
    small-batch sample groups of round 5 (tests/test_gpu_sample_fuse.py) would take the narrow layers first."""
    monkeypatch.setenv("EG_NO_SAMPLE_FUSE", "1")

ACTS = {"relu": layers.relu, "leaky_relu": layers.leaky_relu, "sigmoid": layers.sigmoid, "tanh": layers.tanh}


def mlp(act="relu", dims=(96, 80, 72, 8), rate=0.05):  # hidden widths > 64: not row-fusable
    net = dsl.input("x")
    for i in range(len(dims) - 1):
        net = layers.dense(net, dims[i], dims[i + 1])
        if i + 2 < len(dims):
            net = ACTS[act](net)
    net = net.target("predict")
    net = layers.mse(net, dsl.input("y")).target("loss")
    return [net.backprop(layers.gradient_descent(rate)).target("train")]


def build(gpu_ctx, monkeypatch, min_elems, **kw):
    from oracle import kd
    monkeypatch.setenv("EG_EPILOGUE_MIN_ELEMS", str(min_elems))
    gpu = egm.compile(*mlp(**kw), gpu=gpu_ctx)
    ref = kd.Model(refcases.program_text(mlp(**kw)), threads=4)
    rng = np.random.default_rng(5)
    for tid in sorted(ref.params):
        v = (rng.random(ref.params[tid].shape, dtype=np.float32) * 0.6 - 0.3).astype(np.float32)
        ref.params[tid][...] = v
        gpu.params[tid] = v
    return gpu, ref


def trio(gpu_ctx, monkeypatch, min_elems, graphs, prange):
    monkeypatch.setenv("EG_EPILOGUE_MIN_ELEMS", str(min_elems))
    t = Trio(gpu_ctx, graphs)
    t.init_params(np.random.default_rng(5), -prange, prange)
    return t


@pytest.mark.parametrize("act", sorted(ACTS))
@pytest.mark.parametrize("batch", [64, 37, 300])
def test_fused_epilogue_matches_the_oracle(gpu_ctx, monkeypatch, act, batch):
    t = trio(gpu_ctx, monkeypatch, 0, lambda: mlp(act=act), 0.3)
    rng = np.random.default_rng(batch)
    x = (rng.random((batch, 96), dtype=np.float32) - 0.5).astype(np.float32)
    y = rng.random((batch, 8), dtype=np.float32)
    t.call("predict", {"x": x}, n=96)
    assert t.gpu.launch_plan("predict").count("gemm+epilogue") == 2     # both hidden layers
    t.call("loss", {"x": x, "y": y}, n=batch * 8)
    for _ in range(3):  # eager, captured, replayed — each compared from the backend's own state
        t.step("train", {"x": x, "y": y}, n=batch)
    plan = t.gpu.launch_plan("train")
    # 2 forward + the backward ones (at batch 37 the [37, 72] gradients are "small" tensors and the
    # activation gradients join a single-block small-kernel group instead)
    assert plan.count("gemm+epilogue") >= (3 if batch >= 64 else 2), plan
    t.close()


def test_fused_and_unfused_paths_agree_bit_for_bit(gpu_ctx, monkeypatch):
    # same matrix kernel, same scalar expression: fusing must not change a single bit
    # (the bias-gradient fold is decided per plan and would fold in one launch list and not in the other:
    # another summation order for the bias gradients, equally correct but not bit-identical)
    monkeypatch.setenv("EG_NO_ONES_ROW", "1")
    rng = np.random.default_rng(9)
    # 300 rows: the hidden contractions are beyond the tiny-GEMM kernel in both runs
    x = (rng.random((300, 96), dtype=np.float32) - 0.5).astype(np.float32)
    y = rng.random((300, 8), dtype=np.float32)
    results = []
    for min_elems in (0, 1 << 40):
        gpu, _ = build(gpu_ctx, monkeypatch, min_elems, act="leaky_relu")
        out = gpu.call("predict", {"x": x})
        gpu.apply("train", {"x": x, "y": y})
        fused = "gemm+epilogue" in gpu.launch_plan("train")
        assert fused == (min_elems == 0)
        results.append((out, {t: gpu.params[t].copy() for t in sorted(gpu.params)}))
        gpu.close()
    assert np.array_equal(results[0][0], results[1][0])
    for t in results[0][1]:
        assert np.array_equal(results[0][1][t], results[1][1][t]), t


def test_large_layers_fuse_by_default(gpu_ctx, monkeypatch):
    # >= 2^20 output elements: no environment override needed (the cfg-5 shapes at a reduced batch)
    monkeypatch.delenv("EG_EPILOGUE_MIN_ELEMS", raising=False)
    t = Trio(gpu_ctx, lambda: refcases.dense_softmax_net(n_in=64, n_hidden=512, n_out=10), threads=8)
    rng = np.random.default_rng(1)
    t.init_params(rng, -0.1, 0.1)
    batch = 2048
    x = rng.random((batch, 64), dtype=np.float32)
    y = np.eye(10, dtype=np.float32)[rng.integers(0, 10, size=batch)]
    t.step("train", {"x": x, "y": y}, n=batch)
    plan = t.gpu.launch_plan("train")
    assert plan.count("gemm+epilogue") == 2, plan     # relu forward, relu backward
    t.close()


@pytest.mark.parametrize("dims", [(16, 128, 2048), (20, 96, 1500), (16, 96, 1500), (32, 160, 4096)])
def test_first_k_tile_waits_for_every_waves_loads(gpu_ctx, monkeypatch, dims):
    """An LDS-DMA tile is published to the block by every wave's OWN `s_waitcnt vmcnt(0)` in front of the
    barrier (gemm_f32_mfma.hpp: dma_publish_barrier).  The hiprtc build of the generated-epilogue kernels
    used to put that wait behind the barrier: a block that reads its first k-tile right away — one or two
    k-tiles, 64 x 64 tiles — multiplied stale LDS bytes in most runs (whole 32-column stripes wrong), larger
    problems only once in a while.  Forward of dense + tanh against float64, several fresh models."""
    k, n, batch = dims
    monkeypatch.setenv("EG_EPILOGUE_MIN_ELEMS", "0")

    def graphs():
        return [layers.tanh(layers.dense(dsl.input("x"), k, n)).target("predict")]

    rng = np.random.default_rng(k + n)
    for trial in range(6):
        m = egm.compile(*graphs(), gpu=gpu_ctx)
        vals = {}
        for tid in m.params.ids():
            vals[tid] = (rng.random(m._param_shapes[tid], dtype=np.float32) - 0.5).astype(np.float32)
            m.params[tid] = vals[tid]
        x = (rng.random((batch, k), dtype=np.float32) - 0.5).astype(np.float32)
        got = m.call("predict", {"x": x})
        assert "gemm+epilogue" in m.launch_plan("predict")
        w = [v for v in vals.values() if v.ndim == 2][0].astype(np.float64)
        b = [v for v in vals.values() if v.ndim == 1][0].astype(np.float64)
        want = np.tanh(x.astype(np.float64) @ w + b)
        m.close()
        bad = np.argwhere(np.abs(got - want) > 1e-5)
        assert len(bad) == 0, (trial, len(bad), bad[:, 0].min(), bad[:, 0].max(), bad[:, 1].min(), bad[:, 1].max())


def test_consumer_with_an_inlined_consumer_is_not_taken_as_epilogue(gpu_ctx, monkeypatch):
    """dense -> map -> map on tensors too wide for row fusion: the first map carries the second one as an
    inlined consumer (lower.cpp inline_consumers); fusing that launch into the contraction as its epilogue
    used to drop the second map (seen under EG_NO_ROWFUSE as a violation of plan invariant 1)."""
    def graphs():
        net = layers.dense(dsl.input("x"), 130, 96)
        net = layers.sigmoid(layers.tanh(net))
        net = layers.dense(net, 96, 8).target("predict")
        net = layers.mse(net, dsl.input("y")).target("loss")
        return [net.backprop(layers.gradient_descent(0.05)).target("train")]

    t = trio(gpu_ctx, monkeypatch, 0, graphs, 0.3)
    rng = np.random.default_rng(77)
    x = (rng.random((300, 130), dtype=np.float32) - 0.5).astype(np.float32)
    y = rng.random((300, 8), dtype=np.float32)
    t.call("predict", {"x": x}, n=130)
    for _ in range(3):
        t.step("train", {"x": x, "y": y}, n=300)
    t.close()


def test_generated_kernels_wait_for_their_lds_dma_before_the_barrier(gpu_ctx, monkeypatch, tmp_path):
    """The same property at the instruction level, independent of timing luck: in the code object hiprtc
    builds for a generated-epilogue contraction, no `s_barrier` follows an LDS-DMA load (`global_load_lds` / `buffer_load ... lds`) without an
    `s_waitcnt vmcnt(0)` in between (fall-through order; the K loop issues, computes, waits, then synchronises).
    The build that multiplied stale tiles had `s_barrier` first and the wait behind it — produced by the hiprtc
    the process resolves at run time (the one bundled with PyTorch-ROCm 7.0), not by /opt/rocm's 7.2, whose
    output has the wait in front of the barrier with or without the explicit one: the order is a compiler's
    choice, which is why the kernel spells it out."""
    import os
    import subprocess
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("no llvm-objdump")
    monkeypatch.setenv("EG_EPILOGUE_MIN_ELEMS", "0")
    monkeypatch.setenv("EG_DUMP_CODE", str(tmp_path))
    monkeypatch.setenv("EG_NO_NARROW_K", "1")      # (K = 16 would run on the vector ALUs: this test is about the LDS-DMA loops)
    checked = 0
    for k, n, batch in [(16, 128, 2048), (20, 96, 1500), (784, 512, 4096)]:   # whole tiles; ragged M, N, K; the 256 x 256 tile
        m = egm.compile(layers.tanh(layers.dense(dsl.input("x"), k, n)).target("predict"), gpu=gpu_ctx)
        m.call("predict", {"x": np.zeros((batch, k), dtype=np.float32)})
        assert "gemm+epilogue" in m.launch_plan("predict")
        m.close()
        for name in sorted(os.listdir(tmp_path)):
            if not name.startswith("eg_gemm_epi") or not name.endswith(".co"):
                continue
            text = subprocess.run([objdump, "-d", os.path.join(tmp_path, name)], capture_output=True, text=True, check=True).stdout
            os.remove(os.path.join(tmp_path, name))
            pending, loads = False, 0
            for line in text.splitlines():
                ins = line.split()[0] if line.split() else ""
                if "global_load_lds" in line or ("buffer_load_" in line and " lds" in line):   # either LDS-DMA form
                    pending, loads = True, loads + 1
                elif ins == "s_waitcnt" and "vmcnt(0)" in line:
                    pending = False
                elif ins in ("s_branch", "s_endpgm", "s_setpc_b64"):
                    pending = False     # what follows in the layout is not reached by falling through
                elif ins == "s_barrier":
                    assert not pending, (name, "s_barrier with an LDS-DMA load not waited for", line.strip())
            if "v_mfma" not in text:       # (a narrow-K product runs on the vector ALUs: no LDS-DMA loop in it)
                continue
            assert loads > 0, name      # the LDS-DMA loop is what this test is about
            checked += 1
    assert checked >= 3


# ---- predicate tensors (plan_epilogue.cpp): the pre-activation of a relu layer is stored as one bit per element ----

def _train_state(gpu_ctx, monkeypatch, act, dims, x, y, steps, no_predicate):
    monkeypatch.setenv("EG_EPILOGUE_MIN_ELEMS", "0")
    if no_predicate:
        monkeypatch.setenv("EG_NO_PREDICATE", "1")
    else:
        monkeypatch.delenv("EG_NO_PREDICATE", raising=False)
    gpu = egm.compile(*mlp(act=act, dims=dims), gpu=gpu_ctx)
    rng = np.random.default_rng(5)
    for tid in sorted(gpu.params.ids()):
        gpu.params[tid] = (rng.random(gpu.params[tid].shape, dtype=np.float32) * 0.6 - 0.3).astype(np.float32)
    for _ in range(steps):       # eager, captured, replayed
        gpu.apply("train", {"x": x, "y": y})
    plan = gpu.launch_plan("train")
    params = {t: gpu.params[t].copy() for t in sorted(gpu.params.ids())}
    gpu.close()
    return plan, params


@pytest.mark.parametrize("act", ["relu", "leaky_relu"])
@pytest.mark.parametrize("batch", [256, 300, 37, 1024])
def test_predicate_bits_agree_bit_for_bit_with_stored_values(gpu_ctx, monkeypatch, act, batch):
    """Hidden widths that are multiples of 32: the pre-activations of both hidden layers exist as predicate bits only
    (whole tiles pack eight lanes' nibbles into a word, ragged tiles OR their bits in).  Same comparison on the same
    float32 value, evaluated by the producer instead of the reader: not one bit of any parameter may differ."""
    dims = (96, 128, 96, 8)
    rng = np.random.default_rng(batch)
    x = (rng.random((batch, dims[0]), dtype=np.float32) - 0.5).astype(np.float32)
    x[3] = 0.0                                  # a sample whose pre-activations are exactly the bias
    y = rng.random((batch, dims[-1]), dtype=np.float32)
    plan_bits, with_bits = _train_state(gpu_ctx, monkeypatch, act, dims, x, y, 3, no_predicate=False)
    plan_vals, with_vals = _train_state(gpu_ctx, monkeypatch, act, dims, x, y, 3, no_predicate=True)
    assert "predicate bits" not in plan_vals
    if batch >= 64:                             # (at batch 37 the activation gradients join a small-kernel group: no fused reader)
        assert plan_bits.count("stored as predicate bits") == 2, plan_bits
    for t in with_bits:
        assert np.array_equal(with_bits[t], with_vals[t]), (t, np.max(np.abs(with_bits[t] - with_vals[t])))


def test_keep_values_is_a_per_model_switch(gpu_ctx, monkeypatch):
    """eg_model_keep_values: a tensor the plan keeps as predicate bits is refused by read_tensor; with the switch on the
    same model re-plans with values (no environment variable, no new process), the tensor reads back as the oracle's
    pre-activation, the parameters of the step are bit-identical, and switching off returns to the plan with bits."""
    import re
    from exprgrad_amd._lib import GpuError
    monkeypatch.setenv("EG_EPILOGUE_MIN_ELEMS", "0")
    monkeypatch.delenv("EG_NO_PREDICATE", raising=False)
    dims = (96, 128, 96, 8)
    rng = np.random.default_rng(11)
    x = (rng.random((256, dims[0]), dtype=np.float32) - 0.5).astype(np.float32)
    y = rng.random((256, dims[-1]), dtype=np.float32)
    states = []
    for keep in (False, True):
        gpu = egm.compile(*mlp(act="relu", dims=dims), gpu=gpu_ctx)
        prng = np.random.default_rng(5)
        for tid in sorted(gpu.params.ids()):
            gpu.params[tid] = (prng.random(gpu.params[tid].shape, dtype=np.float32) * 0.6 - 0.3).astype(np.float32)
        if keep:
            gpu.keep_values(True)
        gpu.apply("train", {"x": x, "y": y})
        plan = gpu.launch_plan("train")
        bits = [int(t) for t in re.findall(r"t(\d+) stored as predicate bits", plan)]
        if not keep:
            assert bits, plan
            with pytest.raises(GpuError, match="eg_model_keep_values"):
                gpu.read_tensor("train", bits[0])
            gpu.keep_values(True)                         # the same model, switched: the next run re-plans
            first = {t: gpu.params[t].copy() for t in sorted(gpu.params.ids())}
            gpu.apply("train", {"x": x, "y": y})
            assert "predicate bits" not in gpu.launch_plan("train")
            h = gpu.read_tensor("train", bits[0])
            assert h.shape == (256, dims[1]) and np.isfinite(h).all() and (h < 0).any() and (h > 0).any()
            gpu.keep_values(False)
            gpu.apply("train", {"x": x, "y": y})
            assert "stored as predicate bits" in gpu.launch_plan("train")
            states.append(first)
        else:
            assert not bits, plan
            states.append({t: gpu.params[t].copy() for t in sorted(gpu.params.ids())})
        gpu.close()
    for t in states[0]:
        assert np.array_equal(states[0][t], states[1][t]), t


@pytest.mark.parametrize("batch", [1024, 1000])
def test_narrow_k_streaming_kernel_agrees_with_the_matrix_tile(gpu_ctx, monkeypatch, batch):
    """`ga = gz * W2^T` of a 16-class layer with relu's gradient in the epilogue: K = 16, 128 columns — the streaming
    kernel on the vector ALUs (gemm_narrow_k_block: one fmaf chain in k order per output, the reference's own order)
    against the same launch on the matrix tile (EG_NO_NARROW_K=1), whose 16-deep k-tile reaches the matrix core as
    k = 0, 4, 1, 5, ...: the same 16 products in another order, so the parameters after three steps agree to rounding,
    not to the bit — held to 1e-5 of the largest element, tensor by tensor — and the plan is the same launch list."""
    dims = (64, 128, 16)
    rng = np.random.default_rng(batch)
    x = (rng.random((batch, dims[0]), dtype=np.float32) - 0.5).astype(np.float32)
    y = rng.random((batch, dims[-1]), dtype=np.float32)
    monkeypatch.delenv("EG_NO_NARROW_K", raising=False)
    plan_a, params_a = _train_state(gpu_ctx, monkeypatch, "relu", dims, x, y, 3, no_predicate=False)
    monkeypatch.setenv("EG_NO_NARROW_K", "1")
    plan_b, params_b = _train_state(gpu_ctx, monkeypatch, "relu", dims, x, y, 3, no_predicate=False)
    assert "gemm+epilogue NT" in plan_a and plan_a == plan_b, plan_a
    for t in params_a:
        assert rel_err(params_a[t], params_b[t], what=f"parameter {t}: streaming narrow-K kernel vs matrix tile") <= TOL, t


@pytest.mark.parametrize("act", ["tanh", "sigmoid"])
def test_activations_that_use_the_value_keep_it(gpu_ctx, monkeypatch, act):
    dims = (96, 128, 96, 8)
    rng = np.random.default_rng(1)
    x = (rng.random((256, dims[0]), dtype=np.float32) - 0.5).astype(np.float32)
    y = rng.random((256, dims[-1]), dtype=np.float32)
    plan, _ = _train_state(gpu_ctx, monkeypatch, act, dims, x, y, 1, no_predicate=False)
    assert "predicate bits" not in plan, plan


def test_predicate_bits_on_zero_and_denormal_preactivations(gpu_ctx, monkeypatch):
    """`0 <= h` at h = +0.0 (true: the gradient passes) and at a negative denormal (false) — the bit must be the
    comparison's answer; checked against the oracle's step."""
    monkeypatch.setenv("EG_EPILOGUE_MIN_ELEMS", "0")
    monkeypatch.delenv("EG_NO_PREDICATE", raising=False)
    dims = (32, 128, 8)                         # (a hidden width of 64 or less would be row-fused instead)
    t = Trio(gpu_ctx, lambda: mlp(act="relu", dims=dims), threads=4)
    rng = np.random.default_rng(11)
    t.init_params(rng, -0.3, 0.3)
    bias = sorted(tid for tid in t.ref.params if t.ref.params[tid].shape == (128,))[0]
    b = np.zeros(128, dtype=np.float32)
    b[1::2] = -1e-42                            # negative denormal: 0 <= h is false, relu(h) is 0
    t.set_param(bias, b)
    batch = 128
    x = (rng.random((batch, 32), dtype=np.float32) - 0.5).astype(np.float32)
    x[:16] = 0.0                                # h = bias exactly: +0.0 in the even columns
    y = rng.random((batch, 8), dtype=np.float32)
    t.step("train", {"x": x, "y": y}, n=batch)
    assert "stored as predicate bits" in t.gpu.launch_plan("train")
    t.close()


# ---- row products (plan_epilogue.cpp fold_row_products): the narrow next layer inside the epilogue ----

def _head(gpu_ctx, monkeypatch, dims, x, y, steps, folded):
    monkeypatch.setenv("EG_EPILOGUE_MIN_ELEMS", "0")
    if folded:
        monkeypatch.delenv("EG_NO_ROW_PRODUCT", raising=False)
    else:
        monkeypatch.setenv("EG_NO_ROW_PRODUCT", "1")
    gpu = egm.compile(*mlp(act="relu", dims=dims), gpu=gpu_ctx)
    rng = np.random.default_rng(5)
    start = {}
    for tid in sorted(gpu.params.ids()):
        start[tid] = (rng.random(gpu.params[tid].shape, dtype=np.float32) * 0.2 - 0.1).astype(np.float32)
        gpu.params[tid] = start[tid]
    out = gpu.call("predict", {"x": x}).copy()
    for _ in range(steps):       # eager, captured, replayed
        gpu.apply("train", {"x": x, "y": y})
    plans = gpu.launch_plan("predict"), gpu.launch_plan("train")
    params = {t: gpu.params[t].copy() for t in sorted(gpu.params.ids())}
    gpu.close()
    return plans, start, out, params


@pytest.mark.parametrize("dims,batch,force", [((784, 512, 10), 32768, True), ((64, 256, 16), 65536, True), ((96, 512, 1), 32768, True),
                                              ((32, 512, 7), 65536, True)])
def test_narrow_next_layer_rides_in_the_epilogue(gpu_ctx, monkeypatch, dims, batch, force):
    """dense -> relu -> dense(., <= 16) on whole 256 x 256 tiles: the second contraction is taken of the rows of relu(h)
    while they pass through LDS (gemm_f32_mfma.hpp, RD_N).  Against the float64 product; against the unfolded plan (same
    values to rounding: the 256-column partial sums are formed in another order); run-to-run bit-identical (two N-tiles
    add to a zeroed element in either order)."""
    if force:   # small problems would run on smaller tiles: the row product lives on the 256 x 256 tile
        monkeypatch.setenv("EG_GEMM_FORCE_TILE", "256,256")
    rng = np.random.default_rng(batch + dims[1])
    x = (rng.random((batch, dims[0]), dtype=np.float32) - 0.5).astype(np.float32)
    y = rng.random((batch, dims[-1]), dtype=np.float32)
    plans, start, out, params = _head(gpu_ctx, monkeypatch, dims, x, y, 3, folded=True)
    assert "row product" in plans[0] and "row product" in plans[1], plans
    _, _, out_again, params_again = _head(gpu_ctx, monkeypatch, dims, x, y, 3, folded=True)
    assert np.array_equal(out, out_again)
    for t in params:
        assert np.array_equal(params[t], params_again[t]), t
    plans_plain, _, out_plain, params_plain = _head(gpu_ctx, monkeypatch, dims, x, y, 3, folded=False)
    assert "row product" not in plans_plain[0] + plans_plain[1]
    ids = sorted(start)
    w1, b1, w2, b2 = (start[t].astype(np.float64) for t in ids)
    if w1.ndim == 1:
        w1, b1 = b1, w1
    if w2.ndim == 1:
        w2, b2 = b2, w2
    want = np.maximum(x.astype(np.float64) @ w1 + b1, 0.0) @ w2 + b2
    assert rel_err(out, want) <= TOL
    assert rel_err(out_plain, want) <= TOL
    for t in params:
        assert rel_err(params[t], params_plain[t].astype(np.float64)) <= 2 * TOL, t


def test_row_product_keeps_denormal_sums(gpu_ctx, monkeypatch):
    """The partial products reach their destination through global_atomic_add_f32: a denormal sum must arrive as it is
    (W2 = 0 and a denormal bias: every output is exactly the bias), and so must a sum of two denormal partial products."""
    monkeypatch.setenv("EG_EPILOGUE_MIN_ELEMS", "0")
    monkeypatch.setenv("EG_GEMM_FORCE_TILE", "256,256")
    monkeypatch.delenv("EG_NO_ROW_PRODUCT", raising=False)
    dims = (32, 512, 7)
    gpu = egm.compile(*mlp(act="relu", dims=dims), gpu=gpu_ctx)
    ids = sorted(gpu.params.ids())
    shapes = {t: gpu.params[t].shape for t in ids}
    w1 = next(t for t in ids if shapes[t] == (32, 512))
    b1 = next(t for t in ids if shapes[t] == (512,))
    w2 = next(t for t in ids if shapes[t] == (512, 7))
    b2 = next(t for t in ids if shapes[t] == (7,))
    tiny = np.float32(1e-42)
    gpu.params[w1] = np.zeros((32, 512), dtype=np.float32)
    gpu.params[b1] = np.ones(512, dtype=np.float32)            # relu(h) = 1 everywhere
    gpu.params[w2] = np.zeros((512, 7), dtype=np.float32)
    gpu.params[b2] = np.full(7, tiny, dtype=np.float32)
    x = np.zeros((32768, 32), dtype=np.float32)                 # 256 tiles of 256 x 256: no k-slices, the product is folded
    out = gpu.call("predict", {"x": x})
    assert "row product" in gpu.launch_plan("predict")
    assert np.all(out == tiny), out[:2]
    # 256 columns of 1e-45-sized products per N-tile: 256 * 4e-45 per tile, two tiles, no bias contribution
    step = np.float32(4e-45)                                   # a multiple of the smallest denormal (1.4e-45): 3 ulps
    gpu.params[w2] = np.full((512, 7), step, dtype=np.float32)
    gpu.params[b2] = np.zeros(7, dtype=np.float32)
    out = gpu.call("predict", {"x": x})
    want = np.float32(512) * step                              # exact: integers times the denormal unit
    assert np.all(out == want), (out[:2], want)
    gpu.close()


@pytest.mark.parametrize("seed", range(6))
def test_row_product_with_other_activations_and_widths(gpu_ctx, monkeypatch, seed):
    """Random heads: activation (relu / leaky_relu keep predicate bits, tanh / sigmoid keep the values), hidden width 256 or
    512, 1 .. 16 outputs, input width: folded and unfolded plans agree to rounding, the folded one with the float64 product."""
    rng = np.random.default_rng(900 + seed)
    act = ["relu", "tanh", "leaky_relu", "sigmoid", "relu", "tanh"][seed]
    hidden = int(rng.choice([256, 512]))
    outs = int(rng.integers(1, 17))
    width = int(rng.choice([16, 48, 112]))      # (whole 16-deep k-tiles: a ragged K takes the element-wise epilogue, no row product)
    batch = 65536 if hidden == 256 else 32768          # 256 tiles of 256 x 256: one per compute unit, no k-slices
    monkeypatch.setenv("EG_GEMM_FORCE_TILE", "256,256")
    dims = (width, hidden, outs)
    x = (rng.random((batch, width), dtype=np.float32) - 0.5).astype(np.float32)
    y = rng.random((batch, outs), dtype=np.float32)
    monkeypatch.setenv("EG_EPILOGUE_MIN_ELEMS", "0")
    results = {}
    for folded in (True, False):
        if folded:
            monkeypatch.delenv("EG_NO_ROW_PRODUCT", raising=False)
        else:
            monkeypatch.setenv("EG_NO_ROW_PRODUCT", "1")
        gpu = egm.compile(*mlp(act=act, dims=dims), gpu=gpu_ctx)
        prng = np.random.default_rng(5)
        for tid in sorted(gpu.params.ids()):
            gpu.params[tid] = (prng.random(gpu.params[tid].shape, dtype=np.float32) * 0.2 - 0.1).astype(np.float32)
        out = gpu.call("predict", {"x": x}).copy()
        for _ in range(2):
            gpu.apply("train", {"x": x, "y": y})
        plan = gpu.launch_plan("train")
        params = {t: gpu.params[t].copy() for t in sorted(gpu.params.ids())}
        gpu.close()
        results[folded] = (plan, out, params)
    assert "row product" in results[True][0], results[True][0]
    assert "row product" not in results[False][0]
    assert rel_err(results[True][1], results[False][1].astype(np.float64)) <= 2 * TOL
    for t in results[True][2]:
        assert rel_err(results[True][2][t], results[False][2][t].astype(np.float64)) <= 2 * TOL, t
