"""GPU parity at the FULL sizes of BASELINE.json's configs.

Where the oracle can do the whole problem in seconds (it runs on the GPU box's host cores with the
reference's thread policy) the comparison is direct; for the 4096^3 product the oracle checks a
seeded sample of output rows (each row is the complete, unshortened K = 4096 reduction) and
size-independent properties cover the rest: a checksum of checksums against float64 and linearity.
Tolerance: 1e-5 relative, float32 (BASELINE.json north_star).
"""
import os

import numpy as np
import pytest

import refcases
from conftest import TOL, debug_toggles_active, direct_gate, rel_err
from exprgrad_amd import model as egm
from exprgrad_amd import ops

pytestmark = pytest.mark.gpu
CORES = os.cpu_count() or 1


def dev(ctx, arr):
    t = ctx.allocTensor(arr.shape)
    t.write(arr)
    return t


def test_cfg2_matmul_4096(gpu_ctx, refcpu):
    n = 4096
    rng = np.random.default_rng(2)
    a = rng.random((n, n), dtype=np.float32)          # U[0,1): benchmarks/matmul/matmul_gpu.nim:69-70
    b = rng.random((n, n), dtype=np.float32)
    da, db = dev(gpu_ctx, a), dev(gpu_ctx, b)
    dc = gpu_ctx.allocTensor((n, n))
    ops.sgemm(gpu_ctx, n, n, n, da, n, db, n, dc, n)
    c = dc.read()
    # (1) direct parity on sampled rows: full K = 4096 sequential f32 sums of the reference order
    rows = np.sort(rng.choice(n, size=48, replace=False))
    want = refcpu.sgemm(a[rows], b, threads=min(CORES, 48))
    assert rel_err(c[rows], want) <= TOL
    # (2) checksum of checksums against float64: (A B) 1 == A (B 1)
    rowsum = c.astype(np.float64).sum(axis=1)
    exact = a.astype(np.float64) @ b.astype(np.float64).sum(axis=1)
    assert np.max(np.abs(rowsum - exact)) <= TOL * np.max(np.abs(exact))
    colsum = c.astype(np.float64).sum(axis=0)
    exact_c = a.astype(np.float64).sum(axis=0) @ b.astype(np.float64)
    assert np.max(np.abs(colsum - exact_c)) <= TOL * np.max(np.abs(exact_c))
    # (3) linearity in A through the accumulate path: A1 B + A2 B == (A1 + A2) B
    a2 = rng.random((n, n), dtype=np.float32)
    da2 = dev(gpu_ctx, a2)
    ops.sgemm(gpu_ctx, n, n, n, da2, n, db, n, dc, n, accumulate=True)
    both = dc.read()
    dsum = dev(gpu_ctx, (a + a2).astype(np.float32))
    ops.sgemm(gpu_ctx, n, n, n, dsum, n, db, n, dc, n)
    assert rel_err(both, dc.read()) <= TOL
    # (4) run-to-run determinism
    ops.sgemm(gpu_ctx, n, n, n, dsum, n, db, n, dc, n)
    first = dc.read()
    ops.sgemm(gpu_ctx, n, n, n, dsum, n, db, n, dc, n)
    assert np.array_equal(first, dc.read())


def test_cfg1_matmul_256_direct(gpu_ctx, refcpu):
    # configs[0]: the reference's CPU-runnable case, whole problem against the oracle
    rng = np.random.default_rng(1)
    a = rng.random((256, 256), dtype=np.float32)
    b = rng.random((256, 256), dtype=np.float32)
    model = egm.compile(*refcases.matmul(), gpu=gpu_ctx)
    got = model.call("c", {"a": a, "b": b})
    assert rel_err(got, refcpu.sgemm(a, b)) <= TOL
    model.close()


def test_cfg3_xor_train_step_batch_65536(gpu_ctx):
    from oracle import kd
    batch = 65536
    gpu = egm.compile(*refcases.xor_from_scratch(), gpu=gpu_ctx)
    ref = kd.Model(refcases.program_text(refcases.xor_from_scratch()), threads=CORES)
    rng = np.random.default_rng(3)
    for tid in sorted(ref.params):
        v = (rng.random(ref.params[tid].shape, dtype=np.float32) * 0.2 - 0.1).astype(np.float32)
        ref.params[tid][...] = v
        gpu.params[tid] = v
    x = rng.integers(0, 2, size=(batch, 2)).astype(np.float32)
    y = (x[:, :1] != x[:, 1:]).astype(np.float32)
    before = {t: ref.params[t].copy() for t in ref.params}
    # loss[0] += sq(p - y) over 65 536 samples.  x takes only 4 distinct values, so the reference's
    # sequential f32 accumulation rounds the SAME way at every step and drifts systematically
    # (observed 1.3e-4 relative) — its own rounding, bounded by n*eps/2 = 2e-3.  The exact sum of
    # the reference's f32 terms (float64 accumulation) is the fair target: the GPU's tree sum must
    # be within 1e-5 of it, and the reference-order sum within its own bound.
    loss_gpu = float(gpu.call("loss", {"x": x, "y": y})[0])
    loss_ref = float(ref.call("loss", {"x": x, "y": y})[0])
    d = ref.call("predict", {"x": x}) - y
    exact = float((d * d).astype(np.float64).sum())
    assert abs(loss_gpu - exact) <= TOL * exact
    assert abs(loss_ref - exact) <= batch * 6e-8 / 2 * exact
    # ONE step: with the example's sum-of-squares loss and rate 0.1 the update is a 65 536-sample
    # sum, so a second step overflows to NaN on the reference and on the GPU alike (the example is
    # meant for batch 4; BASELINE.json scales the batch for throughput only)
    gpu.apply("train", {"x": x, "y": y})
    ref.apply("train", {"x": x, "y": y})
    # The same systematic drift hits the reference's batch-long gradient sums.  The loss is a plain
    # sum, so the exact full-batch update is  sum_k count_k * update(sample_k)  over the 4 distinct
    # samples: each per-sample update comes from the oracle (reference arithmetic, batch 1), only
    # the final weighting is float64.
    exact = {t: np.zeros(before[t].shape, dtype=np.float64) for t in before}
    for xs in ([0, 0], [0, 1], [1, 0], [1, 1]):
        one = kd.Model(refcases.program_text(refcases.xor_from_scratch()))
        for t in before:
            one.params[t][...] = before[t]
        xk = np.array([xs], dtype=np.float32)
        one.apply("train", {"x": xk, "y": (xk[:, :1] != xk[:, 1:]).astype(np.float32)})
        count = int(np.sum((x[:, 0] == xs[0]) & (x[:, 1] == xs[1])))
        for t in before:
            exact[t] += count * (one.params[t].astype(np.float64) - before[t])
    for tid in sorted(ref.params):
        du_gpu, du_ref = gpu.params[tid] - before[tid], ref.params[tid] - before[tid]
        assert rel_err(du_gpu, exact[tid], what=f"update of parameter {tid}: backend vs exact") <= TOL, tid
        assert rel_err(du_ref, exact[tid], what=f"update of parameter {tid}: ORACLE vs exact (its own 65 536-term drift)") <= batch * 6e-8 / 2, tid
        # the comparison in BASELINE.json's words (backend against the reference CPU path): bounded by the oracle's own drift
        e_go = rel_err(du_gpu, du_ref, what=f"update of parameter {tid}: backend vs oracle")
        assert e_go <= batch * 6e-8 / 2 + TOL, tid
        direct_gate(e_go, rel_err(du_ref, exact[tid], what="(oracle vs exact, for the gate)"), f"update of parameter {tid}: backend vs oracle")
    for tid in sorted(ref.params):
        assert np.all(np.isfinite(ref.params[tid])) and np.all(np.isfinite(gpu.params[tid]))
    gpu.close()


def test_cfg4_conv2_256x256x64(gpu_ctx, refcpu):
    N, H, W, C, F, FH, FW = 1, 256, 256, 64, 64, 3, 3
    rng = np.random.default_rng(4)
    img = rng.random((N, H, W, C), dtype=np.float32)                          # conv2.nim:337
    flt = (rng.random((F, FH, FW, C), dtype=np.float32) * 4 - 2).astype(np.float32)   # conv2.nim:338
    model = egm.compile(*refcases.conv2_bench(), gpu=gpu_ctx)
    got = model.call("conv2", {"images": img, "filters": flt})
    assert got.shape == (1, 254, 254, 64)
    # direct parity on sampled output rows: rows y..y+2 of the image give exactly output row y
    for y in (0, 1, 100, 253):
        want = refcpu.conv2_nhwc(img[:, y:y + FH], flt)
        assert rel_err(got[:, y:y + 1], want) <= TOL, y
    # checksum against float64: sum over all outputs == sum_taps (window sums of the image) . filters
    img64, flt64 = img.astype(np.float64), flt.astype(np.float64)
    total = 0.0
    for dy in range(FH):
        for dx in range(FW):
            window = img64[0, dy:dy + H - FH + 1, dx:dx + W - FW + 1].sum(axis=(0, 1))   # [C]
            total += float(window @ flt64[:, dy, dx].sum(axis=0))
    # filters are in [-2, 2): the total cancels, so the bound scales with the magnitude summed
    assert abs(got.astype(np.float64).sum() - total) <= TOL * float(np.abs(got).astype(np.float64).sum())
    model.close()


def test_cfg5_dense_train_step_batch_65536(gpu_ctx):
    """One GPU's share of configs[4] (65 536 of the 524 288 samples): whole train step vs the oracle."""
    from parity import Trio
    batch = 65536
    t = Trio(gpu_ctx, refcases.dense_softmax_net, threads=CORES)
    gpu, ref = t.gpu, t.ref
    rng = np.random.default_rng(5)
    t.init_params(rng, -0.1, 0.1)
    x = rng.random((batch, 784), dtype=np.float32)
    y = np.eye(10, dtype=np.float32)[rng.integers(0, 10, size=batch)]
    # gradients: the backend within 1e-5 of the float64 shadow; the reference's batch-long (65 536)
    # sequential float32 reductions within n * 2^-24 of it (round 1 compared the two float32 sides with
    # each other at 1e-4 instead)
    t.step("train", {"x": x, "y": y}, n=batch, sync=False)
    # the small weight gradient runs on the side lane, next to the large weight-gradient contraction, whose
    # last row is the first layer's bias gradient; later steps (eager, captured, replayed) must agree with this one
    plan = gpu.launch_plan("train")
    if not debug_toggles_active():
        assert plan.count("side lane") == 1 and "+ones-row" in plan, plan
        # the pre-activation of the hidden layer exists as predicate bits only, and the 10-wide second layer is computed in
        # the first layer's epilogue (its own launch is gone)
        assert "stored as predicate bits" in plan and "row product NN 65536x10x512" in plan, plan
    serial = egm.compile(*refcases.dense_softmax_net(), gpu=gpu_ctx)
    for tid in sorted(ref.params):
        serial.params[tid] = gpu.params[tid]
    for _ in range(3):                        # replays of the captured two-lane graph ...
        gpu.apply("train", {"x": x, "y": y})
    for _ in range(3):                        # ... against a fresh model's eager, captured and replayed steps
        serial.apply("train", {"x": x, "y": y})
    for tid in sorted(ref.params):
        assert np.array_equal(gpu.params[tid], serial.params[tid]), tid
    serial.close()
    gpu.close()


def test_cfg5_step_with_skewed_waves_equals_the_in_phase_loops(gpu_ctx, monkeypatch):
    """Round 4: the two long contractions of the step (fused forward, weight gradient with its extra rows) run with the
    odd wave of every SIMD pair one k-group late.  Same MFMAs in the same order: three steps from the same state with
    EG_GEMM_NO_SKEW=1 (every wave in phase, the round-3 loops) must leave the same bits in every parameter."""
    torch = pytest.importorskip("torch")
    x = torch.rand((65536, 784), device="cuda")
    y = torch.nn.functional.one_hot(torch.randint(0, 10, (65536,), device="cuda"), 10).to(torch.float32).contiguous()
    states = []
    for no_skew in (False, True):
        if no_skew:
            monkeypatch.setenv("EG_GEMM_NO_SKEW", "1")
        else:
            monkeypatch.delenv("EG_GEMM_NO_SKEW", raising=False)
        m = egm.compile(*refcases.dense_softmax_net(), gpu=gpu_ctx)
        rng = np.random.default_rng(21)
        for tid in m.params.ids():
            m.params[tid] = (rng.random(m.params[tid].shape, dtype=np.float32) * 0.2 - 0.1).astype(np.float32)
        for _ in range(3):
            m.apply("train", [("x", x), ("y", y)])
        gpu_ctx.sync()
        states.append({tid: m.params[tid].copy() for tid in m.params.ids()})
        m.close()
    for tid in states[0]:
        assert np.array_equal(states[0][tid], states[1][tid]), tid


def test_cfg5_split_step_equals_whole_step(gpu_ctx):
    """The data-parallel form of the step (run_backward | exchange | run_update, here on one rank
    without a process group) at the full per-GPU batch, where the side lane is active inside the
    backward range: bit-identical to apply()."""
    torch = pytest.importorskip("torch")
    from exprgrad_amd.parallel import DataParallel, GpuEngine
    import exprgrad_amd as eg
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        ctx = eg.newGpuContext(0, stream=stream.cuda_stream)
        whole = egm.compile(*refcases.dense_softmax_net(), gpu=ctx)
        split = egm.compile(*refcases.dense_softmax_net(), gpu=ctx)
        rng = np.random.default_rng(15)
        for tid in whole.params.ids():
            v = (rng.random(whole.params[tid].shape, dtype=np.float32) * 0.2 - 0.1).astype(np.float32)
            whole.params[tid] = v
            split.params[tid] = v
        x = torch.rand((65536, 784), device="cuda")
        y = torch.nn.functional.one_hot(torch.randint(0, 10, (65536,), device="cuda"), 10).to(torch.float32).contiguous()
        dp = DataParallel(GpuEngine(split, "train"), reduction="mean")
        for _ in range(4):
            whole.apply("train", [("x", x), ("y", y)])
            dp.step([("x", x), ("y", y)])
        stream.synchronize()
        assert debug_toggles_active() or "side lane" in split.launch_plan("train")
        for tid in whole.params.ids():
            assert np.array_equal(whole.params[tid], split.params[tid]), tid
        whole.close()
        split.close()


def test_cfg4_conv2_gradients_256x256x64(gpu_ctx, refcpu):
    """Both gradients of configs[3] at full size.  Direct parity with the oracle on sub-problems that
    are exact slices of the full one (a sub-bank of filters; a band of output-gradient rows), and the
    adjoint identities <conv(img, flt), g> = <img, gradImage(flt, g)> = <flt, gradFilter(img, g)> in
    float64 over the whole problem."""
    N, H, W, C, F, FH, FW = 1, 256, 256, 64, 64, 3, 3
    Ho, Wo = H - FH + 1, W - FW + 1
    rng = np.random.default_rng(44)
    img = rng.random((N, H, W, C), dtype=np.float32)
    flt = (rng.random((F, FH, FW, C), dtype=np.float32) * 4 - 2).astype(np.float32)
    gout = (rng.random((N, Ho, Wo, F), dtype=np.float32) - 0.5).astype(np.float32)
    dimg, dflt, dg = dev(gpu_ctx, img), dev(gpu_ctx, flt), dev(gpu_ctx, gout)
    out, gflt, gimg = gpu_ctx.allocTensor((N, Ho, Wo, F)), gpu_ctx.allocTensor(flt.shape), gpu_ctx.allocTensor(img.shape)
    ops.conv2_nhwc(gpu_ctx, N, H, W, C, F, FH, FW, dimg, dflt, out)
    ops.conv2_nhwc_grad_filter(gpu_ctx, N, H, W, C, F, FH, FW, dimg, dg, gflt)
    ops.conv2_nhwc_grad_image(gpu_ctx, N, H, W, C, F, FH, FW, dflt, dg, gimg)
    conv, gf, gi = out.read(), gflt.read(), gimg.read()

    # (1) filter gradient of a sub-bank: gFlt[f] depends on gOut[..., f] only
    sub = refcpu.conv2_nhwc_grad_filter(img, np.ascontiguousarray(gout[..., :3]), (3, FH, FW, C))
    from parity import check_op, exact_conv2_grad_filter
    # 64 516 sequential f32 additions on the reference side: both against the float64 value
    check_op(gf[:3], sub, exact_conv2_grad_filter(img, gout[..., :3], (3, FH, FW, C)), n=Ho * Wo, what="filter gradient")
    assert rel_err(gf, exact_conv2_grad_filter(img, gout, flt.shape)) <= TOL          # the whole bank against float64
    # (2) image gradient of a band: with gOut zero outside rows [y0, y1) only image rows [y0, y1 + FH - 1) are touched
    y0, y1 = 100, 108
    band = np.zeros_like(gout)
    band[:, y0:y1] = gout[:, y0:y1]
    dband = dev(gpu_ctx, band)
    gimg_band = gpu_ctx.allocTensor(img.shape)
    ops.conv2_nhwc_grad_image(gpu_ctx, N, H, W, C, F, FH, FW, dflt, dband, gimg_band)
    got_band = gimg_band.read()
    want_band = refcpu.conv2_nhwc_grad_image(flt, np.ascontiguousarray(gout[:, y0:y1]), (N, y1 - y0 + FH - 1, W, C))
    assert rel_err(got_band[:, y0:y1 + FH - 1], want_band) <= TOL
    assert not got_band[:, :y0].any() and not got_band[:, y1 + FH - 1:].any()
    # (3) adjoint identities over the whole problem, float64
    lhs = float((conv.astype(np.float64) * gout).sum())
    via_image = float((img.astype(np.float64) * gi).sum())
    via_filter = float((flt.astype(np.float64) * gf).sum())
    scale = float(np.abs(conv.astype(np.float64) * gout).sum())
    assert abs(lhs - via_image) <= TOL * scale and abs(lhs - via_filter) <= TOL * scale


def test_cfg5_native_step_overlaps_the_early_exchange(gpu_ctx):
    """eg_model_step_dp at the full per-GPU batch of configs[4]: the bias gradients and the small weight
    gradient are all-reduced on the side lane WHILE the 784 x 512 weight-gradient contraction runs, only
    that contraction's own gradient after it (two RCCL calls per step instead of one).  One rank here
    (a sum over one rank is the identity), so the step must equal apply() bit for bit; EG_DP_NO_SPLIT=1
    gives the single all-reduce after the backward pass."""
    torch = pytest.importorskip("torch")
    from exprgrad_amd.parallel import NativeDataParallel, RcclGroup
    from exprgrad_amd._lib import call, GpuError
    import exprgrad_amd as eg
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        ctx = eg.newGpuContext(0, stream=stream.cuda_stream)
        group = RcclGroup(ctx, RcclGroup.unique_id(), rank=0, world=1)
        whole = egm.compile(*refcases.dense_softmax_net(), gpu=ctx)
        split = egm.compile(*refcases.dense_softmax_net(), gpu=ctx)
        rng = np.random.default_rng(15)
        for tid in whole.params.ids():
            v = (rng.random(whole.params[tid].shape, dtype=np.float32) * 0.2 - 0.1).astype(np.float32)
            whole.params[tid] = v
            split.params[tid] = v
        x = torch.rand((65536, 784), device="cuda")
        y = torch.nn.functional.one_hot(torch.randint(0, 10, (65536,), device="cuda"), 10).to(torch.float32).contiguous()
        dp = NativeDataParallel(split, "train", group, reduction="mean")
        for _ in range(4):
            whole.apply("train", [("x", x), ("y", y)])
            dp.step([("x", x), ("y", y)])
        stream.synchronize()
        assert debug_toggles_active() or call("eg_dp_last_pieces", group.handle) == 2, split.launch_plan("train")
        for tid in whole.params.ids():
            assert np.array_equal(whole.params[tid], split.params[tid]), tid
        # a model of another context is refused: the stream is what orders the collective
        other = egm.compile(*refcases.dense_softmax_net(n_in=8, n_hidden=8, n_out=2), gpu=gpu_ctx)
        with pytest.raises(GpuError, match="different contexts"):
            call("eg_model_step_dp", other.handle, b"train", group.handle, 1)
        other.close()
        whole.close()
        split.close()
        group.close()


def test_native_step_split_toggle_comparison_collective_and_reserved_units(gpu_ctx, monkeypatch):
    """What the first multi-GPU run relies on, on one GPU: RCCL's own rank count, the split switched on and off inside one
    process (eg_dp_set_split: bench.py times both forms), and — with EG_DP_TEST_AS_MULTI=1 — the code paths a one-rank
    group otherwise skips: the once-per-plan comparison of the exchange plan across the ranks (an int64 MAX all-reduce)
    and the long contraction leaving compute units to the early collective (another slice count: same sums, another
    order, so that variant is compared at 1e-5 instead of bit for bit)."""
    torch = pytest.importorskip("torch")
    from exprgrad_amd.parallel import NativeDataParallel, RcclGroup
    import exprgrad_amd as eg
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        ctx = eg.newGpuContext(0, stream=stream.cuda_stream)
        x = torch.rand((16384, 784), device="cuda")
        y = torch.nn.functional.one_hot(torch.randint(0, 10, (16384,), device="cuda"), 10).to(torch.float32).contiguous()

        def run(env, split_sequence):
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            group = RcclGroup(ctx, RcclGroup.unique_id(), rank=0, world=1)
            assert group.rccl_count() == 1
            model = egm.compile(*refcases.dense_softmax_net(), gpu=ctx)
            rng = np.random.default_rng(15)
            for tid in model.params.ids():
                model.params[tid] = (rng.random(model.params[tid].shape, dtype=np.float32) * 0.2 - 0.1).astype(np.float32)
            dp = NativeDataParallel(model, "train", group, reduction="mean")
            pieces = []
            for allowed in split_sequence:
                group.set_split(allowed)
                for _ in range(3):          # eager, captured, replayed
                    dp.step([("x", x), ("y", y)])
                stream.synchronize()
                pieces.append(group.last_pieces())
            params = {tid: model.params[tid].copy() for tid in model.params.ids()}
            model.close()
            group.close()
            for k in env:
                monkeypatch.delenv(k)
            return params, pieces

        plain, p0 = run({}, [True, False, True])
        if not debug_toggles_active():
            assert p0 == [2, 1, 2], p0
        same_kernels, p1 = run({"EG_DP_TEST_AS_MULTI": "1", "EG_DP_RESERVE_CUS": "0"}, [True, False, True])
        for tid in plain:
            assert np.array_equal(plain[tid], same_kernels[tid]), tid       # the comparison collective changes nothing
        reserved, p2 = run({"EG_DP_TEST_AS_MULTI": "1", "EG_DP_RESERVE_CUS": "8"}, [True, False, True])
        for tid in plain:
            assert rel_err(reserved[tid], plain[tid]) <= TOL, tid
