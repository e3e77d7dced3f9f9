"""GPU parity of kernel-description programs (group 3 of the C ABI) — the drop-in path:
Python DSL mirror -> text -> eg_model_compile -> HIP kernels, compared with
  (a) the reference's own known-answer vectors (tests/golden/known_answers.json), and
  (b) the oracle (oracle/kd.py + refinterp.c: the reference's CPU lowering restated) on seeded
      random inputs, at 1e-5 relative (float32; BASELINE.json north_star).
"""
import numpy as np
import pytest

import refcases
from conftest import TOL, rel_err
from exprgrad_amd import model as egm

pytestmark = pytest.mark.gpu
GOLDEN = refcases.load_golden()


@pytest.mark.parametrize("name", sorted(GOLDEN))
def test_known_answers(gpu_ctx, name):
    model = egm.compile(*refcases.BUILDERS[name](), gpu=gpu_ctx)
    for c in GOLDEN[name]["calls"]:
        got = model.call(c["target"], {k: refcases.arr(v) for k, v in c["inputs"].items()})
        want = refcases.arr(c["expected"])
        assert list(got.shape) == c["expected"]["shape"]
        if c["mode"] == "sumsq" and c["target"] == "x^x":
            # the reference's absolute bound (sum of squares < 0.01 at values up to 5e7) only holds
            # when both sides call the same libm powf; device powf is within 1 ulp: relative bound
            assert rel_err(got, want) <= TOL
        elif c["mode"] == "sumsq":
            assert float(np.sum((got.astype(np.float64) - want) ** 2)) < c["eps"]
        elif name.startswith("derive/") and name not in ("derive/polynomial", "derive/multiply"):
            # the reference compares against host libm; device expf/logf/powf/sinf/cosf are 1-2 ulp
            # implementations of the same functions: float32 tolerance instead of ==
            finite = np.isfinite(want)
            assert np.array_equal(np.isfinite(got), finite)
            assert np.allclose(got[finite], want[finite], rtol=1e-5, atol=1e-6)
            assert np.array_equal(got[~finite], want[~finite])
        else:
            assert np.array_equal(got, want), (got, want)
    model.close()


def set_params(models, seed):
    """Same explicit U[-0.1, 0.1) parameters on every side (the reference draws them from Nim's RNG:
    parity unpinned)."""
    rng = np.random.default_rng(seed)
    gpu, ref = models
    for tid in sorted(ref.params):
        v = (rng.random(ref.params[tid].shape, dtype=np.float32) * 0.2 - 0.1).astype(np.float32)
        ref.params[tid][...] = v
        gpu.params[tid] = v


def pair(graph_builder, gpu_ctx, **kw):
    from oracle import kd
    gpu = egm.compile(*graph_builder(**kw), gpu=gpu_ctx)
    ref = kd.Model(refcases.program_text(graph_builder(**kw)), threads=8)
    return gpu, ref


def test_kernel_lists_match_the_oracle(gpu_ctx):
    for builder in (refcases.xor_from_scratch, refcases.xor_layers, refcases.dense_softmax_net):
        gpu, ref = pair(builder, gpu_ctx)
        for target in ("predict", "loss", "train"):
            assert gpu.kernel_count(target) == ref.kernel_count(target), (builder.__name__, target)
        gpu.close()
    gpu, ref = pair(refcases.xor_from_scratch, gpu_ctx)
    assert gpu.kernel_count("train") == 19
    text = gpu.emit_ir()
    assert "gemm+bias" in text and "seed-fill" in text
    gpu.close()


@pytest.mark.parametrize("batch", [4, 37, 1024])
def test_xor_from_scratch_step_parity(gpu_ctx, batch):
    # configs[2] of BASELINE.json (reduced batch: the oracle is a sequential interpreter)
    gpu, ref = pair(refcases.xor_from_scratch, gpu_ctx)
    set_params((gpu, ref), seed=3)
    rng = np.random.default_rng(batch)
    x = rng.integers(0, 2, size=(batch, 2)).astype(np.float32)
    y = (x[:, :1] != x[:, 1:]).astype(np.float32)
    assert rel_err(gpu.call("predict", {"x": x}), ref.call("predict", {"x": x})) <= TOL
    assert rel_err(gpu.call("loss", {"x": x, "y": y}), ref.call("loss", {"x": x, "y": y})) <= TOL
    for step in range(3):
        gpu.apply("train", {"x": x, "y": y})
        ref.apply("train", {"x": x, "y": y})
        for tid in sorted(ref.params):
            assert rel_err(gpu.params[tid], ref.params[tid]) <= TOL, (step, tid)
    gpu.close()


def test_xor_from_scratch_converges_on_gpu(gpu_ctx):
    # tests/test_model.nim:169-194 (sum of squared errors < 0.1), own seed
    gpu = egm.compile(*refcases.xor_from_scratch(), gpu=gpu_ctx)
    rng = np.random.default_rng(1)
    for tid in gpu.params.ids():
        shape = gpu.params[tid].shape
        gpu.params[tid] = (rng.random(shape, dtype=np.float32) * 0.2 - 0.1).astype(np.float32)
    x = np.array([[0, 0], [0, 1], [1, 0], [1, 1]], dtype=np.float32)
    y = np.array([[0], [1], [1], [0]], dtype=np.float32)
    for _ in range(4000):
        gpu.apply("train", {"x": x, "y": y})
    pred = gpu.call("predict", {"x": x})
    assert float(np.sum((pred - y) ** 2)) < 0.1
    gpu.close()


def test_xor_layers_and_fit(gpu_ctx):
    # tests/test_dnn.nim:23-47 and 51-78: apply and fit(batchSize=4) must do the same thing
    a, ref = pair(refcases.xor_layers, gpu_ctx)
    b = egm.compile(*refcases.xor_layers(), gpu=gpu_ctx)
    set_params((a, ref), seed=1)
    for tid in b.params.ids():
        b.params[tid] = a.params[tid]
    x = np.array([[0, 0], [0, 1], [1, 0], [1, 1]], dtype=np.float32)
    y = np.array([[0], [1], [1], [0]], dtype=np.float32)
    for _ in range(20):
        a.apply("train", {"x": x, "y": y})
        b.fit("train", {"x": x, "y": y}, batch_size=4)
        ref.apply("train", {"x": x, "y": y})
    assert b.epoch == 20
    for tid in a.params.ids():
        assert np.array_equal(a.params[tid], b.params[tid])
        assert rel_err(a.params[tid], ref.params[tid]) <= TOL
    for _ in range(4000):
        a.apply("train", {"x": x, "y": y})
    internal = float(a.call("loss", {"x": x, "y": y}).sum())
    loss = float(np.sum((a.call("predict", {"x": x}) - y) ** 2))
    assert internal < 0.1 and loss < 0.1
    assert abs(loss / y.size - internal) < 1e-4
    a.close()
    b.close()


@pytest.mark.parametrize("dims", [(20, 16, 10, 64), (784, 512, 10, 96)])
def test_dense_softmax_step_parity(gpu_ctx, dims):
    # configs[4] of BASELINE.json at a reduced batch: dense -> relu -> dense -> softmax -> crossEntropy -> GD
    from parity import Trio
    n_in, n_hidden, n_out, batch = dims
    t = Trio(gpu_ctx, lambda: refcases.dense_softmax_net(n_in=n_in, n_hidden=n_hidden, n_out=n_out))
    t.init_params(np.random.default_rng(5), -0.1, 0.1)
    rng = np.random.default_rng(7)
    x = rng.random((batch, n_in), dtype=np.float32)
    labels = rng.integers(0, n_out, size=batch)
    y = np.eye(n_out, dtype=np.float32)[labels]                       # oneHot, tensors.nim:273-276
    t.call("predict", {"x": x}, n=n_in)
    t.call("loss", {"x": x, "y": y}, n=batch * n_out)
    # the gradients themselves (not parameter differences, which sit behind the float32 spacing of the
    # parameters), backend and oracle each against the float64 shadow
    t.step("train", {"x": x, "y": y}, n=max(batch, n_in))
    t.close()


def test_conv2_targets(gpu_ctx, refcpu):
    rng = np.random.default_rng(4)
    img = rng.random((2, 19, 17, 16), dtype=np.float32)
    flt = ((rng.random((24, 3, 3, 16), dtype=np.float32) - 0.5) * 4).astype(np.float32)
    m4 = egm.compile(*refcases.conv2_bench(), gpu=gpu_ctx)
    assert "conv2" in m4.emit_ir()
    want = refcpu.conv2_nhwc(img, flt)
    assert rel_err(m4.call("conv2", {"images": img, "filters": flt}), want) <= TOL
    m3 = egm.compile(*refcases.conv2_3d(), gpu=gpu_ctx)
    assert rel_err(m3.call("conv2", {"image": img[0], "filters": flt}), want[0]) <= TOL
    m4.close()
    m3.close()


def test_shape_changes_between_calls(gpu_ctx):
    # the reference re-infers shapes on every call (model.nim:392-406): different batch sizes must work
    model = egm.compile(*refcases.matmul(), gpu=gpu_ctx)
    rng = np.random.default_rng(0)
    for (m, k, n) in [(3, 4, 5), (64, 32, 16), (3, 4, 5), (1, 1, 1)]:
        a = rng.random((m, k), dtype=np.float32)
        b = rng.random((k, n), dtype=np.float32)
        assert rel_err(model.call("c", {"a": a, "b": b}), a.astype(np.float64) @ b.astype(np.float64)) <= TOL
    model.close()


def test_epoch_inside_a_computed_index_follows_the_epoch(gpu_ctx):
    """`row{i} ++= x[epoch() mod shape, i]`: epoch() inside a tensor index is a host value, evaluated when a plan is made
    (kd.cpp hoist_host_indices).  A plan made at epoch 1 must not serve epoch 2 (it did, and returned row 1
    again): the program is flagged (kd::Program::epoch_in_setup) and plans are then keyed by the epoch.  Checked over
    several epochs against the oracle, which re-evaluates the value on every call, and against the rows themselves;
    written both as `setup` (what dsl.py emits) and as `idx` (what another producer may emit)."""
    from oracle import kd
    from exprgrad_amd import dsl
    from exprgrad_amd.dsl import Fun, iters

    def graphs():
        x = dsl.input("x")
        i = iters("i")
        r = Fun()
        r[i] += x[dsl.epoch() % x.shape[0], i] * 2.0
        r.with_shape(x.shape[1])
        return [r.target("row")]

    text = refcases.program_text(graphs())
    assert "epoch" in text
    variants = [text]
    if "  setup " in text:
        variants.append(text.replace("  setup ", "  idx "))
    xs = np.arange(5 * 7, dtype=np.float32).reshape(5, 7)
    for t in variants:
        gpu = egm.Model(egm._LoadedProgram(t), gpu_ctx)
        ref = kd.Model(t)
        for epoch in (1, 2, 2, 7, 3, 11):
            gpu.epoch = ref.epoch = epoch
            got = gpu.call("row", {"x": xs})
            assert np.array_equal(got, ref.call("row", {"x": xs})), epoch
            assert np.array_equal(got, xs[epoch % 5] * 2.0), epoch
        gpu.close()


def test_device_inputs_are_borrowed(gpu_ctx):
    model = egm.compile(*refcases.matmul(), gpu=gpu_ctx)
    a = np.arange(6, dtype=np.float32).reshape(2, 3)
    b = np.arange(12, dtype=np.float32).reshape(3, 4)

    class Dev:  # anything with data_ptr() and shape (a torch tensor in bench.py)
        def __init__(self, arr):
            self.t = gpu_ctx.allocTensor(arr.shape)
            self.t.write(arr)
            self.shape = arr.shape

        def data_ptr(self):
            return self.t.ptr

    out = model.call("c", {"a": Dev(a), "b": Dev(b)})
    assert np.array_equal(out, a @ b)
    model.close()


def test_errors(gpu_ctx):
    # tests/test_errors.nim:20-59
    import exprgrad_amd as eg
    from exprgrad_amd import dsl
    model = egm.compile(*refcases.matmul(), gpu=gpu_ctx)
    with pytest.raises(eg.RuntimeErrorEG):       # invalidTarget
        model.call("myTarget")
    with pytest.raises(eg.RuntimeErrorEG):       # invalidInput
        model.call("c", {"a": np.zeros((2, 3)), "b": np.zeros((3, 2)), "abc": np.zeros((2, 3))})
    with pytest.raises(eg.RuntimeErrorEG):       # missingInput (a TODO in the reference; defined here)
        model.call("c", {"a": np.zeros((2, 3))})
    with pytest.raises(eg.ShapeError):           # rank mismatch (readDimension)
        model.call("c", {"a": np.zeros((2, 3, 1)), "b": np.zeros((3, 2))})
    model.close()
    y, x, it = dsl.iters("y x it")
    c = dsl.Fun()
    c[y, x] += dsl.input("a", [2, 3])[y, it] * dsl.input("b")[it, x]
    model = egm.compile(c.target("c"), gpu=gpu_ctx)
    with pytest.raises(eg.ShapeError):           # staticShapeMismatch
        model.call("c", {"a": np.zeros((10, 10)), "b": np.zeros((10, 2))})
    model.close()
    ones = dsl.Fun()
    ones.raw[it] += dsl.literal(1.0)
    model = egm.compile(ones.target("ones"), gpu=gpu_ctx)
    with pytest.raises(eg.ShapeError):           # underconstrainedShape ("ones" without withShape)
        model.call("ones")
    model.close()


def test_data_parallel_engine_world1(gpu_ctx):
    """GpuEngine + DataParallel on one rank == plain apply (the bucket lives in a torch tensor the
    all-reduce would work on in place); gradients land in the bound bucket."""
    torch = pytest.importorskip("torch")
    from exprgrad_amd.parallel import DataParallel, GpuEngine
    import exprgrad_amd as eg
    stream = torch.cuda.current_stream()
    ctx = eg.newGpuContext(0, stream=stream.cuda_stream)
    a = egm.compile(*refcases.dense_softmax_net(n_in=20, n_hidden=16, n_out=10), gpu=ctx)
    b = egm.compile(*refcases.dense_softmax_net(n_in=20, n_hidden=16, n_out=10), gpu=ctx)
    rng = np.random.default_rng(2)
    for tid in a.params.ids():
        v = (rng.random(a.params[tid].shape, dtype=np.float32) * 0.2 - 0.1).astype(np.float32)
        a.params[tid] = v
        b.params[tid] = v
    x = torch.rand((64, 20), device="cuda")
    y = torch.nn.functional.one_hot(torch.randint(0, 10, (64,), device="cuda"), 10).to(torch.float32).contiguous()
    engine = GpuEngine(b, "train")
    dp = DataParallel(engine, reduction="mean")
    for _ in range(3):
        a.apply("train", [("x", x), ("y", y)])
        dp.step([("x", x), ("y", y)])
    torch.cuda.synchronize()
    for tid in a.params.ids():
        assert np.array_equal(a.params[tid], b.params[tid])
    assert float(engine.bucket.abs().sum()) > 0
    a.close()
    b.close()


def test_data_parallel_step_over_rccl(gpu_ctx):
    """The step as bench.py --gpus N runs it: side stream (so the launch ranges are captured into HIP
    graphs), gradient bucket in a torch tensor, all-reduce through torch.distributed's "nccl" backend
    (RCCL) between the backward graph and the update graph.  One rank on this one-GPU box — the sum
    over one rank is the identity, so the result must equal plain apply() bit for bit."""
    torch = pytest.importorskip("torch")
    import socket
    import torch.distributed as dist
    from exprgrad_amd.parallel import DataParallel, GpuEngine
    import exprgrad_amd as eg
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", world_size=1, rank=0,
                            device_id=torch.device("cuda", 0))
    try:
        stream = torch.cuda.Stream()
        with torch.cuda.stream(stream):
            ctx = eg.newGpuContext(0, stream=stream.cuda_stream)
            a = egm.compile(*refcases.dense_softmax_net(n_in=96, n_hidden=128, n_out=10), gpu=ctx)
            b = egm.compile(*refcases.dense_softmax_net(n_in=96, n_hidden=128, n_out=10), gpu=ctx)
            rng = np.random.default_rng(5)
            for tid in a.params.ids():
                v = (rng.random(a.params[tid].shape, dtype=np.float32) * 0.2 - 0.1).astype(np.float32)
                a.params[tid] = v
                b.params[tid] = v
            x = torch.rand((512, 96), device="cuda")
            y = torch.nn.functional.one_hot(torch.randint(0, 10, (512,), device="cuda"), 10).to(torch.float32).contiguous()
            engine = GpuEngine(b, "train")
            dp = DataParallel(engine, reduction="mean", always_reduce=True)
            assert dp.world == 1 and dp.always_reduce
            for _ in range(6):     # eager, captured, then replays
                a.apply("train", [("x", x), ("y", y)])
                dp.step([("x", x), ("y", y)])
            stream.synchronize()
            probe = torch.ones(4, device="cuda")
            dist.all_reduce(probe)
            assert float(probe.sum()) == 4.0
            for tid in a.params.ids():
                assert np.array_equal(a.params[tid], b.params[tid]), tid
            assert float(engine.bucket.abs().sum()) > 0
            a.close()
            b.close()
    finally:
        dist.destroy_process_group()


def test_data_parallel_step_through_the_c_abi(gpu_ctx):
    """eg_dp_* + eg_model_step_dp (group 4 of the C ABI): the library itself calls RCCL on the
    context's stream.  One rank here; the sum over one rank is the identity, so the step must equal
    plain apply() bit for bit, and an all-reduce leaves a buffer unchanged."""
    torch = pytest.importorskip("torch")
    from exprgrad_amd.parallel import NativeDataParallel, RcclGroup
    import exprgrad_amd as eg
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        ctx = eg.newGpuContext(0, stream=stream.cuda_stream)
        uid = RcclGroup.unique_id()
        assert len(uid) == 128 and any(uid)
        group = RcclGroup(ctx, uid, rank=0, world=1)
        probe = torch.arange(1000, device="cuda", dtype=torch.float32)
        group.all_reduce(probe)
        stream.synchronize()
        assert torch.equal(probe.cpu(), torch.arange(1000, dtype=torch.float32))
        a = egm.compile(*refcases.dense_softmax_net(n_in=96, n_hidden=128, n_out=10), gpu=ctx)
        b = egm.compile(*refcases.dense_softmax_net(n_in=96, n_hidden=128, n_out=10), gpu=ctx)
        rng = np.random.default_rng(8)
        for tid in a.params.ids():
            v = (rng.random(a.params[tid].shape, dtype=np.float32) * 0.2 - 0.1).astype(np.float32)
            a.params[tid] = v
            b.params[tid] = v
        x = torch.rand((512, 96), device="cuda")
        y = torch.nn.functional.one_hot(torch.randint(0, 10, (512,), device="cuda"), 10).to(torch.float32).contiguous()
        dp = NativeDataParallel(b, "train", group, reduction="mean")
        for _ in range(6):     # eager, captured, then replays
            a.apply("train", [("x", x), ("y", y)])
            dp.step([("x", x), ("y", y)])
        stream.synchronize()
        for tid in a.params.ids():
            assert np.array_equal(a.params[tid], b.params[tid]), tid
        with pytest.raises(Exception):
            RcclGroup(ctx, uid, rank=3, world=2)
        a.close()
        b.close()
        group.close()
