"""reshape, maxpool2 (hand-written gradient with computed indices `y div 2`), avgpool2, upsample2 and
the reference's fashion_mnist network (examples/fashion_mnist/fashion_mnist.nim:39-57).

CPU part: the oracle against closed forms in numpy (the reference pins the customGrad mechanism with
tests/test_model.nim:196-214, which is in tests/golden/known_answers.json; reshape with
tests/test_tensors.nim:111-117).  GPU part: the product against the oracle."""
import numpy as np
import pytest

import refcases
from conftest import TOL, rel_err
from exprgrad_amd import dsl, examples, layers


def oracle(graphs, threads=2):
    from oracle import kd
    return kd.Model(refcases.program_text(graphs), threads=threads)


def pool_graphs(kind):
    img = dsl.input("img")
    layer = {"max": layers.maxpool2, "avg": layers.avgpool2, "up": layers.upsample2}[kind](img)
    it = dsl.iters("it")
    loss = dsl.Fun()
    loss[0] += layer.raw[it] * layer.raw[it]
    return [layer.target("out"), loss.target("loss").backwards().grad(img).target("grad")]


def pool_reference(kind, a):
    n, h, w, c = a.shape
    if kind == "up":
        out = a.repeat(2, axis=1).repeat(2, axis=2)
        grad = (2 * out).reshape(n, h, 2, w, 2, c).sum(axis=(2, 4))
        return out, grad
    blocks = a.reshape(n, h // 2, 2, w // 2, 2, c)
    if kind == "avg":
        out = blocks.sum(axis=(2, 4)) / 4
        grad = (2 * out / 4).repeat(2, axis=1).repeat(2, axis=2)
        return out, grad
    out = blocks.max(axis=(2, 4))
    big = out.repeat(2, axis=1).repeat(2, axis=2)
    return out, np.where(a == big, 2 * big, 0)


@pytest.mark.parametrize("kind", ["max", "avg", "up"])
def test_oracle_pooling_forward_and_gradient(kind):
    a = np.random.default_rng(3).random((2, 4, 6, 3), dtype=np.float32)
    m = oracle(pool_graphs(kind))
    out, grad = pool_reference(kind, a.astype(np.float64))
    assert rel_err(m.call("out", {"img": a}), out) <= TOL
    assert rel_err(m.call("grad", {"img": a}), grad) <= TOL


def test_oracle_reshape_known_answers():
    # tests/test_tensors.nim:111-117 (host-tensor reshape; the graph version must agree)
    b = np.arange(6, dtype=np.float32)
    for shape, want in (([2, 3], (2, 3)), ([2, -1], (2, 3)), ([-1, 3], (2, 3)), ([-1], (6,)), ([-1, 2, 3], (1, 2, 3))):
        m = oracle([dsl.reshape(dsl.input("a"), shape).target("r")])
        got = m.call("r", {"a": b.reshape(3, 2)})
        assert got.shape == want and np.array_equal(got.ravel(), b)


def test_oracle_fashion_mnist_network_learns():
    m = oracle(examples.fashion_mnist_net(size=12, f1=4, f2=8, eta=0.02), threads=4)
    rng = np.random.default_rng(0)
    for tid in m.params:
        m.params[tid][...] = (rng.random(m.params[tid].shape, dtype=np.float32) * 0.4 - 0.2)
    x = rng.random((8, 144), dtype=np.float32)
    y = np.eye(10, dtype=np.float32)[rng.integers(0, 10, 8)]
    first = float(m.call("loss", {"x": x, "y": y})[0])
    for epoch in range(1, 9):
        m.epoch = epoch           # Model.fit bumps the epoch before every pass (model.nim:436)
        m.apply("fit", {"x": x, "y": y})
    assert float(m.call("loss", {"x": x, "y": y})[0]) < 0.8 * first


# ---------------------------------------------------------------------------------------------- GPU

@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["max", "avg", "up"])
@pytest.mark.parametrize("shape", [(2, 4, 6, 3), (3, 16, 20, 8)])
def test_gpu_pooling_matches_the_oracle(gpu_ctx, kind, shape):
    from exprgrad_amd import model as egm
    a = np.random.default_rng(sum(shape)).random(shape, dtype=np.float32)
    ref = oracle(pool_graphs(kind))
    gpu = egm.compile(*pool_graphs(kind), gpu=gpu_ctx)
    assert np.array_equal(gpu.call("out", {"img": a}), ref.call("out", {"img": a}))
    assert rel_err(gpu.call("grad", {"img": a}), ref.call("grad", {"img": a})) <= TOL
    gpu.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["max", "avg"])
def test_gpu_activation_is_inlined_into_the_pooling_kernels(gpu_ctx, kind):
    """conv2 -> leakyRelu -> pool, as in the reference's fashion_mnist network: the activation is
    recomputed inside the pooling kernel (and inside maxpool2's hand-written gradient) instead of
    being stored.  Same operations per element, so the result is bit-identical to the oracle's."""
    from exprgrad_amd import model as egm

    def graphs():
        img = dsl.input("img")
        act = layers.tanh(layers.leaky_relu(img))                 # a chain of two maps
        pool = {"max": layers.maxpool2, "avg": layers.avgpool2}[kind](act)
        it = dsl.iters("it")
        loss = dsl.Fun()
        loss[0] += pool.raw[it] * pool.raw[it]
        return [pool.target("out"), loss.target("loss").backwards().grad(img).target("grad")]

    a = (np.random.default_rng(5).random((4, 64, 64, 16), dtype=np.float32) - 0.5).astype(np.float32)
    ref = oracle(graphs(), threads=8)
    gpu = egm.compile(*graphs(), gpu=gpu_ctx)
    got, want = gpu.call("out", {"img": a}), ref.call("out", {"img": a})
    assert rel_err(got, want) <= TOL
    plan = [ln for ln in gpu.launch_plan("out").splitlines() if ln.startswith("[")]
    assert len(plan) == 1 and gpu.kernel_count("out") == ref.kernel_count("out") == 3, plan
    assert rel_err(gpu.call("grad", {"img": a}), ref.call("grad", {"img": a})) <= TOL
    gpu.close()


@pytest.mark.gpu
def test_gpu_activation_gradient_is_applied_inside_the_pooling_gradient(gpu_ctx, monkeypatch):
    """Backward pass of leakyRelu -> maxpool2 (the reference's fashion_mnist network): maxpool2's
    hand-written gradient kernel applies the activation's gradient to every value before storing it
    (consumer inlining), so the pooled gradient is never stored."""
    from exprgrad_amd import model as egm
    monkeypatch.setenv("EG_NO_SAMPLE_FUSE", "1")   # (the launch chain is what this test looks at; sample groups: test_gpu_sample_fuse.py)

    def graphs():
        img = dsl.input("img")
        pool = layers.maxpool2(layers.leaky_relu(img))
        it = dsl.iters("it")
        loss = dsl.Fun()
        loss[0] += pool.raw[it] * pool.raw[it]
        return [loss.target("loss").backwards().grad(img).target("grad")]

    a = (np.random.default_rng(6).random((4, 64, 64, 16), dtype=np.float32) - 0.5).astype(np.float32)
    ref = oracle(graphs(), threads=8)
    gpu = egm.compile(*graphs(), gpu=gpu_ctx)
    assert rel_err(gpu.call("grad", {"img": a}), ref.call("grad", {"img": a})) <= TOL
    assert "with its consumer" in gpu.launch_plan("grad"), gpu.launch_plan("grad")
    assert gpu.kernel_count("grad") == ref.kernel_count("grad")
    gpu.close()


@pytest.mark.gpu
def test_gpu_reshape_and_custom_grad(gpu_ctx):
    from exprgrad_amd import model as egm
    b = np.arange(24, dtype=np.float32)
    for shape in ([4, 6], [2, -1], [-1, 3], [-1], [-1, 2, 3]):
        gpu = egm.compile(dsl.reshape(dsl.input("a"), shape).target("r"), gpu=gpu_ctx)
        got = gpu.call("r", {"a": b.reshape(3, 8)})
        assert got.shape == tuple(b.reshape(shape).shape) and np.array_equal(got.ravel(), b)
        gpu.close()
    gpu = egm.compile(*refcases.custom_grad(), gpu=gpu_ctx)
    t = np.array([[1, 2], [3, 4]], dtype=np.float32)
    assert np.array_equal(gpu.call("identity", {"inp": t}), t)          # tests/test_model.nim:212-213
    assert np.array_equal(gpu.call("grad", {"inp": t}), t * 2)
    gpu.close()


@pytest.mark.gpu
@pytest.mark.parametrize("dims", [(12, 4, 8, 8), (28, 8, 16, 32)])
def test_gpu_fashion_mnist_network_matches_the_oracle(gpu_ctx, dims):
    from parity import Trio
    size, f1, f2, batch = dims
    t = Trio(gpu_ctx, lambda: examples.fashion_mnist_net(size=size, f1=f1, f2=f2), threads=8)
    rng = np.random.default_rng(size)
    t.init_params(rng, -0.2, 0.2)
    x = rng.random((batch, size * size), dtype=np.float32)
    y = np.eye(10, dtype=np.float32)[rng.integers(0, 10, batch)]
    t.call("predict", {"x": x}, n=size * size)
    t.call("loss", {"x": x, "y": y}, n=batch * 10)
    t.set_epoch(1)
    # adam's first step is eta * g / (|g| + eps): ill-conditioned where g ~ 0, so the step is checked in
    # two links (parity.Trio.step): the gradients against the float64 shadow, then the optimizer kernels
    # of oracle and shadow on the backend's own gradients
    t.step("fit", {"x": x, "y": y}, n=batch * size * size)
    t.close()


def test_front_end_errors_of_the_new_constructs():
    x = dsl.input("x")
    it = dsl.iters("it")
    with pytest.raises(dsl.ParserError):
        dsl.grad_of(x).raw[it] += x.raw[it]            # grad(...) ++= outside a customGrad block
    with pytest.raises(dsl.ParserError):
        dsl.reshape(x, [-1, -1])                       # one free extent at most
    f = dsl.Fun()
    with pytest.raises(dsl.ParserError):
        f.custom_grad()                                # nothing to attach the gradient to
    pooled = layers.maxpool2(x)
    with pytest.raises(dsl.ParserError):
        pooled.raw[it] += x.raw[it]                    # maxpool2 locks its result (dnn.nim:71)


def test_kernel_description_text_of_the_new_constructs():
    text = refcases.program_text(pool_graphs("max"))
    assert "customgrad" in text and "endcustomgrad" in text and "\nidx indexdiv" in text
    assert " -" in [l for l in text.splitlines() if l.startswith("write")][1]   # gradient placeholder: negative id
    text = refcases.program_text([dsl.reshape(dsl.input("a"), [-1, 3]).target("r")])
    assert "shapesetup" in text and "indexdiv" in text


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["max", "avg", "up"])
def test_gpu_four_elements_per_thread_needs_aligned_operands(gpu_ctx, kind):
    """Generated kernels whose operands all end in the fastest iterator handle four channels per
    thread with 16-byte loads and stores — only when every operand is 16-byte aligned.  A
    caller-owned input that starts 4 bytes off must take the element-wise path and give the same
    result, bit for bit."""
    torch = pytest.importorskip("torch")
    from exprgrad_amd import model as egm
    a = np.random.default_rng(11).random((3, 16, 20, 8), dtype=np.float32)
    gpu = egm.compile(*pool_graphs(kind), gpu=gpu_ctx)
    want_out, want_grad = gpu.call("out", {"img": a}), gpu.call("grad", {"img": a})       # arena copy: aligned
    flat = torch.zeros(a.size + 1, device="cuda")
    flat[1:] = torch.from_numpy(a.ravel()).cuda()
    shifted = flat[1:].view(*a.shape)
    assert shifted.data_ptr() % 16 == 4
    assert np.array_equal(gpu.call("out", {"img": shifted}), want_out)
    assert np.array_equal(gpu.call("grad", {"img": shifted}), want_grad)
    ref = oracle(pool_graphs(kind))
    assert np.array_equal(want_out, ref.call("out", {"img": a}))
    assert rel_err(want_grad, ref.call("grad", {"img": a})) <= TOL
    gpu.close()
