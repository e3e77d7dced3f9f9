"""Programs of the reference's examples and of BASELINE.json's configs, in the Python DSL mirror.

Shared by the tests and bench.py (definitions only, no data).  The reference's gan and inverse_rendering examples are
OUT OF SCOPE of the hot path: their programs live with their tests (tests/extra_examples.py),
where they exercise the generic-kernel route, not in the product package.
"""
from . import dsl, layers
from .dsl import Fun, iters, param, select, sq


def xor_from_scratch(rate=0.1):
    """tests/test_model.nim:169-194 == examples/xor_from_scratch/xor_from_scratch.nim:19-31."""
    y, x, it = iters("y x it")
    hidden = Fun()
    hidden[y, x] += dsl.input("x")[y, it] * param([2, 4])[it, x]
    hidden[y, x] += param([4])[x]
    hidden_relu = Fun()
    hidden_relu.raw[it] += select(hidden.raw[it] <= 0.0, 0.1 * hidden.raw[it], hidden.raw[it])
    output = Fun()
    output[y, x] += hidden_relu[y, it] * param([4, 1])[it, x]
    output[y, x] += param([1])[x]
    output_sigmoid = Fun()
    output_sigmoid.raw[it] += 1.0 / (1.0 + dsl.exp(-output.raw[it]))
    pred = output_sigmoid.target("predict")
    loss = Fun()
    loss[0] += sq(pred.raw[it] - dsl.input("y").raw[it])

    def optim(p, g):
        p.raw[it] += -rate * g.raw[it]

    return [loss.target("loss").backprop(optim).target("train")]


def xor_layers(rate=0.2):
    """tests/test_dnn.nim:23-34."""
    net = layers.dense(dsl.input("x"), 2, 4)
    net = layers.leaky_relu(net)
    net = layers.dense(net, 4, 1)
    net = layers.sigmoid(net).target("predict")
    net = layers.mse(net, dsl.input("y")).target("loss")
    return [net.backprop(layers.gradient_descent(rate)).target("train")]


# ---- BASELINE.json configs (definitions: SURVEY.md §8d) ----------------------------------------
def dense_softmax_net(n_in=784, n_hidden=512, n_out=10, rate=0.01):
    """configs[4]: dense(784,512) -> relu -> dense(512,10) -> softmax -> crossEntropy -> GD(0.01)."""
    net = layers.dense(dsl.input("x"), n_in, n_hidden)
    net = layers.relu(net)
    net = layers.dense(net, n_hidden, n_out)
    net = layers.softmax(net).target("predict")
    net = layers.cross_entropy(net, dsl.input("y")).target("loss")
    return [net.backprop(layers.gradient_descent(rate)).target("train")]


def conv2_bench():
    """configs[3] / dnn.nim:45-49 with the filter bank as an input."""
    return [layers.conv2(dsl.input("images"), dsl.input("filters")).target("conv2")]


def conv2_3d():
    """benchmarks/conv2/conv2.nim:128-132 (no batch dimension)."""
    y, x, f, c, dy, dx = iters("y x filter chan dy dx")
    image, filters = dsl.input("image"), dsl.input("filters")
    r = Fun()
    r[y, x, f] += image[y + dy, x + dx, c] * filters[f, dy, dx, c]
    return [r.target("conv2")]




def fashion_mnist_net(eta=0.01, size=28, f1=8, f2=16, classes=10):
    """examples/fashion_mnist/fashion_mnist.nim:39-57: reshape -> conv2(1,5,5,8) -> leakyRelu -> maxpool2
    -> conv2(8,3,3,16) -> leakyRelu -> maxpool2 -> reshape -> dense -> softmax -> crossEntropy -> adam."""
    s1 = (size - 4) // 2          # 5x5 valid convolution, then 2x2 pooling
    s2 = (s1 - 2) // 2            # 3x3 valid convolution, then 2x2 pooling
    net = dsl.reshape(dsl.input("x"), [-1, size, size, 1])
    net = layers.conv2(net, 1, 5, 5, f1)
    net = layers.leaky_relu(net)
    net = layers.maxpool2(net)
    net = layers.conv2(net, f1, 3, 3, f2)
    net = layers.leaky_relu(net)
    net = layers.maxpool2(net)
    net = dsl.reshape(net, [-1, f2 * s2 * s2])
    net = layers.dense(net, f2 * s2 * s2, classes)
    net = layers.softmax(net).target("predict")
    net = layers.cross_entropy(net, dsl.input("y")).target("loss")
    return [net.backwards().optimize(layers.adam(eta=eta)).target("fit")]


def pool_chain_adam(eta=0.01, size=6, filters=2):
    """The first layers of the fashion_mnist example with its optimizer, small: reshape -> conv2(1,3,3,f) -> maxpool2 ->
    mse -> adam (fashion_mnist.nim:39-57; dnn.nim:45-71; base.nim:40-58).  tests/golden/handwritten/pool_chain_adam.kd is
    the text parser.nim + toKd produce for it, derived by hand."""
    net = layers.maxpool2(layers.conv2(dsl.reshape(dsl.input("x"), [-1, size, size, 1]), 1, 3, 3, filters)).target("predict")
    loss = layers.mse(net, dsl.input("y")).target("loss")
    return [loss.backprop(layers.adam(eta=eta)).target("fit")]


def matmul_graph():
    """benchmarks/matmul/matmul_gpu.nim:61-64 / examples/matmul: c*[y, x] ++= a[y, it] * b[it, x]."""
    y, x, it = iters("y x it")
    c = Fun()
    c[y, x] += dsl.input("a")[y, it] * dsl.input("b")[it, x]
    return [c.target("c")]


