"""Programs of the reference's examples and of BASELINE.json's configs, in the Python DSL mirror.

Shared by the tests and bench.py (definitions only, no data).
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


def gan(seed_dim=32, h1=64, h2=128, pixels=28 * 28, rate=0.1):
    """examples/gan/gan.nim:35-61: generator and discriminator MLPs; `cond` feeds the discriminator the
    generator's output in the generator's targets and the `samples` input elsewhere; each side is
    optimised over its own parameters only."""
    it = iters("it")

    def gen_loss(labels):
        r = Fun()
        r[0] += sq(labels.raw[it]) / dsl.to_scalar(labels.shape[0])                  # gan.nim:35-36
        return r

    gen = layers.dense(dsl.input("seed"), seed_dim, h1)
    gen = layers.leaky_relu(gen, 0.01)
    gen = layers.dense(gen, h1, h2)
    gen = layers.leaky_relu(gen, 0.01)
    gen = layers.sigmoid(layers.dense(gen, h2, pixels)).target("gen")
    discr = dsl.cond({"fit.gen": gen, "loss.gen": gen}, dsl.input("samples"))
    discr = layers.leaky_relu(layers.dense(discr, pixels, h2), 0.01)
    discr = layers.leaky_relu(layers.dense(discr, h2, h1), 0.01)
    discr = layers.sigmoid(layers.dense(discr, h1, 1)).target("discr")
    gen_params = gen.params()
    fit_gen = gen_loss(discr).target("loss.gen").backwards().optimize(gen_params, layers.gradient_descent(rate))
    fit_gen = fit_gen.target("fit.gen")
    discr_params = [p for p in discr.params() if p not in gen_params]
    fit_discr = layers.mse(discr, dsl.input("labels")).target("loss.discr").backwards()
    fit_discr = fit_discr.optimize(discr_params, layers.gradient_descent(rate)).target("fit.discr")
    return [gen, discr, fit_gen, fit_discr]


def matmul_graph():
    """benchmarks/matmul/matmul_gpu.nim:61-64 / examples/matmul: c*[y, x] ++= a[y, it] * b[it, x]."""
    y, x, it = iters("y x it")
    c = Fun()
    c[y, x] += dsl.input("a")[y, it] * dsl.input("b")[it, x]
    return [c.target("c")]


def inverse_rendering(size=128, trainable_colors=True, rate=0.01):
    """examples/inverse_rendering/inverse_rendering.nim:33-170: a two-sphere ray tracer written as ONE
    `++=` statement per pixel component (ray/sphere intersection, nearest hit, diffuse shading), and
    gradient descent on the sphere colours against a target image.  One large generated kernel with
    sqrt, min / max, boolean `and` and nested select; its gradient reduces over all pixels."""
    y, x, c = iters("y x c")

    def dot(a, b):
        return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]

    def scale(v, s):
        return [v[0] * s, v[1] * s, v[2] * s]

    def normalize(v):
        length = dsl.sqrt(dot(v, v))
        return [v[0] / length, v[1] / length, v[2] / length]

    def vec3(fun):
        return [fun[0], fun[1], fun[2]]

    def sphere(index):
        geometry = dsl.input(f"sphere{index}.geom", [4])
        color = (dsl.param([3], init_range=(0.0, 1.0), name=f"sphere{index}.color") if trainable_colors
                 else dsl.input(f"sphere{index}.color", [3]))
        return geometry, color

    background, light, camera = dsl.input("background", [3]), dsl.input("light", [3]), dsl.input("camera")
    spheres = [sphere(0), sphere(1)]

    def raycast_sphere(geometry, direction):                     # inverse_rendering.nim:49-75
        pos, radius = vec3(geometry), geometry[3]
        cc = dot(pos, pos) - sq(radius)
        b = 2.0 * dot(pos, direction)
        a = dot(direction, direction)
        d = sq(b) - 4.0 * a * cc
        hit = d >= 0.0
        e = dsl.sqrt(d)
        t = dsl.min((b + e) / (2.0 * a), (b - e) / (2.0 * a))
        along = scale(direction, t)
        normal = normalize([along[0] - pos[0], along[1] - pos[1], along[2] - pos[2]])
        return hit, t, normal

    def raycast(direction, light_dir, comp, view_distance=100.0):   # inverse_rendering.nim:77-93
        result = background[comp]
        min_dist = dsl.literal(view_distance)
        for geometry, color in spheres:
            hit, t, normal = raycast_sphere(geometry, direction)
            closer = hit & (t > 0.0) & (t < min_dist)
            intensity = dsl.max(dot(normal, light_dir), 0.0)
            result = dsl.select(closer, intensity * color[comp], result)
            min_dist = dsl.select(closer, t, min_dist)
        return result

    render = Fun()
    direction = [dsl.to_scalar(x) / float(size) - 0.5, -(dsl.to_scalar(y) / float(size) - 0.5), camera[0]]
    render[y, x, c] += raycast(direction, normalize(vec3(light)), c)  # inverse_rendering.nim:95-106
    render.with_shape(size, size, 3)
    render = render.target("render")
    if not trainable_colors:
        return [render]
    loss = layers.mse(render, dsl.input("target")).target("loss")
    return [loss.backprop(layers.gradient_descent(rate)).target("train")]


def inverse_rendering_scene():
    """The scene of inverse_rendering.nim:125-133 / 172-179 (host values of the inputs)."""
    import numpy as np
    f = np.float32
    return {"camera": np.array([1], f), "background": np.array([0.5, 0.5, 0.5], f),
            "sphere0.geom": np.array([0.5, 0.2, 4, 0.5], f), "sphere1.geom": np.array([-0.6, -0.35, 3, 0.5], f),
            "light": np.array([1, 1, -0.5], f)}
