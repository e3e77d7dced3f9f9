"""The reference's examples that lie OUTSIDE the hot path (examples/gan, examples/inverse_rendering),
in the Python DSL mirror.  Kept with the tests that use them as workloads of the generic-kernel route (`cond` nodes, sqrt /
min / max / nested select, toScalar of iterators); not part of the product package."""
from exprgrad_amd import dsl, layers
from exprgrad_amd.dsl import Fun, iters, param, select, sq  # noqa: F401


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
