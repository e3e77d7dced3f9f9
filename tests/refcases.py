"""Programs of the reference's own tests, restated in the Python DSL mirror.

Each builder returns the graph(s) to compile; the inputs and expected outputs live in
tests/golden/known_answers.json (data transcribed from the reference's tests, with citations).
Used by the CPU tests of the oracle and by the GPU tests of the product, so both are pinned to
the same vectors.
"""
import json
import os

import numpy as np

from exprgrad_amd import dsl, layers
from exprgrad_amd.dsl import Fun, iters, param, select, sq
from exprgrad_amd.examples import (conv2_3d, conv2_bench, dense_softmax_net, xor_from_scratch,  # noqa: F401
                                   xor_layers)

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "known_answers.json")


def load_golden():
    with open(GOLDEN) as f:
        return json.load(f)


def arr(spec):
    return np.array(spec["data"], dtype=np.float32).reshape(spec["shape"])


# ---- tests/test_model.nim ---------------------------------------------------------------------
def identity():
    it = iters("it")
    r = Fun()
    r.raw[it] += dsl.input("x").raw[it]
    return [r.target("y")]


def double():
    it = iters("it")
    r = Fun()
    r.raw[it] += dsl.input("x").raw[it] * 2.0
    return [r.target("y")]


def matmul():
    y, x, it = iters("y x it")
    c = Fun()
    c[y, x] += dsl.input("a")[y, it] * dsl.input("b")[it, x]
    return [c.target("c")]


def relu():
    it = iters("it")
    inp = dsl.input("inp")
    outp = Fun()
    outp.raw[it] += select(0.0 < inp.raw[it], inp.raw[it], 0.0)
    return [outp.target("outp")]


def mean_squared_error():
    it = iters("it")
    loss = Fun()
    loss[0] += sq(dsl.input("pred").raw[it] - dsl.input("labels").raw[it])
    return [loss.target("loss")]


def transpose():
    x, y = iters("x y")
    b = Fun()
    b[y, x] += dsl.input("a")[x, y]
    return [b.target("b")]


def maximum():
    it = iters("it")
    x = dsl.input("x")
    res = Fun()
    res.raw[it] += dsl.max(x.raw[it], dsl.input("y").raw[it])
    res.copy_shape(x)
    return [res.target("z")]


def conv1():
    x, dx = iters("x dx")
    res = Fun()
    res[x] += dsl.input("image")[x + dx] * dsl.input("filter")[dx]
    return [res.target("res")]


def single_write():
    res = Fun()
    res[0] += dsl.literal(10.0)
    return [res.target("y")]


def shape_case():
    it = iters("it")
    res = Fun()
    res.raw[it] += dsl.literal(1.0)
    res.with_shape(3, 2, 1)
    return [res.target("y")]


def dimensions():
    inp = dsl.input("x")
    res = Fun()
    res[0] += dsl.to_scalar(inp.shape[0])
    res[1] += dsl.to_scalar(inp.shape[-2])
    res[2] += dsl.to_scalar(inp.shape[-1])
    res[3] += dsl.to_scalar(dsl.Expr("instr", dsl.INDEX, instr="shapelen", tensor=inp))
    res[4] += dsl.to_scalar(inp.len())
    res.with_shape(5)
    return [res.target("y")]


def loop_bounds():
    res = Fun()
    res[dsl.iter_in("x", 2, 4)] += dsl.literal(1.0)
    res[dsl.iter_in("x", 0, 1)] += dsl.literal(-1.0)
    res[dsl.iter_in("x", 1, 1)] += dsl.literal(-2.0)
    res.with_shape(5)
    return [res.target("res")]


def blur():
    # tests/test_model.nim:99-107: the loop bound names the shape of the tensor being written
    res, image = Fun(), dsl.input("image")
    x = dsl.iter_in("x", 0, res.shape[0])
    res[x] += (image[x] + image[x + 1] + image[x + 2]) / 3.0
    return [res.target("res")]


def blur_center():
    # tests/test_model.nim:109-117
    res, image = Fun(), dsl.input("image")
    x = dsl.iter_in("x", 1, image.shape[0] - 1)
    res[x - 1] += (image[x - 1] + image[x] + image[x + 1]) / 3.0
    return [res.target("res")]


def blur_offset():
    # tests/test_model.nim:119-128
    res, image = Fun(), dsl.input("image")
    x = dsl.iter_in("x", 0, image.shape[0] - 2)
    res[x + 1] += (image[x] + image[x + 1] + image[x + 2]) / 3.0
    res.with_shape(image.shape[0])
    return [res.target("res")]


def extern(factor):
    # tests/test_model.nim:156-167: a host value captured by the kernel
    def build():
        it = iters("it")
        res = Fun()
        res.raw[it] += dsl.input("x").raw[it] * float(factor)
        return [res.target("y")]
    return build


def dynamic_ast(n):
    # tests/test_model.nim:215-231: the expression is assembled by host code (x^n as n multiplications)
    def build():
        it = iters("it")
        x = dsl.input("x")
        prod = dsl.literal(1.0)
        for _ in range(n):
            prod = prod * x.raw[it]
        res = Fun()
        res.raw[it] += prod
        res.copy_shape(x)
        return [res.target("y")]
    return build


def array_case():
    # tests/test_model.nim:233-241
    x = iters("x")
    arr = dsl.array([1.0, 2.0, 3.0])
    res = Fun()
    res[x] += arr[x] + dsl.to_scalar(arr.len())
    res.with_shape(3)
    return [res.target("y")]


def nested_array():
    # tests/test_model.nim:243-255
    y, x = iters("y x")
    arr = dsl.array([[1.0, 2.0, 3.0], [4.0, 5.0, 6.0], [7.0, 8.0, 9.0]])
    res = Fun()
    res[y, x] += arr[y][x]
    res.with_shape(3, 3)
    return [res.target("y")]


def derive_polynomial():
    it = iters("it")
    x = dsl.input("x")
    y = Fun()
    y.raw[it] += sq(x.raw[it]) + 2.0 * x.raw[it] + 1.0
    return [y.backwards().grad(x).target("x^2+2x+1")]


def derive_multiply():
    it = iters("it")
    x = dsl.input("x")
    a, b, c, d = Fun(), Fun(), Fun(), Fun()
    a.raw[it] += x.raw[it] * x.raw[it] * x.raw[it]
    b.raw[it] += x.raw[it] / 2.0
    c.raw[it] += 1.0 / x.raw[it]
    d.raw[it] += x.raw[it] / x.raw[it]
    return [a.backwards().grad(x).target("x^3"), b.backwards().grad(x).target("x/2"),
            c.backwards().grad(x).target("1/x"), d.backwards().grad(x).target("x/x")]


def derive_trigonometry():
    it = iters("it")
    x = dsl.input("x")
    a, b = Fun(), Fun()
    a.raw[it] += dsl.sin(x.raw[it])
    b.raw[it] += dsl.cos(x.raw[it])
    return [a.backwards().grad(x).target("sin"), b.backwards().grad(x).target("cos")]


def derive_exp():
    it = iters("it")
    x = dsl.input("x")
    a, b, c, d, e = Fun(), Fun(), Fun(), Fun(), Fun()
    a.raw[it] += dsl.exp(x.raw[it])
    b.raw[it] += dsl.exp(2.0 * x.raw[it])
    c.raw[it] += dsl.pow(x.raw[it], 3.0)
    d.raw[it] += dsl.pow(2.0, x.raw[it])
    e.raw[it] += dsl.pow(x.raw[it], x.raw[it])
    return [a.backwards().grad(x).target("exp(x)"), b.backwards().grad(x).target("exp(2x)"),
            c.backwards().grad(x).target("x^3"), d.backwards().grad(x).target("2^x"),
            e.backwards().grad(x).target("x^x")]


def derive_log():
    it = iters("it")
    x = dsl.input("x")
    a, b, c, d, e = Fun(), Fun(), Fun(), Fun(), Fun()
    a.raw[it] += dsl.ln(x.raw[it])
    b.raw[it] += dsl.log10(x.raw[it])
    c.raw[it] += dsl.log2(x.raw[it])
    d.raw[it] += dsl.log(x.raw[it], 5.0)
    e.raw[it] += dsl.log(2.0, x.raw[it])
    return [a.backwards().grad(x).target("ln(x)"), b.backwards().grad(x).target("log10(x)"),
            c.backwards().grad(x).target("log2(x)"), d.backwards().grad(x).target("log(x,5)"),
            e.backwards().grad(x).target("log(2,x)")]


# ---- tests/test_talks.nim ---------------------------------------------------------------------
def increment():
    it = iters("it")
    r = Fun()
    r.raw[it] += dsl.input("input").raw[it] + 1.0
    return [r.target("increment")]


def sum_positive():
    it = iters("it")
    inp = dsl.input("input")
    r = Fun()
    r[0] += select(inp.raw[it] > 0.0, inp.raw[it], 0.0)
    return [r.target("sumPositive")]


def linear():
    x, y, it = iters("x y it")
    inp, weights, biases = dsl.input("input"), dsl.input("weights"), dsl.input("biases")
    r = Fun()
    r[y, x] += inp[y, it] * weights[it, x]
    r[y, x] += biases[x]
    return [r.target("predict")]


def multiply_and_square():
    x, y, it = iters("x y it")
    a, b = dsl.input("a"), dsl.input("b")
    c = Fun()
    c[y, x] += a[y, it] * b[it, x]
    d = Fun()
    d.raw[it] += c.raw[it] * c.raw[it]
    return [c.target("multiply"), d.target("multiplyAndSquare")]


# ---- tests/test_gpu.nim -----------------------------------------------------------------------
def leaky_relu_gpu():
    it = iters("it")
    x = dsl.input("x")
    y = Fun()
    y.raw[it] += select(x.raw[it] > 0.0, x.raw[it], 0.01 * x.raw[it])
    return [y.target("y")]


def custom_grad():
    # tests/test_model.nim:196-214: identity with a hand-written gradient inp * 2 * grad(identity)
    x = iters("x")
    inp = dsl.input("inp")
    ident = Fun()
    ident.raw[x] += inp.raw[x]
    with ident.custom_grad():
        dsl.grad_of(inp).raw[x] += inp.raw[x] * 2.0 * dsl.grad_of(ident).raw[x]
    return [ident.target("identity").backwards().grad(inp).target("grad")]


BUILDERS = {
    "identity": identity, "double": double, "matmul": matmul, "relu": relu,
    "meanSquaredError": mean_squared_error, "transpose": transpose, "max": maximum, "conv1": conv1,
    "singleWrite": single_write, "shape": shape_case, "dimensions": dimensions, "loopBounds": loop_bounds,
    "derive/polynomial": derive_polynomial, "derive/multiply": derive_multiply,
    "derive/trigonometry": derive_trigonometry, "derive/exp": derive_exp, "derive/log": derive_log,
    "increment": increment, "sumPositive": sum_positive, "multiple": linear,
    "multiplyAndSquare": multiply_and_square, "leakyReluGpu": leaky_relu_gpu, "matmulTalks": matmul,
    "matmulExample": matmul, "customGrad": custom_grad,
    "blur": blur, "blurCenter": blur_center, "blurOffset": blur_offset,
    **{f"extern/{f}": extern(f) for f in range(-2, 3)},
    "dynamicAst/0": dynamic_ast(0), "dynamicAst/1": dynamic_ast(1),
    "array": array_case, "nestedArray": nested_array,
}


def program_text(graphs):
    return dsl.to_program(*graphs).to_text()
