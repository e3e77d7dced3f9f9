"""The reference's layer library restated in the Python DSL — statement for statement.

Sources: exprgrad/layers/base.nim:19-67 and exprgrad/layers/dnn.nim:19-100.  Iterator
declaration order after `|` does not affect the lowered kernel (loops are created on first
use, parser.nim:183-196), so it is not reproduced.
"""
from . import dsl
from .dsl import Fun, iters, param, select, sq, to_scalar


def _layer(name):
    f = Fun()
    f.name = name
    return f


# ---- base.nim ------------------------------------------------------------------------------
def add(a, b):
    it = iters("it")
    r = _layer("+")
    r.raw[it] += a.raw[it] + b.raw[it]                 # base.nim:19
    return r


def sub(a, b):
    it = iters("it")
    r = _layer("-")
    r.raw[it] += a.raw[it] - b.raw[it]                 # base.nim:20
    return r


def minimum(a, b):
    it = iters("it")
    r = _layer("min")
    r.raw[it] += dsl.min(a.raw[it], b.raw[it])         # base.nim:21
    return r


def maximum(a, b):
    it = iters("it")
    r = _layer("max")
    r.raw[it] += dsl.max(a.raw[it], b.raw[it])         # base.nim:22
    return r


def scale(a, factor):
    it = iters("it")
    r = _layer("*")
    r.raw[it] += a.raw[it] * float(factor)             # base.nim:24
    return r


def divide(a, factor):
    it = iters("it")
    r = _layer("/")
    r.raw[it] += a.raw[it] / float(factor)             # base.nim:25
    return r


def matmul(a, b):
    y, x, it = iters("y x it")
    r = _layer("matmul")
    r[y, x] += a[y, it] * b[it, x]                     # base.nim:27-28
    return r


def transpose(mat):
    y, x = iters("y x")
    r = _layer("transpose")
    r[y, x] += mat[x, y]                               # base.nim:32-33
    return r


def gradient_descent(rate=0.01):
    """makeOpt(gradientDescent, rate=...)  base.nim:37-38."""
    def optim(p, g):
        it = iters("it")
        p.raw[it] += -g.raw[it] * float(rate)
    return optim


gradientDescent = gradient_descent


def adam(eta=0.01, beta1=0.9, beta2=0.999, eps=1e-8):
    """makeOpt(adam, ...)  base.nim:40-53 (Kingma & Ba 2014).  Uses epoch(): drive it with Model.fit,
    which bumps Model.epoch (model.nim:436); at epoch 0 the bias correction divides by zero, as in
    the reference."""
    def optim(p, g):
        it = iters("it")
        m, v = dsl.cache(p, "adam.m"), dsl.cache(p, "adam.v")
        m.raw[it] += m.raw[it] * (beta1 - 1.0) + (1.0 - beta1) * g.raw[it]           # base.nim:47
        v.raw[it] += v.raw[it] * (beta2 - 1.0) + (1.0 - beta2) * sq(g.raw[it])       # base.nim:48
        m_hat = m.raw[it] / (1.0 - dsl.pow(beta1, to_scalar(dsl.epoch())))
        v_hat = v.raw[it] / (1.0 - dsl.pow(beta2, to_scalar(dsl.epoch())))
        p.raw[it] += -eta * m_hat / (dsl.sqrt(v_hat) + eps)                          # base.nim:49-53
    return optim


def mse(a, b):
    it = iters("it")
    r = _layer("mse")
    r[0] += sq(a.raw[it] - b.raw[it]) / to_scalar(a.shape[0])                 # base.nim:57-58
    return r


def binary_cross_entropy(pred, labels):
    it = iters("it")
    r = _layer("binaryCrossEntropy")
    r[0] += -(labels.raw[it] * dsl.ln(pred.raw[it]) +
              (1.0 - labels.raw[it]) * dsl.ln(1.0 - pred.raw[it])) / to_scalar(pred.shape[0])   # base.nim:60-64
    return r


def cross_entropy(pred, labels):
    it = iters("it")
    r = _layer("crossEntropy")
    r[0] += -(labels.raw[it] * dsl.ln(pred.raw[it])) / to_scalar(pred.shape[0])   # base.nim:66-67
    return r


binaryCrossEntropy, crossEntropy = binary_cross_entropy, cross_entropy


# ---- dnn.nim -------------------------------------------------------------------------------
def dense(values, inp, outp, has_bias=True):
    y, x, it = iters("y x it")
    weights = param([inp, outp], name="weights")
    r = _layer("dense")
    r[y, x] += values[y, it] * weights[it, x]          # dnn.nim:21
    if has_bias:
        bias = param([outp], name="bias")
        r[y, x] += bias[x]                             # dnn.nim:22-24
    return r


def relu(inp):
    it = iters("it")
    r = _layer("relu")
    r.raw[it] += select(inp.raw[it] >= 0.0, inp.raw[it], 0.0)           # dnn.nim:26-27
    return r


def leaky_relu(inp, leak=0.01):
    it = iters("it")
    r = _layer("leakyRelu")
    r.raw[it] += select(inp.raw[it] >= 0.0, 1.0, float(leak)) * inp.raw[it]   # dnn.nim:29-30
    return r


leakyRelu = leaky_relu


def sigmoid(inp):
    it = iters("it")
    r = _layer("sigmoid")
    r.raw[it] += 1.0 / (1.0 + dsl.exp(-inp.raw[it]))                    # dnn.nim:32-33
    return r


def tanh(inp):
    it = iters("it")
    r = _layer("tanh")
    a = dsl.exp(inp.raw[it])
    b = dsl.exp(-inp.raw[it])
    r.raw[it] += (a - b) / (a + b)                                      # dnn.nim:35-40
    return r


def sin(inp):
    it = iters("it")
    r = _layer("sin")
    r.raw[it] += dsl.sin(inp.raw[it])                                   # dnn.nim:42-43
    return r


def conv2(images, filters, w=None, h=None, nfilters=None):
    """conv2(images, filters: Fun) or conv2(images, chans, w, h, filters: int)   dnn.nim:45-53."""
    if not isinstance(filters, Fun):
        chans = filters
        filters = param([nfilters, h, w, chans], name="filters")
    image, y, x, flt, dx, dy, chan = iters("image y x filter dx dy chan")
    r = _layer("conv2")
    r[image, y, x, flt] += images[image, y + dy, x + dx, chan] * filters[flt, dy, dx, chan]
    return r


def maxpool2(images):
    """dnn.nim:56-71: 2x2 max pooling with a hand-written gradient (the maximum's position gets the
    output gradient; computed indices `y div 2`)."""
    image, y, x, chan = iters("image y x chan")
    r = _layer("maxpool2")
    r[image, y, x, chan] += dsl.max(dsl.max(images[image, y * 2, x * 2, chan], images[image, y * 2 + 1, x * 2, chan]),
                                    dsl.max(images[image, y * 2, x * 2 + 1, chan], images[image, y * 2 + 1, x * 2 + 1, chan]))
    with r.custom_grad():
        dsl.grad_of(images)[image, y, x, chan] += select(
            images[image, y, x, chan].eq(r[image, y // 2, x // 2, chan]),
            dsl.grad_of(r)[image, y // 2, x // 2, chan], 0.0)
    r.lock()
    return r


def avgpool2(images):
    image, y, x, chan = iters("image y x chan")
    r = _layer("avgpool2")                                              # dnn.nim:73-79
    r[image, y, x, chan] += (images[image, y * 2, x * 2, chan] + images[image, y * 2 + 1, x * 2, chan] +
                             images[image, y * 2, x * 2 + 1, chan] + images[image, y * 2 + 1, x * 2 + 1, chan]) / 4.0
    return r


def upsample2(images):
    image, y, x, chan = iters("image y x chan")
    r = _layer("upsample2")                                             # dnn.nim:81-88
    r[image, y, x, chan] += images[image, y // 2, x // 2, chan]
    r.with_shape(images.shape[0], images.shape[1] * 2, images.shape[2] * 2, images.shape[3])
    return r


def dropout(inp, prob):
    """dnn.nim:96-100: keep an element with probability 1 - prob and rescale it, mask drawn per call."""
    it = iters("it")
    mask = dsl.rand(inp, 0.0, 1.0)
    mask.name = "dropout.rand"
    r = _layer("dropout")
    r.raw[it] += select(dsl.literal(float(prob)) <= mask.raw[it], inp.raw[it] / (1.0 - prob), 0.0)
    r.copy_shape(inp)
    return r


def reshape(fun, shape):
    return dsl.reshape(fun, shape)                                      # parser.nim:786-793


def softmax(inp):
    y, x = iters("y x")
    sums = _layer("softmax.sums")
    sums[y] += dsl.exp(inp[y, x])                                       # dnn.nim:90-92
    r = _layer("softmax")
    r[y, x] += dsl.exp(inp[y, x]) / sums[y]                             # dnn.nim:94
    return r
