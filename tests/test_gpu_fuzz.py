"""Differential fuzzing: random layer chains built with the Python front-end run through the product
(GPU: pattern matcher, generated kernels, row / map / small fusion, fused epilogues, graphs) and
through the oracle's independent restatement of generate / derive / shape inference / execution.
Seeds are fixed: the cases are reproducible.  EG_FUZZ_CHAINS / EG_FUZZ_CNNS = "start:stop" run other
seed ranges (wider sweeps before a release; CNN seeds >= 16 also draw larger images and batches)."""
import os
import random

import numpy as np
import pytest

import refcases
from conftest import rel_err
from exprgrad_amd import dsl, layers
from exprgrad_amd.dsl import Fun, iters

pytestmark = pytest.mark.gpu


def seeds(var, default):
    lo, hi = (int(v) for v in os.environ.get(var, default).split(":"))
    return range(lo, hi)


def unary(rng, x):
    it = iters("it")
    r = Fun()
    kind = rng.choice(["relu", "leaky", "sigmoid", "tanh", "scale", "square", "softsign", "sin", "clip"])
    v = x.raw[it]
    if kind == "relu":
        r.raw[it] += dsl.select(v > 0.0, v, 0.0)
    elif kind == "leaky":
        r.raw[it] += dsl.select(v > 0.0, v, v * 0.1)
    elif kind == "sigmoid":
        r.raw[it] += 1.0 / (1.0 + dsl.exp(-v))
    elif kind == "tanh":
        return layers.tanh(x)
    elif kind == "scale":
        r.raw[it] += v * rng.choice([0.5, -1.5, 2.0]) + rng.choice([0.0, 0.25])
    elif kind == "square":
        r.raw[it] += v * v * 0.5
    elif kind == "softsign":
        r.raw[it] += v / (1.0 + dsl.select(v > 0.0, v, -v))
    elif kind == "sin":
        r.raw[it] += dsl.sin(v)
    else:
        r.raw[it] += dsl.min(dsl.max(v, -0.5), 0.5)
    r.copy_shape(x)
    return r


def build(seed):
    rng = random.Random(seed)
    width = rng.choice([3, 8, 20, 70, 130])
    net, dims = dsl.input("x"), width
    for depth in range(rng.randint(2, 5)):
        op = rng.choice(["dense", "dense", "unary", "unary", "residual", "rowscale"])
        if op == "dense":
            out = rng.choice([1, 4, 10, 33, 96])
            net = layers.dense(net, dims, out, has_bias=rng.random() < 0.7)
            dims = out
        elif op == "unary":
            net = unary(rng, net)
        elif op == "residual":
            total = layers.add(net, unary(rng, net))
            total.copy_shape(net)      # a raw write with two reads has no inferred shape (passes.nim:1059-1066)
            net = total
        else:  # every row divided by (1 + its sum of squares): a per-row reduction feeding a map
            y, x = iters("y x")
            sums = Fun()
            sums[y] += net[y, x] * net[y, x]
            r = Fun()
            r[y, x] += net[y, x] / (1.0 + sums[y])
            net = r
    predict = net.target("predict")
    loss = layers.mse(predict, dsl.input("y")).target("loss")
    opt = layers.gradient_descent(0.05) if rng.random() < 0.6 else layers.adam(eta=0.01)
    return [loss.backprop(opt).target("train"), loss.backwards().grad(dsl.input("x")).target("gx")], width, dims


@pytest.mark.parametrize("seed", seeds("EG_FUZZ_CHAINS", "0:40"))
def test_random_chain_matches_the_oracle(gpu_ctx, monkeypatch, seed):
    from parity import Trio
    monkeypatch.setenv("EG_EPILOGUE_MIN_ELEMS", "0" if seed % 2 else str(1 << 40))
    graphs, width, out_dims = build(seed)
    t = Trio(gpu_ctx, lambda: build(seed)[0], threads=2)
    rng = np.random.default_rng(seed)
    t.init_params(rng, -0.3, 0.3)
    batch = [5, 64, 257, 1500][seed % 4]
    x = (rng.random((batch, width), dtype=np.float32) - 0.5).astype(np.float32)
    y = rng.random((batch, out_dims), dtype=np.float32)
    n = batch * 64      # longest reduction: the loss / a weight gradient over the batch, rows up to 64 wide
    t.call("predict", {"x": x}, n=64)
    t.call("loss", {"x": x, "y": y}, n=n)
    t.call("gx", {"x": x, "y": y}, n=n)
    t.set_epoch(1)
    for _ in range(2):   # eager, then the captured sequence; each from the backend's own state
        t.step("train", {"x": x, "y": y}, n=n)
    t.close()


def build_cnn(seed):
    rng = random.Random(1000 + seed)
    chans = rng.choice([1, 3, 4, 16])
    size = rng.choice([8, 12, 18] if seed < 16 else [8, 12, 18, 30, 44])
    net = dsl.input("x")
    h = w = size
    for depth in range(rng.randint(1, 3)):
        op = rng.choice(["conv", "conv", "unary", "maxpool", "avgpool", "upsample"])
        if op == "conv" and min(h, w) >= 4:
            f = rng.choice([2, 8, 16, 64])
            kh, kw = rng.choice([(1, 1), (3, 3), (2, 3), (3, 1)])
            net = layers.conv2(net, dsl.param([f, kh, kw, chans], name="filters"))
            chans, h, w = f, h - kh + 1, w - kw + 1
        elif op == "maxpool" and h % 2 == 0 and w % 2 == 0:
            net, h, w = layers.maxpool2(net), h // 2, w // 2
        elif op == "avgpool" and h % 2 == 0 and w % 2 == 0:
            net, h, w = layers.avgpool2(net), h // 2, w // 2
        elif op == "upsample" and h * w <= 100:
            net, h, w = layers.upsample2(net), h * 2, w * 2
        else:
            net = unary(rng, net)
    net = dsl.reshape(net, [-1, h * w * chans])
    net = layers.dense(net, h * w * chans, rng.choice([1, 5, 10]))
    predict = net.target("predict")
    loss = layers.mse(predict, dsl.input("y")).target("loss")
    return [loss.backprop(layers.gradient_descent(0.02)).target("train"),
            loss.backwards().grad(dsl.input("x")).target("gx")], (size, size, chans_in(seed)), None


def chans_in(seed):
    return random.Random(1000 + seed).choice([1, 3, 4, 16])


@pytest.mark.parametrize("seed", seeds("EG_FUZZ_CNNS", "0:16"))
def test_random_cnn_matches_the_oracle(gpu_ctx, seed):
    from parity import Trio
    graphs, (h, w, c), _ = build_cnn(seed)
    t = Trio(gpu_ctx, lambda: build_cnn(seed)[0], threads=4)
    rng = np.random.default_rng(seed)
    t.init_params(rng, -0.3, 0.3)
    batch = [2, 9, 40][seed % 3] if seed < 16 else [2, 9, 40, 33, 96][seed % 5]
    x = (rng.random((batch, h, w, c), dtype=np.float32) - 0.5).astype(np.float32)
    n = batch * h * w    # a filter gradient sums over every output pixel of every image
    out = t.call("predict", {"x": x}, n=h * w * 64)
    y = rng.random(out.shape, dtype=np.float32)
    t.call("gx", {"x": x, "y": y}, n=n)
    t.step("train", {"x": x, "y": y}, n=n)
    t.close()
