"""The two-half batch pipeline (csrc/host/plan_pipeline.cpp): the backward range of a dense training step
cut along the batch, long contractions of both halves on the main lane, the streaming launches on the
side lane under them.  Off by default (measured slower on MI355X, see plan_pipeline.cpp); EG_PIPELINE=1
switches it on, EG_PIPELINE_MIN_FLOPS=0 makes small nets qualify; parity against the float64
shadow and the oracle (tests/parity.py), equality with the unpipelined plan to rounding (reductions over
the batch are formed half + half), and run-to-run determinism."""
import numpy as np
import pytest

from conftest import debug_toggles_active

import refcases
from exprgrad_amd import dsl, layers
from exprgrad_amd import model as egm
from parity import Trio

pytestmark = pytest.mark.gpu


def mlp(dims, act):
    def build():
        net = dsl.input("x")
        for i in range(len(dims) - 1):
            net = layers.dense(net, dims[i], dims[i + 1])
            if i + 2 < len(dims):
                net = act(net)
        net = net.target("predict")
        net = layers.mse(net, dsl.input("y")).target("loss")
        return [net.backprop(layers.gradient_descent(0.05)).target("train")]
    return build


@pytest.mark.parametrize("case", [("softmax", 1024), ("softmax", 4096), ("mlp3", 2048), ("mlp-tanh", 1536)])
def test_pipelined_step_matches_the_oracle(gpu_ctx, monkeypatch, case):
    name, batch = case
    monkeypatch.setenv("EG_PIPELINE", "1")
    monkeypatch.setenv("EG_PIPELINE_MIN_FLOPS", "0")
    monkeypatch.setenv("EG_EPILOGUE_MIN_ELEMS", "0")      # activations ride on the contractions: nothing but cuttable launches
    if name == "softmax":
        graphs, n_in, n_out, onehot = (lambda: refcases.dense_softmax_net(n_in=200, n_hidden=136, n_out=10)), 200, 10, True
    elif name == "mlp3":
        graphs, n_in, n_out, onehot = mlp((96, 160, 72, 8), layers.relu), 96, 8, False
    else:
        graphs, n_in, n_out, onehot = mlp((64, 256, 4), layers.tanh), 64, 4, False
    t = Trio(gpu_ctx, graphs, threads=8)
    rng = np.random.default_rng(batch)
    t.init_params(rng, -0.1, 0.1)
    x = (rng.random((batch, n_in), dtype=np.float32) - 0.5).astype(np.float32)
    y = np.eye(n_out, dtype=np.float32)[rng.integers(0, n_out, size=batch)] if onehot else rng.random((batch, n_out), dtype=np.float32)
    for _ in range(3):          # eager, captured, replayed
        t.step("train", {"x": x, "y": y}, n=batch)
    plan = t.gpu.launch_plan("train")
    # nets whose backward range holds a launch that cannot be cut (here: the map kernels of an 8-wide last
    # layer, too small for an epilogue) keep the plain plan; the others must be pipelined
    if not debug_toggles_active():      # (without row fusion the softmax chain is not pipelinable)
        assert ("batch pipeline" in plan) == (name != "mlp3"), plan
    t.close()


def test_pipelined_and_plain_plans_agree_and_repeat(gpu_ctx, monkeypatch):
    graphs = lambda: refcases.dense_softmax_net(n_in=256, n_hidden=192, n_out=10)
    rng = np.random.default_rng(0)
    batch = 2048
    x = rng.random((batch, 256), dtype=np.float32)
    y = np.eye(10, dtype=np.float32)[rng.integers(0, 10, size=batch)]
    results = []
    for mode in ("plain", "pipe", "pipe"):
        monkeypatch.setenv("EG_EPILOGUE_MIN_ELEMS", "0")
        if mode == "plain":
            monkeypatch.delenv("EG_PIPELINE", raising=False)
        else:
            monkeypatch.setenv("EG_PIPELINE", "1")
            monkeypatch.setenv("EG_PIPELINE_MIN_FLOPS", "0")
        gpu = egm.compile(*graphs(), gpu=gpu_ctx)
        prng = np.random.default_rng(7)
        for tid in gpu.params.ids():
            gpu.params[tid] = (prng.random(gpu.params[tid].shape, dtype=np.float32) * 0.2 - 0.1).astype(np.float32)
        for _ in range(4):
            gpu.apply("train", {"x": x, "y": y})
        if not debug_toggles_active():
            assert ("batch pipeline" in gpu.launch_plan("train")) == (mode == "pipe")
        results.append({t: gpu.params[t] for t in gpu.params.ids()})
        gpu.close()
    for t in results[0]:
        assert np.array_equal(results[1][t], results[2][t]), t                     # run-to-run: bit for bit
        scale = np.max(np.abs(results[0][t]))
        assert np.max(np.abs(results[0][t] - results[1][t])) <= 1e-5 * scale, t      # halves vs whole: rounding only
