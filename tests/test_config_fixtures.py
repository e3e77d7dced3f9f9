"""Committed fixtures of downsized BASELINE configurations (tests/golden/config_fixtures.json, written
by tests/golden/make_config_fixtures.py from the oracle): the oracle must keep reproducing them
(CPU: bit for bit where only + and * are involved, 1e-6 where libm is), the HIP path must match them at 1e-5 relative (GPU)."""
import importlib.util
import json
import os

import numpy as np
import pytest

import refcases
from conftest import TOL, rel_err

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("make_config_fixtures", os.path.join(HERE, "golden", "make_config_fixtures.py"))
fx = importlib.util.module_from_spec(spec)
spec.loader.exec_module(fx)
with open(os.path.join(HERE, "golden", "config_fixtures.json")) as f:
    FIXTURES = json.load(f)


def arr(spec_):
    return np.array(spec_["data"], dtype=np.float32).reshape(spec_["shape"])


def run(model, name, is_oracle):
    entry = FIXTURES[name]
    inputs = fx.inputs_of(name)
    params = model.params
    ids = sorted(params) if is_oracle else params.ids()
    got = {}
    if ids:
        for t, v in fx.params_of(name, {t: np.asarray(params[t]) for t in ids}).items():
            if is_oracle:
                params[t][...] = v
            else:
                params[t] = v
        model.apply(entry["target"], inputs)
        model.apply(entry["target"], inputs)
        got["params"] = {str(t): np.asarray(params[t]) for t in ids}
        got["predict"] = model.call("predict", {k: v for k, v in inputs.items() if k == "x"})
    else:
        got["output"] = model.call(entry["target"], inputs)
    return entry, got


def check(entry, got, exact):
    def same(a, b):
        if exact:
            assert np.array_equal(a, b)
        else:
            assert rel_err(a, b) <= TOL
    if "params" in got:
        for t, want in entry["params_after_two_steps"].items():
            same(got["params"][t], arr(want))
        same(got["predict"], arr(entry["predict"]))
    elif "rows" in entry:
        same(got["output"][entry["rows"]], arr(entry["output_rows"]))
        sums = got["output"].astype(np.float64).sum(axis=0)
        assert np.max(np.abs(sums - np.array(entry["column_sums_f64"]))) <= (0 if exact else TOL * np.max(np.abs(sums)))
    else:
        same(got["output"], arr(entry["output"]))


@pytest.mark.parametrize("name", sorted(fx.GRAPHS))
def test_oracle_reproduces_the_committed_fixtures(name):
    from oracle import kd
    model = kd.Model(refcases.program_text(fx.GRAPHS[name][0]()), threads=1)
    entry, got = run(model, name, True)
    if "params_after_two_steps" in entry:
        # exp / ln come from the host libm, whose last bit may depend on the CPU it dispatches for
        def close(a, b):
            assert rel_err(a, b) <= 1e-6
        for t, want in entry["params_after_two_steps"].items():
            close(got["params"][t], arr(want))
        close(got["predict"], arr(entry["predict"]))
    else:
        check(entry, got, exact=True)      # multiplications and additions only: bit for bit


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(fx.GRAPHS))
def test_gpu_matches_the_committed_fixtures(gpu_ctx, name):
    from exprgrad_amd import model as egm
    model = egm.compile(*fx.GRAPHS[name][0](), gpu=gpu_ctx)
    entry, got = run(model, name, False)
    check(entry, got, exact=False)
    model.close()
