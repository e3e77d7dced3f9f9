"""The reference's GAN example (examples/gan/gan.nim:35-61): `cond` nodes (a different sub-graph per
target), several optimisers over disjoint parameter sets.  CPU: the oracle; GPU: parity with it."""
import numpy as np
import pytest

import refcases
from conftest import TOL, rel_err
import extra_examples
from exprgrad_amd import dsl

DIMS = dict(seed_dim=8, h1=12, h2=16, pixels=20)


def oracle_model():
    from oracle import kd
    return kd.Model(refcases.program_text(extra_examples.gan(**DIMS)), threads=2)


def data(rng, n=6):
    seed = rng.random((n, DIMS["seed_dim"]), dtype=np.float32)
    real = rng.random((n, DIMS["pixels"]), dtype=np.float32)
    labels = np.concatenate([np.ones((n, 1), np.float32), np.zeros((n, 1), np.float32)])
    return seed, real, labels


def test_cond_selects_the_branch_of_the_target():
    text = refcases.program_text(extra_examples.gan(**DIMS))
    m = oracle_model()
    rng = np.random.default_rng(0)
    for tid in m.params:
        m.params[tid][...] = rng.random(m.params[tid].shape, dtype=np.float32) * 0.4 - 0.2
    seed, real, labels = data(rng)
    fake = m.call("gen", {"seed": seed})
    assert fake.shape == (6, DIMS["pixels"])
    # loss.gen sees the generator's output, discr sees the `samples` input: same numbers through both routes
    direct = m.call("discr", {"samples": fake})
    assert rel_err(m.call("loss.gen", {"seed": seed}), np.array([np.sum(direct.astype(np.float64) ** 2) / 6])) <= TOL
    # each optimiser touches its own parameters only (gan.nim:50-58)
    before = {t: m.params[t].copy() for t in m.params}
    m.apply("fit.gen", {"seed": seed})
    changed_gen = {t for t in m.params if not np.array_equal(before[t], m.params[t])}
    before = {t: m.params[t].copy() for t in m.params}
    m.apply("fit.discr", {"samples": np.concatenate([fake, real]), "labels": labels})
    changed_discr = {t for t in m.params if not np.array_equal(before[t], m.params[t])}
    assert len(changed_gen) == 6 and len(changed_discr) == 6 and not (changed_gen & changed_discr)
    assert "cond" not in text                          # resolved by the front-end, nothing new crosses the boundary
    with pytest.raises(dsl.ParserError):
        dsl.to_program(dsl.cond({"a": dsl.input("x")}).target("b"))


@pytest.mark.gpu
def test_gpu_gan_matches_the_oracle(gpu_ctx):
    from parity import Trio
    t = Trio(gpu_ctx, lambda: extra_examples.gan(**DIMS))
    rng = np.random.default_rng(1)
    t.init_params(rng, -0.2, 0.2)
    seed, real, labels = data(rng)
    n = max(DIMS.values()) * len(seed) * 2
    for step in range(3):
        fake = t.call("gen", {"seed": seed}, n=n)
        samples = np.concatenate([fake, real])
        t.call("loss.discr", {"samples": samples, "labels": labels}, n=n)
        t.step("fit.discr", {"samples": samples, "labels": labels}, n=n)
        t.call("loss.gen", {"seed": seed}, n=n)
        t.step("fit.gen", {"seed": seed}, n=n)
    t.close()
