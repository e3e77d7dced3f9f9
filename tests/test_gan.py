"""The reference's GAN example (examples/gan/gan.nim:35-61): `cond` nodes (a different sub-graph per
target), several optimisers over disjoint parameter sets.  CPU: the oracle; GPU: parity with it."""
import numpy as np
import pytest

import refcases
from conftest import TOL, rel_err
from exprgrad_amd import dsl, examples

DIMS = dict(seed_dim=8, h1=12, h2=16, pixels=20)


def oracle_model():
    from oracle import kd
    return kd.Model(refcases.program_text(examples.gan(**DIMS)), threads=2)


def data(rng, n=6):
    seed = rng.random((n, DIMS["seed_dim"]), dtype=np.float32)
    real = rng.random((n, DIMS["pixels"]), dtype=np.float32)
    labels = np.concatenate([np.ones((n, 1), np.float32), np.zeros((n, 1), np.float32)])
    return seed, real, labels


def test_cond_selects_the_branch_of_the_target():
    text = refcases.program_text(examples.gan(**DIMS))
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
    from exprgrad_amd import model as egm
    gpu = egm.compile(*examples.gan(**DIMS), gpu=gpu_ctx)
    ref = oracle_model()
    rng = np.random.default_rng(1)
    for tid in sorted(ref.params):
        v = (rng.random(ref.params[tid].shape, dtype=np.float32) * 0.4 - 0.2).astype(np.float32)
        ref.params[tid][...] = v
        gpu.params[tid] = v
    seed, real, labels = data(rng)
    for step in range(3):
        fake_g, fake_r = gpu.call("gen", {"seed": seed}), ref.call("gen", {"seed": seed})
        assert rel_err(fake_g, fake_r) <= TOL
        samples = np.concatenate([fake_r, real])
        assert rel_err(gpu.call("loss.discr", {"samples": samples, "labels": labels}),
                       ref.call("loss.discr", {"samples": samples, "labels": labels})) <= TOL
        gpu.apply("fit.discr", {"samples": samples, "labels": labels})
        ref.apply("fit.discr", {"samples": samples, "labels": labels})
        assert rel_err(gpu.call("loss.gen", {"seed": seed}), ref.call("loss.gen", {"seed": seed})) <= TOL
        gpu.apply("fit.gen", {"seed": seed})
        ref.apply("fit.gen", {"seed": seed})
        for tid in sorted(ref.params):
            assert rel_err(gpu.params[tid], ref.params[tid]) <= 2e-5, (step, tid)
    gpu.close()
