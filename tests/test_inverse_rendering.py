"""The reference's inverse_rendering example (examples/inverse_rendering/inverse_rendering.nim): a
two-sphere ray tracer as ONE `++=` statement per pixel component — sqrt, min / max, boolean `and`,
nested select, toScalar of the iterators — and gradient descent on the sphere colours through its
derivative (a reduction over all pixels).  No known answers in the reference (it writes images);
CPU part: the oracle converges to the target colours.  GPU part: the product against the oracle."""
import numpy as np
import pytest

import refcases
import extra_examples

RED, BLUE = np.array([1, 0, 0], np.float32), np.array([0, 0, 1], np.float32)


def oracle(graphs, threads=8):
    from oracle import kd
    return kd.Model(refcases.program_text(graphs), threads=threads)


def target_image(size):
    scene = extra_examples.inverse_rendering_scene()
    ref = oracle(extra_examples.inverse_rendering(size=size, trainable_colors=False))
    img = ref.call("render", {**scene, "sphere0.color": RED, "sphere1.color": BLUE})
    return scene, ref, np.clip(img, 0, 1)          # renderTargetImage clamps (inverse_rendering.nim:133)


def test_oracle_recovers_the_sphere_colours():
    scene, _, target = target_image(32)
    assert (target[:, :, 0] > 0.6).sum() > 5 and (target[:, :, 2] > 0.6).sum() > 5      # both spheres are visible
    ref = oracle(extra_examples.inverse_rendering(size=32, rate=0.5))
    for tid in ref.params:
        ref.params[tid][...] = 0.5
    first = float(ref.call("loss", {**scene, "target": target})[0])
    for _ in range(8):
        ref.apply("train", {**scene, "target": target})
    assert float(ref.call("loss", {**scene, "target": target})[0]) < 0.01 * first
    found = sorted(tuple(np.round(ref.params[t]).astype(int)) for t in ref.params)
    assert found == [(0, 0, 1), (1, 0, 0)]


@pytest.mark.gpu
def test_gpu_render_matches_the_oracle(gpu_ctx):
    from exprgrad_amd import model as egm
    size = 96
    scene, ref, _ = target_image(size)
    gpu = egm.compile(*extra_examples.inverse_rendering(size=size, trainable_colors=False), gpu=gpu_ctx)
    args = {**scene, "sphere0.color": RED, "sphere1.color": BLUE}
    got, want = gpu.call("render", args), ref.call("render", args)
    assert got.shape == want.shape == (size, size, 3)
    # a pixel on a silhouette may fall on the other side of `t < minDist` / `d >= 0` when sqrt or the
    # division differ in the last bit: allow a handful of such pixels, everything else at 1e-5
    err = np.abs(got.astype(np.float64) - want).max(axis=2)
    off = err > 1e-5
    assert off.sum() <= 4, off.sum()
    assert err[~off].max() <= 1e-5
    gpu.close()


@pytest.mark.gpu
def test_gpu_training_matches_the_oracle(gpu_ctx):
    from parity import Trio
    size = 64
    scene, _, target = target_image(size)
    t = Trio(gpu_ctx, lambda: extra_examples.inverse_rendering(size=size, rate=0.25))
    for tid in sorted(t.ref.params):
        t.set_param(tid, np.full(t.ref.params[tid].shape, 0.5, dtype=np.float32))
    args = {**scene, "target": target}
    n = size * size * 3            # the loss and every colour gradient sum over all pixel components
    t.call("loss", args, n=n)
    for _ in range(4):             # eager, captured, replayed: each step from the backend's own colours
        t.step("train", args, n=n)
    assert t.gpu.kernel_count("train") == t.ref.kernel_count("train")
    t.close()
