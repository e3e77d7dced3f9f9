"""CPU tests of the data-parallel step (exprgrad_amd/parallel.py), gloo backend, world_size 2.

No GPU here, so the compute engine is the ORACLE wrapped to the engine protocol DataParallel
drives (bucket / set_grad_scale / run_backward / run_update).  What is tested is the host logic
the product ships: sharding, the B_local/B_global seed scale, one SUM all-reduce of the flat
gradient bucket, identical updates on every rank — against the single-process full-batch step
(the exactness condition of SURVEY.md §8e).
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import refcases
from exprgrad_amd.parallel import DataParallel, shard


class OracleEngine:
    def __init__(self, text, target):
        from oracle import kd
        self.model, self.target = kd.Model(text), target
        self.pg = self.model.param_grads(target)
        self.sizes = [int(np.prod(self.model.params[p].shape)) for p, _ in self.pg]
        self.bucket = torch.zeros(sum(self.sizes), dtype=torch.float32)
        self.scale = 1.0

    def set_grad_scale(self, s):
        self.scale = s

    def run_backward(self, args):
        self.model.run_backward(self.target, dict(args), grad_scale=self.scale)
        off = 0
        for (_, g), n in zip(self.pg, self.sizes):
            self.bucket[off:off + n] = torch.from_numpy(self.model.last[g].reshape(-1))
            off += n

    def run_update(self):
        off = 0
        for (_, g), n in zip(self.pg, self.sizes):
            self.model.last[g].reshape(-1)[:] = self.bucket[off:off + n].numpy()
            off += n
        self.model.run_update(self.target)


def make_data(kind, batch, seed=0):
    rng = np.random.default_rng(seed)
    if kind == "xor":
        x = rng.integers(0, 2, size=(batch, 2)).astype(np.float32)
        y = (x[:, :1] != x[:, 1:]).astype(np.float32)
    else:
        x = rng.random((batch, 12), dtype=np.float32)
        y = np.eye(5, dtype=np.float32)[rng.integers(0, 5, size=batch)]
    return x, y


def program(kind):
    if kind == "xor":
        return refcases.program_text(refcases.xor_from_scratch()), "sum"
    if kind == "xor_mse":
        return refcases.program_text(refcases.xor_layers()), "mean"
    return refcases.program_text(refcases.dense_softmax_net(n_in=12, n_hidden=8, n_out=5, rate=0.5)), "mean"


def init_params(model, seed=11):
    rng = np.random.default_rng(seed)
    for tid in sorted(model.params):
        model.params[tid][...] = (rng.random(model.params[tid].shape, dtype=np.float32) * 0.2 - 0.1).astype(np.float32)


def _worker(rank, world, port, kind, steps, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        text, reduction = program(kind)
        engine = OracleEngine(text, "train")
        init_params(engine.model)
        dp = DataParallel(engine, reduction=reduction)
        x, y = make_data("xor" if kind.startswith("xor") else "dense", 16 * world)
        for _ in range(steps):
            dp.step(shard({"x": x, "y": y}, rank, world))
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **{str(t): p for t, p in engine.model.params.items()})
    finally:
        dist.destroy_process_group()


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("kind", ["xor", "xor_mse", "dense_softmax"])
def test_two_ranks_equal_full_batch(refcpu, tmp_path, kind):
    world, steps = 2, 3
    mp.spawn(_worker, args=(world, free_port(), kind, steps, str(tmp_path)), nprocs=world, join=True)
    from oracle import kd
    text, _ = program(kind)
    full = kd.Model(text)
    init_params(full)
    x, y = make_data("xor" if kind.startswith("xor") else "dense", 16 * world)
    for _ in range(steps):
        full.apply("train", {"x": x, "y": y})
    r0 = np.load(tmp_path / "rank0.npz")
    r1 = np.load(tmp_path / "rank1.npz")
    for tid, p in full.params.items():
        assert np.array_equal(r0[str(tid)], r1[str(tid)]), "replicas diverged"
        # shard sums are added in a different order than the full-batch reduction: 1e-5, not bitwise
        denom = np.abs(p).max()
        assert np.abs(r0[str(tid)] - p).max() <= 1e-5 * denom, (kind, tid)


def test_virtual_shards_sum_to_full_batch_gradient(refcpu):
    """No processes: sum over W shard gradients (seed scaled by 1/W for mean losses) == full-batch gradient."""
    from oracle import kd
    for kind in ("xor", "dense_softmax"):
        text, reduction = program(kind)
        x, y = make_data("xor" if kind == "xor" else "dense", 32)
        full = kd.Model(text)
        init_params(full)
        full.run_backward("train", {"x": x, "y": y})
        want = {g: full.last[g].copy() for _, g in full.param_grads("train")}
        for world in (2, 4, 8):
            acc = {g: np.zeros_like(v) for g, v in want.items()}
            for r in range(world):
                m = kd.Model(text)
                init_params(m)
                m.run_backward("train", dict(shard({"x": x, "y": y}, r, world)),
                               grad_scale=1.0 / world if reduction == "mean" else 1.0)
                for g in acc:
                    acc[g] += m.last[g]
            for g in want:
                assert np.abs(acc[g] - want[g]).max() <= 1e-5 * max(np.abs(want[g]).max(), 1e-12), (kind, world, g)


def test_shard_rejects_ragged_batches():
    with pytest.raises(ValueError):
        shard({"x": np.zeros((10, 2))}, 0, 4)
    parts = [shard({"x": np.arange(8).reshape(8, 1)}, r, 4)[0][1] for r in range(4)]
    assert np.array_equal(np.concatenate(parts), np.arange(8).reshape(8, 1))
