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

    def with_trailer(self, world):
        """One float per rank behind the bucket, in the same storage (what GpuEngine allocates): the rows of every
        rank's shard travel in the step's one all-reduce (DataParallel check_shards)."""
        n = self.bucket.numel()
        self.exchange = torch.zeros(n + world, dtype=torch.float32)
        self.bucket, self.trailer = self.exchange[:n], self.exchange[n:]
        return self

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


def _early_worker(rank, world, port, kind, out_dir):
    """Rank 1 moves to a smaller shard ONE STEP BEFORE rank 0 does (VERDICT r4 next #4b: the likeliest first-contact bug
    of an N > 1 run).  Every step is one all-reduce of the same shape on both ranks, so nothing can pair mismatched
    collectives; with a batch-mean loss the step in which the shards differ is reported as an error (by both ranks, in
    the same step), with a sum-type loss it is simply the sum over both shards."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=__import__("datetime").timedelta(seconds=60))
    outcome = []
    try:
        text, reduction = program(kind)
        engine = OracleEngine(text, "train").with_trailer(world)
        init_params(engine.model)
        dp = DataParallel(engine, reduction=reduction, check_shards=True)
        x, y = make_data("xor" if kind.startswith("xor") else "dense", 32)
        for step in range(5):
            rows = 8 if step >= (2 if rank == 1 else 3) else 16          # rank 1 shrinks at step 2, rank 0 at step 3
            lo = rank * 16
            try:
                dp.step([("x", x[lo:lo + rows]), ("y", y[lo:lo + rows])])
                outcome.append("ok")
            except RuntimeError as exc:
                outcome.append("error: " + str(exc)[:60])
        np.savez(os.path.join(out_dir, f"early{rank}.npz"), outcome=np.array(outcome),
                 **{str(t): p for t, p in engine.model.params.items()})
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["xor_mse", "xor"])
def test_a_rank_that_changes_its_shard_one_step_early_neither_hangs_nor_goes_unnoticed(refcpu, tmp_path, kind):
    world = 2
    mp.spawn(_early_worker, args=(world, free_port(), kind, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = np.load(tmp_path / "early0.npz"), np.load(tmp_path / "early1.npz")
    o0, o1 = list(r0["outcome"]), list(r1["outcome"])
    assert o0 == o1                                  # both ranks see the same thing in the same step
    assert [o.startswith("ok") for o in o0] == [True, True, False, True, True], o0   # step 2: shards of 16 and 8 rows
    assert "16" in o0[2] and "8" in o0[2]
    for key in r0.files:
        if key != "outcome":
            assert np.array_equal(r0[key], r1[key]), "replicas diverged"    # the all-reduce itself stayed well-formed
