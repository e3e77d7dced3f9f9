"""Data-parallel training step: batch sharded over ranks, one all-reduce of the flat
parameter-gradient bucket before the optimizer kernels.

The reference has no multi-device code at all; this step is defined by
BASELINE.json's north_star and pinned by equivalence with the single-device full-batch step
(SURVEY.md §8e):

  * every forward/backward kernel is independent per sample, except the parameter-gradient
    reductions over the batch — so rank r runs the target's kernel list on rows
    [r*B/W, (r+1)*B/W) of every input (the slicing of viewFirst, tensors.nim:290-297) up to,
    not including, the optimizer kernels (gradientDescent, base.nim:37-38);
  * one SUM all-reduce of the gradient bucket (RCCL over xGMI on GPUs: torch.distributed's
    "nccl" backend; gloo in the CPU tests);
  * losses that divide by the batch (`mse`, `crossEntropy`: toScalar(shape[0]) evaluated at run
    time, base.nim:57-67) see B/W instead of B, so the seed gradient gradLoss (passes.nim:594-596)
    is scaled by B_local/B_global ("mean"); sum-type losses (the XOR example) use 1 ("sum");
  * every rank then applies the identical optimizer update to its replica of the parameters.

One process per GPU; the engine's HIP stream is torch's current stream, so the collective is
ordered after the backward kernels and before the update kernels without host synchronisation.
"""
import torch
import torch.distributed as dist


def shard(args, rank, world):
    """Rows [rank*B/world, (rank+1)*B/world) of every input (B must divide evenly)."""
    items = args.items() if isinstance(args, dict) else args
    out = []
    for name, a in items:
        b = a.shape[0]
        if b % world != 0:
            raise ValueError(f"batch {b} of input {name} does not divide over {world} ranks")
        per = b // world
        out.append((name, a[rank * per:(rank + 1) * per]))
    return out


def _check_stream(model):
    """The collective is ordered against the backward and update kernels by the STREAM they share: the
    model's context must have been created on torch's current stream (newGpuContext(dev, stream=...))."""
    want = torch.cuda.current_stream().cuda_stream
    have = model.ctx.stream
    if int(have or 0) != int(want or 0):
        raise RuntimeError("data-parallel step: the model's context runs on another HIP stream than torch's current one; "
                           "the all-reduce would race with the backward / update kernels")


class GpuEngine:
    """Adapter: exprgrad_amd.model.Model -> the engine protocol DataParallel drives."""

    def __init__(self, model, target, rank=None):
        self.model, self.target = model, target
        _check_stream(model)
        if rank is None:
            rank = dist.get_rank() if dist.is_initialized() else 0
        # random tensors (dropout masks) are drawn per element from (seed, draw, tensor, index): with one
        # seed every shard would draw the SAME mask for its rows; fold the rank into the seed
        model.set_seed(0x5EED5EED + 7919 * int(rank))
        _, count = model.grad_bucket(target)
        # gradients live in a torch tensor so torch.distributed can reduce them in place; behind them one float per rank
        # that travels in the SAME all-reduce: the rows of every rank's shard (DataParallel's check of equal shards)
        world = dist.get_world_size() if dist.is_initialized() else 1
        # (a compile[float64] model's bucket holds doubles: eg_model_grad_bucket counts elements)
        import numpy as _np
        dtype = torch.float64 if getattr(model, "dtype", _np.float32) == _np.float64 else torch.float32
        self.exchange = torch.zeros(max(count, 1) + world, dtype=dtype, device="cuda")
        self.bucket = self.exchange[:max(count, 1)]
        self.trailer = self.exchange[max(count, 1):]
        if count > 0:
            model.bind_grad_bucket(target, self.bucket)

    def set_grad_scale(self, s):
        self.model.set_grad_scale(s)

    def run_backward(self, args):
        self.model.run_backward(self.target, args)

    def run_update(self):
        self.model.run_update(self.target)


class DataParallel:
    def __init__(self, engine, reduction="mean", group=None, always_reduce=False, check_shards=None):
        """always_reduce: issue the all-reduce on a one-rank group too (exercises the RCCL path on a
        one-GPU box; a sum over one rank leaves the bucket unchanged).
        check_shards (default: on for "mean"): every step carries the row count of each rank's shard in the step's ONE
        all-reduce (engines that expose `exchange` / `trailer`, one float per rank behind the bucket: no extra
        collective, and the call has the same shape on every rank whatever the shards look like — a rank that changes
        its shard one step before the others cannot pair mismatched collectives).  With a batch-mean loss every shard is
        weighted 1 / world, which is the full-batch step only for EQUAL shards: unequal ones raise RuntimeError —
        checked without stalling the stream, i.e. at a later step or at finish()."""
        if reduction not in ("mean", "sum"):
            raise ValueError("reduction must be 'mean' (loss divides by the batch) or 'sum'")
        self.engine, self.reduction, self.group = engine, reduction, group
        self.always_reduce = always_reduce and dist.is_initialized()
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        if check_shards is None:
            check_shards = reduction == "mean"
        self.check_shards = bool(check_shards) and self.world > 1 and hasattr(engine, "trailer") and \
            engine.trailer.numel() == self.world
        self.steps = 0
        self._rows_src, self._rows = None, None
        self._pending = []     # (step, host copy of the trailer, event or None)

    def _raise_if_unequal(self, step, rows):
        rows = [int(round(float(v))) for v in rows]
        if len(set(rows)) > 1:
            raise RuntimeError(f"data-parallel step {step}: the ranks ran shards of {rows} rows; a batch-mean loss weights "
                               "every shard 1 / world, which equals the full-batch step only for equal shards")

    def _poll(self, block=False):
        while self._pending:
            step, host, event = self._pending[0]
            if event is not None and not block and not event.query():
                return
            if event is not None:
                event.synchronize()
            self._pending.pop(0)
            self._raise_if_unequal(step, host.tolist())

    def finish(self):
        """Wait for the steps queued so far and report a shard mismatch of any of them."""
        self._poll(block=True)

    def step(self, local_args):
        """One training step on this rank's shard.  Asynchronous on GPUs."""
        e = self.engine
        if self.check_shards:
            self._poll()
            items = local_args.items() if isinstance(local_args, dict) else local_args
            rows = int(next(iter(items))[1].shape[0])
            if rows != self._rows:      # (device-side constant, rewritten only when the shard changes)
                src = torch.zeros(self.world, dtype=torch.float32)
                src[self.rank] = float(rows)
                self._rows_src, self._rows = src.to(e.trailer.device), rows
        e.set_grad_scale(1.0 / self.world if self.reduction == "mean" else 1.0)
        e.run_backward(local_args)
        if self.world > 1 or self.always_reduce:
            if self.check_shards:
                e.trailer.copy_(self._rows_src)
                dist.all_reduce(e.exchange, op=dist.ReduceOp.SUM, group=self.group)
                if e.trailer.is_cuda:
                    host = torch.empty(self.world, dtype=torch.float32, pin_memory=True)
                    host.copy_(e.trailer, non_blocking=True)
                    event = torch.cuda.Event()
                    event.record()
                    self._pending.append((self.steps, host, event))
                else:
                    self._pending.append((self.steps, e.trailer.clone(), None))
            else:
                dist.all_reduce(e.bucket, op=dist.ReduceOp.SUM, group=self.group)
        e.run_update()
        self.steps += 1
        if self.check_shards and not e.trailer.is_cuda:
            self._poll()


class RcclGroup:
    """One rank of the C ABI's own data-parallel group (eg_dp_*, include/exprgrad_hip.h group 4):
    RCCL called from the library on the context's stream — what a Nim host binds.  Rank 0 creates
    the id with RcclGroup.unique_id() and the host hands its 128 bytes to the other ranks."""

    def __init__(self, ctx, unique_id, rank, world):
        import ctypes
        from ._lib import call
        if len(unique_id) != 128:
            raise ValueError("the RCCL unique id has 128 bytes")
        self.ctx = ctx
        self._id = (ctypes.c_char * 128).from_buffer_copy(bytes(unique_id))
        h = ctypes.c_void_p()
        call("eg_dp_init", ctx.handle, ctypes.cast(self._id, ctypes.c_void_p), int(rank), int(world), ctypes.byref(h))
        self.handle = h
        self.rank, self.world = int(rank), int(world)

    @staticmethod
    def unique_id():
        import ctypes
        from ._lib import call
        buf = (ctypes.c_char * 128)()
        call("eg_dp_unique_id", ctypes.cast(buf, ctypes.c_void_p))
        return bytes(buf)

    def rccl_count(self):
        """Ranks of the communicator as RCCL reports them (ncclCommCount), -1 if the library cannot say."""
        from ._lib import call
        return call("eg_dp_rccl_count", self.handle)

    def set_split(self, enabled):
        """Allow / forbid the early exchange under the last long contraction (every rank the same)."""
        from ._lib import call
        call("eg_dp_set_split", self.handle, int(bool(enabled)))

    def last_pieces(self):
        from ._lib import call
        return call("eg_dp_last_pieces", self.handle)

    def all_reduce(self, tensor):
        """In-place SUM of a float32 device tensor, asynchronous on the context's stream."""
        import ctypes
        from ._lib import call
        call("eg_dp_allreduce_sum_f32", self.handle, ctypes.c_void_p(tensor.data_ptr()), int(tensor.numel()))

    def close(self):
        from ._lib import call
        if self.handle:
            call("eg_dp_free", self.handle)
            self.handle = None


class NativeDataParallel:
    """DataParallel.step through one C-ABI call (eg_model_step_dp): backward | RCCL all-reduce of the
    library-owned gradient bucket | update, all on the context's stream."""

    def __init__(self, model, target, group, reduction="mean"):
        if reduction not in ("mean", "sum"):
            raise ValueError("reduction must be 'mean' (loss divides by the batch) or 'sum'")
        self.model, self.target, self.group, self.mean = model, target, group, int(reduction == "mean")
        self.world, self.rank = group.world, group.rank
        model.set_seed(0x5EED5EED + 7919 * int(group.rank))   # distinct dropout masks per shard (see GpuEngine)

    def step(self, local_args):
        from ._lib import call
        self.model._bind_all(local_args)
        call("eg_model_step_dp", self.model.handle, self.target.encode(), self.group.handle, self.mean)
