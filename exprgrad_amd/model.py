"""Host-side mirror of exprgrad/model.nim for GPU targets, over group 3 of the C ABI.

    model = compile(net, gpu=ctx)                 # compile[float32](net, gpu=ctx)   model.nim:270-273
    out = model.call("predict", {"x": x})         # model.nim:392-406
    model.apply("train", {"x": x, "y": y})        # model.nim:408-411
    model.fit("train", {"x": x, "y": y}, batch_size=32)                             # model.nim:413-454
    model.params[tensor_id]                       # Model.params                     model.nim:37

Inputs may be numpy arrays (copied host->device on every call, as the reference's writeInput
does, model.nim:364-368) or device tensors (anything with .data_ptr() and .shape, e.g. a torch
tensor: borrowed, no copy).  Outputs are fresh numpy arrays (readOutput, model.nim:375-376).
"""
import ctypes

import numpy as np

from . import dsl
from ._lib import call, GpuError, RuntimeErrorEG  # noqa: F401
from .runtime import GpuContext


def _check_device_tensor(name, tensor, ctx, want="float32"):
    """Only the address and the shape of a borrowed device tensor cross the C ABI: anything that is not
    packed float32 on the context's device would be read as if it were (a transposed view, a float64
    or half tensor), silently.  Refuse it instead."""
    dtype = getattr(tensor, "dtype", None)
    if dtype is not None and str(dtype) not in ("torch." + want, want):
        raise GpuError(f"input {name}: device tensors must be {want}, got {dtype}")
    if hasattr(tensor, "is_contiguous") and not tensor.is_contiguous():
        raise GpuError(f"input {name}: device tensors must be contiguous (row-major, packed); call .contiguous()")
    dev = getattr(tensor, "device", None)
    if dev is not None and hasattr(dev, "type"):
        if dev.type != "cuda":
            raise GpuError(f"input {name}: tensor lives on {dev}, not on a GPU")
        want = getattr(ctx, "device", None)
        if want is not None and dev.index is not None and dev.index != want:
            raise GpuError(f"input {name}: tensor lives on GPU {dev.index}, the context on GPU {want}")


class _Params:
    """dict-like view of the device-resident parameters: get copies D2H, set copies H2D."""

    def __init__(self, model, shapes=None):
        self._m = model
        self._shapes = model._param_shapes if shapes is None else shapes
        self._dtype = model.dtype
        self._sfx = model._sfx

    def ids(self):
        return list(self._shapes)

    def __iter__(self):
        return iter(self._shapes)

    def __len__(self):
        return len(self._shapes)

    def __getitem__(self, tid):
        shape = self._shapes[tid]
        out = np.empty(shape, dtype=self._dtype)
        call("eg_model_param_read" + self._sfx, self._m.handle, int(tid), out.ctypes.data_as(ctypes.c_void_p), out.size)
        return out

    def __setitem__(self, tid, value):
        shape = self._shapes[tid]
        arr = np.ascontiguousarray(value, dtype=self._dtype)
        if list(arr.shape) != list(shape):
            raise GpuError(f"parameter {tid} has shape {list(shape)}, got {list(arr.shape)}")
        call("eg_model_param_write" + self._sfx, self._m.handle, int(tid), arr.ctypes.data_as(ctypes.c_void_p), arr.size)

    def items(self):
        return [(t, self[t]) for t in self.ids()]


class Model:
    def __init__(self, program, ctx, handle=None):
        if not isinstance(ctx, GpuContext):
            raise GpuError("compile(..., gpu=ctx) needs a GpuContext: this backend has no CPU path")
        self.ctx = ctx
        self.program = program
        self.source_text = program.to_text()
        if handle is None:
            handle = ctypes.c_void_p()
            call("eg_model_compile", ctx.handle, self.source_text.encode(), ctypes.byref(handle))
        self.handle = handle
        # Model[T] (model.nim:21-43): float32, or float64 for a program whose header says `kd 1 f64`
        self.dtype = np.float64 if call("eg_model_scalar_bytes", handle) == 8 else np.float32
        self._sfx = "_f64" if self.dtype == np.float64 else ""
        self._param_shapes = {}
        self._cache_shapes = {}
        for tid, t in enumerate(program.tensors, 1):
            if t["kind"] == "param":
                self._param_shapes[tid] = list(t["shape"])
            elif t["kind"] == "cache":
                self._cache_shapes[tid] = list(t["shape"])
        self.params = _Params(self)                          # Model.params (model.nim:37)
        self.caches = _Params(self, self._cache_shapes)      # Model.caches (model.nim:38): adam.m / adam.v
        self._keep = {}

    # ---- introspection -----------------------------------------------------------------------
    def emit_ir(self):
        """emitIr (model.nim:262-264): the lowered plan, one line per kernel."""
        from . import _lib
        return (_lib.lib().eg_model_plan_text(self.handle) or b"").decode()

    emitIr = emit_ir

    def launch_plan(self, target):
        """The launch sequence the target last ran with (fusion groups, fused epilogues), one line per launch."""
        from . import _lib
        return (_lib.lib().eg_model_launch_text(self.handle, target.encode()) or b"").decode()

    def kernel_count(self, target):
        return call("eg_model_kernel_count", self.handle, target.encode())

    @property
    def epoch(self):
        from . import _lib
        return _lib.lib().eg_model_epoch(self.handle)

    @epoch.setter
    def epoch(self, v):
        call("eg_model_set_epoch", self.handle, int(v))

    def keep_values(self, on=True):
        """Plans keep the values of every result tensor (no predicate-bit tensors): read_tensor works for any
        intermediate, as in the reference where every kernel's output is a tensor (model.nim:295-300)."""
        call("eg_model_keep_values", self.handle, 1 if on else 0)

    def set_seed(self, seed):
        """Seed of the model's random tensors (`rand`, dropout masks); the reference uses Nim's global
        generator (`randomize(seed)`)."""
        call("eg_model_set_seed", self.handle, int(seed) & 0xFFFFFFFFFFFFFFFF)

    # ---- inputs ------------------------------------------------------------------------------
    def _bind(self, name, tensor):
        if hasattr(tensor, "data_ptr"):  # device tensor: borrow
            _check_device_tensor(name, tensor, self.ctx, np.dtype(self.dtype).name)
            shape = [int(s) for s in tensor.shape]
            arr = (ctypes.c_int64 * max(len(shape), 1))(*shape)
            self._keep[name] = tensor
            call("eg_model_set_input_device" + self._sfx, self.handle, name.encode(), ctypes.c_void_p(tensor.data_ptr()), len(shape), arr)
        else:
            a = np.ascontiguousarray(tensor, dtype=self.dtype)
            arr = (ctypes.c_int64 * max(a.ndim, 1))(*a.shape)
            call("eg_model_set_input_host" + self._sfx, self.handle, name.encode(), a.ctypes.data_as(ctypes.c_void_p), a.ndim, arr)

    def _bind_all(self, args):
        items = list(args.items() if isinstance(args, dict) else args)
        # The same device tensors as in the previous call (a training loop on resident data): the library still has
        # them bound — skip three C-ABI calls per step (a launch-bound step is 14 us of GPU time; the host must not be
        # the slower side).  Host arrays are uploaded on every call: their contents may have changed.
        sig = None
        if items and all(hasattr(t, "data_ptr") for _, t in items):
            sig = tuple((name, t.data_ptr(), tuple(t.shape)) for name, t in items)
            if sig == getattr(self, "_bound_sig", None):
                return
        self._bound_sig = None
        call("eg_model_clear_inputs", self.handle)
        self._keep = {}
        for name, tensor in items:
            self._bind(name, tensor)
        self._bound_sig = sig

    # ---- call / apply / fit ------------------------------------------------------------------
    def call(self, target, args=()):
        """call (model.nim:392-406)."""
        self._bind_all(args)
        call("eg_model_run", self.handle, target.encode())
        return self._read_output(target)

    def apply(self, target, args=()):
        """apply = discard call (model.nim:408-411), without the device->host copy."""
        self._bind_all(args)
        call("eg_model_run", self.handle, target.encode())

    def _read_output(self, target):
        tgt = self.program.targets[target]
        if tgt.output == 0:
            return None
        rank = ctypes.c_int(0)
        shape = (ctypes.c_int64 * 8)()
        call("eg_model_output_shape", self.handle, target.encode(), ctypes.byref(rank), shape)
        from .runtime import pinned_pool
        if self.dtype == np.float64:
            out = np.empty([shape[i] for i in range(rank.value)], dtype=np.float64)
        else:
            out = pinned_pool.empty([shape[i] for i in range(rank.value)])   # fresh array, recycled pinned block when large
        call("eg_model_read_output" + self._sfx, self.handle, target.encode(), out.ctypes.data_as(ctypes.c_void_p), out.size)
        return out

    def read_tensor(self, target, tensor_id):
        rank = ctypes.c_int(0)
        shape = (ctypes.c_int64 * 8)()
        call("eg_model_tensor_shape", self.handle, target.encode(), int(tensor_id), ctypes.byref(rank), shape)
        out = np.empty([shape[i] for i in range(rank.value)], dtype=self.dtype)
        call("eg_model_read_tensor" + self._sfx, self.handle, target.encode(), int(tensor_id), out.ctypes.data_as(ctypes.c_void_p),
             out.size)
        return out

    def fit(self, target, args, batch_size=32, log_status=False):
        """fit (model.nim:413-454): one epoch of mini-batches; the tail that does not fill a batch is
        dropped (batchCount = shape[0] div batchSize), Model.epoch is bumped once.  The loop itself
        is eg_model_fit: the data set goes to the device once, every batch is two launches."""
        items = list(args.items() if isinstance(args, dict) else args)
        if not items:
            raise RuntimeErrorEG("Model.fit requires at least one input tensor. Use Model.apply instead if the "
                                 "target has zero inputs.")
        n = len(items)
        names = (ctypes.c_char_p * n)(*[name.encode() for name, _ in items])
        data = (ctypes.c_void_p * n)()
        on_device = (ctypes.c_int * n)()
        ranks = (ctypes.c_int * n)()
        shapes = (ctypes.c_int64 * (8 * n))()
        keep = []
        for i, (_, tensor) in enumerate(items):
            if hasattr(tensor, "data_ptr"):      # device tensor: read in place
                if hasattr(tensor, "is_contiguous") and not tensor.is_contiguous():
                    tensor = tensor.contiguous()
                _check_device_tensor(items[i][0], tensor, self.ctx, np.dtype(self.dtype).name)
                data[i], on_device[i], shape = tensor.data_ptr(), 1, [int(s) for s in tensor.shape]
            else:
                tensor = np.ascontiguousarray(tensor, dtype=self.dtype)
                data[i], on_device[i], shape = tensor.ctypes.data, 0, list(tensor.shape)
            keep.append(tensor)
            if len(shape) > 8:
                raise GpuError("inputs have at most 8 dimensions")
            ranks[i] = len(shape)
            shapes[8 * i:8 * i + len(shape)] = shape
        self._keep = {"fit": keep}
        self._bound_sig = None   # the library binds batch slices of its own during the epoch
        call("eg_model_fit" + self._sfx, self.handle, target.encode(), n, names, data, on_device, ranks, shapes, int(batch_size))

    # ---- data-parallel hooks ---------------------------------------------------
    def grad_bucket(self, target):
        """(device pointer, float count) of the flat parameter-gradient bucket of `target`."""
        p = ctypes.c_void_p()
        n = ctypes.c_int64(0)
        call("eg_model_grad_bucket", self.handle, target.encode(), ctypes.byref(p), ctypes.byref(n))
        return p.value or 0, n.value

    def bind_grad_bucket(self, target, tensor):
        """Let the gradients of `target` live in caller-owned device memory (a torch tensor)."""
        self._bucket_keep = tensor
        call("eg_model_bind_grad_bucket", self.handle, target.encode(), ctypes.c_void_p(tensor.data_ptr()),
             int(tensor.numel()))

    def run_backward(self, target, args=None):
        if args is not None:
            self._bind_all(args)
        call("eg_model_run_backward", self.handle, target.encode())

    def run_update(self, target):
        call("eg_model_run_update", self.handle, target.encode())

    def set_grad_scale(self, scale):
        call("eg_model_set_grad_scale", self.handle, float(scale))

    # ---- save / load (io/serialize.nim:344-379) ------------------------------------------------
    def save(self, path):
        """model.save(path) (serialize.nim:366-369): isNil, program, params, caches in the reference's
        byte layout, written by the library (eg_model_save) from the DEVICE state — the reference never
        copies GPU-side updates back (model.nim:326-345).  The program field holds the
        kernel-description text; Model.epoch follows the caches (the reference forgets it)."""
        call("eg_model_save", self.handle, str(path).encode())

    def state_bytes(self):
        """The `params` + `caches` tables as serialize.nim:348-349 writes them (what a Nim host appends
        to its own `store(program)`)."""
        n = ctypes.c_size_t(0)
        call("eg_model_state_bytes", self.handle, ctypes.byref(n))
        buf = (ctypes.c_ubyte * max(n.value, 1))()
        call("eg_model_store_state", self.handle, buf, n.value, ctypes.byref(n))
        return bytes(buf[:n.value])

    def load_state(self, data):
        buf = (ctypes.c_ubyte * max(len(data), 1)).from_buffer_copy(bytes(data) or b"\0")
        used = ctypes.c_size_t(0)
        call("eg_model_load_state", self.handle, buf, len(data), ctypes.byref(used))
        return used.value

    def close(self):
        if self.handle:
            call("eg_model_free", self.handle)
            self.handle = None


class _LoadedProgram:
    """What Model needs from a Program when it is rebuilt from saved text."""

    def __init__(self, text):
        self._text = text
        self.tensors = []
        self.targets = {}
        for line in text.splitlines():
            tok = line.split()
            if not tok:
                continue
            if tok[0] == "tensor":
                rank = int(tok[4])
                self.tensors.append({"kind": tok[2], "name": "" if tok[3] == "-" else tok[3],
                                     "shape": [int(v) for v in tok[5:5 + rank]] if rank >= 0 else None})
            elif tok[0] == "target":
                self.targets[tok[1]] = type("T", (), {"output": int(tok[2])})()

    def to_text(self):
        return self._text


def load_model(path, gpu=None):
    """loadModel(path, gpu) (serialize.nim:351-364) through eg_model_load: read the file, rebuild the
    kernels from the stored program, restore parameters, caches and epoch on the device."""
    if not isinstance(gpu, GpuContext):
        raise GpuError("load_model(path, gpu=ctx) needs a GpuContext: this backend has no CPU path")
    from . import _lib
    h = ctypes.c_void_p()
    call("eg_model_load", gpu.handle, str(path).encode(), ctypes.byref(h))
    text = (_lib.lib().eg_model_source_text(h) or b"").decode()
    return Model(_LoadedProgram(text), gpu, handle=h)


loadModel = load_model


def compile(*graphs, gpu=None, dtype=np.float32):  # noqa: A001 - mirrors compile[T](graphs, gpu) model.nim:270-273
    """dtype: the T of compile[T] — np.float32 or np.float64 (model.nim:253-260: any other type raises ValueError)."""
    if len(graphs) == 1 and isinstance(graphs[0], (list, tuple)):
        graphs = tuple(graphs[0])
    dt = np.dtype(dtype)
    if dt not in (np.dtype(np.float32), np.dtype(np.float64)):
        raise ValueError(f"{dt} is not a valid scalar type")   # model.nim:259
    program = dsl.to_program(*graphs)
    program.scalar = "f64" if dt == np.dtype(np.float64) else "f32"
    return Model(program, gpu)
