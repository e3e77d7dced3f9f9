"""Host-side mirror of exprgrad/runtimes/gpu.nim over the C ABI.

Same names, argument meaning and error behaviour as the reference procs
(gpu.nim:24-76, implemented for OpenCL in cl.nim:45-207): listDevices, newGpuContext,
allocBuffer, write, fill, readInto, read, compile, arg, run, GpuTensor.  snake_case aliases are
provided (Nim identifiers are style-insensitive, the reference itself mixes both).
"""
import ctypes
import os

import numpy as np

from . import _lib
from ._lib import GpuError, call


class GpuDevice:
    """gpu.nim:26 / cl.nim:20-21."""

    def __init__(self, index):
        self.index = index

    def _info(self):
        name = ctypes.create_string_buffer(256)
        vendor = ctypes.create_string_buffer(256)
        version = ctypes.create_string_buffer(256)
        is_gpu = ctypes.c_int(0)
        call("eg_device_info", self.index, name, 256, vendor, 256, version, 256, ctypes.byref(is_gpu))
        return name.value.decode(), vendor.value.decode(), version.value.decode(), bool(is_gpu.value)

    @property
    def name(self):
        return self._info()[0]

    @property
    def vendor(self):
        return self._info()[1]

    @property
    def version(self):
        return self._info()[2]

    @property
    def isGpu(self):
        return self._info()[3]

    is_gpu = isGpu


def listDevices():
    """gpu.nim:34 / cl.nim:64-66."""
    n = ctypes.c_int(0)
    call("eg_device_count", ctypes.byref(n))
    return [GpuDevice(i) for i in range(n.value)]


list_devices = listDevices


class GpuContext:
    """gpu.nim:27 / cl.nim:23-26: a device plus one in-order queue (here: one HIP stream)."""

    def __init__(self, device=None, stream=None):
        if device is None:
            devices = listDevices()
            if not devices:
                raise GpuError("Unable to find device")  # cl.nim:97-98
            device = devices[0]
        if isinstance(device, GpuDevice):
            device = device.index
        self.device = int(device)
        h = ctypes.c_void_p()
        if stream is None:
            call("eg_ctx_create", self.device, ctypes.byref(h))
        else:
            call("eg_ctx_create_on_stream", self.device, ctypes.c_void_p(int(stream)), ctypes.byref(h))
        self.handle = h

    def sync(self):
        call("eg_ctx_sync", self.handle)

    @property
    def stream(self):
        return _lib.lib().eg_ctx_stream(self.handle) or 0

    # allocBuffer(ctx, size): gpu.nim:41
    def allocBuffer(self, size):
        return GpuBuffer(self, int(size))

    alloc_buffer = allocBuffer

    # compile(ctx, name, source) / compile(ctx, GpuKernelSource): gpu.nim:48-49
    def compile(self, name, source=None):
        if source is None and isinstance(name, GpuKernelSource):
            name, source = name.name, name.source
        return GpuKernel(self, name, source)

    def allocTensor(self, shape):
        return GpuTensor(self, shape)

    alloc_tensor = allocTensor

    def close(self):
        if self.handle:
            _lib.lib().eg_ctx_destroy(self.handle)
            self.handle = None
            pinned_pool.trim()      # (page-locked result blocks nobody references any more)


def newGpuContext(device=None, stream=None):
    """gpu.nim:39-40 / cl.nim:83-99."""
    return GpuContext(device, stream)


new_gpu_context = newGpuContext


class GpuKernelSource:
    """cl.nim:33-35."""

    def __init__(self, name, source):
        self.name, self.source = name, source


class GpuBuffer:
    """gpu.nim:28 / cl.nim:28-31."""

    def __init__(self, ctx, size, wrap_ptr=None):
        self.ctx, self.size = ctx, size
        h = ctypes.c_void_p()
        if wrap_ptr is None:
            call("eg_buf_alloc", ctx.handle, size, ctypes.byref(h))
        else:
            call("eg_buf_wrap", ctx.handle, ctypes.c_void_p(int(wrap_ptr)), size, ctypes.byref(h))
        self.handle = h

    @property
    def ptr(self):
        return _lib.lib().eg_buf_ptr(self.handle) or 0

    def write(self, data):
        """write(buffer, data): gpu.nim:42-43 (blocking; size must match, cl.nim:112-113)."""
        arr = np.ascontiguousarray(data)
        call("eg_buf_write", self.handle, arr.ctypes.data_as(ctypes.c_void_p), arr.nbytes)

    def fill(self, value, dtype=np.float32):
        """fill[T](buffer, value): gpu.nim:44 (asynchronous)."""
        v = np.array([value], dtype=dtype)
        call("eg_buf_fill", self.handle, v.ctypes.data_as(ctypes.c_void_p), v.nbytes)

    def readInto(self, out):
        """readInto(buffer, data): gpu.nim:45-46 (blocking; cl.nim:134-135 size check)."""
        assert out.flags.c_contiguous
        call("eg_buf_read", self.handle, out.ctypes.data_as(ctypes.c_void_p), out.nbytes)
        return out

    read_into = readInto

    def read(self, dtype=np.float32):
        """read[T](buffer): gpu.nim:47 / cl.nim:141-147."""
        item = np.dtype(dtype).itemsize
        if self.size % item != 0:
            raise GpuError("Buffer size is not divisible by item type size")
        return self.readInto(np.empty(self.size // item, dtype=dtype))

    def dealloc(self):
        if self.handle:
            call("eg_buf_free", self.handle)
            self.handle = None


class GpuKernel:
    """gpu.nim:30 / cl.nim:37-39.  Arguments are sticky, `arg` returns the kernel for chaining."""

    def __init__(self, ctx, name, source):
        self.ctx, self.name = ctx, name
        h = ctypes.c_void_p()
        call("eg_kernel_compile", ctx.handle, name.encode(), source.encode(), ctypes.byref(h))
        self.handle = h
        self._keep = {}

    def arg(self, index, value):
        if isinstance(value, GpuBuffer):
            self._keep[index] = value  # buffers must outlive the kernel's sticky args
            call("eg_kernel_set_arg_buf", self.handle, index, value.handle)
        elif isinstance(value, (float, np.floating)):
            if isinstance(value, np.float64):
                call("eg_kernel_set_arg_f64", self.handle, index, float(value))
            else:
                call("eg_kernel_set_arg_f32", self.handle, index, float(value))
        else:
            call("eg_kernel_set_arg_i64", self.handle, index, int(value))
        return self

    def run(self, groupSize, localSize):
        """run(kernel, groupSize, localSize): gpu.nim:52 / cl.nim:190-207."""
        if len(groupSize) == 0:
            raise GpuError("Group size must have at least one dimension")
        if len(groupSize) != len(localSize):
            raise GpuError("Dimension of group size must equal dimension of local size")
        n = len(groupSize)
        g = (ctypes.c_int64 * n)(*[int(v) for v in groupSize])
        l = (ctypes.c_int64 * n)(*[int(v) for v in localSize])
        call("eg_kernel_launch", self.handle, n, g, l)

    def free(self):
        if self.handle:
            call("eg_kernel_free", self.handle)
            self.handle = None


class GpuTensor:
    """gpu.nim:54-76: shape + buffer, float32."""

    def __init__(self, ctx, shape):
        self.shape = [int(s) for s in shape]
        self.buffer = ctx.allocBuffer(int(np.prod(self.shape, dtype=np.int64)) * 4)

    def write(self, tensor):
        self.buffer.write(np.ascontiguousarray(tensor, dtype=np.float32))

    def fill(self, value):
        self.buffer.fill(value, np.float32)

    def readInto(self, tensor):
        assert list(tensor.shape) == self.shape
        self.buffer.readInto(tensor)
        return tensor

    def read(self):
        return self.readInto(np.empty(self.shape, dtype=np.float32))

    @property
    def ptr(self):
        return self.buffer.ptr



class PinnedPool:
    """Result arrays in pinned host memory (eg_host_alloc), recycled: `call` returns a FRESH array every
    time (readOutput, model.nim:375-376), but a fresh 64 MiB numpy array page-faults inside the
    device-to-host copy.  Arrays handed out here are numpy views of pinned blocks; when the last view
    of a block is garbage collected the block goes back to the pool and the next result of that size
    class reuses it: direct DMA, no page faults.

    Blocks come in size classes (powers of two and 1.5 x powers of two: at most 33 % slack), so a model
    whose output size varies from call to call does not page-lock a new block (milliseconds) per size.
    `trim()` gives the free blocks back; EG_NO_PINNED_POOL=1 switches the pool off (plain numpy arrays).
    The arrays are ordinary numpy arrays to the caller with one caveat that comes with pinned memory: a
    host that starts its OWN asynchronous copy out of such an array on another stream must keep the array
    alive until that copy has finished — once garbage collected, its block may be the target of the next
    result's DMA."""

    MIN_BYTES = 1 << 20
    MAX_FREE_BYTES = 1 << 30

    def __init__(self):
        self._free = {}
        self._free_bytes = 0
        self.enabled = os.environ.get("EG_NO_PINNED_POOL", "0") in ("", "0")

    @staticmethod
    def size_class(nbytes):
        p = 1 << (nbytes - 1).bit_length()          # next power of two
        return p * 3 // 4 if nbytes <= p * 3 // 4 else p

    def empty(self, shape):
        n = int(np.prod(shape, dtype=np.int64))
        nbytes = n * 4
        if nbytes < self.MIN_BYTES or not self.enabled:
            return np.empty(shape, dtype=np.float32)
        import weakref
        cls = self.size_class(nbytes)
        blocks = self._free.get(cls)
        if blocks:
            ptr = blocks.pop()
            self._free_bytes -= cls
        else:
            p = ctypes.c_void_p()
            call("eg_host_alloc", cls, ctypes.byref(p))
            ptr = p.value
        buf = (ctypes.c_float * n).from_address(ptr)
        weakref.finalize(buf, self._release, ptr, cls)
        return np.frombuffer(buf, dtype=np.float32).reshape(shape)

    def _release(self, ptr, cls):
        if self._free_bytes + cls > self.MAX_FREE_BYTES:
            self._free_block(ptr)
            return
        self._free.setdefault(cls, []).append(ptr)
        self._free_bytes += cls

    @staticmethod
    def _free_block(ptr):
        try:
            call("eg_host_free", ctypes.c_void_p(ptr))
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass

    def trim(self):
        """Give every free block back to the system (blocks still referenced by live arrays stay)."""
        for blocks in self._free.values():
            for ptr in blocks:
                self._free_block(ptr)
        self._free.clear()
        self._free_bytes = 0


pinned_pool = PinnedPool()
