"""Blocking copies of large pageable host arrays (csrc/host_copy.cpp: pinned staging buffers filled by a
thread pool, DMA overlapped with the host copies) — the write / readInto of the runtime group
(cl.nim:111-131) and Model.call's input upload / output download (model.nim:364-376)."""
import numpy as np
import pytest

import refcases
from exprgrad_amd import model as egm

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("floats", [(4 << 20) // 4, (8 << 20) // 4 + 12345, 3 * (8 << 20) // 4, (48 << 20) // 4 + 7, (33 << 20) // 4 - 1])
def test_large_buffers_round_trip_exactly(gpu_ctx, floats):
    rng = np.random.default_rng(floats)
    a = rng.random(floats, dtype=np.float32)
    t = gpu_ctx.allocTensor((floats,))
    t.write(a)
    got = t.read()
    assert np.array_equal(got, a)
    # a second, different array through the same staging buffers; reads interleaved with writes
    b = (a[::-1] * 3).astype(np.float32)
    t.write(b)
    assert np.array_equal(t.read(), b)
    t.buffer.dealloc()


def test_unaligned_and_offset_host_arrays(gpu_ctx):
    base = np.arange((20 << 20) // 4 + 9, dtype=np.float32)
    for off in (0, 1, 3):            # host pointers 0 / 4 / 12 bytes into an allocation
        a = base[off:off + (20 << 20) // 4]
        t = gpu_ctx.allocTensor(a.shape)
        t.write(a)
        out = np.empty_like(base)
        view = out[off:off + a.size]
        t.readInto(view)
        assert np.array_equal(view, a)
        t.buffer.dealloc()


def test_model_call_with_large_host_arrays(gpu_ctx, refcpu):
    n = 1536                        # 9 MiB per operand: the staged path on both directions
    rng = np.random.default_rng(1)
    a = rng.random((n, n), dtype=np.float32)
    b = rng.random((n, n), dtype=np.float32)
    m = egm.compile(*refcases.matmul(), gpu=gpu_ctx)
    for _ in range(3):
        got = m.call("c", {"a": a, "b": b})
    want = a.astype(np.float64) @ b.astype(np.float64)
    assert np.max(np.abs(got - want)) <= 1e-5 * np.max(np.abs(want))
    a2 = np.ascontiguousarray(a.T)
    got2 = m.call("c", {"a": a2, "b": b})
    want2 = a2.astype(np.float64) @ b.astype(np.float64)
    assert np.max(np.abs(got2 - want2)) <= 1e-5 * np.max(np.abs(want2))
    m.close()


def test_pinned_result_blocks_come_in_size_classes_and_can_be_trimmed(gpu_ctx):
    """Results of varying sizes reuse page-locked blocks by size class (no eg_host_alloc per distinct size), and trim()
    returns what nobody references."""
    from exprgrad_amd.runtime import PinnedPool
    pool = PinnedPool()
    assert pool.size_class(3 << 20) == 3 << 20 and pool.size_class((3 << 20) + 4) == 4 << 20 and pool.size_class(1 << 20) == 1 << 20
    a = pool.empty((1 << 18) + 10,)            # 1 MiB + 40 bytes -> the 1.5 MiB class
    a[:] = 1.0
    addr = a.__array_interface__["data"][0]
    del a
    import gc
    gc.collect()
    b = pool.empty((1 << 18) + 4000,)          # another size, same class: the same block
    assert b.__array_interface__["data"][0] == addr
    del b
    gc.collect()
    assert pool._free_bytes == (3 << 19)
    pool.trim()
    assert pool._free_bytes == 0 and not pool._free
