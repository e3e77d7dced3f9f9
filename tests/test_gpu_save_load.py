"""eg_model_save / eg_model_load / eg_model_store_state / eg_model_load_state in the
byte layout of io/serialize.nim:21-75, 344-379, against a fixture assembled by hand from the format
description (tests/golden/make_model_fixture.py) and against the oracle."""
import os

import numpy as np
import pytest

from conftest import TOL, rel_err
from exprgrad_amd import model as egm
from test_model_file_format import read_model_file

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURE = os.path.join(HERE, "golden", "model_fixture.bin")


def test_load_fixture_step_and_compare_with_the_oracle(gpu_ctx, tmp_path):
    from oracle import kd
    with open(FIXTURE, "rb") as f:
        data = f.read()
    text, params, caches, epoch, state_offset = read_model_file(data)
    gpu = egm.load_model(FIXTURE, gpu=gpu_ctx)
    assert gpu.epoch == epoch == 1
    for t, a in params.items():
        assert np.array_equal(gpu.params[t], a)
    for t, a in caches.items():
        assert np.array_equal(gpu.caches[t], a)
    # written back unchanged: the library's writer produces the hand-assembled bytes
    again = tmp_path / "again.bin"
    gpu.save(again)
    assert again.read_bytes() == data
    assert gpu.state_bytes() == data[state_offset:-8]
    # continue training where the file left off, next to the oracle started from the same state
    ref = kd.Model(text)
    for t, a in params.items():
        ref.params[t][...] = a
    for t, a in caches.items():
        ref.caches[t][...] = a
    tgt = np.array([0.5, 0.25, -1.0], np.float32)
    for step in (2, 3, 4):
        gpu.epoch = ref.epoch = step
        gpu.apply("train", {"t": tgt})
        ref.apply("train", {"t": tgt})
    assert rel_err(gpu.params[1], ref.params[1]) <= TOL
    for c in (5, 6):
        assert rel_err(gpu.caches[c], ref.caches[c]) <= TOL
    # the device state (not the initial one) is what gets saved: reload and compare bit for bit
    trained = tmp_path / "trained.bin"
    gpu.save(trained)
    back = egm.load_model(trained, gpu=gpu_ctx)
    assert back.epoch == 4
    assert np.array_equal(back.params[1], gpu.params[1])
    for c in (5, 6):
        assert np.array_equal(back.caches[c], gpu.caches[c])
    _, p2, c2, e2, _ = read_model_file(trained.read_bytes())
    assert e2 == 4 and np.array_equal(p2[1], gpu.params[1]) and np.array_equal(c2[6], gpu.caches[6])
    gpu.close()
    back.close()


def test_state_section_round_trip_into_a_fresh_model(gpu_ctx):
    """What a Nim host does: its own Program in front, the library's params + caches behind it."""
    with open(FIXTURE, "rb") as f:
        data = f.read()
    text, params, caches, _, state_offset = read_model_file(data)
    a = egm.Model(egm._LoadedProgram(text), gpu_ctx)
    used = a.load_state(data[state_offset:])
    assert used == len(data) - state_offset - 8          # stops in front of the epoch extension
    assert np.array_equal(a.params[1], params[1]) and np.array_equal(a.caches[5], caches[5])
    a.close()


def test_load_errors(gpu_ctx, tmp_path):
    from exprgrad_amd._lib import GpuError
    with pytest.raises(GpuError, match="cannot open"):
        egm.load_model(tmp_path / "missing.bin", gpu=gpu_ctx)
    with open(FIXTURE, "rb") as f:
        data = f.read()
    cut = tmp_path / "cut.bin"
    cut.write_bytes(data[:len(data) - 30])
    with pytest.raises(GpuError, match="truncated"):
        egm.load_model(cut, gpu=gpu_ctx)
    nil = tmp_path / "nil.bin"
    nil.write_bytes(b"\x01")
    with pytest.raises(GpuError, match="nil model"):
        egm.load_model(nil, gpu=gpu_ctx)
    text, params, caches, _, state_offset = read_model_file(data)
    m = egm.Model(egm._LoadedProgram(text), gpu_ctx)
    bad = bytearray(data[state_offset:])
    bad[8] = 2                                            # tensor id 2 is an input, not a parameter
    with pytest.raises(GpuError, match="not one of the model's parameters"):
        m.load_state(bytes(bad))
    m.close()


def test_state_tables_must_hold_every_tensor_exactly_once(gpu_ctx):
    """A table that omits a parameter (or names one twice) is refused: it would load as a partly random model, where the
    reference always writes and reads the full tables (io/serialize.nim:344-349)."""
    import struct
    from exprgrad_amd import dsl, layers
    from exprgrad_amd._lib import GpuError
    net = layers.dense(dsl.input("x"), 3, 2).target("predict")
    net = layers.mse(net, dsl.input("y")).target("loss")
    model = egm.compile(net.backprop(layers.gradient_descent(0.1)).target("train"), gpu=gpu_ctx)
    state = model.state_bytes()
    count = struct.unpack_from("<q", state, 0)[0]
    assert count == 2                                   # weights [3, 2] and bias [2]

    def entry_end(pos):                                 # int64 id, bool isNil, seq[int] shape, float32 elements
        pos += 8 + 1
        rank = struct.unpack_from("<q", state, pos)[0]
        dims = struct.unpack_from("<%dq" % rank, state, pos + 8)
        n = 1
        for d in dims:
            n *= d
        return pos + 8 + 8 * rank + 4 * n
    first_end = entry_end(8)
    second_end = entry_end(first_end)
    rest = state[second_end:]                           # the (empty) cache table
    model.load_state(state)                             # the full tables load
    only_first = struct.pack("<q", 1) + state[8:first_end] + rest
    with pytest.raises(GpuError, match="lacks tensor"):
        model.load_state(only_first)
    twice = struct.pack("<q", 2) + state[8:first_end] + state[8:first_end] + rest
    with pytest.raises(GpuError, match="appears twice"):
        model.load_state(twice)
    model.close()
