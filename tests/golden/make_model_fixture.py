"""A saved model assembled BY HAND from the format description of io/serialize.nim — not written by
the library under test.

    store(model)         serialize.nim:344-349   bool isNil; program; params; caches
    integers             serialize.nim:21-39     little endian; `int` and int64 are 8 bytes
    bool                 serialize.nim:27        one byte
    string               serialize.nim:41-44     int64 length, then the bytes
    seq[T]               serialize.nim:46-50     int64 length, then the items
    Table[K, V]          serialize.nim:58-63     int64 count, then key, value pairs
    Tensor[float32]      serialize.nim:65-70     bool isNil, seq[int] shape, then every element (4 bytes)
    TensorId             serialize.nim:333       distinct int -> 8 bytes

The program field of a file this backend reads holds the kernel-description text as a string (a Nim
host stores its own `Program` there and uses eg_model_store_state / _load_state for the rest:
INTEGRATION.md); one int64 with Model.epoch follows the caches.

Content: tests/golden/handwritten/adam_step.kd, parameters p = [1, -2, 3], caches adam.m (tensor 5) and
adam.v (tensor 6) as after one step from zero, epoch 1.
Run:  python tests/golden/make_model_fixture.py  ->  tests/golden/model_fixture.bin (+ .json with the values)
"""
import json
import os
import struct

HERE = os.path.dirname(os.path.abspath(__file__))


def i64(v):
    return struct.pack("<q", v)


def tensor(shape, values):
    out = b"\x00" + i64(len(shape)) + b"".join(i64(d) for d in shape)
    return out + b"".join(struct.pack("<f", v) for v in values)


def table(entries):
    out = i64(len(entries))
    for tid, shape, values in entries:
        out += i64(tid) + tensor(shape, values)
    return out


with open(os.path.join(HERE, "handwritten", "adam_step.kd"), "rb") as f:
    text = f.read()
params = [(1, [3], [1.0, -2.0, 3.0])]
caches = [(5, [3], [0.2, -0.4, 0.6]), (6, [3], [0.004, 0.016, 0.036])]
epoch = 1
blob = b"\x00" + i64(len(text)) + text + table(params) + table(caches) + i64(epoch)
with open(os.path.join(HERE, "model_fixture.bin"), "wb") as f:
    f.write(blob)
with open(os.path.join(HERE, "model_fixture.json"), "w") as f:
    json.dump({"params": {str(t): v for t, _, v in params}, "caches": {str(t): v for t, _, v in caches}, "epoch": epoch,
               "bytes": len(blob), "state_offset": 1 + 8 + len(text)}, f, indent=1)
print(len(blob), "bytes")
