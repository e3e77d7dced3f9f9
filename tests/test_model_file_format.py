"""The saved-model fixture (tests/golden/model_fixture.bin, assembled by hand from the layout of
io/serialize.nim:21-75, 344-349) read back with an independent pure-Python reader: the fixture is
what its description says, byte for byte, before the GPU test trusts it."""
import json
import os
import struct

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


class Reader:
    def __init__(self, data):
        self.d, self.p = data, 0

    def i64(self):
        v = struct.unpack_from("<q", self.d, self.p)[0]
        self.p += 8
        return v

    def byte(self):
        self.p += 1
        return self.d[self.p - 1]

    def table(self):
        out = {}
        for _ in range(self.i64()):
            tid = self.i64()
            assert self.byte() == 0                      # isNil
            shape = [self.i64() for _ in range(self.i64())]
            n = int(np.prod(shape, dtype=np.int64))
            out[tid] = np.array(struct.unpack_from("<%df" % n, self.d, self.p), dtype=np.float32).reshape(shape)
            self.p += 4 * n
        return out


def read_model_file(data):
    r = Reader(data)
    assert r.byte() == 0                                 # model.isNil
    n = r.i64()
    text = data[r.p:r.p + n].decode()
    r.p += n
    state_offset = r.p
    params, caches = r.table(), r.table()
    epoch = r.i64() if r.p + 8 <= len(data) else None
    assert r.p == len(data)
    return text, params, caches, epoch, state_offset


def test_fixture_is_what_its_description_says():
    with open(os.path.join(HERE, "golden", "model_fixture.bin"), "rb") as f:
        data = f.read()
    with open(os.path.join(HERE, "golden", "model_fixture.json")) as f:
        want = json.load(f)
    with open(os.path.join(HERE, "golden", "handwritten", "adam_step.kd")) as f:
        kd_text = f.read()
    text, params, caches, epoch, state_offset = read_model_file(data)
    assert text == kd_text and epoch == want["epoch"] and len(data) == want["bytes"] and state_offset == want["state_offset"]
    assert {str(t): [float(np.float32(v)) for v in vs] for t, vs in ((1, [1.0, -2.0, 3.0]),)} == \
        {str(t): [float(v) for v in a] for t, a in params.items()}
    for t, vs in want["caches"].items():
        assert np.array_equal(caches[int(t)], np.array(vs, dtype=np.float32))
    # the oracle accepts the stored program and continues from the stored state
    from oracle import kd
    ref = kd.Model(text)
    for t, a in params.items():
        ref.params[t][...] = a
    for t, a in caches.items():
        ref.caches[t][...] = a
    ref.epoch = 2
    ref.apply("train", {"t": np.zeros(3, np.float32)})
    assert np.all(np.abs(ref.params[1]) < np.abs(params[1]))      # a second adam step towards the target
