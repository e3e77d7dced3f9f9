// save / loadModel for GPU models, in the byte layout of io/serialize.nim.
//
// The reference writes a model as (serialize.nim:344-349)
//     store(model.isNil); store(model.program); store(model.params); store(model.caches)
// with the primitive rules of serialize.nim:21-75: integers little endian (`int` = 8 bytes), bool one
// byte, string = int64 length + bytes, seq = int64 length + items, Table = int64 count + (key, value)
// pairs, Tensor[T] = bool isNil, then seq[int] shape, then the elements (float32: 4 bytes each).
// Its GPU path never copies trained parameters back (model.nim:326-345: stateLocation only grows),
// so a model trained on the GPU is saved with its initial parameters; here the device state is what
// gets written ("flush first").
//
// Two levels:
//   eg_model_state_bytes / eg_model_store_state / eg_model_load_state
//       the `params` and `caches` tables exactly as serialize.nim:348-349 / 360-361 write and read
//       them.  A Nim host stores `isNil` and its own `Program` with the reference's procs and puts
//       these bytes behind them; the library never sees a Nim `Program`.
//   eg_model_save / eg_model_load
//       a whole file for hosts without the Nim front-end: same container, the `program` field holds
//       what this backend compiles from — the kernel-description text, as a serialize.nim string —
//       and one trailing int64 carries Model.epoch (the reference forgets it; a reader that stops
//       after `caches`, as serialize.nim:351-364 does, never sees it).
#include <cstdio>

#include "model_types.hpp"

using namespace eg::kd;
using namespace eg::model;
using eg::set_error;

namespace {

struct Writer {
  std::vector<unsigned char>* out = nullptr;  // nullptr: count only
  size_t n = 0;
  void byte(unsigned char b) {
    if (out) out->push_back(b);
    ++n;
  }
  void u64(uint64_t v) {  // serialize.nim:21-25
    for (int i = 0; i < 8; ++i) byte((unsigned char)((v >> (8 * i)) & 0xff));
  }
  void i64(int64_t v) { u64((uint64_t)v); }
  void u32(uint32_t v) {
    for (int i = 0; i < 4; ++i) byte((unsigned char)((v >> (8 * i)) & 0xff));
  }
  void str(const std::string& s) {  // serialize.nim:41-44
    i64((int64_t)s.size());
    for (char c : s) byte((unsigned char)c);
  }
};

struct Reader {
  const unsigned char* p = nullptr;
  size_t n = 0, pos = 0;
  bool ok = true;
  unsigned char byte() {
    if (pos >= n) {
      ok = false;
      return 0;
    }
    return p[pos++];
  }
  uint64_t u64() {  // serialize.nim:83-85
    uint64_t v = 0;
    for (int i = 0; i < 8; ++i) v |= (uint64_t)byte() << (8 * i);
    return v;
  }
  int64_t i64() { return (int64_t)u64(); }
  uint32_t u32() {
    uint32_t v = 0;
    for (int i = 0; i < 4; ++i) v |= (uint32_t)byte() << (8 * i);
    return v;
  }
};

// Table[TensorId, Tensor[float32]] of every tensor of `kind`, ascending id (a Nim Table iterates in
// hash order; any order loads the same).
int store_table(eg_model* m, TK kind, Writer& w) {
  int64_t count = 0;
  for (auto& p : m->params)
    if (m->prog.tensors[p.first].kind == kind) ++count;
  w.i64(count);
  std::vector<float> host;
  for (auto& p : m->params) {
    if (m->prog.tensors[p.first].kind != kind) continue;
    w.i64(p.first);                       // TensorId = distinct int (serialize.nim:333)
    w.byte(0);                            // tensor.isNil = false (serialize.nim:64)
    w.i64((int64_t)p.second.shape.size());  // seq[int] shape
    for (long d : p.second.shape) w.i64(d);
    if (!w.out) {
      w.n += (size_t)p.second.count * 4 * m->esz;
      continue;
    }
    // (a float64 model's elements are 8 bytes, little endian like everything else — serialize.nim:35 — i.e. two of the
    //  32-bit words this loop writes, low word first)
    host.resize((size_t)p.second.count * m->esz);
    if (p.second.count > 0) {
      int rc = eg::copy_d2h(m->ctx, host.data(), p.second.ptr, host.size() * sizeof(float));  // D2H on the context's stream + sync
      if (rc) return rc;
    }
    for (float f : host) {
      uint32_t bits;
      memcpy(&bits, &f, 4);
      w.u32(bits);
    }
  }
  return EG_OK;
}

int load_table(eg_model* m, TK kind, Reader& r, const char* what) {
  const int64_t count = r.i64();
  EG_REQUIRE(r.ok && count >= 0, EG_ERR_INVALID, "model state: truncated %s table", what);
  std::vector<float> host;
  std::set<int> seen;  // every tensor of this kind exactly once: a table that leaves some out would load as a partly
                       // random (mt19937-initialised) or partly zero model; the reference always writes the full tables
  for (int64_t e = 0; e < count; ++e) {
    const int64_t tid = r.i64();
    const bool is_nil = r.byte() != 0;
    EG_REQUIRE(r.ok, EG_ERR_INVALID, "model state: truncated %s table", what);
    EG_REQUIRE(seen.insert((int)tid).second, EG_ERR_INVALID, "model state: tensor %ld appears twice in the %s table", (long)tid, what);
    auto it = m->params.find((int)tid);
    EG_REQUIRE(tid >= 1 && it != m->params.end() && m->prog.tensors[(size_t)tid].kind == kind, EG_ERR_INVALID,
               "model state: tensor %ld is not one of the model's %s", (long)tid, what);
    EG_REQUIRE(!is_nil, EG_ERR_INVALID, "model state: tensor %ld of the %s table is nil", (long)tid, what);
    const int64_t rank = r.i64();
    EG_REQUIRE(r.ok, EG_ERR_INVALID, "model state: truncated %s table", what);
    EG_REQUIRE(rank >= 0 && rank <= 64, EG_ERR_INVALID, "model state: bad rank for tensor %ld", (long)tid);
    std::vector<long> shape;
    for (int64_t d = 0; d < rank; ++d) shape.push_back((long)r.i64());
    EG_REQUIRE(r.ok, EG_ERR_INVALID, "model state: truncated shape of tensor %ld", (long)tid);
    EG_REQUIRE(shape == it->second.shape, EG_ERR_SHAPE, "model state: tensor %ld has another shape than the model's", (long)tid);
    const long n = it->second.count * m->esz;  // 32-bit words
    EG_REQUIRE(r.pos + (size_t)n * 4 <= r.n, EG_ERR_INVALID, "model state: truncated data of tensor %ld", (long)tid);
    host.resize((size_t)n);
    for (long i = 0; i < n; ++i) {
      const uint32_t bits = r.u32();
      memcpy(&host[(size_t)i], &bits, 4);
    }
    if (n > 0) {
      int rc = eg::copy_h2d(m->ctx, it->second.ptr, host.data(), (size_t)n * sizeof(float));
      if (rc) return rc;
    }
  }
  for (auto& p : m->params)
    if (m->prog.tensors[(size_t)p.first].kind == kind)
      EG_REQUIRE(seen.count(p.first), EG_ERR_INVALID, "model state: the %s table lacks tensor %d of the model", what, p.first);
  return EG_OK;
}

int read_file(const char* path, std::vector<unsigned char>& data) {
  FILE* f = fopen(path, "rb");
  EG_REQUIRE(f, EG_ERR_RUNTIME, "cannot open %s for reading", path);
  fseek(f, 0, SEEK_END);
  const long size = ftell(f);
  fseek(f, 0, SEEK_SET);
  data.resize(size > 0 ? (size_t)size : 0);
  const size_t got = data.empty() ? 0 : fread(data.data(), 1, data.size(), f);
  fclose(f);
  EG_REQUIRE(got == data.size(), EG_ERR_RUNTIME, "short read from %s", path);
  return EG_OK;
}

}  // namespace

extern "C" {

int eg_model_state_bytes(eg_model* m, size_t* bytes) try {
  EG_REQUIRE(m && bytes, EG_ERR_INVALID, "eg_model_state_bytes: NULL argument");
  Writer w;
  int rc = store_table(m, TK::Param, w);
  if (rc) return rc;
  rc = store_table(m, TK::Cache, w);
  if (rc) return rc;
  *bytes = w.n;
  return EG_OK;
}
EG_CATCH_ALL

int eg_model_store_state(eg_model* m, void* buf, size_t cap, size_t* written) try {
  EG_REQUIRE(m && (buf || cap == 0), EG_ERR_INVALID, "eg_model_store_state: NULL argument");
  std::vector<unsigned char> out;
  Writer w;
  w.out = &out;
  int rc = store_table(m, TK::Param, w);
  if (rc) return rc;
  rc = store_table(m, TK::Cache, w);
  if (rc) return rc;
  if (written) *written = out.size();
  EG_REQUIRE(out.size() <= cap, EG_ERR_SIZE, "eg_model_store_state: the state needs %zu bytes, the buffer has %zu", out.size(), cap);
  if (!out.empty()) memcpy(buf, out.data(), out.size());
  return EG_OK;
}
EG_CATCH_ALL

int eg_model_load_state(eg_model* m, const void* buf, size_t bytes, size_t* consumed) try {
  EG_REQUIRE(m && (buf || bytes == 0), EG_ERR_INVALID, "eg_model_load_state: NULL argument");
  Reader r;
  r.p = static_cast<const unsigned char*>(buf);
  r.n = bytes;
  int rc = load_table(m, TK::Param, r, "parameters");
  if (rc) return rc;
  rc = load_table(m, TK::Cache, r, "caches");
  if (rc) return rc;
  if (consumed) *consumed = r.pos;
  return EG_OK;
}
EG_CATCH_ALL

int eg_model_save(eg_model* m, const char* path) try {
  EG_REQUIRE(m && path, EG_ERR_INVALID, "eg_model_save: NULL argument");
  std::vector<unsigned char> out;
  Writer w;
  w.out = &out;
  w.byte(0);                // model.isNil = false   serialize.nim:345
  w.str(m->source_text);    // the program field: kernel-description text
  int rc = store_table(m, TK::Param, w);
  if (rc) return rc;
  rc = store_table(m, TK::Cache, w);
  if (rc) return rc;
  w.i64(m->epoch);          // extension behind everything the reference writes
  FILE* f = fopen(path, "wb");
  EG_REQUIRE(f, EG_ERR_RUNTIME, "cannot open %s for writing", path);
  const size_t put = fwrite(out.data(), 1, out.size(), f);
  const int closed = fclose(f);
  EG_REQUIRE(put == out.size() && closed == 0, EG_ERR_RUNTIME, "short write to %s", path);
  return EG_OK;
}
EG_CATCH_ALL

int eg_model_load(eg_ctx* ctx, const char* path, eg_model** out) try {
  EG_REQUIRE(ctx && path && out, EG_ERR_INVALID, "eg_model_load: NULL argument");
  std::vector<unsigned char> data;
  int rc = read_file(path, data);
  if (rc) return rc;
  Reader r;
  r.p = data.data();
  r.n = data.size();
  const bool is_nil = r.byte() != 0;
  EG_REQUIRE(r.ok && !is_nil, EG_ERR_INVALID, "%s holds a nil model", path);  // serialize.nim:353-355
  const int64_t len = r.i64();
  EG_REQUIRE(r.ok && len >= 0 && r.pos + (size_t)len <= r.n, EG_ERR_INVALID, "%s: truncated program", path);
  const std::string text(reinterpret_cast<const char*>(r.p + r.pos), (size_t)len);
  r.pos += (size_t)len;
  EG_REQUIRE(text.compare(0, 3, "kd ") == 0, EG_ERR_INVALID,
             "%s: the program field is not kernel-description text (a file written by the reference holds a Nim "
             "Program there; load it on the Nim side and pass the rest to eg_model_load_state)", path);
  eg_model* m = nullptr;
  rc = eg_model_compile(ctx, text.c_str(), &m);
  if (rc) return rc;
  size_t used = 0;
  rc = eg_model_load_state(m, r.p + r.pos, r.n - r.pos, &used);
  if (rc) {
    const std::string msg = eg_last_error();
    eg_model_free(m);
    set_error("%s", msg.c_str());
    return rc;
  }
  r.pos += used;
  if (r.pos + 8 <= r.n) m->epoch = r.i64();
  *out = m;
  return EG_OK;
}
EG_CATCH_ALL

const char* eg_model_source_text(eg_model* m) { return m ? m->source_text.c_str() : ""; }

}  // extern "C"
