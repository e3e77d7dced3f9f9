// Blocking device -> host copies into large PAGEABLE host arrays, and pinned host memory for callers.
//
// The reference uploads every input with a blocking write and downloads every output with a blocking
// read on each call (model.nim:364-376 -> cl.nim:111-131); its matmul benchmark times exactly that
// (benchmarks/matmul/matmul_gpu.nim:35-46: 128 MiB up, 64 MiB down around a 4096^3 product).  A host
// array from a Nim seq or numpy is pageable: hipMemcpy stages it through one pinned bounce buffer
// with one thread (~11 GB/s measured: the 4096^3 call took 16.6 ms for a 1 ms kernel).  Here:
//   * a few pinned staging buffers per context (hipHostMalloc, 8 MiB each), used round robin;
//   * a small pool of worker threads copies pageable <-> pinned in parallel slices (one core moves
//     ~12 GB/s, PCIe 5 x16 wants ~55);
//   * the DMA of chunk i overlaps the host copy of chunk i - 1;
//   * everything is enqueued on the context's stream, so ordering against kernels stays the
//     in-order-queue contract of cl.nim:92, and the call returns when the data has arrived.
// Copies below 4 MiB take the plain hipMemcpyAsync + synchronize path.
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "eg_internal.hpp"

namespace eg {

namespace {

constexpr size_t CHUNK = 8u << 20;
constexpr int NBUF = 4;
constexpr size_t THRESHOLD = 4u << 20;

// Fork-join pool: run(n, fn) executes fn(0..n-1) on the workers and the calling thread.
class Pool {
 public:
  explicit Pool(int workers) {
    for (int i = 0; i < workers; ++i) threads_.emplace_back([this] { loop(); });
  }
  ~Pool() {
    {
      std::lock_guard<std::mutex> lock(mu_);
      stop_ = true;
    }
    cv_.notify_all();
    for (auto& t : threads_) t.join();
  }
  template <class F>
  void run(int n, F fn) {
    {
      std::lock_guard<std::mutex> lock(mu_);
      fn_ = [&fn](int i) { fn(i); };
      next_ = 0;
      total_ = n;
      pending_ = n;
      ++generation_;
    }
    cv_.notify_all();
    work();  // the caller takes slices too
    std::unique_lock<std::mutex> lock(mu_);
    done_.wait(lock, [this] { return pending_ == 0; });
    fn_ = nullptr;
  }
  int size() const { return (int)threads_.size() + 1; }

 private:
  void work() {
    for (;;) {
      int i;
      {
        std::lock_guard<std::mutex> lock(mu_);
        if (next_ >= total_) return;
        i = next_++;
      }
      fn_(i);
      std::lock_guard<std::mutex> lock(mu_);
      if (--pending_ == 0) done_.notify_all();
    }
  }
  void loop() {
    unsigned long seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> lock(mu_);
        cv_.wait(lock, [&] { return stop_ || generation_ != seen; });
        if (stop_) return;
        seen = generation_;
      }
      work();
    }
  }
  std::vector<std::thread> threads_;
  std::mutex mu_;
  std::condition_variable cv_, done_;
  std::function<void(int)> fn_;
  int next_ = 0, total_ = 0, pending_ = 0;
  unsigned long generation_ = 0;
  bool stop_ = false;
};

}  // namespace

struct HostStager {
  void* buf[NBUF] = {};
  hipEvent_t ev[NBUF] = {};
  std::unique_ptr<Pool> pool;
  std::mutex mu;  // one staged copy at a time per context
  int device = 0;
  ~HostStager() {
    pool.reset();
    hipSetDevice(device);
    for (int i = 0; i < NBUF; ++i) {
      if (ev[i]) hipEventDestroy(ev[i]);
      if (buf[i]) hipHostFree(buf[i]);
    }
  }
};

void host_stager_free(HostStager* s) { delete s; }

static int stager(eg_ctx* ctx, HostStager** out) {
  if (!ctx->stager) {
    std::unique_ptr<HostStager> s(new HostStager());
    s->device = ctx->device;
    for (int i = 0; i < NBUF; ++i) {
      EG_HIP_CHECK(hipHostMalloc(&s->buf[i], CHUNK, hipHostMallocDefault));
      EG_HIP_CHECK(hipEventCreateWithFlags(&s->ev[i], hipEventDisableTiming));
    }
    int threads = 6;
    const int hw = (int)std::thread::hardware_concurrency();
    if (hw > 0 && threads > hw) threads = hw;
    if (threads < 1) threads = 1;
    s->pool.reset(new Pool(threads - 1));
    ctx->stager = s.release();
  }
  *out = ctx->stager;
  return EG_OK;
}

static void parallel_copy(Pool& pool, void* dst, const void* src, size_t bytes) {
  const int parts = std::max(1, std::min(pool.size(), (int)(bytes >> 20)));  // at least 1 MiB per slice
  const size_t per = ((bytes + parts - 1) / parts + 63) & ~(size_t)63;
  pool.run(parts, [&](int i) {
    const size_t off = (size_t)i * per;
    if (off < bytes) memcpy(static_cast<char*>(dst) + off, static_cast<const char*>(src) + off, std::min(per, bytes - off));
  });
}

static bool staging_enabled() {
  static const bool on = [] {
    const char* e = eg::sw::raw("EG_NO_STAGED_COPY");
    return !(e && e[0] && e[0] != '0');
  }();
  return on;
}

static bool is_pageable(const void* host) {
  hipPointerAttribute_t attr;
  if (hipPointerGetAttributes(&attr, host) != hipSuccess) {
    (void)hipGetLastError();  // an ordinary malloc'ed pointer is "invalid value" to the runtime
    return true;
  }
  return attr.type == hipMemoryTypeUnregistered;
}

// Blocking upload: returns when `bytes` of `host` have arrived at `device` (ordered on ctx->stream).
// Measured (4096^2 floats, pageable numpy arrays that have been touched): the runtime's own pageable
// path reaches 56 GB/s, the staged path above 46 — uploads therefore go straight to hipMemcpyAsync.
// What made the reference-style call slow was the DOWNLOAD into a freshly allocated result array
// (page faults inside the copy): see copy_d2h and the pinned result arrays of eg_host_alloc.
int copy_h2d(eg_ctx* ctx, void* device, const void* host, size_t bytes) {
  if (bytes == 0) return EG_OK;
  EG_HIP_CHECK(hipSetDevice(ctx->device));
  EG_HIP_CHECK(hipMemcpyAsync(device, host, bytes, hipMemcpyHostToDevice, ctx->stream));
  EG_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  return EG_OK;
}

// Blocking download: returns when `bytes` at `device` (as of everything queued on ctx->stream) are in `host`.
int copy_d2h(eg_ctx* ctx, void* host, const void* device, size_t bytes) {
  if (bytes == 0) return EG_OK;
  EG_HIP_CHECK(hipSetDevice(ctx->device));
  if (bytes < THRESHOLD || !staging_enabled() || !is_pageable(host)) {
    EG_HIP_CHECK(hipMemcpyAsync(host, device, bytes, hipMemcpyDeviceToHost, ctx->stream));
    EG_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return EG_OK;
  }
  HostStager* s;
  int rc = stager(ctx, &s);
  if (rc) return rc;
  std::lock_guard<std::mutex> lock(s->mu);
  const int chunks = (int)((bytes + CHUNK - 1) / CHUNK);
  auto issue = [&](int c) {
    const size_t off = (size_t)c * CHUNK, len = std::min(CHUNK, bytes - off);
    EG_HIP_CHECK(hipMemcpyAsync(s->buf[c % NBUF], static_cast<const char*>(device) + off, len, hipMemcpyDeviceToHost, ctx->stream));
    EG_HIP_CHECK(hipEventRecord(s->ev[c % NBUF], ctx->stream));
    return (int)EG_OK;
  };
  // keep NBUF - 1 DMAs in flight ahead of the host copy
  int issued = 0;
  for (; issued < chunks && issued < NBUF - 1; ++issued) {
    rc = issue(issued);
    if (rc) return rc;
  }
  for (int c = 0; c < chunks; ++c) {
    if (issued < chunks) {  // buffer (issued % NBUF) was drained by the host copy of chunk issued - NBUF (< c)
      rc = issue(issued++);
      if (rc) return rc;
    }
    const size_t off = (size_t)c * CHUNK, len = std::min(CHUNK, bytes - off);
    EG_HIP_CHECK(hipEventSynchronize(s->ev[c % NBUF]));
    parallel_copy(*s->pool, static_cast<char*>(host) + off, s->buf[c % NBUF], len);
  }
  return EG_OK;
}

}  // namespace eg

// ---- pinned host memory for callers (group 1): result tensors allocated here are written by DMA at
// PCIe speed and, recycled by the host, never page-fault again (the Python mirror keeps a pool).
extern "C" int eg_host_alloc(size_t bytes, void** out) {
  EG_REQUIRE(out, EG_ERR_INVALID, "eg_host_alloc: out is NULL");
  *out = nullptr;
  if (bytes == 0) return EG_OK;
  EG_HIP_CHECK(hipHostMalloc(out, bytes, hipHostMallocDefault));
  return EG_OK;
}

extern "C" int eg_host_free(void* p) {
  if (!p) return EG_OK;
  EG_HIP_CHECK(hipHostFree(p));
  return EG_OK;
}
