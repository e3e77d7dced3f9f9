// Data-parallel exchange of the C ABI: one process per GPU, one RCCL
// communicator per context, one in-place SUM all-reduce of the flat parameter-gradient bucket on
// the context's stream between the backward kernels and the optimizer kernels.
//
// The reference has no multi-device code; this is what a Nim host binds for the north-star's
// "RCCL all-reduce of parameter gradients before the gradientDescent step" (the Python mirror
// uses torch.distributed's "nccl" backend, which is the same RCCL).  librccl is opened lazily
// with dlopen — a process that never calls eg_dp_* never loads it, and a process that already
// has one (torch ships its own) reuses that copy instead of loading a second one.
#include <dlfcn.h>

#include <chrono>
#include <condition_variable>
#include <cstring>
#include <memory>
#include <mutex>
#include <thread>

#include "eg_internal.hpp"
#include "host/dp_schedule.hpp"

namespace {

// The slice of rccl.h this file needs (rccl/rccl.h: ncclUniqueId is 128 opaque bytes, passed by
// value; ncclFloat32 = 7; ncclSum = 0).
struct UniqueId {
  char internal[128];
};
using Comm = void*;
struct Rccl {
  int (*GetUniqueId)(UniqueId*) = nullptr;
  int (*CommInitRank)(Comm*, int, UniqueId, int) = nullptr;
  int (*CommDestroy)(Comm) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  int (*CommCount)(Comm, int*) = nullptr;     // optional: what RCCL itself says the group is
  int (*CommUserRank)(Comm, int*) = nullptr;
  bool ok = false;
  std::string why;
};

const Rccl& rccl() {
  static const Rccl lib = [] {
    Rccl r;
    void* h = nullptr;
    const char* names[] = {"librccl.so", "librccl.so.1"};
    for (const char* n : names)  // a copy that is already in the process wins
      if (!h) h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
    const char* paths[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* p : paths)
      if (!h) h = dlopen(p, RTLD_NOW | RTLD_GLOBAL);
    if (!h) {
      r.why = std::string("librccl.so not found: ") + (dlerror() ? dlerror() : "");
      return r;
    }
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(h, "ncclAllReduce"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    r.CommCount = reinterpret_cast<decltype(r.CommCount)>(dlsym(h, "ncclCommCount"));
    r.CommUserRank = reinterpret_cast<decltype(r.CommUserRank)>(dlsym(h, "ncclCommUserRank"));
    r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllReduce && r.GetErrorString;
    if (!r.ok) r.why = "librccl.so lacks an expected symbol";
    return r;
  }();
  return lib;
}

#define EG_RCCL_CHECK(expr)                                                                         \
  do {                                                                                              \
    int _r = (expr);                                                                                \
    if (_r != 0) {                                                                                  \
      ::eg::set_error("%s failed: %s (%d)", #expr, rccl().GetErrorString(_r), _r);                  \
      return EG_ERR_HIP;                                                                            \
    }                                                                                               \
  } while (0)

}  // namespace

struct eg_dp {
  eg_ctx* ctx = nullptr;
  Comm comm = nullptr;
  int rank = 0, world = 1;
  int last_pieces = 0;  // all-reduce calls of the last eg_model_step_dp (2+: early gradients went under the contraction)
  bool split = true;    // early / late split of the bucket allowed (eg_dp_set_split; EG_DP_NO_SPLIT=1 starts with false)
  int reserve_cus = 8;  // compute units the last long contraction leaves to the early collective (EG_DP_RESERVE_CUS)
  int64_t* agree_buf = nullptr;  // device scratch of the cross-rank comparison: 2 x 32 int64
  uint64_t identity = 0;         // hash of the communicator's unique id: the same on every rank, new for every group
};

namespace {
// ncclCommInitRank blocks until every rank has joined.  A rank that never arrives (crashed, took another code path)
// would leave the others inside it for ever; the call runs on a helper thread and is given up after
// EG_DP_INIT_TIMEOUT_S seconds (default 180) with an error the host can act on (bench.py: every rank falls back to the
// torch.distributed exchange).  The helper thread of a call that was given up stays blocked and is abandoned.
struct InitState {
  std::mutex mu;
  std::condition_variable cv;
  bool done = false;
  int status = 0;
  Comm comm = nullptr;
};

int init_with_watchdog(int device, int world, const UniqueId& id, int rank, Comm* out) {
  double timeout_s = 180;
  if (const char* e = eg::sw::raw("EG_DP_INIT_TIMEOUT_S")) timeout_s = atof(e);
  auto st = std::make_shared<InitState>();
  std::thread([st, device, world, id, rank] {
    hipSetDevice(device);
    Comm c = nullptr;
    const int r = rccl().CommInitRank(&c, world, id, rank);
    std::lock_guard<std::mutex> lock(st->mu);
    st->status = r;
    st->comm = c;
    st->done = true;
    st->cv.notify_all();
  }).detach();
  std::unique_lock<std::mutex> lock(st->mu);
  if (!st->cv.wait_for(lock, std::chrono::duration<double>(timeout_s > 0 ? timeout_s : 1e9), [&] { return st->done; })) {
    ::eg::set_error("ncclCommInitRank (rank %d of %d) did not return within %.0f s: another rank never joined", rank, world,
                    timeout_s);
    return EG_ERR_RUNTIME;
  }
  if (st->status != 0) {
    ::eg::set_error("ncclCommInitRank failed: %s (%d)", rccl().GetErrorString(st->status), st->status);
    return EG_ERR_HIP;
  }
  *out = st->comm;
  return EG_OK;
}

bool env_on(const char* name) {
  const char* e = eg::sw::raw(name);
  return e && e[0] && e[0] != '0';
}
}  // namespace

extern "C" {

int eg_dp_unique_id(void* id128) {
  EG_REQUIRE(id128, EG_ERR_INVALID, "eg_dp_unique_id: NULL buffer");
  EG_REQUIRE(rccl().ok, EG_ERR_RUNTIME, "%s", rccl().why.c_str());
  UniqueId id;
  EG_RCCL_CHECK(rccl().GetUniqueId(&id));
  memcpy(id128, &id, sizeof(id));
  return EG_OK;
}

int eg_dp_init(eg_ctx* ctx, const void* id128, int rank, int world, eg_dp** out) {
  EG_REQUIRE(ctx && id128 && out, EG_ERR_INVALID, "eg_dp_init: NULL argument");
  EG_REQUIRE(world >= 1 && rank >= 0 && rank < world, EG_ERR_INVALID, "eg_dp_init: rank %d of %d", rank, world);
  EG_REQUIRE(rccl().ok, EG_ERR_RUNTIME, "%s", rccl().why.c_str());
  int rc = eg::set_device(ctx);
  if (rc) return rc;
  UniqueId id;
  memcpy(&id, id128, sizeof(id));
  Comm comm = nullptr;
  rc = init_with_watchdog(ctx->device, world, id, rank, &comm);
  if (rc) return rc;
  eg_dp* dp = new eg_dp();
  dp->ctx = ctx;
  dp->comm = comm;
  dp->rank = rank;
  dp->world = world;
  dp->identity = eg::dp::group_identity(id128, sizeof(UniqueId));
  dp->split = !env_on("EG_DP_NO_SPLIT");
  if (const char* e = eg::sw::raw("EG_DP_RESERVE_CUS")) dp->reserve_cus = atoi(e);
  *out = dp;
  return EG_OK;
}

int eg_dp_free(eg_dp* dp) {
  if (!dp) return EG_OK;
  if (dp->comm) {
    hipSetDevice(dp->ctx->device);
    hipStreamSynchronize(dp->ctx->stream);
    rccl().CommDestroy(dp->comm);
  }
  if (dp->agree_buf) hipFree(dp->agree_buf);
  delete dp;
  return EG_OK;
}

int eg_dp_world(const eg_dp* dp) { return dp ? dp->world : 0; }
int eg_dp_rank(const eg_dp* dp) { return dp ? dp->rank : -1; }

// What RCCL itself reports for the communicator (ncclCommCount / ncclCommUserRank): -1 when the library lacks the symbols.
int eg_dp_rccl_count(const eg_dp* dp) {
  int n = -1;
  if (dp && dp->comm && rccl().CommCount && rccl().CommCount(dp->comm, &n) != 0) n = -1;
  return n;
}
int eg_dp_rccl_rank(const eg_dp* dp) {
  int r = -1;
  if (dp && dp->comm && rccl().CommUserRank && rccl().CommUserRank(dp->comm, &r) != 0) r = -1;
  return r;
}

int eg_dp_set_split(eg_dp* dp, int enabled) {
  EG_REQUIRE(dp, EG_ERR_INVALID, "eg_dp_set_split: NULL group");
  dp->split = enabled != 0;
  return EG_OK;
}

int eg_dp_allreduce_sum_f32(eg_dp* dp, float* device_buf, int64_t count) {
  EG_REQUIRE(dp && dp->comm, EG_ERR_INVALID, "eg_dp_allreduce_sum_f32: NULL communicator");
  EG_REQUIRE(count >= 0, EG_ERR_INVALID, "eg_dp_allreduce_sum_f32: negative count");
  if (count == 0) return EG_OK;
  EG_REQUIRE(device_buf, EG_ERR_INVALID, "eg_dp_allreduce_sum_f32: NULL buffer");
  int rc = eg::set_device(dp->ctx);
  if (rc) return rc;
  EG_RCCL_CHECK(rccl().AllReduce(device_buf, device_buf, (size_t)count, /*ncclFloat32*/ 7, /*ncclSum*/ 0, dp->comm,
                                 dp->ctx->stream));
  return EG_OK;
}

int eg_dp_allreduce_sum_f64(eg_dp* dp, double* device_buf, int64_t count) {
  EG_REQUIRE(dp && dp->comm, EG_ERR_INVALID, "eg_dp_allreduce_sum_f64: NULL communicator");
  EG_REQUIRE(count >= 0, EG_ERR_INVALID, "eg_dp_allreduce_sum_f64: negative count");
  if (count == 0) return EG_OK;
  EG_REQUIRE(device_buf, EG_ERR_INVALID, "eg_dp_allreduce_sum_f64: NULL buffer");
  int rc = eg::set_device(dp->ctx);
  if (rc) return rc;
  EG_RCCL_CHECK(rccl().AllReduce(device_buf, device_buf, (size_t)count, /*ncclFloat64*/ 8, /*ncclSum*/ 0, dp->comm,
                                 dp->ctx->stream));
  return EG_OK;
}

// One data-parallel training step on this rank's shard (the inputs are bound already):
//   [forward + backward kernels] | all-reduce of the gradient bucket | [optimizer kernels]
// with the bucket split where the plan allows it: the gradients that are complete before the last
// long contraction of the backward pass (the first layer's weight gradient) are reduced on the side
// lane while it runs, only its own gradient after it (eg::model_backward_with_exchange).
// mean != 0: the loss divides by the batch (mse, crossEntropy: base.nim:57-67), so the seed
// gradient is scaled by B_local / B_global = 1 / world; sum-type losses use 1.
int eg_model_step_dp(eg_model* model, const char* target, eg_dp* dp, int mean) {
  EG_REQUIRE(model && target && dp, EG_ERR_INVALID, "eg_model_step_dp: NULL argument");
  // the collective is ordered against the backward and update kernels by the stream they share
  EG_REQUIRE(eg::model_context(model) == dp->ctx, EG_ERR_INVALID,
             "eg_model_step_dp: the model and the data-parallel group belong to different contexts");
  int rc = eg_model_set_grad_scale(model, mean ? 1.0f / (float)dp->world : 1.0f);
  if (rc) return rc;
  eg::GradExchange gx;
  gx.user = dp;
  gx.group = dp->identity;
  gx.split = dp->split;
  // (EG_DP_TEST_AS_MULTI=1: a one-rank group takes the multi-rank code paths — comparison collective, reserved compute
  // units — so that a one-GPU box exercises them)
  const bool multi = dp->world > 1 || env_on("EG_DP_TEST_AS_MULTI");
  gx.reserve_cus = multi ? dp->reserve_cus : 0;
  gx.allreduce = [](void* user, float* buf, long count) {
    return eg_dp_allreduce_sum_f32(static_cast<eg_dp*>(user), buf, (int64_t)count);
  };
  if (eg_model_scalar_bytes(model) == 8)  // a float64 model's bucket: the segments are counted in 4-byte units (host/model_types.hpp)
    gx.allreduce = [](void* user, float* buf, long count) {
      return eg_dp_allreduce_sum_f64(static_cast<eg_dp*>(user), reinterpret_cast<double*>(buf), (int64_t)(count / 2));
    };
  if (multi)
    gx.agree = [](void* user, const int64_t* values, int n, int* same) {
      // MAX over the ranks of [v, -v]: all ranks hold the same v  <=>  max(v) == -max(-v) element by element
      eg_dp* d = static_cast<eg_dp*>(user);
      EG_REQUIRE(n >= 1 && n <= 32, EG_ERR_INVALID, "eg_dp: comparison of %d values", n);
      if (!d->agree_buf) EG_HIP_CHECK(hipMalloc((void**)&d->agree_buf, 64 * sizeof(int64_t)));
      int64_t host[64];
      for (int i = 0; i < n; ++i) {
        host[i] = values[i];
        host[n + i] = -values[i];
      }
      hipStream_t s = d->ctx->stream;
      EG_HIP_CHECK(hipMemcpyAsync(d->agree_buf, host, 2 * n * sizeof(int64_t), hipMemcpyHostToDevice, s));
      EG_RCCL_CHECK(rccl().AllReduce(d->agree_buf, d->agree_buf, (size_t)(2 * n), /*ncclInt64*/ 4, /*ncclMax*/ 2, d->comm, s));
      EG_HIP_CHECK(hipMemcpyAsync(host, d->agree_buf, 2 * n * sizeof(int64_t), hipMemcpyDeviceToHost, s));
      EG_HIP_CHECK(hipStreamSynchronize(s));
      *same = 1;
      for (int i = 0; i < n; ++i)
        if (host[i] != -host[n + i]) *same = 0;
      return (int)EG_OK;
    };
  rc = eg::model_backward_with_exchange(model, target, gx, &dp->last_pieces);
  if (rc) return rc;
  return eg_model_run_update(model, target);
}

int eg_dp_last_pieces(const eg_dp* dp) { return dp ? dp->last_pieces : 0; }

}  // extern "C"
