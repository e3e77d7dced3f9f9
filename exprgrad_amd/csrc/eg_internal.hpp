// Internal declarations shared by the translation units of libexprgrad_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <map>
#include <string>
#include <utility>
#include <vector>

#include "exprgrad_hip.h"
#include "switches.hpp"

namespace eg {

// Thread-local error text behind eg_last_error().
void set_error(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
void clear_error();

#define EG_HIP_CHECK(expr)                                                             \
  do {                                                                                 \
    hipError_t _e = (expr);                                                            \
    if (_e != hipSuccess) {                                                            \
      ::eg::set_error("%s failed: %s (%d) at %s:%d", #expr, hipGetErrorString(_e),     \
                      (int)_e, __FILE__, __LINE__);                                    \
      return EG_ERR_HIP;                                                               \
    }                                                                                  \
  } while (0)

#define EG_REQUIRE(cond, code, ...)   \
  do {                                \
    if (!(cond)) {                    \
      ::eg::set_error(__VA_ARGS__);   \
      return (code);                  \
    }                                 \
  } while (0)

}  // namespace eg

namespace eg {
struct HostStager;  // host_copy.cpp: pinned staging buffers + copy threads for large pageable host arrays
void host_stager_free(HostStager* s);
}  // namespace eg

struct eg_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool owns_stream = false;
  // Scratch for split-K partials and two-stage reductions; grown on demand, never shrunk.
  void* workspace = nullptr;
  size_t workspace_bytes = 0;
  // Second scratch block for operands a library call prepares for an inner call that may itself
  // grow the workspace (padded output gradient + flipped filters of the convolution's image gradient).
  void* aux = nullptr;
  size_t aux_bytes = 0;
  // Side lane (host/model.cpp, plan_overlap): a second stream with scratch blocks of its own for
  // the bandwidth-bound launches that run next to a long contraction.  Swapped in around those
  // launches, so the library calls they make need not know.
  hipStream_t side_stream = nullptr;
  void* side_workspace = nullptr;
  size_t side_workspace_bytes = 0;
  void* side_aux = nullptr;
  size_t side_aux_bytes = 0;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  bool on_side_lane = false;  // true while the side lane's stream / scratch are swapped in (LaneSwap)
  std::vector<hipEvent_t> pipe_events;  // batch pipeline (host/plan_pipeline.cpp): per-stage, per-half dependencies between the lanes
  eg::HostStager* stager = nullptr;  // created by the first large host copy
  float* ones = nullptr;             // {1,1,1,1, 1,0,0,0}: source of a contraction's virtual row of ones (GemmArgs::ones)
  int compute_units = 256;
  std::string arch;
  // kernels a library call specialises at run time (hiprtc) and keeps: by name
  std::map<std::string, eg_kernel*> jit;
};

struct eg_kernel;

struct eg_buf {
  eg_ctx* ctx = nullptr;
  void* ptr = nullptr;
  size_t bytes = 0;
  bool owned = true;
};

namespace eg {
// Ensure ctx->workspace holds at least `bytes`; synchronises the stream if it has to grow.
int ensure_workspace(eg_ctx* ctx, size_t bytes);
void* kernel_function(eg_kernel* kernel);  // what a captured kernel node of it carries as `func`
int ensure_aux(eg_ctx* ctx, size_t bytes);
// Blocking copies between a host array and device memory, ordered on ctx->stream (host_copy.cpp): large
// pageable arrays go through pinned staging buffers filled by several threads, the rest is a plain copy.
int copy_h2d(eg_ctx* ctx, void* device, const void* host, size_t bytes);
int copy_d2h(eg_ctx* ctx, void* host, const void* device, size_t bytes);
// EG_POISON=1: scratch and to-be-overwritten result storage is filled with NaN patterns before use.
bool poison_enabled();
// Zero several float ranges with one launch per ZeroRanges::MAX ranges (kernels/elementwise.hip).
struct ZeroRanges {
  static constexpr int MAX = 8;
  float* ptr[MAX];
  long floats[MAX];
  int blocks[MAX];
  int count;
};
int zero_ranges(eg_ctx* ctx, const std::vector<std::pair<float*, long>>& ranges);
namespace rtc {
// Source text -> code object through the library's own hiprtc (rtc.cpp), cached on disk.
int compile(const char* label, const char* source, const std::string& arch, std::vector<char>& code);
std::string compiler_info();
}  // namespace rtc

// Build several kernels (extern "C" names) from one source text in a single hiprtc program.
int kernels_compile_batch(eg_ctx* ctx, const char* label, const char* source, const std::vector<std::string>& names,
                          std::vector<eg_kernel*>& out);
// Launch a hiprtc-built kernel with an explicit argument array (bypasses the sticky arguments).
int kernel_launch_raw(eg_kernel* kernel, unsigned gx, unsigned gy, unsigned gz, unsigned block, void** args);
// Column sum with caller-provided scratch (colsum_scratch_floats(...) floats); see reduce.hip.
long colsum_scratch_floats(const eg_ctx* ctx, long rows, long cols);
// One-launch sum of k-slice slabs into a small output (reduce.hip); fixed order.
bool slab_sum_supported(long total, const float* slab, const float* out);
int slab_sum(eg_ctx* ctx, long slabs, long total, const float* slab, float* out, int accumulate);
int colsum_with_scratch(eg_ctx* ctx, long rows, long cols, const float* in, float* out, int accumulate,
                        float* scratch);
// float64 twin (kernels/gemm_f64_mfma.hip): same geometry, scratch of colsum_scratch_floats(...) DOUBLES
int colsum_f64_with_scratch(eg_ctx* ctx, long rows, long cols, const double* in, double* out, int accumulate, double* scratch);
// Second stage of a row-fused kernel's batch reductions (host/rowfuse.hpp): partial is
// [nblocks][E]; column e belongs to segment s with offset[s] <= e < offset[s+1] and is summed over
// the blocks (fixed tree order) into dst[s][e - offset[s]].
struct RowFinalizeArgs {
  float* dst[16];
  int offset[17];
  int accumulate[16];
  int nseg;
};
int row_finalize(eg_ctx* ctx, const float* partial, int nblocks, int E, int stride, const RowFinalizeArgs& args);
// Direct per-pixel kernels for few input channels (kernels/conv2_direct.cpp), same convention.
int conv2_direct_try(eg_ctx* ctx, long N, long H, long W, long C, long F, long FH, long FW, const float* img,
                     const float* flt, float* out, int accumulate, bool* launched);
// kernels/conv2_band.cpp: convolutions with at most 16 channels and 16 filters on 16 x 16 x 4 matrix instructions, float32 and float64
int conv2_band_forward_try(eg_ctx* ctx, bool f64, long N, long H, long W, long C, long F, long FH, long FW, const void* img, const void* flt, void* out,
                           int accumulate, bool* launched);
int conv2_band_grad_image_try(eg_ctx* ctx, bool f64, long N, long H, long W, long C, long F, long FH, long FW, const void* flt, const void* gout, void* gimg,
                              int accumulate, bool* launched);
int conv2_band_grad_filter_try(eg_ctx* ctx, bool f64, long N, long H, long W, long C, long F, long FH, long FW, const void* img, const void* gout, void* gflt,
                               int accumulate, bool* launched);
int conv2_direct_f64_try(eg_ctx* ctx, long N, long H, long W, long C, long F, long FH, long FW, const double* img, const double* flt,
                         double* out, int accumulate, bool* launched);
int conv2_direct_grad_filter_try(eg_ctx* ctx, long N, long H, long W, long C, long F, long FH, long FW, const float* img,
                                 const float* gout, float* gflt, int accumulate, bool* launched);
// LDS-halo convolution (kernels/conv2_halo.hip); *launched = false when the problem does not suit it.
// kernels/conv2_gradf_halo.hip: the filter gradient of a 3 x 3 convolution with the halo in LDS (C, F multiples of 32)
int conv2_gradf_halo_try(eg_ctx* ctx, long N, long H, long W, long C, long F, long FH, long FW, const float* img, const float* gout,
                         float* gflt, int accumulate, bool* launched);
// kernels/gemm_f32_mfma.hip: contractions small enough for one wave per output element; two independent ones in one launch
struct SmallGemm {
  const float *A, *B;
  float* C;
  const float* bias;
  long M, N, K, a_sm, a_sk, b_sk, b_sn, ldc;
  int accumulate;
};
bool gemm_small_suits(long M, long N, long K);
SmallGemm small_gemm(int trans_a, int trans_b, long M, long N, long K, const float* A, long lda, const float* B, long ldb, float* C,
                     long ldc, int accumulate, const float* bias);
int gemm_small_pair(eg_ctx* ctx, const SmallGemm& g0, const SmallGemm& g1);
// kernels/conv2_tiny.hip: conv2 and its gradients for problems of a few million multiply-adds (*launched says whether it ran)
int conv2_tiny_forward_try(eg_ctx* ctx, long N, long H, long W, long C, long F, long FH, long FW, const float* img,
                           const float* flt, float* out, int accumulate, bool* launched);
int conv2_tiny_grad_image_try(eg_ctx* ctx, long N, long H, long W, long C, long F, long FH, long FW, const float* flt,
                              const float* gout, float* gimg, int accumulate, bool* launched);
int conv2_tiny_grad_filter_try(eg_ctx* ctx, long N, long H, long W, long C, long F, long FH, long FW, const float* img,
                               const float* gout, float* gflt, int accumulate, bool* launched);
bool conv2_halo_suits(eg_ctx* ctx, long N, long H, long W, long C, long F, long FH, long FW, long py, long px, const float* img,
                      bool flt_aligned);
int conv2_halo_try_padded(eg_ctx* ctx, long N, long H, long W, long C, long F, long FH, long FW, long py, long px,
                          const float* img, const float* flt, float* out, int accumulate, bool* launched);
int conv2_halo_try(eg_ctx* ctx, long N, long H, long W, long C, long F, long FH, long FW, const float* img,
                   const float* flt, float* out, int accumulate, bool* launched);
// Several contiguous device-to-device float copies in one launch (kernels/elementwise.hip).
struct CopySegments {
  const float* src[8];
  float* dst[8];
  long count[8];
  int n;
};
int copy_segments(eg_ctx* ctx, const CopySegments& seg);
bool copy_segments_node_params(eg_ctx* ctx, const CopySegments& seg, hipKernelNodeParams* out, void** arg);
const void* copy_segments_function();
// Data-parallel step (dp_rccl.cpp -> host/model_api.cpp): run the backward range of `target` and
// exchange the gradient bucket through `allreduce(user, device pointer, float count)`, which must
// enqueue an in-place SUM on the context's CURRENT stream.  Gradients that are complete before the
// last long contraction of the backward pass are exchanged on the side lane, under that contraction
// (host/run.cpp plan_exchange); the rest after it.  *pieces (optional) = all-reduce calls issued.
struct GradExchange {
  int (*allreduce)(void* user, float* buf, long count) = nullptr;
  // Do all ranks hold the same `n` numbers?  (*same = 1 / 0; one small collective + a host read, once per plan.)
  // Ranks that would cut the bucket differently — unequal shards through the C ABI, an environment toggle on one rank —
  // would issue different sequences of all-reduce calls on one communicator: a hang or silently mixed gradients.
  // NULL: not checked (a one-rank group).
  int (*agree)(void* user, const int64_t* values, int n, int* same) = nullptr;
  void* user = nullptr;
  // identity of the group, the same number on every rank (dp::group_identity of the communicator's unique id): the key
  // of the target's exchange schedule.  0 = none (the schedule is keyed by `user`; only sound for a one-rank group)
  uint64_t group = 0;
  bool split = true;    // allow the early / late split of the bucket (EG_DP_NO_SPLIT, eg_dp_set_split)
  int reserve_cus = 0;  // compute units the last long contraction leaves free for the collective's kernel
};
int model_backward_with_exchange(eg_model* model, const char* target, const GradExchange& gx, int* pieces);
eg_ctx* model_context(eg_model* model);
inline int set_device(eg_ctx* ctx) {
  EG_HIP_CHECK(hipSetDevice(ctx->device));
  return EG_OK;
}
}  // namespace eg
