// Contraction with a generated epilogue: the host plans the launch exactly as eg_sgemm would, the
// kernel is gemm_block<..., Epi> from gemm_f32_mfma.hpp compiled at run time (hiprtc) together with
// the `Epi` functor that host/epilogue.cpp generates from the fused consumer kernel.
//
// Role in the reference: none on its GPU target — every lowered kernel is its own launch
// (llvmgen.nim:455-500); on the CPU target fuseLoops (passes.nim:1929-2004) merges `dense` with
// the following activation into one loop nest.  Here the activation (and, in the backward pass,
// the activation gradient) runs on the accumulator registers of the matrix kernel, so the
// intermediate tensor is written at most once and never re-read.
#pragma once
#include <string>

#include "../eg_internal.hpp"

namespace eg {
namespace gemm {

struct FusedLaunch {
  int bm = 0, bn = 0, bk = 0, wm = 0, wn = 0, minb = 0, nt = 0;
  bool a_kc = false, b_kc = false, edge = false, dma = false;
  int vec = 1;
  int splits = 1;   // > 1: the problem needs split-K, run it unfused
  unsigned grid = 0;
  // K <= 16 and a row-major output whose rows 256 threads cover evenly: a store stream, not matrix work — the streaming
  // kernel on the vector ALUs (gemm_f32_mfma.hpp, gemm_narrow_k_block) instead of the matrix tile; narrow_grid blocks
  // of 256 threads.  Withdrawn by set_epilogue_operands for unaligned operands and by the caller for a row product.
  bool narrow = false;
  int narrow_k = 0;
  unsigned narrow_grid = 0;
  long matrix_k_per_split = 0;   // GemmArgs::k_per_split of the matrix tile (the streaming kernel reads its rows per block there)
  alignas(8) unsigned char args[320];  // the kernel's GemmArgs (opaque to host-only translation units)
  unsigned args_size = 0;
  unsigned waves = 0;   // per block (EG_GEMM_TRACE)
};
constexpr int MAX_EPILOGUE_OPERANDS = 6;

// Plan C = op(A) * op(B) (+ bias) like eg_sgemm(accumulate = 0).  `out.args` is ready to launch
// except for the epilogue operands.
int plan_fused(eg_ctx* ctx, int trans_a, int trans_b, long M, long N, long K, const float* A, long lda, const float* B,
               long ldb, float* C, long ldc, const float* bias, FusedLaunch& out);

// Weight gradient + bias gradient in one contraction (gemm_f32_mfma.hip): C[M + 1, N] = [op(A); 1] * op(B).
bool ones_row_supported(int trans_a, int trans_b, long M, long N, long K, const float* A, long lda, const float* B, long ldb);
int sgemm_ones_row(eg_ctx* ctx, int trans_a, int trans_b, long M, long N, long K, const float* A, long lda, const float* B,
                   long ldb, float* C, long ldc, int accumulate);

// Tensors the generated epilogue reads / writes (a.epi[i]), the seed-gradient scale and the epoch.
void set_epilogue_operands(FusedLaunch& f, void* const* ptrs, int count, float grad_scale, long epoch);

// Back to the matrix tile (an operand turned out unaligned, a row product rides on the launch, the batch pipeline).
void fused_withdraw_narrow(FusedLaunch& f);

// Does every tile of the planned launch leave through the wide-store pass (after set_epilogue_operands)?
bool fused_wide_store(const FusedLaunch& f);

// Identifies the template instantiation (cache key / kernel name suffix).
std::string fused_variant(const FusedLaunch& f);

// Complete hiprtc translation unit: the kernel header, `epi_struct` (which must define
// `struct <epi_name>` with ACTIVE and apply) and an extern "C" kernel `kernel_name(GemmArgs)`.
std::string fused_source(const FusedLaunch& f, const std::string& epi_struct, const std::string& epi_name,
                         const std::string& kernel_name);

// EG_GEMM_TRACE=1: cycle stamps of every wave of a launch (entry, k loop begins, k loop ends, epilogue done) — trace_begin
// allocates and zeroes the buffer (the pointer goes into the launch's GemmArgs), trace_end waits, prints mean / max and frees.
long long* trace_begin(eg_ctx* ctx, unsigned blocks, unsigned waves);
void fused_set_trace(FusedLaunch& f, long long* buffer);
void trace_end(eg_ctx* ctx, long long* buffer, unsigned blocks, unsigned waves, const char* what);

}  // namespace gemm
}  // namespace eg
