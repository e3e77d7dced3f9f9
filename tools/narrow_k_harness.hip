// The narrow-K streaming kernel (exprgrad_amd/csrc/kernels/gemm_f32_mfma.hpp, gemm_narrow_k_block) stand-alone, with the functor
// host/epilogue.cpp generates for relu's gradient on predicate bits: the cfg-5 activation-gradient product 65 536 x 512 x 10.
//   hipcc -O3 -ffp-contract=off -std=c++17 --offload-arch=gfx950 tools/narrow_k_harness.hip -o tools/bin/nk_harness
//   [BURN=<iterations of an MFMA burner in front of every timed launch>] [OUT_OFFSET=<floats>] [RANDOM=1] tools/bin/nk_harness
// Prints microseconds per launch for several grids, one output buffer and four in rotation (beyond the Infinity Cache).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../exprgrad_amd/csrc/kernels/gemm_f32_mfma.hpp"

struct EgEpi {
  static constexpr bool ACTIVE = true;
  static constexpr int NX = 1;
  static constexpr bool STORE_C = false;
  static constexpr int OUT = 1;
  static constexpr int PRED = -1;
  static constexpr int RD_N = 0, RD_W = 0, RD_OUT = 0, RD_BIAS = -1, RD_LDW = 0, RD_LDO = 0;
  __device__ __forceinline__ static bool predicate(float) { return false; }
  __device__ __forceinline__ static void prefetch(const eg::gemm::GemmArgs& a, long idx, float (&x)[1]) {
    x[0] = (float)((((const unsigned*)a.epi[0])[idx >> 5] >> (idx & 31)) & 1u);
  }
  __device__ __forceinline__ static void prefetch4(const eg::gemm::GemmArgs& a, long idx, eg::gemm::f32x4 (&x)[1]) {
    const unsigned w = ((const unsigned*)a.epi[0])[idx >> 5] >> (idx & 31);
    for (int e = 0; e < 4; ++e) x[0][e] = (float)((w >> e) & 1u);
  }
  __device__ __forceinline__ static float compute(const eg::gemm::GemmArgs&, long, float v, const float (&x)[1]) {
    return 0.0f + (x[0] != 0.0f ? v : 0.0f);
  }
};
extern "C" __global__ __launch_bounds__(256) void narrow_k(eg::gemm::GemmArgs a) {
  eg::gemm::gemm_narrow_k_block<10, 128, true, true, EgEpi>(a);
}
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(512) void burner(float* sink, int iters) {
  f16v acc[4];
  for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  const float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f;
  for (int i = 0; i < iters; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
  float s = 0;
  for (int j = 0; j < 4; ++j) s += acc[j][0];
  if (s == 123.456f) sink[0] = s;
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
int main() {
  const long M = 65536, N = 512, K = 10;
  const int burn = getenv("BURN") ? atoi(getenv("BURN")) : 0;
  const long off = getenv("OUT_OFFSET") ? atol(getenv("OUT_OFFSET")) : 0;
  float *gz, *w2, *out[4], *sink;
  unsigned* bits;
  CK(hipMalloc(&gz, M * K * 4)); CK(hipMalloc(&w2, N * K * 4)); CK(hipMalloc(&bits, M * N / 8)); CK(hipMalloc(&sink, 64));
  if (getenv("RANDOM")) {
    std::vector<float> h(M * K); for (auto& v : h) v = ((float)rand() / RAND_MAX - 0.5f) * 1e-3f;
    CK(hipMemcpy(gz, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    std::vector<float> w(N * K); for (auto& v : w) v = ((float)rand() / RAND_MAX - 0.5f) * 0.1f;
    CK(hipMemcpy(w2, w.data(), w.size() * 4, hipMemcpyHostToDevice));
    std::vector<unsigned> b(M * N / 32); for (auto& v : b) v = (unsigned)rand() * 2654435761u;
    CK(hipMemcpy(bits, b.data(), b.size() * 4, hipMemcpyHostToDevice));
  } else {
    CK(hipMemset(gz, 0, M * K * 4)); CK(hipMemset(w2, 0, N * K * 4)); CK(hipMemset(bits, 0x5a, M * N / 8));
  }
  for (int i = 0; i < 4; ++i) CK(hipMalloc(&out[i], M * N * 4 + 4096));
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int sets = 1; sets <= 4; sets += 3)
    for (long grid : {1024L, 2048L, 4096L, 8192L}) {
      const long rpt = 8, per = ((M + grid - 1) / grid + rpt - 1) / rpt * rpt;   // four rows of two per trip
      auto launch = [&](int s) {
        eg::gemm::GemmArgs a = {};
        a.A = gz; a.B = w2; a.C = out[s] + off; a.M = M; a.N = N; a.K = K; a.lda = K; a.ldb = K; a.ldc = N; a.a_rows = M;
        a.epi[0] = bits; a.epi[1] = out[s] + off; a.k_per_split = per;
        hipLaunchKernelGGL(narrow_k, dim3((unsigned)((M + per - 1) / per)), dim3(256), 0, st, a);
      };
      for (int i = 0; i < 10; ++i) launch(i % sets);
      float total = 0;
      for (int i = 0; i < 30; ++i) {
        if (burn) hipLaunchKernelGGL(burner, dim3(256), dim3(512), 0, st, sink, burn);
        CK(hipEventRecord(e0, st));
        launch(i % sets);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        total += ms;
      }
      printf("narrow-K kernel, %d output set(s), %ld blocks%s: %.2f us\n", sets, grid, burn ? ", behind an MFMA burner" : "", total * 1e3 / 30);
    }
  return 0;
}
