// A/B of generated-epilogue forms on the 256 x 256 tile (NN, whole tiles): what the forward product of a relu layer costs
// when it stores h and relu(h), relu(h) only, or relu(h) and the predicate bits of h.  tools/bin/epi_ab [M N K]
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../exprgrad_amd/csrc/kernels/gemm_f32_mfma.hpp"
using namespace eg::gemm;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <bool STORE, int PREDV, int MODE>
struct Relu {
  static constexpr bool ACTIVE = true;
  static constexpr int NX = 1;
  static constexpr bool STORE_C = STORE;
  static constexpr int OUT = 0;
  static constexpr int PRED = PREDV;
  static constexpr int RD_N = 0, RD_W = 0, RD_OUT = 0, RD_BIAS = -1, RD_LDW = 0, RD_LDO = 0;
  __device__ __forceinline__ static bool predicate(float v) { return 0.0f <= v; }
  __device__ __forceinline__ static void prefetch(const GemmArgs&, long, float (&)[1]) {}
  __device__ __forceinline__ static void prefetch4(const GemmArgs&, long, f32x4 (&)[1]) {}
  __device__ __forceinline__ static float compute(const GemmArgs&, long, float v, const float (&)[1]) { return 0.0f <= v ? v : 0.0f; }
};
template <class Epi>
__global__ __launch_bounds__(512, 2) void fused(GemmArgs a) { gemm_block<256, 256, 16, 128, 64, true, false, 4, false, 0, 0, true, Epi>(a); }

int main(int argc, char** argv) {
  const long M = argc > 1 ? atol(argv[1]) : 65536, N = argc > 2 ? atol(argv[2]) : 512, K = argc > 3 ? atol(argv[3]) : 784;
  float *A, *B, *C, *R, *bias; unsigned* bits;
  CHECK(hipMalloc(&A, M * K * 4)); CHECK(hipMalloc(&B, K * N * 4)); CHECK(hipMalloc(&C, M * N * 4)); CHECK(hipMalloc(&R, M * N * 4));
  CHECK(hipMalloc(&bias, N * 4)); CHECK(hipMalloc(&bits, M * N / 8));
  std::vector<float> h((size_t)M * K);
  srand(1);
  for (auto& v : h) v = (float)rand() / (float)RAND_MAX - 0.5f;
  CHECK(hipMemcpy(A, h.data(), M * K * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(B, h.data(), K * N * 4, hipMemcpyHostToDevice));
  CHECK(hipMemset(bias, 0, N * 4));
  GemmArgs a = {};
  a.A = A; a.B = B; a.C = C; a.bias = bias; a.M = a.a_rows = M; a.N = N; a.K = K; a.lda = K; a.ldb = N; a.ldc = N;
  a.tiles_m = (int)(M / 256); a.tiles_n = (int)(N / 256); a.k_per_split = K; a.splits = 1; a.wide_store = 1; a.nt_store = 1;
  a.epi[0] = R; a.epi[1] = bits;
  const dim3 grid((unsigned)(a.tiles_m * a.tiles_n));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  if (argc > 4) {  // placement sweep: the output tensor at byte offsets from one large allocation
    char* big;
    CHECK(hipMalloc(&big, (size_t)M * N * 4 + (64u << 20)));
    printf("allocation at %p\n", (void*)big);
    for (int ai = 4; ai < argc; ++ai) {
      const long off = atol(argv[ai]);
      a.epi[0] = big + off;
      void (*k)(GemmArgs) = fused<Relu<false, 1, 0>>;
      float best = 1e30f;
      for (int round = 0; round < 3; ++round) {
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, grid, dim3(512), 0, 0, a);
        CHECK(hipEventRecord(e0, 0));
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k, grid, dim3(512), 0, 0, a);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms * 100);
      }
      printf("  output at +%-10ld %8.1f us\n", off, best);
    }
    return 0;
  }
  struct V { const char* name; void (*k)(GemmArgs); } vs[] = {
      {"h and relu(h)          ", fused<Relu<true, -1, 0>>},
      {"relu(h) only           ", fused<Relu<false, -1, 0>>},
      {"relu(h) + bits of h    ", fused<Relu<false, 1, 0>>},
  };
  for (int round = 0; round < 3; ++round)
    for (auto& v : vs) {
      for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(v.k, grid, dim3(512), 0, 0, a);
      CHECK(hipEventRecord(e0, 0));
      for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(v.k, grid, dim3(512), 0, 0, a);
      CHECK(hipEventRecord(e1, 0));
      CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (round == 2) printf("%s %8.1f us\n", v.name, ms * 100);
    }
  return 0;
}
