// Tile shapes for the fused activation-gradient product of a dense layer with few outputs:
//   dH[M, N] = select(bits of h, dL[M, K] * W2[N, K]^T, 0)   (NT, K = 10: one ragged k-tile; the launch is bound by writing dH)
// hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -Iexprgrad_amd/csrc/kernels tools/dgrad_ab.hip -o tools/bin/dgrad_ab
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../exprgrad_amd/csrc/kernels/gemm_f32_mfma.hpp"
using namespace eg::gemm;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct Sel {
  static constexpr bool ACTIVE = true;
  static constexpr int NX = 1;
  static constexpr bool STORE_C = false;
  static constexpr int OUT = 0;
  static constexpr int PRED = -1;
  static constexpr int RD_N = 0, RD_W = 0, RD_OUT = 0, RD_BIAS = -1, RD_LDW = 0, RD_LDO = 0;
  __device__ __forceinline__ static bool predicate(float) { return false; }
  __device__ __forceinline__ static void prefetch(const GemmArgs& a, long idx, float (&x)[1]) {
    x[0] = (float)((((const unsigned*)a.epi[1])[idx >> 5] >> (idx & 31)) & 1u);
  }
  __device__ __forceinline__ static void prefetch4(const GemmArgs& a, long idx, f32x4 (&x)[1]) {
    const unsigned w = ((const unsigned*)a.epi[1])[idx >> 5] >> (idx & 31);
    for (int e = 0; e < 4; ++e) x[0][e] = (float)((w >> e) & 1u);
  }
  __device__ __forceinline__ static float compute(const GemmArgs&, long, float v, const float (&x)[1]) { return x[0] != 0.0f ? v : 0.0f; }
};

template <int BM, int BN, int WM, int WN, int MINB>
__global__ __launch_bounds__((BM / WM) * (BN / WN) * 64, MINB) void fused(GemmArgs a) {
  gemm_block<BM, BN, 16, WM, WN, true, true, 1, true, 0, 0, false, Sel>(a);
}

template <int BM, int BN, int WM, int WN, int MINB>
static void run(const char* name, GemmArgs a) {
  a.tiles_m = (int)((a.M + BM - 1) / BM);
  a.tiles_n = (int)((a.N + BN - 1) / BN);
  const dim3 grid((unsigned)(a.tiles_m * a.tiles_n)), block((BM / WM) * (BN / WN) * 64);
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int round = 0; round < 3; ++round) {
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((fused<BM, BN, WM, WN, MINB>), grid, block, 0, 0, a);
    CHECK(hipEventRecord(e0, 0));
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((fused<BM, BN, WM, WN, MINB>), grid, block, 0, 0, a);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    best = std::min(best, ms * 50);
  }
  printf("%-28s %8.1f us  (%.2f TB/s written)\n", name, best, (double)a.M * a.N * 4 / best * 1e-6);
}

int main(int argc, char** argv) {
  const long M = argc > 1 ? atol(argv[1]) : 65536, N = argc > 2 ? atol(argv[2]) : 512, K = argc > 3 ? atol(argv[3]) : 10;
  float *A, *B, *R;
  unsigned* bits;
  CHECK(hipMalloc(&A, M * K * 4)); CHECK(hipMalloc(&B, N * K * 4)); CHECK(hipMalloc(&R, M * N * 4)); CHECK(hipMalloc(&bits, M * N / 8));
  std::vector<float> h((size_t)M * K);
  srand(1);
  for (auto& v : h) v = (float)rand() / (float)RAND_MAX - 0.5f;
  CHECK(hipMemcpy(A, h.data(), M * K * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(B, h.data(), N * K * 4, hipMemcpyHostToDevice));
  CHECK(hipMemset(bits, 0x5a, M * N / 8));
  GemmArgs a = {};
  a.A = A; a.B = B; a.C = R; a.M = a.a_rows = M; a.N = N; a.K = K; a.lda = K; a.ldb = K; a.ldc = N;
  a.k_per_split = 16; a.splits = 1; a.nt_store = 1;
  a.epi[0] = R; a.epi[1] = bits;
  for (int wide = 1; wide >= 0; --wide) {
    a.wide_store = wide;
    printf("wide_store %d\n", wide);
    run<128, 128, 64, 64, 4>("128 x 128 (the planner's)", a);
    run<256, 256, 128, 64, 1>("256 x 256", a);
    run<128, 64, 64, 32, 4>("128 x 64", a);
    run<64, 64, 32, 32, 4>("64 x 64", a);
    run<256, 64, 64, 32, 2>("256 x 64", a);
    run<128, 32, 32, 32, 4>("128 x 32", a);
  }
  return 0;
}
