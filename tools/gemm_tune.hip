// Tuning harness for the f32 MFMA contraction kernel (exprgrad_amd/csrc/kernels/gemm_f32_mfma.hpp).
// Build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude tools/gemm_tune.hip -o gpurun_out/gemm_tune
// Run on the GPU box; prints TFLOP/s per variant (interleaved rounds, median and best).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../exprgrad_amd/csrc/kernels/gemm_f32_mfma.hpp"

using namespace eg::gemm;

#define CHECK(x)                                                                        \
  do {                                                                                  \
    hipError_t e = (x);                                                                 \
    if (e != hipSuccess) {                                                              \
      fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e));                     \
      exit(1);                                                                          \
    }                                                                                   \
  } while (0)

struct Variant {
  const char* name;
  int bm, bn, bk;
  void (*launch)(const GemmArgs&, dim3 grid, hipStream_t);
  size_t threads;
};

template <int BM, int BN, int BK, int WM, int WN, int MINB, bool AKC, bool BKC, int ABL = 0>
void launch_variant(const GemmArgs& a, dim3 grid, hipStream_t s) {
  constexpr int NT = Geometry<BM, BN, WM, WN>::NT;
  hipLaunchKernelGGL((gemm_f32_mfma_kernel<BM, BN, BK, WM, WN, MINB, AKC, BKC, 4, false, false, ABL>), grid, dim3(NT), 0,
                     s, a);
}
#define VA(BM, BN, BK, WM, WN, MINB, AKC, BKC, ABL) \
  { #BM "x" #BN "x" #BK " w" #WM "x" #WN " b" #MINB " abl" #ABL, BM, BN, BK, launch_variant<BM, BN, BK, WM, WN, MINB, AKC, BKC, ABL>, 0 }

template <int BM, int BN, int WM, int WN, int MINB, bool AKC, bool BKC>
void launch_dma(const GemmArgs& a, dim3 grid, hipStream_t s) {
  constexpr int NT = Geometry<BM, BN, WM, WN>::NT;
  hipLaunchKernelGGL((gemm_f32_mfma_kernel<BM, BN, 16, WM, WN, MINB, AKC, BKC, 4, false, false, 0, true>), grid, dim3(NT),
                     0, s, a);
}
#define VD(BM, BN, WM, WN, MINB, AKC, BKC) \
  { #BM "x" #BN "x16 w" #WM "x" #WN " b" #MINB " DMA", BM, BN, 16, launch_dma<BM, BN, WM, WN, MINB, AKC, BKC>, 0 }

#define V(BM, BN, BK, WM, WN, MINB, AKC, BKC) \
  { #BM "x" #BN "x" #BK " w" #WM "x" #WN " b" #MINB, BM, BN, BK, launch_variant<BM, BN, BK, WM, WN, MINB, AKC, BKC>, 0 }

template <bool AKC, bool BKC>
std::vector<Variant> variants() {
  return {
      V(128, 128, 16, 64, 64, 4, AKC, BKC),
      V(128, 128, 16, 64, 64, 3, AKC, BKC),
      V(128, 128, 32, 64, 64, 2, AKC, BKC),
      V(128, 128, 8, 64, 64, 4, AKC, BKC),
      V(256, 128, 16, 64, 64, 2, AKC, BKC),
      V(128, 256, 16, 64, 64, 2, AKC, BKC),
      V(256, 256, 16, 64, 64, 1, AKC, BKC),
      V(256, 128, 16, 128, 64, 2, AKC, BKC),
      V(128, 256, 16, 64, 128, 2, AKC, BKC),
      V(256, 256, 16, 128, 64, 1, AKC, BKC),
      V(256, 256, 16, 128, 128, 1, AKC, BKC),
      V(256, 128, 8, 64, 64, 2, AKC, BKC),
      VD(128, 128, 64, 64, 4, AKC, BKC),
      VD(256, 256, 64, 64, 1, AKC, BKC),
      VD(256, 128, 64, 64, 2, AKC, BKC),
      VD(128, 256, 64, 64, 2, AKC, BKC),
      VD(256, 256, 128, 64, 1, AKC, BKC),
      VA(128, 128, 16, 64, 64, 4, AKC, BKC, 1),
      VA(128, 128, 16, 64, 64, 4, AKC, BKC, 2),
      VA(128, 128, 16, 64, 64, 4, AKC, BKC, 4),
      VA(128, 128, 16, 64, 64, 4, AKC, BKC, 7),
      VA(256, 256, 16, 64, 64, 1, AKC, BKC, 1),
      VA(256, 256, 16, 64, 64, 1, AKC, BKC, 2),
      VA(256, 256, 16, 64, 64, 1, AKC, BKC, 7),
      VA(256, 256, 16, 64, 64, 1, AKC, BKC, 8),
      VA(256, 256, 16, 64, 64, 1, AKC, BKC, 16),
      VA(256, 256, 16, 64, 64, 1, AKC, BKC, 4),
      VA(128, 128, 16, 64, 64, 4, AKC, BKC, 8),
      VA(128, 128, 16, 64, 64, 4, AKC, BKC, 16),
  };
}

int main(int argc, char** argv) {
  long M = argc > 1 ? atol(argv[1]) : 4096, N = argc > 2 ? atol(argv[2]) : 4096, K = argc > 3 ? atol(argv[3]) : 4096;
  const int layout = argc > 4 ? atoi(argv[4]) : 0;  // 0 NN, 1 NT, 2 TN
  const int splits = argc > 5 ? atoi(argv[5]) : 1;
  const int rounds = argc > 6 ? atoi(argv[6]) : 7;
  const bool akc = layout != 2, bkc = layout == 1;
  std::vector<float> ha((size_t)M * K), hb((size_t)K * N);
  srand(1);
  for (auto& v : ha) v = (float)rand() / RAND_MAX;
  for (auto& v : hb) v = (float)rand() / RAND_MAX * 2.f - 1.f;
  float *A, *B, *C, *Cref, *P;
  CHECK(hipMalloc(&A, ha.size() * 4));
  CHECK(hipMalloc(&B, hb.size() * 4));
  CHECK(hipMalloc(&C, (size_t)M * N * 4));
  CHECK(hipMalloc(&Cref, (size_t)M * N * 4));
  CHECK(hipMalloc(&P, (size_t)M * N * 4 * (splits > 1 ? splits : 1)));
  CHECK(hipMemcpy(A, ha.data(), ha.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(B, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
  hipStream_t s;
  CHECK(hipStreamCreate(&s));
  std::vector<Variant> vs = layout == 0 ? variants<true, false>() : layout == 1 ? variants<true, true>() : variants<false, false>();
  std::vector<std::vector<float>> times(vs.size());
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  std::vector<float> href((size_t)M * N), hc((size_t)M * N);
  for (int round = -1; round < rounds; ++round) {
    for (size_t vi = 0; vi < vs.size(); ++vi) {
      const Variant& v = vs[vi];
      if (M % v.bm || N % v.bn || K % v.bk) continue;
      GemmArgs a = {};
      a.A = A;
      a.B = B;
      a.C = C;
      a.M = M;
      a.N = N;
      a.K = K;
      a.lda = akc ? K : M;
      a.ldb = bkc ? K : N;
      a.ldc = N;
      a.tiles_m = (int)(M / v.bm);
      a.tiles_n = (int)(N / v.bn);
      long k_tiles = K / v.bk, per = (k_tiles + splits - 1) / splits;
      a.k_per_split = per * v.bk;
      a.partial = splits > 1 ? P : nullptr;
      a.splits = splits;
      dim3 grid(a.tiles_m * a.tiles_n * splits, 1, 1);
      const int reps = round < 0 ? 1 : 5;
      CHECK(hipEventRecord(e0, s));
      for (int r = 0; r < reps; ++r) {
        v.launch(a, grid, s);
        if (splits > 1)
          hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3(2048), dim3(256), 0, s, P, C, (const float*)nullptr, M, N, N, splits, 0, M, 0);
      }
      CHECK(hipEventRecord(e1, s));
      CHECK(hipStreamSynchronize(s));
      CHECK(hipGetLastError());
      float ms;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (round >= 0) times[vi].push_back(ms / reps);
      if (round < 0) {  // correctness vs variant 0
        if (vi == 0) {
          CHECK(hipMemcpy(href.data(), C, href.size() * 4, hipMemcpyDeviceToHost));
        } else {
          CHECK(hipMemcpy(hc.data(), C, hc.size() * 4, hipMemcpyDeviceToHost));
          double md = 0, mx = 0;
          for (size_t i = 0; i < hc.size(); i += 97) {
            md = std::max(md, (double)fabsf(hc[i] - href[i]));
            mx = std::max(mx, (double)fabsf(href[i]));
          }
          if (md > 1e-5 * mx && !strstr(v.name, "abl")) printf("!! variant %s differs from variant 0: %g (max %g)\n", v.name, md, mx);
        }
      }
    }
  }
  const double flops = 2.0 * M * N * K;
  printf("M=%ld N=%ld K=%ld layout=%d splits=%d\n", M, N, K, layout, splits);
  for (size_t vi = 0; vi < vs.size(); ++vi) {
    if (times[vi].empty()) continue;
    std::sort(times[vi].begin(), times[vi].end());
    const float med = times[vi][times[vi].size() / 2], best = times[vi][0];
    printf("%-26s median %8.4f ms  %7.2f TF   best %8.4f ms  %7.2f TF\n", vs[vi].name, med, flops / med / 1e9, best,
           flops / best / 1e9);
  }
  return 0;
}
