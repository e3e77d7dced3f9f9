// A/B harness for the LDS-DMA K loop of the f32 MFMA contraction kernel (exprgrad_amd/csrc/kernels/gemm_f32_mfma.hpp):
// the software-pipelined loop (ABL bit 6) against the straight one the library ships (ABL = 0), same tiles, interior problems only.
// Build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemm_pipe.hip -o tools/bin/gemm_pipe
// Run:    tools/bin/gemm_pipe M N K layout(0 NN, 1 NT, 2 TN) [launches per timing = 10] [rounds = 5]
// Prints microseconds / TFLOP/s per variant and whether the two variants agree bit for bit.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <ctime>

#include "../exprgrad_amd/csrc/kernels/gemm_f32_pair.hpp"

using namespace eg::gemm;

#define CHECK(x)                                                                    \
  do {                                                                              \
    hipError_t e = (x);                                                             \
    if (e != hipSuccess) {                                                          \
      fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e));                 \
      exit(1);                                                                      \
    }                                                                               \
  } while (0)

struct Variant {
  const char* name;
  int bm, bn, bk;
  void (*launch)(const GemmArgs&, dim3, hipStream_t);
};

template <int BM, int BN, int BK, int WM, int WN, int MINB, bool AKC, bool BKC, int ABL>
void launch(const GemmArgs& a, dim3 grid, hipStream_t s) {
  constexpr int NT = Geometry<BM, BN, WM, WN>::NT;
  hipLaunchKernelGGL((gemm_f32_mfma_kernel<BM, BN, BK, WM, WN, MINB, AKC, BKC, 4, false, 0, ABL, true>), grid, dim3(NT), 0, s, a);
}

template <int BM, int BN, int WM, int WN, bool AKC, bool BKC, int ABL, int ST = 3, int KB = 32, bool EDGE = false>
void launch_pair(const GemmArgs& a, dim3 grid, hipStream_t s) {
  hipLaunchKernelGGL((gemm_pair_kernel<BM, BN, WM, WN, AKC, BKC, ABL, ST, KB, EDGE>), grid, dim3(PairGeometry<BM, BN, WM, WN, ST, KB, EDGE>::NT), 0, s, a);
}

template <bool AKC, bool BKC>
std::vector<Variant> variants() {
  return {
      {"256x256x16 unskewed ", 256, 256, 16, launch<256, 256, 16, 128, 64, 1, AKC, BKC, 256>},
      {"256x256x16 skewed   ", 256, 256, 16, launch<256, 256, 16, 128, 64, 1, AKC, BKC, 0>},
      {"256x256x16 skew half", 256, 256, 16, launch<256, 256, 16, 128, 64, 1, AKC, BKC, 512>},
      {"256x128 4w 128x64 b2", 256, 128, 16, launch<256, 128, 16, 128, 64, 2, AKC, BKC, 0>},
      {"128x256 4w 128x64 b2", 128, 256, 16, launch<128, 256, 16, 128, 64, 2, AKC, BKC, 0>},
      {"256x128 4w 128x64 b1", 256, 128, 16, launch<256, 128, 16, 128, 64, 1, AKC, BKC, 0>},
      {"128x128 8w 64x32 b1 ", 128, 128, 16, launch<128, 128, 16, 64, 32, 1, AKC, BKC, 0>},
      {"128x128 8w 64x32 b2 ", 128, 128, 16, launch<128, 128, 16, 64, 32, 2, AKC, BKC, 0>},
      {"128x128 8w 64x32 uns", 128, 128, 16, launch<128, 128, 16, 64, 32, 2, AKC, BKC, 256>},
      {"256x256x16 p no-reads", 256, 256, 16, launch<256, 256, 16, 128, 64, 1, AKC, BKC, 65>},
      {"256x256x16 p no-dma  ", 256, 256, 16, launch<256, 256, 16, 128, 64, 1, AKC, BKC, 66>},
      {"256x256x16 p no-barr ", 256, 256, 16, launch<256, 256, 16, 128, 64, 1, AKC, BKC, 68>},
      {"256x256x16 p none    ", 256, 256, 16, launch<256, 256, 16, 128, 64, 1, AKC, BKC, 71>},
      {"256x256x16 no stores", 256, 256, 16, launch<256, 256, 16, 128, 64, 1, AKC, BKC, 128>},
      {"128x128x16 no stores", 128, 128, 16, launch<128, 128, 16, 64, 64, 4, AKC, BKC, 128>},
      {"128x128x16 pipelined", 128, 128, 16, launch<128, 128, 16, 64, 64, 4, AKC, BKC, 64>},
      {"128x128x16 straight ", 128, 128, 16, launch<128, 128, 16, 64, 64, 4, AKC, BKC, 0>},
      {"64x64x32   pipelined", 64, 64, 32, launch<64, 64, 32, 32, 32, 4, AKC, BKC, 64>},
      {"64x64x32   straight ", 64, 64, 32, launch<64, 64, 32, 32, 32, 4, AKC, BKC, 0>},
      {"64x64 pair skewed   ", 64, 64, 32, launch_pair<64, 64, 32, 32, AKC, BKC, 0>},
      {"64x64 pair in phase ", 64, 64, 32, launch_pair<64, 64, 32, 32, AKC, BKC, 256>},
      {"64x64 pair loads only", 64, 64, 32, launch_pair<64, 64, 32, 32, AKC, BKC, 257>},
      {"64x64 pair mfma only ", 64, 64, 32, launch_pair<64, 64, 32, 32, AKC, BKC, 258>},
      {"64x64 pair mfma nobar", 64, 64, 32, launch_pair<64, 64, 32, 32, AKC, BKC, 262>},
      {"64x64 pair mfma noread", 64, 64, 32, launch_pair<64, 64, 32, 32, AKC, BKC, 266>},
      {"64x64 pair mfma bare ", 64, 64, 32, launch_pair<64, 64, 32, 32, AKC, BKC, 270>},
      {"64x64 pair 2st bk64  ", 64, 64, 64, launch_pair<64, 64, 32, 32, AKC, BKC, 0, 2, 64>},
      {"64x64 pair bk64 EDGE ", 64, 64, 64, launch_pair<64, 64, 32, 32, AKC, BKC, 0, 2, 64, true>},
      {"64x64 pair 2st bk64 i", 64, 64, 64, launch_pair<64, 64, 32, 32, AKC, BKC, 256, 2, 64>},
      {"64x64 pair 3st bk64  ", 64, 64, 64, launch_pair<64, 64, 32, 32, AKC, BKC, 0, 3, 64>},
      {"64x64 pair 2st bk128 ", 64, 64, 128, launch_pair<64, 64, 32, 32, AKC, BKC, 0, 2, 128>},
      {"128x128 pair 2st bk64", 128, 128, 64, launch_pair<128, 128, 64, 64, AKC, BKC, 0, 2, 64>},
      {"128x64 pair 2st bk64 ", 128, 64, 64, launch_pair<128, 64, 64, 32, AKC, BKC, 0, 2, 64>},
      {"64x64 pair 2 stages  ", 64, 64, 32, launch_pair<64, 64, 32, 32, AKC, BKC, 0, 2>},
      {"64x64 pair 2st loads ", 64, 64, 32, launch_pair<64, 64, 32, 32, AKC, BKC, 257, 2>},
      {"64x64 pair 5 stages  ", 64, 64, 32, launch_pair<64, 64, 32, 32, AKC, BKC, 0, 5>},
      {"64x64 pair 5st loads ", 64, 64, 32, launch_pair<64, 64, 32, 32, AKC, BKC, 257, 5>},
      {"128x128 pair skewed ", 128, 128, 32, launch_pair<128, 128, 64, 64, AKC, BKC, 0>},
      {"128x128 pair in phas", 128, 128, 32, launch_pair<128, 128, 64, 64, AKC, BKC, 256>},
      {"128x64 pair skewed  ", 128, 64, 32, launch_pair<128, 64, 64, 32, AKC, BKC, 0>},
      {"128x64 pair in phase", 128, 64, 32, launch_pair<128, 64, 64, 32, AKC, BKC, 256>},
  };
}

int main(int argc, char** argv) {
  const long M = argc > 1 ? atol(argv[1]) : 4096, N = argc > 2 ? atol(argv[2]) : 4096, K = argc > 3 ? atol(argv[3]) : 4096;
  const int layout = argc > 4 ? atoi(argv[4]) : 0;
  const int per = argc > 5 ? atoi(argv[5]) : 10, rounds = argc > 6 ? atoi(argv[6]) : 5;
  const bool akc = layout != 2, bkc = layout == 1;
  std::vector<float> ha((size_t)M * K), hb((size_t)K * N);
  srand(1);
  for (auto& v : ha) v = (float)rand() / RAND_MAX;
  for (auto& v : hb) v = (float)rand() / RAND_MAX * 2.f - 1.f;
  float *A, *B, *C;
  CHECK(hipMalloc(&A, ha.size() * 4));
  CHECK(hipMalloc(&B, hb.size() * 4));
  CHECK(hipMalloc(&C, (size_t)M * N * 4));
  CHECK(hipMemcpy(A, ha.data(), ha.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(B, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
  hipStream_t s;
  CHECK(hipStreamCreate(&s));
  auto vs = layout == 0 ? variants<true, false>() : layout == 1 ? variants<true, true>() : variants<false, false>();
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  std::vector<std::vector<float>> out(vs.size());
  std::vector<float> best(vs.size(), 1e30f);
  auto args_for = [&](const Variant& v) {
    GemmArgs a = {};
    a.A = A; a.B = B; a.C = C;
    a.M = a.a_rows = M; a.N = N; a.K = K;
    a.lda = akc ? K : M; a.ldb = bkc ? K : N; a.ldc = N;
    a.tiles_m = (int)(M / v.bm); a.tiles_n = (int)(N / v.bn);
    a.k_per_split = K; a.splits = 1;
    a.wide_store = 1; a.nt_store = 1;
    return a;
  };
  for (int round = -1; round < rounds; ++round)
    for (size_t vi = 0; vi < vs.size(); ++vi) {
      const Variant& v = vs[vi];
      if (M % v.bm || N % v.bn || K % v.bk) continue;
      const GemmArgs a = args_for(v);
      const dim3 grid((unsigned)(a.tiles_m * a.tiles_n));
      if (round < 0) {  // correctness pass: keep the output
        CHECK(hipMemsetAsync(C, 0xff, (size_t)M * N * 4, s));
        v.launch(a, grid, s);
        out[vi].resize((size_t)M * N);
        CHECK(hipMemcpyAsync(out[vi].data(), C, (size_t)M * N * 4, hipMemcpyDeviceToHost, s));
        CHECK(hipStreamSynchronize(s));
        continue;
      }
      for (int i = 0; i < 3; ++i) v.launch(a, grid, s);
      if (getenv("SPIN_MS")) {  // sustained clocks: keep the device busy with this launch for SPIN_MS first
        const double spin = atof(getenv("SPIN_MS")) * 1e-3;
        timespec t0, t1;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        do {
          for (int i = 0; i < 50; ++i) v.launch(a, grid, s);
          CHECK(hipStreamSynchronize(s));
          clock_gettime(CLOCK_MONOTONIC, &t1);
        } while ((t1.tv_sec - t0.tv_sec) + (t1.tv_nsec - t0.tv_nsec) * 1e-9 < spin);
      }
      CHECK(hipEventRecord(e0, s));
      for (int i = 0; i < per; ++i) v.launch(a, grid, s);
      CHECK(hipEventRecord(e1, s));
      CHECK(hipEventSynchronize(e1));
      float ms;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      best[vi] = std::min(best[vi], ms / per);
    }
  // spot check against a float64 product on a sample of outputs
  double worst = 0;
  for (int t = 0; t < 64 && !out[0].empty(); ++t) {
    const long m = (long)rand() % M, n = (long)rand() % N;
    double ref = 0;
    for (long k = 0; k < K; ++k) ref += (double)(akc ? ha[m * K + k] : ha[k * M + m]) * (double)(bkc ? hb[n * K + k] : hb[k * N + n]);
    worst = std::max(worst, std::abs(ref - out[0][m * N + n]) / (std::abs(ref) + 1e-30));
  }
  printf("%ld x %ld x %ld layout %d: sampled rel err of variant 0 vs float64 %.2e\n", M, N, K, layout, worst);
  for (size_t vi = 0; vi < vs.size(); ++vi) {
    if (out[vi].empty()) continue;
    const size_t other = vi ^ 1;
    const bool same = other < out.size() && !out[other].empty() && out[other].size() == out[vi].size() && memcmp(out[vi].data(), out[other].data(), out[vi].size() * 4) == 0;
    size_t ref = 0;
    while (out[ref].empty()) ++ref;
    double dmax = 0, scale = 0;
    for (size_t e = 0; e < out[vi].size(); ++e) {
      dmax = std::max(dmax, (double)std::abs(out[vi][e] - out[ref][e]));
      scale = std::max(scale, (double)std::abs(out[ref][e]));
    }
    printf("  %s  %9.1f us  %7.2f TFLOP/s   %s   vs first variant %.1e\n", vs[vi].name, best[vi] * 1e3, 2.0 * M * N * K / best[vi] / 1e9,
           same ? "bit-identical to its twin" : "DIFFERS from its twin", dmax / scale);
  }
  return 0;
}
