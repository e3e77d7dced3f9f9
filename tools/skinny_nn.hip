// Microbenchmark of gemm_skinny_nn_kernel (exprgrad_amd/csrc/kernels/gemm_skinny.hpp) at the classifier shape
// 65536 x 10 x 512 and a few neighbours; checks against a float64 host product.
// hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -Iexprgrad_amd/csrc/kernels tools/skinny_nn.hip -o tools/bin/skinny_nn
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "gemm_skinny.hpp"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int W>
static float run(const float* A, const float* B, float* C, const float* bias, long M, int N, int K, int blocks_per_cu, int reps) {
  long blocks = ((M + 15) / 16 + 3) / 4;
  if (blocks > 256L * blocks_per_cu) blocks = 256L * blocks_per_cu;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i)
    hipLaunchKernelGGL((eg_skinny::gemm_skinny_nn_kernel<W>), dim3((unsigned)blocks), dim3(256), (size_t)K * 64, 0, A, B, C, bias, M, N, K, (long)K, (long)N, (long)N, 0);
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i)
    hipLaunchKernelGGL((eg_skinny::gemm_skinny_nn_kernel<W>), dim3((unsigned)blocks), dim3(256), (size_t)K * 64, 0, A, B, C, bias, M, N, K, (long)K, (long)N, (long)N, 0);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1000.f / reps;
}

int main() {
  const long shapes[][3] = {{65536, 10, 512}, {65536, 16, 512}, {65536, 10, 1024}, {8192, 10, 512}, {65536 * 4, 10, 256}};
  for (auto& sh : shapes) {
    const long M = sh[0]; const int N = (int)sh[1], K = (int)sh[2];
    std::vector<float> a(M * K), b((size_t)K * N), bias(N), c(M * N);
    unsigned s = 12345;
    auto rnd = [&] { s = s * 1664525u + 1013904223u; return (float)(s >> 8) / (1 << 24) - 0.5f; };
    for (auto& v : a) v = rnd();
    for (auto& v : b) v = rnd();
    for (auto& v : bias) v = rnd();
    float *dA, *dB, *dC, *dbias;
    CK(hipMalloc(&dA, a.size() * 4)); CK(hipMalloc(&dB, b.size() * 4)); CK(hipMalloc(&dC, c.size() * 4)); CK(hipMalloc(&dbias, N * 4));
    CK(hipMemcpy(dA, a.data(), a.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, b.data(), b.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dbias, bias.data(), N * 4, hipMemcpyHostToDevice));
    const double bytes = (double)M * K * 4 + (double)M * N * 4;
    for (int bpc : {2, 4, 5, 8}) {
      const float t4 = run<4>(dA, dB, dC, dbias, M, N, K, bpc, 50);
      const float t8 = run<8>(dA, dB, dC, dbias, M, N, K, bpc, 50);
      printf("M %ld N %d K %d  blocks/CU %d : W4 %.1f us (%.2f TB/s)  W8 %.1f us (%.2f TB/s)\n", M, N, K, bpc, t4, bytes / t4 * 1e-6, t8, bytes / t8 * 1e-6);
    }
    CK(hipMemcpy(c.data(), dC, c.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0, scale = 0;
    for (long m = 0; m < M; m += 997) for (int n = 0; n < N; ++n) {
      double w = bias[n];
      for (int k = 0; k < K; ++k) w += (double)a[m * K + k] * b[(size_t)k * N + n];
      worst = fmax(worst, fabs(w - c[m * N + n])); scale = fmax(scale, fabs(w));
    }
    printf("   max |err| %.3g of max |c| %.3g\n", worst, scale);
    CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC)); CK(hipFree(dbias));
  }
  return 0;
}
