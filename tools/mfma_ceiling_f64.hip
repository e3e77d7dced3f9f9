// What the float64 matrix cores sustain with nothing but MFMAs — the ceiling eg_dgemm is measured against.
// v_mfma_f64_16x16x4_f64, 16 independent accumulator blocks per wave (the 128 x 128 tile's wave: 4 x 4), operands in
// registers, W waves per SIMD, every CU busy.  Prints TFLOP/s and cycles per MFMA (wave clock ticks / MFMAs issued).
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_ceiling_f64.hip -o tools/bin/mfma_ceiling_f64
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                    \
  do {                                                              \
    hipError_t e = (x);                                             \
    if (e != hipSuccess) {                                          \
      fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e)); \
      exit(1);                                                      \
    }                                                               \
  } while (0)

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64, 1) void mfma_loop(const double* __restrict__ in, double* __restrict__ out, int iters,
                                                           long long* __restrict__ ticks) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  double a[4], b[4];
  for (int i = 0; i < 4; ++i) a[i] = in[(t * 8 + i) & 0xfffff];
  for (int i = 0; i < 4; ++i) b[i] = in[(t * 8 + 4 + i) & 0xfffff];
  d4 acc[4][4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) acc[i][j] = d4{0, 0, 0, 0};
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
  }
  const long long t1 = __builtin_readcyclecounter();
  double s = 0;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j)
      for (int r = 0; r < 4; ++r) s += acc[i][j][r];
  out[t] = s;
  if (t == 0) *ticks = t1 - t0;
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 2000;  // 16 MFMAs per iteration per wave
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  std::vector<double> h(1 << 20);
  double *in, *out;
  long long* ticks;
  CHECK(hipMalloc(&in, h.size() * 8));
  CHECK(hipMalloc(&out, (size_t)cus * 1024 * 8));
  CHECK(hipMalloc(&ticks, 8));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int data = 0; data < 2; ++data) {
    srand(7);
    for (auto& v : h) v = data == 0 ? (double)rand() / (double)RAND_MAX * 2. - 1. : 0.;
    CHECK(hipMemcpy(in, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    for (int waves = 4; waves <= 8; waves += 4) {
      auto launch = [&] {
        if (waves == 4) hipLaunchKernelGGL(mfma_loop<4>, dim3(cus), dim3(256), 0, 0, in, out, iters, ticks);
        else hipLaunchKernelGGL(mfma_loop<8>, dim3(cus), dim3(512), 0, 0, in, out, iters, ticks);
      };
      for (int i = 0; i < 20; ++i) launch();
      float best = 1e30f;
      long long tk = 0;
      for (int rep = 0; rep < 5; ++rep) {
        CHECK(hipEventRecord(e0, 0));
        launch();
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
        CHECK(hipMemcpy(&tk, ticks, 8, hipMemcpyDeviceToHost));
      }
      const double flops = (double)cus * waves * iters * 16 * 2048.0;
      printf("%s operands, %d waves/SIMD: %.3f ms  %.2f TFLOP/s  (%.1f counter ticks per MFMA of one wave)\n", data == 0 ? "random" : "zero", waves / 4,
             best, flops / best / 1e9, (double)tk / ((double)iters * 16));
    }
  }
  return 0;
}
