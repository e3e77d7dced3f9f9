// What the vector ALUs sustain in float64 multiply-adds (v_fma_f64), W waves per SIMD: the ceiling of the direct float64 convolution.
// Build: hipcc --offload-arch=gfx950 -O3 tools/fma64_ceiling.hip -o tools/bin/fma64_ceiling
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

template <int WAVES, bool SCALAR_B>
__global__ __launch_bounds__(WAVES * 64) void fma_loop(const double* __restrict__ in, double* __restrict__ out, int iters) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  double acc[16], a[4];
  for (int i = 0; i < 16; ++i) acc[i] = 0.0;
  for (int i = 0; i < 4; ++i) a[i] = in[(t * 4 + i) & 0xffff];
  for (int it = 0; it < iters; ++it) {
    double b[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) b[j] = SCALAR_B ? in[(it * 4 + j) & 0xffff] : a[(j + 1) & 3] + (double)it;  // scalar (uniform) load or vector value
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = __builtin_fma(a[i & 3], b[i >> 2], acc[i]);
  }
  double s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i];
  out[t] = s;
}

int main() {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount, iters = 4000;
  double *in, *out;
  hipMalloc(&in, 65536 * 8);
  hipMemset(in, 0, 65536 * 8);
  hipMalloc(&out, (size_t)cus * 16 * 1024 * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int scalar = 0; scalar < 2; ++scalar)
    for (int waves = 4; waves <= 16; waves *= 2) {
      auto launch = [&] {
        const dim3 g(cus), b(waves * 64);
        if (scalar) {
          if (waves == 4) hipLaunchKernelGGL((fma_loop<4, true>), g, b, 0, 0, in, out, iters);
          else if (waves == 8) hipLaunchKernelGGL((fma_loop<8, true>), g, b, 0, 0, in, out, iters);
          else hipLaunchKernelGGL((fma_loop<16, true>), g, b, 0, 0, in, out, iters);
        } else {
          if (waves == 4) hipLaunchKernelGGL((fma_loop<4, false>), g, b, 0, 0, in, out, iters);
          else if (waves == 8) hipLaunchKernelGGL((fma_loop<8, false>), g, b, 0, 0, in, out, iters);
          else hipLaunchKernelGGL((fma_loop<16, false>), g, b, 0, 0, in, out, iters);
        }
      };
      for (int i = 0; i < 10; ++i) launch();
      float best = 1e30f;
      for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0, 0);
        launch();
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      const double flops = (double)cus * waves * 64 * iters * 16 * 2.0;
      printf("%s operand, %d waves/SIMD: %.3f ms  %.2f TFLOP/s\n", scalar ? "scalar-loaded" : "vector", waves / 4, best, flops / best / 1e9);
    }
  return 0;
}
