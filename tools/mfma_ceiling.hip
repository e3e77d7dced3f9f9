// What the f32 matrix cores sustain with nothing but MFMAs — the ceiling the contraction kernels are measured against.
// v_mfma_f32_32x32x2_f32, 8 independent accumulator blocks per wave (the 256 x 256 tile's wave), operands in registers,
// W waves per SIMD, every CU busy for ~1 ms.  Operands: random (what a real product toggles), zero, or constant —
// the chip clocks to its power budget (MI355X_MICROARCH.md, "DVFS give-back"), so the sustained rate depends on the data.
// Prints TFLOP/s and the effective shader clock (s_memtime ticks of one wave / wall time).
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_ceiling.hip -o tools/bin/mfma_ceiling
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CHECK(x)                                                                    \
  do {                                                                              \
    hipError_t e = (x);                                                             \
    if (e != hipSuccess) {                                                          \
      fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e));                 \
      exit(1);                                                                      \
    }                                                                               \
  } while (0)

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64, 1) void mfma_loop(const float* __restrict__ in, float* __restrict__ out, int iters,
                                                           long long* __restrict__ ticks) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  float a[4], b[2];
  for (int i = 0; i < 4; ++i) a[i] = in[(t * 6 + i) & 0xfffff];
  for (int i = 0; i < 2; ++i) b[i] = in[(t * 6 + 4 + i) & 0xfffff];
  f32x16 acc[4][2];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 2; ++j)
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 2; ++j)
      for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  out[t] = s;
  if (t == 0) *ticks = t1 - t0;
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 1200;  // 32 MFMAs per iteration per wave
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  std::vector<float> h(1 << 20);
  float *in, *out;
  long long* ticks;
  CHECK(hipMalloc(&in, h.size() * 4));
  CHECK(hipMalloc(&out, (size_t)cus * 1024 * 4));
  CHECK(hipMalloc(&ticks, 8));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int data = 0; data < 3; ++data) {
    srand(7);
    for (auto& v : h) v = data == 0 ? (float)rand() / (float)RAND_MAX * 2.f - 1.f : (data == 1 ? 0.f : 1.0f);
    CHECK(hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    for (int waves = 4; waves <= 8; waves += 4) {  // 4 = one wave per SIMD, 8 = two
      auto launch = [&] {
        if (waves == 4) hipLaunchKernelGGL(mfma_loop<4>, dim3(cus), dim3(256), 0, 0, in, out, iters, ticks);
        else hipLaunchKernelGGL(mfma_loop<8>, dim3(cus), dim3(512), 0, 0, in, out, iters, ticks);
      };
      for (int i = 0; i < 20; ++i) launch();  // clock ramp
      float best = 1e30f;
      long long tk = 0;
      for (int rep = 0; rep < 5; ++rep) {
        CHECK(hipEventRecord(e0, 0));
        for (int i = 0; i < 10; ++i) launch();
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms / 10 < best) {
          best = ms / 10;
          CHECK(hipMemcpy(&tk, ticks, 8, hipMemcpyDeviceToHost));
        }
      }
      const double flops = (double)cus * waves * iters * 32.0 * (32 * 32 * 2 * 2);
      // s_memtime / readcyclecounter on gfx9 counts at a fixed 100 MHz reference on some parts; print both readings
      printf("%-8s %d waves/SIMD: %8.1f us  %7.2f TFLOP/s   wave ticks %lld (%.3f ticks/ns)  ideal cycles/wave %d\n",
             data == 0 ? "random" : (data == 1 ? "zeros" : "ones"), waves / 4, best * 1e3, flops / best / 1e9, tk,
             tk / (best * 1e6), iters * 32 * 64 * (waves / 4));
    }
  }
  return 0;
}
