// Prototype (tuning harness, not part of the library): the two 10-wide products of the dense net's backward pass in ONE
// streaming launch — the activation-gradient product with its mask,  gh[y, j] = bit(y, j) ? sum_c gz[y, c] * W2[j, c] : 0
// (dense, dnn.nim:19-24, differentiated: passes.nim:519-549, with relu's select, dnn.nim:26-27), and the weight gradient
// gW2[j, c] = sum_y a[y, j] * gz[y, c] — which today are a 38.6 us launch that writes 134 MB and a 33 + 6 us pair of
// launches that reads 134 MB.  One pass: read a (134 MB) + bits (4 MB), write gh (134 MB); VALU only (80 FMAs per 32 bytes
// moved: ~17 us of issue on the whole chip, under the memory time).
// Build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/skinny_pair.hip -o tools/bin/skinny_pair
// Run:    tools/bin/skinny_pair [rows = 65536]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                        \
  do {                                                                  \
    hipError_t e = (x);                                                 \
    if (e != hipSuccess) {                                              \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));            \
      exit(1);                                                          \
    }                                                                   \
  } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int H = 512, O = 10;

template <int U>
__global__ __launch_bounds__(256) void pair_kernel(const float* __restrict__ a, const float* __restrict__ gz,
                                                   const float* __restrict__ w2, const unsigned* __restrict__ bits,
                                                   float* __restrict__ gh, float* __restrict__ partial, int M) {
  __shared__ float fold[128 * 4 * O];
  const int t = threadIdx.x, col4 = (t & 127) * 4;
  const int rowpar = __builtin_amdgcn_readfirstlane(t >> 7);
  float w[4][O], acc[4][O];
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int c = 0; c < O; ++c) {
      w[e][c] = w2[(col4 + e) * O + c];
      acc[e][c] = 0.f;
    }
  const int stride = gridDim.x * 2;
  for (int row0 = blockIdx.x * 2 + rowpar; row0 < M; row0 += stride * U) {
    f32x4 av[U];
    unsigned word[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int row = row0 + u * stride;
      if (row < M) {
        av[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a + (long)row * H + col4));
        word[u] = bits[((long)row * H + col4) >> 5];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int row = __builtin_amdgcn_readfirstlane(row0 + u * stride);
      if (row >= M) break;
      float g[O];
#pragma unroll
      for (int c = 0; c < O; ++c) g[c] = gz[(long)row * O + c];   // wave-uniform: scalar loads
      f32x4 out;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = 0.f;
#pragma unroll
        for (int c = 0; c < O; ++c) v = fmaf(g[c], w[e][c], v);
        out[e] = (word[u] >> ((col4 + e) & 31) & 1u) ? v : 0.f;
#pragma unroll
        for (int c = 0; c < O; ++c) acc[e][c] = fmaf(av[u][e], g[c], acc[e][c]);
      }
      __builtin_nontemporal_store(out, reinterpret_cast<f32x4*>(gh + (long)row * H + col4));
    }
  }
  // the two row parities of a block meet in LDS (fixed order: parity 0 + parity 1), one slab per block
  if (rowpar == 1) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int c = 0; c < O; ++c) fold[((t & 127) * 4 + e) * O + c] = acc[e][c];
  }
  __syncthreads();
  if (rowpar == 0) {
    float* slab = partial + (long)blockIdx.x * H * O;
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int c = 0; c < O; ++c) slab[(col4 + e) * O + c] = acc[e][c] + fold[((t & 127) * 4 + e) * O + c];
  }
}

__global__ __launch_bounds__(256) void slab_sum(const float* __restrict__ partial, float* __restrict__ out, int count, int slabs) {
  // 16 threads per output element walk the slabs in 16 interleaved chains, folded by a fixed butterfly
  const int idx = (blockIdx.x * 256 + threadIdx.x) >> 4, lane = threadIdx.x & 15;
  float s = 0.f;
  if (idx < count)
    for (int z = lane; z < slabs; z += 16) s += partial[(long)z * count + idx];
  for (int d = 8; d >= 1; d >>= 1) s += __shfl_xor(s, d, 16);
  if (idx < count && lane == 0) out[idx] = s;
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 65536;
  std::vector<float> ha((size_t)M * H), hgz((size_t)M * O), hw(H * O);
  std::vector<unsigned> hb((size_t)M * H / 32);
  srand(3);
  for (auto& v : ha) v = (float)rand() / RAND_MAX;
  for (auto& v : hgz) v = ((float)rand() / RAND_MAX - 0.5f) * 1e-4f;
  for (auto& v : hw) v = (float)rand() / RAND_MAX * 0.2f - 0.1f;
  for (auto& v : hb) v = (unsigned)rand() * 2654435761u;
  float *a, *gz, *w2, *gh, *partial, *gw;
  unsigned* bits;
  CHECK(hipMalloc(&a, ha.size() * 4));
  CHECK(hipMalloc(&gz, hgz.size() * 4));
  CHECK(hipMalloc(&w2, hw.size() * 4));
  CHECK(hipMalloc(&bits, hb.size() * 4));
  CHECK(hipMalloc(&gh, ha.size() * 4));
  CHECK(hipMalloc(&partial, (size_t)4096 * H * O * 4));
  CHECK(hipMalloc(&gw, H * O * 4));
  CHECK(hipMemcpy(a, ha.data(), ha.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(gz, hgz.data(), hgz.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(w2, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(bits, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
  hipStream_t s;
  CHECK(hipStreamCreate(&s));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int grid : {256, 512, 1024, 2048})
    for (int unroll : {2, 4, 8}) {
      auto run = [&] {
        if (unroll == 2) hipLaunchKernelGGL(pair_kernel<2>, dim3(grid), dim3(256), 0, s, a, gz, w2, bits, gh, partial, M);
        else if (unroll == 4) hipLaunchKernelGGL(pair_kernel<4>, dim3(grid), dim3(256), 0, s, a, gz, w2, bits, gh, partial, M);
        else hipLaunchKernelGGL(pair_kernel<8>, dim3(grid), dim3(256), 0, s, a, gz, w2, bits, gh, partial, M);
        hipLaunchKernelGGL(slab_sum, dim3((H * O * 16 + 255) / 256), dim3(256), 0, s, partial, gw, H * O, grid);
      };
      for (int i = 0; i < 3; ++i) run();
      float best = 1e9f;
      for (int r = 0; r < 5; ++r) {
        CHECK(hipEventRecord(e0, s));
        for (int i = 0; i < 10; ++i) run();
        CHECK(hipEventRecord(e1, s));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms / 10);
      }
      printf("grid %4d unroll %d: %6.1f us for both launches (%.2f TB/s over %.0f MB)\n", grid, unroll, best * 1e3,
             (2.0 * M * H * 4 + M * H / 8.0) / best / 1e9, (2.0 * M * H * 4 + M * H / 8.0) / 1e6);
    }
  // check against a float64 restatement on a sample
  std::vector<float> ogh((size_t)M * H), ogw(H * O);
  CHECK(hipMemcpy(ogh.data(), gh, ogh.size() * 4, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(ogw.data(), gw, ogw.size() * 4, hipMemcpyDeviceToHost));
  double worst_gh = 0, worst_gw = 0, scale = 0;
  for (int k = 0; k < 2000; ++k) {
    const long row = rand() % M, j = rand() % H;
    double v = 0;
    for (int c = 0; c < O; ++c) v += (double)hgz[row * O + c] * hw[j * O + c];
    const long idx = row * H + j;
    if (!(hb[idx >> 5] >> (idx & 31) & 1u)) v = 0;
    worst_gh = std::max(worst_gh, std::abs(v - ogh[idx]));
    scale = std::max(scale, std::abs(v));
  }
  double gscale = 0;
  for (int j = 0; j < H; j += 37)
    for (int c = 0; c < O; ++c) {
      double v = 0;
      for (long row = 0; row < M; ++row) v += (double)ha[row * H + j] * hgz[row * O + c];
      worst_gw = std::max(worst_gw, std::abs(v - ogw[j * O + c]));
      gscale = std::max(gscale, std::abs(v));
    }
  printf("gh: max abs err %.2e of %.2e   gW2: max abs err %.2e of %.2e\n", worst_gh, scale, worst_gw, gscale);
  return 0;
}
