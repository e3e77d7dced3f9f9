// Tall-and-skinny float32 products on the matrix cores (gfx950): see the comment at the kernel.
#pragma once
#include <hip/hip_runtime.h>

namespace eg_skinny {
// ---- tall-and-skinny products ------------------------------------------------------------------
// C[M, N] (+)= A[M, K] * B[K, N] (+ bias) with N <= 16: the last layer of a classifier (cfg 5: 65536 x 10 x 512 — every
// sample's 512 activations against a 512 x 10 weight matrix).  The work is reading A once; a matrix-core tile of
// 128 x 32 with a two-stage K loop keeps only 2 x 8 KiB per block in flight (512 blocks: 3.7 TB/s, 36 us).  Here a
// wave owns 16 rows, B sits in LDS for the whole block, and the wave streams its rows with eight 1 KiB loads in flight
// at any time (two register sets of WINDOWS x float4; the library uses 4: 32.5 us against 35 with 8): v_mfma_f32_16x16x4_f32 on 16-float windows of k,
//   lane (r = l % 16, g = l / 16) loads A[row0 + r][16 s + 4 g .. + 3]  (16 rows x 64 contiguous bytes per instruction)
//   MFMA j of window s multiplies k = 16 s + 4 g + j on both operands (any assignment is valid if A and B agree);
//   B is stored in LDS as [k / 4][16 columns][k % 4]: one conflict-free ds_read_b128 per window.
// Fixed summation order (windows in sequence, the four k-groups of a window inside one MFMA): run-to-run identical.
typedef float f32x4v __attribute__((ext_vector_type(4)));

template <int WINDOWS>  // windows (16 k each) per register set
__global__ __launch_bounds__(256) void gemm_skinny_nn_kernel(const float* __restrict__ A, const float* __restrict__ B, float* C,
                                                             const float* __restrict__ bias, long M, int N, int K, long lda,
                                                             long ldb, long ldc, int accumulate) {
  extern __shared__ __attribute__((aligned(16))) float bs[];  // [K / 4][16][4]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int e = tid; e < K * 16; e += 256) {
    const int k = e >> 4, c = e & 15;
    bs[(((k >> 2) * 16 + c) << 2) + (k & 3)] = c < N ? B[(long)k * ldb + c] : 0.f;
  }
  __syncthreads();
  const int r = lane & 15, g = lane >> 4;
  const int nwin = K / 16;
  const long groups = (M + 15) / 16;
  for (long grp = (long)blockIdx.x * 4 + wave; grp < groups; grp += (long)gridDim.x * 4) {
    const long row0 = grp * 16;
    long row = row0 + r;
    if (row > M - 1) row = M - 1;  // a ragged last group re-reads the last row; its outputs are not stored
    const float* a = A + row * lda + 4 * g;
    f32x4v acc = {0.f, 0.f, 0.f, 0.f};
    f32x4v cur[WINDOWS], nxt[WINDOWS];
#pragma unroll
    for (int w = 0; w < WINDOWS; ++w) cur[w] = w < nwin ? *reinterpret_cast<const f32x4v*>(a + 16 * w) : f32x4v{0.f, 0.f, 0.f, 0.f};
    for (int s0 = 0; s0 < nwin; s0 += WINDOWS) {
#pragma unroll
      for (int w = 0; w < WINDOWS; ++w) {
        const int s = s0 + WINDOWS + w;
        nxt[w] = s < nwin ? *reinterpret_cast<const f32x4v*>(a + 16 * s) : f32x4v{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int w = 0; w < WINDOWS; ++w) {
        const int s = s0 + w;
        if (s >= nwin) break;
        const f32x4v b = *reinterpret_cast<const f32x4v*>(bs + (((4 * s + g) * 16 + r) << 2));
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[w][j], b[j], acc, 0, 0, 0);
      }
#pragma unroll
      for (int w = 0; w < WINDOWS; ++w) cur[w] = nxt[w];
    }
    // 16 x 16 result: lane l holds column l % 16 of rows 4 (l / 16) + v
    if (r < N) {
      const float bv = bias ? bias[r] : 0.f;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const long m = row0 + 4 * g + v;
        if (m >= M) continue;
        float* c = C + m * ldc + r;
        float out = acc[v];
        if (accumulate) out = *c + out;
        *c = out + bv;
      }
    }
  }
}

}  // namespace eg_skinny
