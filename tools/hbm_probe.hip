// Variants of the bandwidth-bound library kernels (kernels/elementwise.hip, reduce.hip) measured against each other on
// working sets that do not fit the 256 MB Infinity Cache (four operand sets in rotation): which access pattern reaches
// the float4-copy ceiling of the guide (6.29 TB/s).   hipcc -O3 --offload-arch=gfx950 tools/hbm_probe.hip -o hbm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ float relu(float x) { return x >= 0.f ? x : 0.f; }
__device__ __forceinline__ float sig_bwd(float x, float g) {
  const float e = expf(-x), s = 1.f + e;
  return -(((-1.f) * (g / (s * s))) * e);
}

// ---- map: 1 in, 1 out --------------------------------------------------------------------------------------
template <int U, bool NT_>
__global__ __launch_bounds__(256) void map_k(const f4* __restrict__ in, f4* __restrict__ out, long n4) {
  const long stride = (long)gridDim.x * 256;
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  for (; i + (U - 1) * stride < n4; i += U * stride) {
    f4 x[U];
#pragma unroll
    for (int u = 0; u < U; ++u) x[u] = NT_ ? __builtin_nontemporal_load(in + i + u * stride) : in[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      f4 y;
#pragma unroll
      for (int j = 0; j < 4; ++j) y[j] = relu(x[u][j]);
      if (NT_) __builtin_nontemporal_store(y, out + i + u * stride); else out[i + u * stride] = y;
    }
  }
  for (; i < n4; i += stride) {
    f4 x = in[i], y;
#pragma unroll
    for (int j = 0; j < 4; ++j) y[j] = relu(x[j]);
    out[i] = y;
  }
}

// contiguous chunk per block instead of grid stride
template <int U, bool NT_>
__global__ __launch_bounds__(256) void map_chunk_k(const f4* __restrict__ in, f4* __restrict__ out, long n4) {
  // block b owns [b * per, (b + 1) * per), per a multiple of 256 * U
  const long per = ((n4 + gridDim.x - 1) / gridDim.x + 256 * U - 1) / (256 * U) * (256 * U);
  const long lo = (long)blockIdx.x * per, hi = min(n4, lo + per);
  for (long i = lo + threadIdx.x; i < hi; i += 256 * U) {
    f4 x[U];
#pragma unroll
    for (int u = 0; u < U; ++u) if (i + u * 256 < hi) x[u] = NT_ ? __builtin_nontemporal_load(in + i + u * 256) : in[i + u * 256];
#pragma unroll
    for (int u = 0; u < U; ++u) if (i + u * 256 < hi) {
      f4 y;
#pragma unroll
      for (int j = 0; j < 4; ++j) y[j] = relu(x[u][j]);
      if (NT_) __builtin_nontemporal_store(y, out + i + u * 256); else out[i + u * 256] = y;
    }
  }
}

// ---- map_grad: 2 in, 1 out ---------------------------------------------------------------------------------
template <int U, bool NT_, bool SIG>
__global__ __launch_bounds__(256) void mapg_k(const f4* __restrict__ in, const f4* __restrict__ g, f4* __restrict__ out, long n4) {
  const long stride = (long)gridDim.x * 256;
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  for (; i + (U - 1) * stride < n4; i += U * stride) {
    f4 x[U], gg[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      x[u] = NT_ ? __builtin_nontemporal_load(in + i + u * stride) : in[i + u * stride];
      gg[u] = NT_ ? __builtin_nontemporal_load(g + i + u * stride) : g[i + u * stride];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      f4 y;
#pragma unroll
      for (int j = 0; j < 4; ++j) y[j] = SIG ? sig_bwd(x[u][j], gg[u][j]) : (x[u][j] >= 0.f ? gg[u][j] : 0.f);
      if (NT_) __builtin_nontemporal_store(y, out + i + u * stride); else out[i + u * stride] = y;
    }
  }
  for (; i < n4; i += stride) {
    f4 x = in[i], gv = g[i], y;
#pragma unroll
    for (int j = 0; j < 4; ++j) y[j] = SIG ? sig_bwd(x[j], gv[j]) : (x[j] >= 0.f ? gv[j] : 0.f);
    out[i] = y;
  }
}

template <int U, bool NT_, bool SIG>
__global__ __launch_bounds__(256) void mapg_chunk_k(const f4* __restrict__ in, const f4* __restrict__ g, f4* __restrict__ out, long n4) {
  const long per = ((n4 + gridDim.x - 1) / gridDim.x + 256 * U - 1) / (256 * U) * (256 * U);
  const long lo = (long)blockIdx.x * per, hi = min(n4, lo + per);
  for (long i = lo + threadIdx.x; i < hi; i += 256 * U) {
    f4 x[U], gg[U];
#pragma unroll
    for (int u = 0; u < U; ++u) if (i + u * 256 < hi) {
      x[u] = NT_ ? __builtin_nontemporal_load(in + i + u * 256) : in[i + u * 256];
      gg[u] = NT_ ? __builtin_nontemporal_load(g + i + u * 256) : g[i + u * 256];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) if (i + u * 256 < hi) {
      f4 y;
#pragma unroll
      for (int j = 0; j < 4; ++j) y[j] = SIG ? sig_bwd(x[u][j], gg[u][j]) : (x[u][j] >= 0.f ? gg[u][j] : 0.f);
      if (NT_) __builtin_nontemporal_store(y, out + i + u * 256); else out[i + u * 256] = y;
    }
  }
}

// ---- the activation-gradient product of cfg 5: gh[y, j] = bit(y, j) ? sum_c gz[y, c] * W2[j, c] : 0  (65 536 x 512 x 10) ----
// thread = 4 consecutive columns of a row, its 4 x 10 values of W2 in registers; 128 threads per row, 2 rows per block-step.
// STRIDE: rows of a step gridDim * 2 apart (the round-4 experiment); else a block owns a contiguous run of rows.
template <int U, bool CHUNK, bool NT_>
__global__ __launch_bounds__(256) void gh_k(const float* __restrict__ gz, const float* __restrict__ w2, const unsigned* __restrict__ bits,
                                            float* __restrict__ out, long M) {
  constexpr int K = 10, N = 512, TPR = N / 4, RPB = 256 / TPR;
  const int tid = threadIdx.x, c4 = tid % TPR;
  const long n = (long)c4 * 4;
  float w[4][K];
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int k = 0; k < K; ++k) w[e][k] = w2[(n + e) * K + k];
  long lo, hi, step;
  if (CHUNK) {
    const long per = ((M + gridDim.x - 1) / gridDim.x + RPB * U - 1) / (RPB * U) * (RPB * U);
    lo = (long)blockIdx.x * per; hi = min(M, lo + per); step = RPB;
  } else {
    lo = (long)blockIdx.x * RPB; hi = M; step = (long)gridDim.x * RPB;
  }
  for (long m0 = lo + tid / TPR; m0 < hi; m0 += step * U) {
    float ar[U][K];
    unsigned bw[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      long m = m0 + u * step;
      if (m > hi - 1) m = hi - 1;
      const long mu = ((long)__builtin_amdgcn_readfirstlane((int)(m >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)m);
#pragma unroll
      for (int k = 0; k < K; ++k) ar[u][k] = gz[mu * K + k];
      bw[u] = bits[(m * N + n) >> 5] >> ((m * N + n) & 31);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long m = m0 + u * step;
      if (m >= hi) break;
      f4 res;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < K; ++k) acc = __builtin_fmaf(ar[u][k], w[e][k], acc);
        res[e] = ((bw[u] >> e) & 1u) ? acc : 0.f;
      }
      if (NT_) __builtin_nontemporal_store(res, reinterpret_cast<f4*>(out + m * N + n)); else *reinterpret_cast<f4*>(out + m * N + n) = res;
    }
  }
}

// ---- full sum ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
template <int U, bool NT_>
__global__ __launch_bounds__(256) void sum_k(const f4* __restrict__ in, float* __restrict__ partial, long n4) {
  __shared__ float red[4];
  const long stride = (long)gridDim.x * 256;
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  f4 acc[U];
#pragma unroll
  for (int u = 0; u < U; ++u) acc[u] = f4{0.f, 0.f, 0.f, 0.f};
  for (; i + (U - 1) * stride < n4; i += U * stride) {
    f4 x[U];
#pragma unroll
    for (int u = 0; u < U; ++u) x[u] = NT_ ? __builtin_nontemporal_load(in + i + u * stride) : in[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) acc[u] += x[u];
  }
  for (; i < n4; i += stride) acc[0] += in[i];
  f4 t = acc[0];
#pragma unroll
  for (int u = 1; u < U; ++u) t += acc[u];
  float a = wave_sum((t[0] + t[1]) + (t[2] + t[3]));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
}

// ---- column sum, cols % 4 == 0: a thread owns one float4 column group, row phases over the rest of the block --------
template <int U, bool NT_>
__global__ __launch_bounds__(256) void colsum_k(const float* __restrict__ in, float* __restrict__ partial, long rows, long cols,
                                                long rows_per_block) {
  __shared__ f4 red[256];
  const int cg = (int)min(cols / 4, 256L);   // column groups per block (blockIdx.y walks further ones)
  const int phases = 256 / cg;
  const int gi = threadIdx.x % cg, ph = threadIdx.x / cg;
  const long c4 = (long)blockIdx.y * cg + gi;
  const long r0 = (long)blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  f4 acc[U];
#pragma unroll
  for (int u = 0; u < U; ++u) acc[u] = f4{0.f, 0.f, 0.f, 0.f};
  if (ph < phases && c4 * 4 < cols) {
    const f4* p = reinterpret_cast<const f4*>(in) + c4;
    const long ld4 = cols / 4;
    long r = r0 + ph;
    for (; r + (long)(U - 1) * phases < r1; r += (long)U * phases) {
      f4 x[U];
#pragma unroll
      for (int u = 0; u < U; ++u) x[u] = NT_ ? __builtin_nontemporal_load(p + (r + (long)u * phases) * ld4) : p[(r + (long)u * phases) * ld4];
#pragma unroll
      for (int u = 0; u < U; ++u) acc[u] += x[u];
    }
    for (; r < r1; r += phases) acc[0] += p[r * ld4];
  }
  f4 t = acc[0];
#pragma unroll
  for (int u = 1; u < U; ++u) t += acc[u];
  red[threadIdx.x] = t;
  __syncthreads();
  if (ph == 0 && c4 * 4 < cols) {
    f4 s = red[gi];
    for (int q = 1; q < phases; ++q) s += red[q * cg + gi];
    reinterpret_cast<f4*>(partial + (long)blockIdx.x * cols)[c4] = s;
  }
}

// ---- row sum, cols % 4 == 0: a wave takes R rows per trip --------------------------------------------------------
template <int R, bool NT_>
__global__ __launch_bounds__(256) void rowsum_k(const float* __restrict__ in, float* __restrict__ out, long rows, long cols) {
  const int lane = threadIdx.x & 63;
  const long wave = ((long)blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = ((long)gridDim.x * 256) >> 6;
  const long ld4 = cols / 4;
  for (long r = wave * R; r < rows; r += nwaves * R) {
    float s[R];
#pragma unroll
    for (int q = 0; q < R; ++q) {
      f4 a = {0.f, 0.f, 0.f, 0.f};
      if (r + q < rows)
        for (long c = lane; c < ld4; c += 64) {
          const f4* p = reinterpret_cast<const f4*>(in) + (r + q) * ld4 + c;
          a += NT_ ? __builtin_nontemporal_load(p) : *p;
        }
      s[q] = (a[0] + a[1]) + (a[2] + a[3]);
    }
#pragma unroll
    for (int q = 0; q < R; ++q) s[q] = wave_sum(s[q]);
    if (lane == 0)
#pragma unroll
      for (int q = 0; q < R; ++q) if (r + q < rows) out[r + q] = s[q];
  }
}

int main() {
  const long rows = 65536, cols = 512, n = rows * cols, n4 = n / 4;
  const int SETS = 4;
  float *x[SETS], *g[SETS], *y[SETS], *part;
  for (int s = 0; s < SETS; ++s) {
    CK(hipMalloc(&x[s], n * 4)); CK(hipMalloc(&g[s], n * 4)); CK(hipMalloc(&y[s], n * 4));
    CK(hipMemset(x[s], 0, n * 4)); CK(hipMemset(g[s], 0, n * 4));
  }
  CK(hipMalloc(&part, 64 << 20));
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time = [&](const char* name, double bytes, auto launch) {
    for (int i = 0; i < 40; ++i) launch(i % SETS);
    CK(hipStreamSynchronize(st));
    const int K = 40;
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < K; ++i) launch(i % SETS);
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / K;
    printf("%-44s %8.2f us  %6.3f TB/s  %.3f of 8\n", name, us, bytes / us * 1e-6, bytes / us * 1e-6 / 8.0);
    fflush(stdout);
  };
#define MAPV(U, NTV, G) { char nm[64]; snprintf(nm, 64, "map relu stride U=%d nt=%d grid=%d/CU", U, NTV, G); \
    time(nm, 2.0 * n * 4, [&](int s) { hipLaunchKernelGGL((map_k<U, NTV>), dim3(256 * G), dim3(256), 0, st, (const f4*)x[s], (f4*)y[s], n4); }); }
  if (getenv("PROBE_GH")) {
    float *gz, *w2; unsigned* bits;
    CK(hipMalloc(&gz, rows * 10 * 4)); CK(hipMalloc(&w2, 512 * 10 * 4)); CK(hipMalloc(&bits, n / 8));
    CK(hipMemset(gz, 0, rows * 10 * 4)); CK(hipMemset(w2, 0, 512 * 10 * 4)); CK(hipMemset(bits, 0x5a, n / 8));
#define GHV(U, CH, NTV, G) { char nm[64]; snprintf(nm, 64, "gh %s U=%d nt=%d grid=%d/CU", CH ? "chunk " : "stride", U, NTV, G); \
    time(nm, 1.0 * n * 4 + rows * 40.0 + n / 8.0, [&](int s) { hipLaunchKernelGGL((gh_k<U, CH, NTV>), dim3(256 * G), dim3(256), 0, st, gz, w2, bits, y[s], rows); }); }
    GHV(4, false, true, 8) GHV(4, false, false, 8) GHV(4, true, true, 8) GHV(4, true, true, 16) GHV(4, true, true, 32) GHV(8, true, true, 8) GHV(8, true, true, 16)
    GHV(2, true, true, 16) GHV(2, true, true, 32) GHV(8, true, true, 4) GHV(4, true, false, 16)
  } else if (getenv("PROBE_ROUND1")) {
  MAPV(1, false, 8) MAPV(1, true, 8) MAPV(2, false, 8) MAPV(4, false, 8) MAPV(4, true, 8) MAPV(2, false, 16) MAPV(4, false, 4) MAPV(8, false, 4)
  MAPV(1, false, 16) MAPV(1, false, 32) MAPV(2, true, 16)
#define MAPC(U, NTV, G) { char nm[64]; snprintf(nm, 64, "map relu chunk  U=%d nt=%d grid=%d/CU", U, NTV, G); \
    time(nm, 2.0 * n * 4, [&](int s) { hipLaunchKernelGGL((map_chunk_k<U, NTV>), dim3(256 * G), dim3(256), 0, st, (const f4*)x[s], (f4*)y[s], n4); }); }
  MAPC(4, false, 8) MAPC(4, true, 8) MAPC(4, false, 16) MAPC(8, false, 8)
#define MAPG(U, NTV, SIGV, G) { char nm[64]; snprintf(nm, 64, "map_grad %s U=%d nt=%d grid=%d/CU", SIGV ? "sigmoid" : "relu", U, NTV, G); \
    time(nm, 3.0 * n * 4, [&](int s) { hipLaunchKernelGGL((mapg_k<U, NTV, SIGV>), dim3(256 * G), dim3(256), 0, st, (const f4*)x[s], (const f4*)g[s], (f4*)y[s], n4); }); }
  MAPG(1, false, false, 8) MAPG(2, false, false, 8) MAPG(4, false, false, 8) MAPG(2, true, false, 8) MAPG(2, false, false, 16) MAPG(1, false, false, 16)
  MAPG(1, false, true, 8) MAPG(2, false, true, 8) MAPG(2, true, true, 8) MAPG(2, false, true, 16)
#define SUMV(U, NTV, G) { char nm[64]; snprintf(nm, 64, "sum U=%d nt=%d grid=%d/CU", U, NTV, G); \
    time(nm, 1.0 * n * 4, [&](int s) { hipLaunchKernelGGL((sum_k<U, NTV>), dim3(256 * G), dim3(256), 0, st, (const f4*)x[s], part, n4); }); }
  SUMV(1, false, 4) SUMV(2, false, 8) SUMV(4, false, 8) SUMV(4, true, 8) SUMV(8, false, 8) SUMV(4, false, 16) SUMV(8, false, 4)
#define COLV(U, NTV, NB) { char nm[64]; snprintf(nm, 64, "colsum U=%d nt=%d blocks=%d", U, NTV, NB); const long rpb = (rows + NB - 1) / NB; \
    time(nm, 1.0 * n * 4, [&](int s) { hipLaunchKernelGGL((colsum_k<U, NTV>), dim3(NB, (unsigned)((cols / 4 + 255) / 256)), dim3(256), 0, st, x[s], part, rows, cols, rpb); }); }
  COLV(4, false, 1024) COLV(8, false, 1024) COLV(8, true, 1024) COLV(8, false, 2048) COLV(16, false, 1024) COLV(4, false, 2048) COLV(8, false, 512)
#define ROWV(R, NTV, G) { char nm[64]; snprintf(nm, 64, "rowsum R=%d nt=%d grid=%d/CU", R, NTV, G); \
    time(nm, 1.0 * n * 4, [&](int s) { hipLaunchKernelGGL((rowsum_k<R, NTV>), dim3(256 * G), dim3(256), 0, st, x[s], y[s], rows, cols); }); }
  ROWV(1, false, 8) ROWV(2, false, 8) ROWV(4, false, 8) ROWV(4, true, 8) ROWV(8, false, 8) ROWV(4, false, 16)
  } else {
#define MAPGC(U, NTV, SIGV, G) { char nm[64]; snprintf(nm, 64, "map_grad chunk %s U=%d nt=%d grid=%d/CU", SIGV ? "sigmoid" : "relu", U, NTV, G); \
    time(nm, 3.0 * n * 4, [&](int s) { hipLaunchKernelGGL((mapg_chunk_k<U, NTV, SIGV>), dim3(256 * G), dim3(256), 0, st, (const f4*)x[s], (const f4*)g[s], (f4*)y[s], n4); }); }
  MAPGC(1, true, false, 8) MAPGC(1, true, false, 16) MAPGC(1, true, false, 32) MAPGC(2, true, false, 8) MAPGC(2, true, false, 16) MAPGC(2, true, false, 32) MAPGC(4, true, false, 8) MAPGC(4, true, false, 16) MAPGC(4, true, false, 32)
  MAPGC(2, true, true, 16) MAPGC(2, true, true, 32) MAPGC(4, true, true, 16)
  MAPC(4, true, 32) MAPC(4, true, 64) MAPC(2, true, 32) MAPC(2, true, 64)
  SUMV(1, true, 8) SUMV(1, true, 16) COLV(4, true, 2048) COLV(2, true, 2048) COLV(4, true, 4096) ROWV(1, true, 8) ROWV(1, true, 16)
  }
  // a pure copy for reference (runtime)
  time("hipMemcpyAsync d2d (read + write)", 2.0 * n * 4, [&](int s) { CK(hipMemcpyAsync(y[s], x[s], n * 4, hipMemcpyDeviceToDevice, st)); });
  return 0;
}
