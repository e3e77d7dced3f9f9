// float64 contraction on the matrix cores — the `compile[float64]` form of `c[y,x] ++= a[y,it] * b[it,x]`
// (base.nim:27-28; model.nim:253-260 instantiates every kernel of a program over Scalar64) and of the two
// gradient contractions derive makes of it (passes.nim:519-549).
//
// `v_mfma_f64_16x16x4_f64`: A fragment = one double per lane (row lane % 16, k lane / 16), B likewise
// (k lane / 16, column lane % 16), C / D four doubles per lane (column lane & 15, row (lane >> 4) + 4 r).
// One such instruction is 2 048 FLOP in 64 cycles (78.6 TFLOP/s over 1 024 SIMDs at 2.4 GHz): a wave needs
// 16 bytes of operand per lane every 64 cycles, so — unlike the float32 kernel, whose 32 x 32 x 2 tiles are
// bound by what stands between two MFMAs — a plain register-staged, double-buffered tile loop keeps the
// matrix pipe busy: 4 waves per block, a wave owns WM x WN of the BM x BN tile, 16-deep k-tiles,
// operands in LDS as [k][m | n] rows padded by 16 doubles (the four k rows a fragment read touches then
// fall into different bank halves).  All four storage orders go through one loader (strides), ragged
// tiles are zero filled, K is cut into slices with a fixed-order second pass when the tiles alone
// cannot fill the chip (deterministic: no atomics).
#include "../eg_internal.hpp"

namespace {

typedef double d4 __attribute__((ext_vector_type(4)));

constexpr int BK = 16;
constexpr int PAD = 16;

struct DgemmArgs {
  const double* A;
  const double* B;
  double* C;        // destination, or the slab block when splits > 1
  const double* bias;
  long M, N, K;
  long a_sm, a_sk, b_sk, b_sn, ldc;
  int accumulate;
  int splits;       // k-slices (grid.y)
  long k_per_split; // multiple of BK
  int tiles_m, tiles_n;
  int remap;        // tiles % 8 == 0: contiguous tile ranges per XCD
};

// Tile loader: ELEMS doubles per thread of a [BK][BMN] tile whose (mn, k) element lives at base[mn * s_mn + k * s_k].
// k-contiguous operands walk k with the lanes (16 lanes = one 128-byte row piece), mn-contiguous ones walk mn.
template <int BMN>
struct TileLoader {
  static constexpr int ELEMS = BMN * BK / 256;
  double v[ELEMS];
  __device__ __forceinline__ void load(const double* __restrict__ base, long s_mn, long s_k, long mn0, long k0, long MN, long Kend,
                                       bool kc, int tid) {
#pragma unroll
    for (int j = 0; j < ELEMS; ++j) {
      int mn, k;
      if (kc) {
        k = tid & 15;
        mn = (tid >> 4) + 16 * j;
      } else {
        mn = tid % BMN;
        k = tid / BMN + (256 / BMN) * j;
      }
      const long gm = mn0 + mn, gk = k0 + k;
      v[j] = (gm < MN && gk < Kend) ? base[gm * s_mn + gk * s_k] : 0.0;
    }
  }
  __device__ __forceinline__ void store(double* lds, bool kc, int tid) const {
#pragma unroll
    for (int j = 0; j < ELEMS; ++j) {
      int mn, k;
      if (kc) {
        k = tid & 15;
        mn = (tid >> 4) + 16 * j;
      } else {
        mn = tid % BMN;
        k = tid / BMN + (256 / BMN) * j;
      }
      lds[k * (BMN + PAD) + mn] = v[j];
    }
  }
};

template <int BM, int BN>
__global__ __launch_bounds__(256) void dgemm_kernel(DgemmArgs a) {
  constexpr int WM = BM / 2, WN = BN / 2;  // 2 x 2 waves
  constexpr int FM = WM / 16, FN = WN / 16;
  constexpr int LDA = BM + PAD, LDB = BN + PAD;
  extern __shared__ double lds[];
  double* As = lds;                      // [2][BK][LDA]
  double* Bs = lds + 2 * BK * LDA;       // [2][BK][LDB]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = (wave >> 1) * WM, wn = (wave & 1) * WN;
  long tile = blockIdx.x;
  if (a.remap) {
    const long per = (long)gridDim.x >> 3;
    tile = (tile & 7) * per + (tile >> 3);
  }
  const long tm = tile / a.tiles_n, tn = tile % a.tiles_n;
  const long m0 = tm * BM, n0 = tn * BN;
  const long kbeg = (long)blockIdx.y * a.k_per_split;
  const long kend = min(a.K, kbeg + a.k_per_split);
  const bool a_kc = a.a_sk == 1, b_kc = a.b_sk == 1;

  d4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = d4{0.0, 0.0, 0.0, 0.0};

  TileLoader<BM> la;
  TileLoader<BN> lb;
  const long ktiles = kend > kbeg ? (kend - kbeg + BK - 1) / BK : 0;
  if (ktiles > 0) {
    la.load(a.A, a.a_sm, a.a_sk, m0, kbeg, a.M, kend, a_kc, tid);
    lb.load(a.B, a.b_sn, a.b_sk, n0, kbeg, a.N, kend, b_kc, tid);
    la.store(As, a_kc, tid);
    lb.store(Bs, b_kc, tid);
  }
  __syncthreads();
  const int fr = lane & 15, fk = lane >> 4;
  for (long kt = 0; kt < ktiles; ++kt) {
    const int cur = (int)(kt & 1);
    const bool more = kt + 1 < ktiles;
    if (more) {
      la.load(a.A, a.a_sm, a.a_sk, m0, kbeg + (kt + 1) * BK, a.M, kend, a_kc, tid);
      lb.load(a.B, a.b_sn, a.b_sk, n0, kbeg + (kt + 1) * BK, a.N, kend, b_kc, tid);
    }
    const double* Ac = As + cur * BK * LDA;
    const double* Bc = Bs + cur * BK * LDB;
#pragma unroll
    for (int s = 0; s < BK / 4; ++s) {
      double af[FM], bf[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i) af[i] = Ac[(4 * s + fk) * LDA + wm + 16 * i + fr];
#pragma unroll
      for (int j = 0; j < FN; ++j) bf[j] = Bc[(4 * s + fk) * LDB + wn + 16 * j + fr];
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    if (more) {
      la.store(As + (cur ^ 1) * BK * LDA, a_kc, tid);
      lb.store(Bs + (cur ^ 1) * BK * LDB, b_kc, tid);
    }
    __syncthreads();
  }

  // C / D: column lane & 15, row (lane >> 4) + 4 r
  const bool slab = a.splits > 1;
  double* C = slab ? a.C + (long)blockIdx.y * a.M * a.N : a.C;
  const long ldc = slab ? a.N : a.ldc;
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const long col = n0 + wn + 16 * j + fr;
      if (col >= a.N) continue;
      const double b = (!slab && a.bias) ? a.bias[col] : 0.0;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long row = m0 + wm + 16 * i + fk + 4 * r;
        if (row >= a.M) continue;
        double v = acc[i][j][r];
        if (!slab) {
          if (a.bias) v = v + b;
          if (a.accumulate) v = C[row * ldc + col] + v;
        }
        C[row * ldc + col] = v;
      }
    }
}

// Second pass of a sliced product: out = (accumulate ? out : 0) + (slab 0 + slab 1 + ...) + bias, slabs in order.
__global__ __launch_bounds__(256) void dgemm_reduce_kernel(const double* __restrict__ slabs, double* __restrict__ C, const double* __restrict__ bias,
                                                           long M, long N, long ldc, int splits, int accumulate) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= M * N) return;
  const long row = i / N, col = i % N;
  double s = slabs[i];
  for (int z = 1; z < splits; ++z) s = s + slabs[(long)z * M * N + i];
  if (bias) s = s + bias[col];
  double* dst = C + row * ldc + col;
  *dst = accumulate ? *dst + s : s;
}

template <typename T>
__global__ __launch_bounds__(256) void fill_kernel_t(T* __restrict__ out, long n, T value) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) out[i] = value;
}

// ---- column sum (the float64 twin of reduce.hip's two-stage tree; same geometry, same order) -------------------------
__global__ __launch_bounds__(256) void colsum_partial_f64_kernel(const double* __restrict__ in, double* __restrict__ partial, long rows, long cols,
                                                                 int colsP, long rows_per_block) {
  __shared__ double red[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rpw = 64 / colsP;
  const int r_in_wave = lane / colsP;
  const long c = (long)blockIdx.y * 64 + (lane % colsP);
  const long row_begin = (long)blockIdx.x * rows_per_block;
  const long row_end = min(rows, row_begin + rows_per_block);
  double acc = 0.0;
  if (c < cols)
    for (long r = row_begin + wave * rpw + r_in_wave; r < row_end; r += 4 * rpw) acc += in[r * cols + c];
  for (int off = 32; off >= colsP; off >>= 1) acc += __shfl_xor(acc, off, 64);
  red[wave][lane] = acc;
  __syncthreads();
  if (wave == 0 && lane < colsP && c < cols) partial[(long)blockIdx.x * cols + c] = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
}

__global__ __launch_bounds__(256) void colsum_final_f64_kernel(const double* __restrict__ partial, double* __restrict__ out, long cols, int nparts,
                                                               int accumulate) {
  const long c = (long)blockIdx.x * 256 + threadIdx.x;
  if (c >= cols) return;
  double s = 0.0;
  for (int p = 0; p < nparts; ++p) s += partial[(long)p * cols + c];
  out[c] = accumulate ? out[c] + s : s;
}

// element i = lo + (hi - lo) * u, u in [0, 1) from 53 bits of the same counter hash the float32 fill uses
__device__ __forceinline__ uint64_t mix64(uint64_t x) {
  x ^= x >> 30;
  x *= 0xbf58476d1ce4e5b9ULL;
  x ^= x >> 27;
  x *= 0x94d049bb133111ebULL;
  x ^= x >> 31;
  return x;
}
__global__ __launch_bounds__(256) void fill_uniform_f64_kernel(double* __restrict__ out, long n, double lo, double hi, const uint64_t* __restrict__ state,
                                                               uint64_t stream) {
  const uint64_t seed = state[0], draw = state[1];
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const uint64_t h = mix64(mix64(seed ^ (draw * 0x9e3779b97f4a7c15ULL)) ^ mix64(stream * 0xd1b54a32d192ed03ULL + (uint64_t)i));
    const double u = (double)(h >> 11) * (1.0 / 9007199254740992.0);
    out[i] = lo + (hi - lo) * u;
  }
}

template <int BM, int BN>
int launch_dgemm(eg_ctx* ctx, DgemmArgs& a) {
  constexpr size_t lds = (size_t)2 * BK * ((BM + PAD) + (BN + PAD)) * sizeof(double);
  static bool attr_set = false;
  if (!attr_set) {
    EG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&dgemm_kernel<BM, BN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  a.tiles_m = (int)((a.M + BM - 1) / BM);
  a.tiles_n = (int)((a.N + BN - 1) / BN);
  const long tiles = (long)a.tiles_m * a.tiles_n;
  a.remap = tiles % 8 == 0 && tiles >= 16;
  hipLaunchKernelGGL((dgemm_kernel<BM, BN>), dim3((unsigned)tiles, (unsigned)a.splits), dim3(256), lds, ctx->stream, a);
  EG_HIP_CHECK(hipGetLastError());
  return EG_OK;
}

}  // namespace

namespace eg {

long colsum_f64_scratch_doubles(const eg_ctx* ctx, long rows, long cols) { return colsum_scratch_floats(ctx, rows, cols); }

int colsum_f64_with_scratch(eg_ctx* ctx, long rows, long cols, const double* in, double* out, int accumulate, double* scratch) {
  if (cols == 0) return EG_OK;
  int rc = set_device(ctx);
  if (rc) return rc;
  int colsP = 1;
  while (colsP < cols && colsP < 64) colsP <<= 1;
  // the geometry of reduce.hip's colsum_geometry (its scratch size is what the caller reserved)
  const long col_tiles = (cols + 63) / 64;
  long nparts = (4L * ctx->compute_units + col_tiles - 1) / col_tiles;
  const long max_parts = (rows + 63) / 64;
  if (nparts > max_parts) nparts = max_parts;
  if (nparts < 1) nparts = 1;
  const long rows_per_block = (rows + nparts - 1) / nparts;
  nparts = rows_per_block > 0 ? (rows + rows_per_block - 1) / rows_per_block : 1;
  if (nparts < 1) nparts = 1;
  hipLaunchKernelGGL(colsum_partial_f64_kernel, dim3((unsigned)nparts, (unsigned)col_tiles), dim3(256), 0, ctx->stream, in, scratch, rows, cols, colsP,
                     rows_per_block);
  hipLaunchKernelGGL(colsum_final_f64_kernel, dim3((unsigned)((cols + 255) / 256)), dim3(256), 0, ctx->stream, scratch, out, cols, (int)nparts,
                     accumulate);
  EG_HIP_CHECK(hipGetLastError());
  return EG_OK;
}

}  // namespace eg

extern "C" int eg_dgemm(eg_ctx* ctx, int trans_a, int trans_b, int64_t M, int64_t N, int64_t K, const double* A, int64_t lda, const double* B,
                        int64_t ldb, double* C, int64_t ldc, int accumulate, const double* bias) {
  EG_REQUIRE(ctx, EG_ERR_INVALID, "eg_dgemm: ctx is NULL");
  EG_REQUIRE(M >= 0 && N >= 0 && K >= 0, EG_ERR_INVALID, "eg_dgemm: negative extent");
  if (M == 0 || N == 0) return EG_OK;
  EG_REQUIRE(C, EG_ERR_INVALID, "eg_dgemm: C is NULL");
  EG_REQUIRE(K == 0 || (A && B), EG_ERR_INVALID, "eg_dgemm: NULL operand");
  EG_REQUIRE(lda >= (trans_a ? M : K) && ldb >= (trans_b ? K : N) && ldc >= N, EG_ERR_INVALID,
             "eg_dgemm: leading dimension smaller than the row length");
  int rc = eg::set_device(ctx);
  if (rc) return rc;
  DgemmArgs a = {};
  a.A = A;
  a.B = B;
  a.C = C;
  a.bias = bias;
  a.M = M;
  a.N = N;
  a.K = K;
  a.a_sm = trans_a ? 1 : lda;
  a.a_sk = trans_a ? lda : 1;
  a.b_sk = trans_b ? 1 : ldb;
  a.b_sn = trans_b ? ldb : 1;
  // (an operand with a single column / row has stride 1 both ways: treat it as k-contiguous only if it really is)
  a.ldc = ldc;
  a.accumulate = accumulate;
  a.splits = 1;
  a.k_per_split = ((K + BK - 1) / BK) * BK;
  if (a.k_per_split == 0) a.k_per_split = BK;
  const long cus = ctx->compute_units;
  const long tiles128 = ((M + 127) / 128) * ((N + 127) / 128);
  const bool big = tiles128 >= 2 * cus;
  const long tiles = big ? tiles128 : ((M + 63) / 64) * ((N + 63) / 64);
  if (!big && tiles < cus && K >= 1024) {
    long splits = (2 * cus + tiles - 1) / tiles;
    const long max_splits = K / 256;
    if (splits > max_splits) splits = max_splits;
    if (splits > 1) {
      long per = (K + splits - 1) / splits;
      per = ((per + BK - 1) / BK) * BK;
      splits = (K + per - 1) / per;
      if (splits > 1) {
        rc = eg::ensure_workspace(ctx, (size_t)splits * M * N * sizeof(double));
        if (rc) return rc;
        a.splits = (int)splits;
        a.k_per_split = per;
        a.C = static_cast<double*>(ctx->workspace);
      }
    }
  }
  rc = big ? launch_dgemm<128, 128>(ctx, a) : launch_dgemm<64, 64>(ctx, a);
  if (rc) return rc;
  if (a.splits > 1) {
    hipLaunchKernelGGL(dgemm_reduce_kernel, dim3((unsigned)((M * N + 255) / 256)), dim3(256), 0, ctx->stream, static_cast<const double*>(ctx->workspace), C,
                       bias, (long)M, (long)N, (long)ldc, a.splits, accumulate);
    EG_HIP_CHECK(hipGetLastError());
  }
  return EG_OK;
}

extern "C" int eg_fill_f64(eg_ctx* ctx, int64_t n, double value, double* out) {
  EG_REQUIRE(ctx, EG_ERR_INVALID, "eg_fill_f64: ctx is NULL");
  EG_REQUIRE(n >= 0, EG_ERR_INVALID, "eg_fill_f64: negative count");
  if (n == 0) return EG_OK;
  EG_REQUIRE(out, EG_ERR_INVALID, "eg_fill_f64: NULL tensor");
  int rc = eg::set_device(ctx);
  if (rc) return rc;
  long blocks = (n + 255) / 256;
  if (blocks > 8L * ctx->compute_units) blocks = 8L * ctx->compute_units;
  hipLaunchKernelGGL((fill_kernel_t<double>), dim3((unsigned)blocks), dim3(256), 0, ctx->stream, out, (long)n, value);
  EG_HIP_CHECK(hipGetLastError());
  return EG_OK;
}

extern "C" int eg_fill_uniform_f64(eg_ctx* ctx, int64_t n, double lo, double hi, const uint64_t* state, uint64_t stream, double* out) {
  EG_REQUIRE(ctx && state, EG_ERR_INVALID, "eg_fill_uniform_f64: NULL argument");
  EG_REQUIRE(n >= 0, EG_ERR_INVALID, "eg_fill_uniform_f64: negative count");
  if (n == 0) return EG_OK;
  EG_REQUIRE(out, EG_ERR_INVALID, "eg_fill_uniform_f64: NULL tensor");
  int rc = eg::set_device(ctx);
  if (rc) return rc;
  long blocks = (n + 255) / 256;
  if (blocks > 8L * ctx->compute_units) blocks = 8L * ctx->compute_units;
  hipLaunchKernelGGL(fill_uniform_f64_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, out, (long)n, lo, hi, state, stream);
  EG_HIP_CHECK(hipGetLastError());
  return EG_OK;
}

extern "C" int eg_colsum_f64(eg_ctx* ctx, int64_t rows, int64_t cols, const double* in, double* out, int accumulate) {
  EG_REQUIRE(ctx, EG_ERR_INVALID, "eg_colsum_f64: ctx is NULL");
  EG_REQUIRE(rows >= 0 && cols >= 0, EG_ERR_INVALID, "eg_colsum_f64: negative extent");
  if (cols == 0) return EG_OK;
  EG_REQUIRE(out && (rows == 0 || in), EG_ERR_INVALID, "eg_colsum_f64: NULL tensor");
  int rc = eg::ensure_workspace(ctx, (size_t)eg::colsum_f64_scratch_doubles(ctx, rows, cols) * sizeof(double));
  if (rc) return rc;
  return eg::colsum_f64_with_scratch(ctx, rows, cols, in, out, accumulate, static_cast<double*>(ctx->workspace));
}
