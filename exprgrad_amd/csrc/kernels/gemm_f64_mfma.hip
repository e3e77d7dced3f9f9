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
#include <cstdlib>
#include <cstring>

#include "../eg_internal.hpp"

namespace {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

constexpr int BK = 16;
constexpr int PAD = 16;

struct DgemmArgs {
  const double* A;
  const double* B;
  double* C;        // destination, or the slab block when splits > 1
  const double* bias;
  long M, N, K;
  long lda, ldb, ldc;
  int accumulate;
  int splits;       // k-slices (grid.y)
  long k_per_split; // multiple of BK
  int tiles_m, tiles_n;
  int remap;        // tiles % 8 == 0: contiguous tile ranges per XCD
};

// Tile loader.  A k-contiguous operand ((mn, k) at base[mn * ld + k]) is staged as [mn][LDK = 18] rows, an mn-contiguous one
// ((mn, k) at base[k * ld + mn]) as [k][BMN + 16] rows; either way a thread moves 16-byte pieces (two doubles that are
// neighbours in memory AND in LDS: one global_load_dwordx4, one ds_write_b128) and consecutive lanes walk consecutive
// memory.  Both row strides put the two half-waves of a fragment read (ds_read_b64) on disjoint bank sets.
// VEC = false (an odd leading dimension or a base that is not 16-byte aligned): the same pieces as two 8-byte loads.
// (round 6: row paddings of 4 / 6 / 10 doubles instead of 2 measured SLOWER — 4096^3 NT 0.836 -> 0.76 / 0.69 / 0.69 of peak, NN
//  0.78 -> 0.74 / 0.78 / 0.70 — although the counters report bank conflicts for this layout and none for [k][mn + 16]: NT, both
//  operands in this layout, is the fastest order; the conflicts counted are the 16-byte writes of two rows per pass, not the reads)
constexpr int LDK = BK + 2;

template <int BMN, int NT, bool KC, bool VEC>
struct TileLoader {
  static constexpr int PIECES = BMN * (BK / 2) / NT;
  static constexpr int LDM = BMN + PAD;
  static constexpr int LDS_DOUBLES = KC ? BMN * LDK : BK * LDM;
  double v[PIECES][2];
  // piece -> (mn, k) of its first element
  static __device__ __forceinline__ void where(int p, int& mn, int& k) {
    if (KC) {
      mn = p / (BK / 2);
      k = (p % (BK / 2)) * 2;
    } else {
      k = p / (BMN / 2);
      mn = (p % (BMN / 2)) * 2;
    }
  }
  __device__ __forceinline__ void load(const double* __restrict__ base, long ld, long mn0, long k0, long MN, long Kend, int tid) {
#pragma unroll
    for (int j = 0; j < PIECES; ++j) {
      int mn, k;
      where(tid + NT * j, mn, k);
      const long gm = mn0 + mn, gk = k0 + k;
      const double* src = KC ? base + gm * ld + gk : base + gk * ld + gm;
      const bool in0 = gm < MN && gk < Kend;
      const bool in1 = KC ? (gm < MN && gk + 1 < Kend) : (gm + 1 < MN && gk < Kend);
      if (VEC && in1) {  // (in1 implies in0)
        const d2 t = *reinterpret_cast<const d2*>(src);
        v[j][0] = t[0];
        v[j][1] = t[1];
      } else {
        v[j][0] = in0 ? src[0] : 0.0;
        v[j][1] = in1 ? src[1] : 0.0;
      }
    }
  }
  __device__ __forceinline__ void store(double* lds, int tid) const {
#pragma unroll
    for (int j = 0; j < PIECES; ++j) {
      int mn, k;
      where(tid + NT * j, mn, k);
      double* dst = KC ? lds + mn * LDK + k : lds + k * LDM + mn;
      *reinterpret_cast<d2*>(dst) = d2{v[j][0], v[j][1]};
    }
  }
  // fragment element (mn, k) of the staged tile
  static __device__ __forceinline__ double at(const double* lds, int mn, int k) { return KC ? lds[mn * LDK + k] : lds[k * LDM + mn]; }
};

// One wave alone issues a float64 MFMA every ~146 cycles, two waves of a SIMD together one every 64 (tools/mfma_ceiling_f64.hip:
// 34 against 77.8 TFLOP/s) — the matrix pipe needs several waves per SIMD that are multiplying at the same time.  So a tile
// is shared by WR x WC waves with small sub-tiles (128 x 128: eight waves of 64 x 32; 64 x 64: four of 32 x 32) and
// blocks are small enough in LDS for two (three) of them per CU, which meet their barriers at different times.
template <int BM, int BN, int WR, int WC, bool AKC, bool BKC, bool VEC>
__global__ __launch_bounds__(WR* WC * 64, 4) void dgemm_kernel(DgemmArgs a) {
  constexpr int NT = WR * WC * 64;
  constexpr int WM = BM / WR, WN = BN / WC;
  constexpr int FM = WM / 16, FN = WN / 16;
  using LA = TileLoader<BM, NT, AKC, VEC>;
  using LB = TileLoader<BN, NT, BKC, VEC>;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double* As = lds;                             // [2][LA::LDS_DOUBLES]
  double* Bs = lds + 2 * LA::LDS_DOUBLES;       // [2][LB::LDS_DOUBLES]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = (wave / WC) * WM, wn = (wave % WC) * WN;
  long tile = blockIdx.x;
  if (a.remap) {
    const long per = (long)gridDim.x >> 3;
    tile = (tile & 7) * per + (tile >> 3);
  }
  const long tm = tile / a.tiles_n, tn = tile % a.tiles_n;
  const long m0 = tm * BM, n0 = tn * BN;
  const long kbeg = (long)blockIdx.y * a.k_per_split;
  const long kend = min(a.K, kbeg + a.k_per_split);

  d4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = d4{0.0, 0.0, 0.0, 0.0};

  LA la;
  LB lb;
  const long ktiles = kend > kbeg ? (kend - kbeg + BK - 1) / BK : 0;
  if (ktiles > 0) {
    la.load(a.A, a.lda, m0, kbeg, a.M, kend, tid);
    lb.load(a.B, a.ldb, n0, kbeg, a.N, kend, tid);
    la.store(As, tid);
    lb.store(Bs, tid);
  }
  __syncthreads();
  const int fr = lane & 15, fk = lane >> 4;
  for (long kt = 0; kt < ktiles; ++kt) {
    const int cur = (int)(kt & 1);
    const bool more = kt + 1 < ktiles;
    if (more) {
      la.load(a.A, a.lda, m0, kbeg + (kt + 1) * BK, a.M, kend, tid);
      lb.load(a.B, a.ldb, n0, kbeg + (kt + 1) * BK, a.N, kend, tid);
    }
    const double* Ac = As + cur * LA::LDS_DOUBLES;
    const double* Bc = Bs + cur * LB::LDS_DOUBLES;
    // (two k-steps per unrolled body: with all four the k-contiguous variants held every fragment of the k-tile at once and
    //  spilled 2 - 11 registers at the 128 that four waves per SIMD allow — 4096^3 NN ran 0.72 of peak where TN, which did not spill, ran 0.84)
#pragma unroll 2
    for (int s = 0; s < BK / 4; ++s) {
      double af[FM], bf[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i) af[i] = LA::at(Ac, wm + 16 * i + fr, 4 * s + fk);
#pragma unroll
      for (int j = 0; j < FN; ++j) bf[j] = LB::at(Bc, wn + 16 * j + fr, 4 * s + fk);
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    if (more) {
      la.store(As + (cur ^ 1) * LA::LDS_DOUBLES, tid);
      lb.store(Bs + (cur ^ 1) * LB::LDS_DOUBLES, tid);
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

template <int BM, int BN, int WR, int WC, bool AKC, bool BKC, bool VEC>
int launch_one(eg_ctx* ctx, DgemmArgs& a) {
  using LA = TileLoader<BM, WR * WC * 64, AKC, VEC>;
  using LB = TileLoader<BN, WR * WC * 64, BKC, VEC>;
  constexpr size_t lds = (size_t)2 * (LA::LDS_DOUBLES + LB::LDS_DOUBLES) * sizeof(double);
  static bool attr_set = false;
  if (!attr_set) {
    EG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&dgemm_kernel<BM, BN, WR, WC, AKC, BKC, VEC>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  a.tiles_m = (int)((a.M + BM - 1) / BM);
  a.tiles_n = (int)((a.N + BN - 1) / BN);
  const long tiles = (long)a.tiles_m * a.tiles_n;
  a.remap = tiles % 8 == 0 && tiles >= 16;
  hipLaunchKernelGGL((dgemm_kernel<BM, BN, WR, WC, AKC, BKC, VEC>), dim3((unsigned)tiles, (unsigned)a.splits), dim3(WR * WC * 64), lds, ctx->stream, a);
  EG_HIP_CHECK(hipGetLastError());
  return EG_OK;
}

template <int BM, int BN, int WR, int WC>
int launch_dgemm(eg_ctx* ctx, DgemmArgs& a, bool akc, bool bkc, bool vec) {
#define EG_DGEMM_CASE(AK, BK_, V) \
  if (akc == AK && bkc == BK_ && vec == V) return launch_one<BM, BN, WR, WC, AK, BK_, V>(ctx, a);
  EG_DGEMM_CASE(true, true, true)
  EG_DGEMM_CASE(true, false, true)
  EG_DGEMM_CASE(false, true, true)
  EG_DGEMM_CASE(false, false, true)
  EG_DGEMM_CASE(true, true, false)
  EG_DGEMM_CASE(true, false, false)
  EG_DGEMM_CASE(false, true, false)
  EG_DGEMM_CASE(false, false, false)
#undef EG_DGEMM_CASE
  return EG_ERR_INVALID;
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
  a.lda = lda;
  a.ldb = ldb;
  a.ldc = ldc;
  // A(m, k): [M, K] rows are k-contiguous unless transposed; B(k, n): [K, N] rows are n-contiguous unless transposed
  const bool akc = !trans_a, bkc = trans_b != 0;
  const bool vec = lda % 2 == 0 && ldb % 2 == 0 && (reinterpret_cast<uintptr_t>(A) & 15) == 0 && (reinterpret_cast<uintptr_t>(B) & 15) == 0;
  a.accumulate = accumulate;
  a.splits = 1;
  a.k_per_split = ((K + BK - 1) / BK) * BK;
  if (a.k_per_split == 0) a.k_per_split = BK;
  // Tile shape by a small time model.  What a SIMD's matrix pipe delivers depends on how many waves multiply on it at the
  // same time (tools/mfma_ceiling_f64.hip: one wave 0.44 of peak, two 0.99 in a bare loop), so a launch is priced as
  // rounds of resident blocks, each round at the rate of the waves it puts on a SIMD:
  //   config        waves  blocks / CU (LDS)   relative loop efficiency
  //   128 x 128     8      2 (74 KB)           1.00
  //   128 x  64     8      2 (55 KB)           0.97
  //    64 x  64     8      4 (37 KB)           0.92
  const long cus = ctx->compute_units;
  struct Cfg { int bm, bn, cap; double eff; };
  static const Cfg cfgs[3] = {{128, 128, 2, 1.0}, {128, 64, 2, 0.97}, {64, 64, 4, 0.92}};
  auto rate = [](long waves_per_simd) { return waves_per_simd <= 1 ? 0.44 : waves_per_simd == 2 ? 0.80 : waves_per_simd == 3 ? 0.88 : 0.92; };
  auto cost = [&](const Cfg& c, long splits) {
    const long tiles = ((M + c.bm - 1) / c.bm) * ((N + c.bn - 1) / c.bn) * splits;
    const double t_tile = (double)c.bm * c.bn * ((double)K / splits) / c.eff;  // matrix work of one block at the pipe's full rate
    const long slots = cus * c.cap;
    const long full = tiles / slots, rem = tiles % slots;
    double t = (double)full * c.cap * t_tile / rate(2L * c.cap);  // 8-wave blocks: two waves per SIMD each
    if (rem > 0) {
      const long per_cu = (rem + cus - 1) / cus;
      t += (double)per_cu * t_tile / rate(2L * per_cu);
    }
    // slabs out and back (16 bytes per element and slice at ~3 TB/s, in units of a CU's 64 multiply-adds per clock) and a second launch (~5 us)
    if (splits > 1) t += (double)M * N * splits * 1.0 + 8.0e5;
    return t;
  };
  int best = 0;
  long best_splits = 1;
  double best_t = 1e300;
  for (int c = 0; c < 3; ++c)
    for (long splits = 1; splits <= 64 && (splits == 1 || K / splits >= 256); splits *= 2) {
      const double t = cost(cfgs[c], splits);
      if (t < best_t) {
        best_t = t;
        best = c;
        best_splits = splits;
      }
    }
  if (const char* e = eg::sw::raw("EG_DGEMM_TILE")) {  // measurement aid: "<config>[,<splits>]"
    best = atoi(e) % 3;
    if (const char* comma = strchr(e, ',')) best_splits = atol(comma + 1) > 0 ? atol(comma + 1) : 1;
  }
  if (best_splits > 1) {
    long per = (K + best_splits - 1) / best_splits;
    per = ((per + BK - 1) / BK) * BK;
    const long splits = (K + per - 1) / per;
    if (splits > 1) {
      rc = eg::ensure_workspace(ctx, (size_t)splits * M * N * sizeof(double));
      if (rc) return rc;
      a.splits = (int)splits;
      a.k_per_split = per;
      a.C = static_cast<double*>(ctx->workspace);
    }
  }
  rc = best == 0   ? launch_dgemm<128, 128, 2, 4>(ctx, a, akc, bkc, vec)
       : best == 1 ? launch_dgemm<128, 64, 4, 2>(ctx, a, akc, bkc, vec)
                   : launch_dgemm<64, 64, 2, 4>(ctx, a, akc, bkc, vec);
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
