// f32 contraction  C[m,n] (+)= sum_k opA(m,k) * opB(k,n) (+ bias[n])  on CDNA4 matrix cores.
//
// Replaces what the reference emits for `c[y,x] ++= a[y,it] * b[it,x]` on its GPU target
// (tests/cache/matmul_basic.ir: one work-item per output, global-memory RMW per k; or the
// user-scheduled 16x16x16 LDS tiling of tests/cache/matmul_schedule_tiled16.ir) and for the two
// gradient contractions passes.nim:519-549 derives from it.
//
// Design (gfx950):
//   * v_mfma_f32_32x32x2_f32: exact f32 products and accumulation (an fmaf chain), 64 cycles per
//     instruction per SIMD = the 157 TFLOP/s f32 matrix peak.  No reduced-precision path exists
//     or is wanted (parity is 1e-5 relative against the reference's f32 CPU path).
//   * block tile BM x BN x 16, 4 waves (256 threads); every wave owns a WM x WN sub-tile made of
//     32x32 MFMA blocks, accumulators stay in registers for the whole K loop (the reference
//     re-reads and re-writes C once per k).
//   * both operand tiles are staged in LDS as [k][m|n] so that an MFMA operand fetch is one
//     ds_read_b32 of 32 consecutive dwords per half-wave (bank-conflict free).  An operand whose
//     global layout is k-contiguous (A of NN/NT, B of NT) is transposed on the way in: coalesced
//     16-byte global loads along k, four ds_write_b32; the row stride is chosen so those writes
//     spread over all 32 banks.
//   * double-buffered LDS + register prefetch of the next k-tile: one barrier per k-tile, global
//     latency hidden behind 32 MFMAs (2048 cycles) per wave and by the co-resident blocks.
//   * XCD-aware tile order: consecutive block ids land on different XCDs (id % 8), so ids are
//     remapped to give each XCD's private L2 a contiguous, squarish patch of output tiles.
//   * split-K (grid.z) with a deterministic second pass for contractions whose output is small
//     and whose K is the batch (the weight gradients): no float atomics, fixed summation order.
#include "../eg_internal.hpp"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct GemmArgs {
  const float* A;
  const float* B;
  float* C;
  const float* bias;
  float* partial;  // split-K slabs [splits][M][N] or nullptr
  long M, N, K;
  long lda, ldb, ldc;
  long k_per_split;  // multiple of BK
  int tiles_m, tiles_n;
  int accumulate;
  // Implicit-GEMM convolution (A gathered from an NHWC image): m = (n, y, x), k = (dy, dx, c).
  long cH, cW, cC, cFW, cHo, cWo;
};

constexpr int BK = 16;
constexpr int NT = 256;  // threads per block (4 waves)

// Row stride (in floats) of an LDS operand tile [BK][stride].
//  - mn-contiguous operand: 16-byte ds_write_b128 rows, no padding needed.
//  - k-contiguous operand (transposed while staging): with 16-byte global loads 4 lanes share a
//    row and write k = 4c+j; stride % 8 == 2 puts the 32 lanes of a half-wave on 32 distinct banks.
template <int BMN, bool KC>
struct LdsStride {
  static constexpr int value = KC ? BMN + 2 : BMN;
};

// One operand tile: BMN rows/cols along m|n, BK along k.
//   KC == true : global element (mn, k) at  base[mn * ld + k]   (k contiguous)
//   KC == false: global element (mn, k) at  base[k * ld + mn]   (m|n contiguous)
//   CONV (KC only): element (m, k) of the virtual im2col matrix, read straight from the image:
//       base[((n*H + y + dy)*W + x + dx)*C + c],  m = (n*Ho + y)*Wo + x,  k = (dy*FW + dx)*C + c
//     (dnn.nim:45-49: images[image, y + dy, x + dx, chan], valid padding, stride 1).
template <int BMN, bool KC, int VEC, bool EDGE, bool CONV>
struct TileLoader {
  static_assert(!CONV || KC, "the gathered operand is k(channel)-contiguous");
  static constexpr int STRIDE = LdsStride<BMN, KC>::value;
  static constexpr int ELEMS = BMN * BK;
  static constexpr int CHUNKS = ELEMS / VEC;
  static constexpr int NVEC = (CHUNKS + NT - 1) / NT;  // load instructions per thread
  static constexpr int PER_THREAD = NVEC * VEC;         // floats per thread
  static constexpr bool PARTIAL = CHUNKS % NT != 0;     // small tile: trailing threads idle
  static constexpr int CPR = (KC ? BK : BMN) / VEC;     // chunks per tile row

  float regs[PER_THREAD];
  long row_off[NVEC];  // CONV: element offset of the output pixel's top-left input pixel

  __device__ __forceinline__ static void coords(int idx, int& mn, int& k) {
    if (KC) {
      mn = idx / CPR;
      k = (idx % CPR) * VEC;
    } else {
      k = idx / CPR;
      mn = (idx % CPR) * VEC;
    }
  }

  __device__ __forceinline__ void init(const GemmArgs& a, long mn0, int tid) {
    if (CONV) {
#pragma unroll
      for (int i = 0; i < NVEC; ++i) {
        int mn, k;
        coords(tid + i * NT, mn, k);
        const long m = mn0 + mn;
        const long img = m / (a.cHo * a.cWo), rem = m % (a.cHo * a.cWo);
        const long y = rem / a.cWo, x = rem % a.cWo;
        row_off[i] = ((img * a.cH + y) * a.cW + x) * a.cC;
      }
    }
  }

  // mn0/k0: tile origin; mn_lim/k_lim: exclusive global limits (only read when EDGE).
  __device__ __forceinline__ void load(const GemmArgs& a, const float* __restrict__ base, long ld, long mn0, long k0,
                                       long mn_lim, long k_lim, int tid) {
#pragma unroll
    for (int i = 0; i < NVEC; ++i) {
      const int idx = tid + i * NT;
      int mn, k;
      coords(idx, mn, k);
      const long gmn = mn0 + mn, gk = k0 + k;
      const float* p;
      if (CONV) {
        const long tap = gk / a.cC, c = gk % a.cC;
        const long dy = tap / a.cFW, dx = tap % a.cFW;
        p = base + row_off[i] + (dy * a.cW + dx) * a.cC + c;
      } else {
        p = KC ? base + gmn * ld + gk : base + gk * ld + gmn;
      }
      bool ok = !PARTIAL || idx < CHUNKS;
      if (EDGE) ok = ok && (gmn < mn_lim) && (gk < k_lim);
      if (VEC == 4) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (ok) v = *reinterpret_cast<const f32x4*>(p);
        regs[i * 4 + 0] = v[0];
        regs[i * 4 + 1] = v[1];
        regs[i * 4 + 2] = v[2];
        regs[i * 4 + 3] = v[3];
      } else {
        regs[i] = ok ? *p : 0.f;
      }
    }
  }

  __device__ __forceinline__ void store(float* __restrict__ lds, int tid) const {
#pragma unroll
    for (int i = 0; i < NVEC; ++i) {
      const int idx = tid + i * NT;
      if (PARTIAL && idx >= CHUNKS) continue;
      int mn, k;
      coords(idx, mn, k);
      if (KC) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) lds[(k + j) * STRIDE + mn] = regs[i * VEC + j];
      } else {
        if (VEC == 4) {
          f32x4 v = {regs[i * 4 + 0], regs[i * 4 + 1], regs[i * 4 + 2], regs[i * 4 + 3]};
          *reinterpret_cast<f32x4*>(&lds[k * STRIDE + mn]) = v;
        } else {
          lds[k * STRIDE + mn] = regs[i];
        }
      }
    }
  }
};

// Bijective XCD remap (block id b runs on XCD b % 8): give every XCD a contiguous id range.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  constexpr int NXCD = 8;
  const int q = nwg / NXCD, r = nwg % NXCD;
  const int xcd = bid % NXCD, local = bid / NXCD;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + local;
}

template <int BM, int BN, int WM, int WN, bool A_KC, bool B_KC, int VEC, bool EDGE, bool CONV>
__global__ __launch_bounds__(NT, 4) void gemm_f32_mfma_kernel(GemmArgs a) {
  constexpr int WAVES_N = BN / WN;
  constexpr int MI = WM / 32, NI = WN / 32;
  static_assert((BM / WM) * WAVES_N * 64 == NT, "4 waves per block");
  using LoadA = TileLoader<BM, A_KC, VEC, EDGE, CONV>;
  using LoadB = TileLoader<BN, B_KC, VEC, EDGE, false>;
  constexpr int SA = LoadA::STRIDE, SB = LoadB::STRIDE;

  __shared__ __attribute__((aligned(16))) float lds[2 * BK * (SA + SB)];
  constexpr int BUF = BK * (SA + SB);  // one stage: A tile then B tile

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm0 = (wave / WAVES_N) * WM;
  const int wn0 = (wave % WAVES_N) * WN;

  // ---- tile coordinates: XCD-contiguous ids, then 8-row groups so co-resident tiles share
  //      A row-panels and B column-panels inside one L2.
  const int nwg = a.tiles_m * a.tiles_n;
  const int wgid = xcd_remap(blockIdx.x, nwg);
  constexpr int GROUP = 8;
  const int per_group = GROUP * a.tiles_n;
  const int group = wgid / per_group;
  const int first_m = group * GROUP;
  const int gsize = min(a.tiles_m - first_m, GROUP);
  const int in_group = wgid % per_group;
  const long m_blk = (long)(first_m + in_group % gsize) * BM;
  const long n_blk = (long)(in_group / gsize) * BN;

  const long k_begin = (long)blockIdx.z * a.k_per_split;
  const long k_end = min(a.K, k_begin + a.k_per_split);
  const int nk = (int)((k_end - k_begin + BK - 1) / BK);

  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  LoadA la;
  LoadB lb;
  la.init(a, m_blk, tid);
  lb.init(a, n_blk, tid);
  if (nk > 0) {
    la.load(a, a.A, a.lda, m_blk, k_begin, a.M, k_end, tid);
    lb.load(a, a.B, a.ldb, n_blk, k_begin, a.N, k_end, tid);
    la.store(lds, tid);
    lb.store(lds + BK * SA, tid);
  }
  __syncthreads();

  const int a_off = (lane >> 5) * SA + wm0 + (lane & 31);
  const int b_off = (lane >> 5) * SB + wn0 + (lane & 31);

  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    const bool more = kt + 1 < nk;
    if (more) {
      const long k0 = k_begin + (long)(kt + 1) * BK;
      la.load(a, a.A, a.lda, m_blk, k0, a.M, k_end, tid);
      lb.load(a, a.B, a.ldb, n_blk, k0, a.N, k_end, tid);
    }
    const float* as = lds + cur * BUF + a_off;
    const float* bs = lds + cur * BUF + BK * SA + b_off;
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      float av[MI], bv[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) av[i] = as[kk * 2 * SA + i * 32];
#pragma unroll
      for (int j = 0; j < NI; ++j) bv[j] = bs[kk * 2 * SB + j * 32];
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
    }
    if (more) {
      la.store(lds + (cur ^ 1) * BUF, tid);
      lb.store(lds + (cur ^ 1) * BUF + BK * SA, tid);
    }
    __syncthreads();
  }

  // ---- epilogue.  32x32 accumulator block: register r of lane l holds
  //      row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31.
  const bool to_partial = a.partial != nullptr;
  float* out = to_partial ? a.partial + (long)blockIdx.z * a.M * a.N : a.C;
  const long ldo = to_partial ? a.N : a.ldc;
#pragma unroll
  for (int i = 0; i < MI; ++i) {
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const long n = n_blk + wn0 + j * 32 + (lane & 31);
      float bias = 0.f;
      if (!to_partial && a.bias != nullptr && (!EDGE || n < a.N)) bias = a.bias[n];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long m = m_blk + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (EDGE && (m >= a.M || n >= a.N)) continue;
        float* p = out + m * ldo + n;
        float v = acc[i][j][r];
        if (!to_partial) {
          if (a.accumulate) v = *p + v;
          v += bias;
        }
        *p = v;
      }
    }
  }
}

// Second pass of split-K: C[m,n] = (accumulate ? C : 0) + sum_z partial[z][m][n] + bias[n],
// slabs added in increasing z (fixed order => run-to-run deterministic).
__global__ __launch_bounds__(256) void gemm_splitk_reduce_kernel(const float* __restrict__ partial, float* C,
                                                                 const float* __restrict__ bias, long M, long N,
                                                                 long ldc, int splits, int accumulate) {
  const long total = M * N;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long m = i / N, n = i % N;
    float s = 0.f;
    for (int z = 0; z < splits; ++z) s += partial[(long)z * total + i];
    float* p = C + m * ldc + n;
    if (accumulate) s = *p + s;
    if (bias) s += bias[n];
    *p = s;
  }
}

template <int BM, int BN, int WM, int WN>
int launch_config(eg_ctx* ctx, bool a_kc, bool b_kc, const GemmArgs& args, int splits, bool vec, bool edge,
                  bool conv) {
  dim3 grid((unsigned)(args.tiles_m * args.tiles_n), 1, (unsigned)splits);
  dim3 block(NT);
  hipStream_t s = ctx->stream;
#define EG_GEMM_LAUNCH(AKC, BKC, V, E, CV) \
  hipLaunchKernelGGL((gemm_f32_mfma_kernel<BM, BN, WM, WN, AKC, BKC, V, E, CV>), grid, block, 0, s, args)
#define EG_GEMM_LAYOUT(AKC, BKC)                  \
  do {                                            \
    if (!edge)                                    \
      EG_GEMM_LAUNCH(AKC, BKC, 4, false, false);  \
    else if (vec)                                 \
      EG_GEMM_LAUNCH(AKC, BKC, 4, true, false);   \
    else                                          \
      EG_GEMM_LAUNCH(AKC, BKC, 1, true, false);   \
  } while (0)
  if (conv) {
    if (vec)
      EG_GEMM_LAUNCH(true, true, 4, true, true);
    else
      EG_GEMM_LAUNCH(true, true, 1, true, true);
  } else if (a_kc && !b_kc) {
    EG_GEMM_LAYOUT(true, false);  // NN
  } else if (a_kc && b_kc) {
    EG_GEMM_LAYOUT(true, true);  // NT
  } else if (!a_kc && !b_kc) {
    EG_GEMM_LAYOUT(false, false);  // TN
  } else {
    EG_GEMM_LAYOUT(false, true);  // TT
  }
#undef EG_GEMM_LAYOUT
#undef EG_GEMM_LAUNCH
  EG_HIP_CHECK(hipGetLastError());
  return EG_OK;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// Shared host-side planning: tile shape, split-K, vector/edge variant, launch, second pass.
int run_gemm(eg_ctx* ctx, bool a_kc, bool b_kc, GemmArgs args, bool conv, bool vec_ok) {
  const long M = args.M, N = args.N, K = args.K;
  // Narrow outputs (bias-sized N: the N = 1/4/10 layers of the XOR and dense nets, F = 64 filter
  // banks) get narrower tiles so the padding wasted in the matrix core stays small.
  const int BM = 128;
  const int BN = N <= 32 ? 32 : (N <= 64 ? 64 : 128);
  args.tiles_m = (int)((M + BM - 1) / BM);
  args.tiles_n = (int)((N + BN - 1) / BN);
  args.partial = nullptr;

  // Split K when the output alone cannot fill the chip and K is long (weight gradients:
  // K = batch).  Target ~4 blocks per CU; every slice keeps at least 8 k-tiles.
  const long tiles = (long)args.tiles_m * args.tiles_n;
  const long k_tiles = (K + BK - 1) / BK;
  int splits = 1;
  const long target_blocks = 4L * ctx->compute_units;
  if (tiles < 2L * ctx->compute_units && k_tiles >= 32) {
    long want = (target_blocks + tiles - 1) / tiles;
    long max_by_k = k_tiles / 8;
    splits = (int)(want < max_by_k ? want : max_by_k);
    if (splits < 1) splits = 1;
    if (splits > 1024) splits = 1024;
  }
  long tiles_per_split = (k_tiles + splits - 1) / splits;
  if (tiles_per_split < 1) tiles_per_split = 1;
  args.k_per_split = tiles_per_split * BK;
  splits = (int)((k_tiles + tiles_per_split - 1) / tiles_per_split);
  if (splits < 1) splits = 1;
  if (splits > 1) {
    int rc = eg::ensure_workspace(ctx, (size_t)splits * M * N * sizeof(float));
    if (rc) return rc;
    args.partial = static_cast<float*>(ctx->workspace);
  }

  const bool vec = vec_ok;
  const bool edge = conv || !(vec && M % BM == 0 && N % BN == 0 && K % BK == 0 && K > 0);
  int rc;
  if (BN == 32)
    rc = launch_config<128, 32, 32, 32>(ctx, a_kc, b_kc, args, splits, vec, edge, conv);
  else if (BN == 64)
    rc = launch_config<128, 64, 64, 32>(ctx, a_kc, b_kc, args, splits, vec, edge, conv);
  else
    rc = launch_config<128, 128, 64, 64>(ctx, a_kc, b_kc, args, splits, vec, edge, conv);
  if (rc) return rc;

  if (splits > 1) {
    const long total = M * N;
    long blocks = (total + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, args.partial,
                       args.C, args.bias, M, N, args.ldc, splits, args.accumulate);
    EG_HIP_CHECK(hipGetLastError());
  }
  return EG_OK;
}

}  // namespace

extern "C" int eg_sgemm(eg_ctx* ctx, int trans_a, int trans_b, int64_t M, int64_t N, int64_t K, const float* A,
                        int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int accumulate,
                        const float* bias) {
  EG_REQUIRE(ctx, EG_ERR_INVALID, "eg_sgemm: ctx is NULL");
  EG_REQUIRE(M >= 0 && N >= 0 && K >= 0, EG_ERR_INVALID, "eg_sgemm: negative extent");
  if (M == 0 || N == 0) return EG_OK;
  EG_REQUIRE(C, EG_ERR_INVALID, "eg_sgemm: C is NULL");
  EG_REQUIRE(K == 0 || (A && B), EG_ERR_INVALID, "eg_sgemm: NULL operand");
  EG_REQUIRE(lda >= (trans_a ? M : K) && ldb >= (trans_b ? K : N) && ldc >= N, EG_ERR_INVALID,
             "eg_sgemm: leading dimension smaller than the row length");
  int rc = eg::set_device(ctx);
  if (rc) return rc;

  // Operand "k-contiguous" flags: A[M,K] row-major has k contiguous unless transposed;
  // B[K,N] row-major has n contiguous unless transposed.
  const bool a_kc = !trans_a, b_kc = trans_b != 0;
  GemmArgs args = {};
  args.A = A;
  args.B = B;
  args.C = C;
  args.bias = bias;
  args.M = M;
  args.N = N;
  args.K = K;
  args.lda = lda;
  args.ldb = ldb;
  args.ldc = ldc;
  args.accumulate = accumulate;
  // 16-byte global loads need every row start and every chunk 16-byte aligned and whole.
  const long a_contig = a_kc ? K : M, b_contig = b_kc ? K : N;
  const bool vec = (lda % 4 == 0) && (ldb % 4 == 0) && (a_contig % 4 == 0) && (b_contig % 4 == 0) &&
                   (A == nullptr || aligned16(A)) && (B == nullptr || aligned16(B));
  return run_gemm(ctx, a_kc, b_kc, args, /*conv=*/false, vec);
}

// Direct convolution as an implicit GEMM:  M = N*Ho*Wo output pixels, N = F filters,
// K = FH*FW*C taps; A is gathered from the NHWC image inside the tile loader (no im2col
// buffer), B = the filter bank [F][FH*FW*C] is already "N x K, k-contiguous".
extern "C" int eg_conv2_nhwc(eg_ctx* ctx, int64_t N, int64_t H, int64_t W, int64_t C, int64_t F, int64_t FH,
                             int64_t FW, const float* img, const float* flt, float* out, int accumulate) {
  EG_REQUIRE(ctx, EG_ERR_INVALID, "eg_conv2_nhwc: ctx is NULL");
  EG_REQUIRE(N >= 0 && H >= 0 && W >= 0 && C >= 0 && F >= 0 && FH >= 1 && FW >= 1, EG_ERR_INVALID,
             "eg_conv2_nhwc: bad extent");
  // Output shape as the reference's linear shape solver finds it (passes.nim:1420-1436):
  // max(y) + max(dy) = H - 1.
  const long Ho = H - FH + 1, Wo = W - FW + 1;
  EG_REQUIRE(Ho >= 0 && Wo >= 0, EG_ERR_SHAPE, "eg_conv2_nhwc: filter larger than image");
  if (N == 0 || Ho == 0 || Wo == 0 || F == 0) return EG_OK;
  EG_REQUIRE(out && (C == 0 || (img && flt)), EG_ERR_INVALID, "eg_conv2_nhwc: NULL tensor");
  int rc = eg::set_device(ctx);
  if (rc) return rc;
  GemmArgs args = {};
  args.A = img;
  args.B = flt;
  args.C = out;
  args.bias = nullptr;
  args.M = N * Ho * Wo;
  args.N = F;
  args.K = FH * FW * C;
  args.lda = 0;
  args.ldb = args.K;
  args.ldc = F;
  args.accumulate = accumulate;
  args.cH = H;
  args.cW = W;
  args.cC = C;
  args.cFW = FW;
  args.cHo = Ho;
  args.cWo = Wo;
  const bool vec = (C % 4 == 0) && aligned16(img) && aligned16(flt);
  return run_gemm(ctx, true, true, args, /*conv=*/true, vec);
}
