// f32 contraction kernel template for CDNA4 (gfx950) matrix cores — see gemm_f32_mfma.hip for the
// role of this kernel in the backend.  Kept in a header so the tuning harness
// (tools/gemm_tune.hip) instantiates exactly the code the library ships.
//
// Design (gfx950):
//   * v_mfma_f32_32x32x2_f32: exact f32 products and accumulation (an fmaf chain), 64 cycles per
//     instruction per SIMD = the 157 TFLOP/s f32 matrix peak.  No reduced-precision path exists
//     or is wanted (parity is 1e-5 relative against the reference's f32 CPU path).
//   * block tile BM x BN x BK, (BM/WM)*(BN/WN) waves; every wave owns a WM x WN sub-tile made of
//     32x32 MFMA blocks, accumulators stay in registers for the whole K loop (the reference
//     re-reads and re-writes C once per k: tests/cache/matmul_basic.ir).
//   * both operand tiles are staged in LDS as [k][m|n] so that an MFMA operand fetch is a
//     ds_read of 32 consecutive dwords per half-wave (bank-conflict free).  An operand whose
//     global layout is k-contiguous (A of NN/NT, B of NT) is transposed on the way in: coalesced
//     16-byte global loads along k, four ds_write_b32; the row stride is chosen so those writes
//     spread over all 32 banks.
//   * double-buffered LDS + register prefetch of the next k-tile: one barrier per k-tile; operand
//     fragments of k-step kk+1 are fetched before the MFMAs of k-step kk issue.
//   * XCD-aware tile order: consecutive block ids land on different XCDs (id % 8), so ids are
//     remapped to give each XCD's private L2 a contiguous, squarish patch of output tiles.
//   * split-K (grid.z) with a deterministic second pass for contractions whose output is small
//     and whose K is the batch (the weight gradients): no float atomics, fixed summation order.
#ifndef __HIPCC_RTC__  // hiprtc (kernels/gemm_fused.hpp) compiles this text as the main file, runtime built in
#pragma once
#include <hip/hip_runtime.h>
#endif

namespace eg {
namespace gemm {

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
  int splits;  // k-splits (grid = tiles_m * tiles_n * splits blocks); 0 / 1 = none
  // Ragged last tile row with split-K: its tiles hold little matrix work (empty 32x32 sub-blocks
  // are skipped), so they are cut into fewer, longer k-slices to finish with the full tiles.
  // edge_splits > 0: tiles of row tiles_m - 1 use edge_splits slices of k_per_split_edge.
  int edge_splits;
  long k_per_split_edge;
  // Tail tiles (splits <= 1 only): when the tiles do not fill a whole number of rounds of the chip, the
  // last tail_tiles of them are cut into tail_splits k-slices each, so the partial last round is short
  // instead of a whole tile long (4100^3: 289 tiles on 256 CUs = two rounds for 1.13 rounds of work).
  // Their blocks write whole-tile slabs partial[(tile - first tail tile) * tail_splits + slice][BM][BN];
  // gemm_tail_reduce_kernel folds them into C in slice order.
  int tail_tiles, tail_splits;
  long tail_k_per_split;
  // Implicit-GEMM convolution (A gathered from an NHWC image): m = (n, y, x), k = (dy, dx, c).
  long cH, cW, cC, cFW, cHo, cWo;
  // Operands of a generated epilogue (fused elementwise consumer, host/epilogue.cpp): tensors of
  // the same [M, N] shape as C, the seed-gradient scale and the epoch.
  void* epi[6];
  float epi_gs;
  long epi_ep;
  // Virtual last row of ones: with ones_row != 0 the A operand has a_rows = M - 1 physical rows and
  // A(M - 1, k) = 1 for every k, so row M - 1 of C is the column sum of B.  This is how a bias gradient
  // `gb[x] ++= g[y,x]` rides along with the weight gradient `gW[it,x] ++= a[y,it] * g[y,x]` that
  // reduces over the same batch (dnn.nim:19-24 differentiated, passes.nim:519-549): gb is row M - 1
  // of [gW; gb] when the two are adjacent in memory.  LDS-DMA loop only (the host checks): the lanes
  // whose 16-byte chunk holds the virtual row fetch it from `ones` instead of from A — no extra LDS
  // pass, no extra barrier (a first version wrote the ones into LDS after every k-tile landed: the
  // barrier that needed cost the edge blocks 0.3 us per k-tile, 86 us on the cfg-5 weight gradient).
  int ones_row;
  long a_rows;  // physical rows of A (= M unless ones_row)
  const float* ones;  // device constants {1,1,1,1, 1,0,0,0}: what the loaders fetch for the virtual row
  // Whole tiles leave through LDS: the accumulator blocks of 32 tile rows per wave row are transposed
  // into [row][BN] order and written (and, for a generated epilogue, its operands read) as 16-byte
  // accesses, a full tile row per wave instruction instead of 128-byte pieces of 32 different rows.
  // Requires N, ldc multiples of 4 and 16-byte aligned C / partial / epilogue operands (host checks).
  int wide_store;
  // Launched on the side lane, next to a long contraction of the main lane: the waves raise their issue
  // priority (s_setprio 3).  The instruction arbiter otherwise serves the older waves of the long
  // contraction first, whose MFMAs are always ready: a 128x32 product that takes 18 us alone was
  // measured at 208 us next to a 256x256 one (rocprofv3 timeline, profiles/r02_pipeline_trace.txt).
  int prio;
  int nt_store;  // wide stores as nontemporal stores (host: nt_store_enabled)
  int no_skew;   // EG_GEMM_NO_SKEW=1: every wave of a block runs the k loop in phase (the round-3 loop; A/B aid)
  // Extra rows (XR kernels; split-K launches of the TN form): M = tiles_m * BM + x_rows with 0 < x_rows <= 32.  The blocks of
  // the LAST tile row carry a ninth accumulator block per wave for rows [tiles_m * BM, M) — a [k][32] strip of A staged
  // next to the tile, multiplied with the B tile the block stages anyway — instead of a ragged tile row of its own that
  // would stage whole B tiles for 1/8 of the matrix work (784 x 512 x 65536: the 16 (+1 virtual) rows beyond 768 cost
  // 47 us as a tile row).  Those blocks get edge_splits (more, shorter) k-slices.
  int x_rows;
  // EG_GEMM_TRACE=1 (detector): four cycle stamps per wave — entry, in front of the k loop, behind it, behind the epilogue
  long long* trace;
};

// Epilogue functor of the library kernels: plain store.  A generated epilogue (ACTIVE = true)
// receives every finished element v = acc + bias of a non-accumulating, non-split launch together
// with its flat index m * ldc + n.  The work is split so the kernel can issue the loads of all
// elements it holds before the first dependent store: prefetch() reads the NX other operands of the
// element, compute() produces the consumer's value.
struct EpiNone {
  static constexpr bool ACTIVE = false;
  static constexpr bool STORE_C = true;
  static constexpr int NX = 1;
  static constexpr int OUT = 0;
  static constexpr int PRED = -1;
  static constexpr int RD_N = 0, RD_W = 0, RD_OUT = 0, RD_BIAS = -1, RD_LDW = 0, RD_LDO = 0;
  __device__ __forceinline__ static bool predicate(float) { return false; }
  __device__ __forceinline__ static void prefetch(const GemmArgs&, long, float (&)[1]) {}
  __device__ __forceinline__ static void prefetch4(const GemmArgs&, long, f32x4 (&)[1]) {}
  __device__ __forceinline__ static float compute(const GemmArgs&, long, float v, const float (&)[1]) { return v; }
};

// A generated epilogue (host/epilogue.cpp) provides: NX operands to read per element (prefetch /
// prefetch4 = four consecutive elements as 16-byte loads), compute() = the consumer's value for one
// element given the contraction result v = acc + bias, STORE_C = whether v itself is stored too, OUT =
// the index in a.epi[] of the tensor the consumer writes.  The kernel does the stores.
// PRED >= 0 (a "predicate tensor", host/plan_epilogue.cpp): every later reader of the contraction result v only asks
// one yes / no question of it (the relu gradient's `0 <= h`, dnn.nim:26-27 differentiated), so instead of v the kernel
// stores predicate(v), one bit per element — bit (idx & 31) of word idx >> 5 of a.epi[PRED], zeroed before the run —
// and the readers fetch the bit: 1/32 of the bytes in both directions.
// RD_N > 0 (a "row product", host/plan_epilogue.cpp fold_row_products): the NEXT layer's contraction
// `out2[y, x] ++= res[y, it] * W2[it, x]` (dense, dnn.nim:19-24) with x < RD_N <= 16 columns runs on the rows of the
// consumer's result while they sit in LDS on their way out (wide-store pass of a 256 x 256 tile): a.epi[RD_W] = W2
// (row stride RD_LDW), a.epi[RD_OUT] = out2 (row stride RD_LDO, zeroed before the run), a.epi[RD_BIAS] = its bias or
// RD_BIAS < 0.  Every N-tile adds its 256-column partial product to out2 with one float atomic per element; the host
// folds only when there are at most TWO N-tiles, so an element is 0 + p + q in either order: the same bits.
template <class Epi>
__device__ __forceinline__ void epi_apply(const GemmArgs& a, long idx, float v, const float (&x)[Epi::NX]) {
  const float r = Epi::compute(a, idx, v, x);
  if (Epi::STORE_C) a.C[idx] = v;
  static_cast<float*>(a.epi[Epi::OUT])[idx] = r;
  if constexpr (Epi::PRED >= 0) {  // element-wise path (ragged tiles, unaligned operands): one atomic OR per set bit
    if (Epi::predicate(v)) atomicOr(static_cast<unsigned*>(a.epi[Epi::PRED]) + (idx >> 5), 1u << (idx & 31));
  }
}

// Row stride (in floats) of an LDS operand tile [BK][stride].
//  - m|n-contiguous operand: 16-byte ds_write_b128 rows, no padding needed.
//  - k-contiguous operand (transposed while staging): BK/4 lanes share a row and write
//    k = 4c+j, so a half-wave covers R = 128/BK rows; stride % 8 == R/4 puts its 32 lanes on 32
//    distinct banks (BK 16 -> +2, BK 32 -> +1, BK 8 -> +4).
template <int BMN, int BK, bool KC>
struct LdsStride {
  static constexpr int value = KC ? BMN + 32 / BK : BMN;
};

// One operand tile: BMN rows/cols along m|n, BK along k.
//   KC == true : global element (mn, k) at  base[mn * ld + k]   (k contiguous)
//   KC == false: global element (mn, k) at  base[k * ld + mn]   (m|n contiguous)
//   CONV && KC : element (m, k) of the virtual im2col matrix, read straight from the image:
//       base[((n*H + y + dy)*W + x + dx)*C + c],  m = (n*Ho + y)*Wo + x,  k = (dy*FW + dx)*C + c
//     (dnn.nim:45-49: images[image, y + dy, x + dx, chan], valid padding, stride 1).
//   CONV && !KC: the same matrix as the k x n operand of the filter-gradient contraction
//     gF[f, (dy,dx,c)] = sum_pixels gOut[pixel, f] * im2col[pixel, (dy,dx,c)]: k = pixel, n = tap.
template <int BMN, int BK, int NT, bool KC, int VEC, bool EDGE, bool CONV>
struct TileLoader {
  static constexpr int STRIDE = LdsStride<BMN, BK, KC>::value;
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
        const long img = m / (a.cHo * a.cWo);
        const unsigned rem = (unsigned)(m - img * (a.cHo * a.cWo));
        const unsigned y = rem / (unsigned)a.cWo, x = rem % (unsigned)a.cWo;
        if (KC) {
          row_off[i] = ((img * a.cH + y) * a.cW + x) * a.cC;
        } else {  // n = (dy*FW + dx)*C + c is fixed per thread: its offset inside the window
          const unsigned C = (unsigned)a.cC, FW = (unsigned)a.cFW;
          const unsigned tap = (unsigned)m / C, c = (unsigned)m % C;
          row_off[i] = (long)((tap / FW) * (unsigned)a.cW + tap % FW) * C + c;
        }
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
      if (CONV && !KC) {
        // k -> output pixel (n, y, x) -> offset of its window's top-left input pixel
        const unsigned hw = (unsigned)(a.cHo * a.cWo);
        const unsigned img = (unsigned)gk / hw, rem = (unsigned)gk % hw;
        const unsigned y = rem / (unsigned)a.cWo, x = rem % (unsigned)a.cWo;
        p = base + (((long)img * a.cH + y) * a.cW + x) * a.cC + row_off[i];
      } else if (CONV) {
        // k -> (dy, dx, c).  The k-tile origin k0 is block-uniform: when C is a multiple of BK the
        // whole tile lies inside one filter tap and the split is scalar work; otherwise a
        // 32-bit division per chunk (64-bit division is a ~100 instruction software routine).
        const unsigned C = (unsigned)a.cC, FW = (unsigned)a.cFW;
        unsigned tap, c;
        if (C % BK == 0) {
          tap = (unsigned)k0 / C;
          c = (unsigned)k0 % C + (unsigned)k;
        } else {
          tap = (unsigned)gk / C;
          c = (unsigned)gk % C;
        }
        const unsigned dy = tap / FW, dx = tap % FW;
        p = base + row_off[i] + ((long)(dy * (unsigned)a.cW + dx) * C + c);
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

// Tile id -> tile origin: groups of 8 tile rows, column-major inside a group, so co-resident tiles share
// A row-panels and B column-panels inside one L2.
template <int BM, int BN>
__device__ __forceinline__ void tile_origin(int wgid, int rows_m, int tiles_n, long& m_blk, long& n_blk) {
  constexpr int GROUP = 8;
  const int per_group = GROUP * tiles_n;
  const int group = wgid / per_group;
  const int first_m = group * GROUP;
  const int gsize = min(rows_m - first_m, GROUP);
  const int in_group = wgid % per_group;
  m_blk = (long)(first_m + in_group % gsize) * BM;
  n_blk = (long)(in_group / gsize) * BN;
}

// Which rows (columns) of its sub-tile a wave's accumulator block mi (ni) holds.  For an operand that sits
// in LDS as [k][m|n] rows (m|n-contiguous in memory) the blocks are INTERLEAVED: lane i of block mi owns row
// CNT * i + mi, so the CNT values a lane needs for one k are adjacent in LDS and arrive with ONE
// ds_read_b64 / b128 instead of CNT ds_read_b32 (the weight-gradient contractions, both operands of this
// kind, issued 24 LDS reads per 32 MFMAs; 6 now).  k-contiguous operands ([m|n][16] rows, one b128 per
// row already) keep whole 32-row blocks.  Only INTERIOR tiles interleave: a ragged tile keeps whole blocks so
// that the blocks outside the problem can be skipped (an interleaved block always has some row inside).
// The epilogue follows whichever map the main loop of its tile used.
template <bool KC, int CNT>
struct Interleaved {
  static constexpr bool value = !KC && (CNT == 2 || CNT == 4);
};
template <int CNT>
__device__ __forceinline__ int sub_index(bool il, int block, int lane31) {
  return il ? CNT * lane31 + block : block * 32 + lane31;
}

template <int BM, int BN, int WM, int WN>
struct Geometry {
  static constexpr int WAVES = (BM / WM) * (BN / WN);
  static constexpr int NT = WAVES * 64;
};

// K loop of one block.  E = predicate every global load against the problem edges.
// ABL (tuning harness only; the library always instantiates 0): bit 0 = no LDS fragment reads in
// the k loop, bit 1 = no global loads / LDS stores after the first tile, bit 2 = no barriers,
// bit 3 = no LDS stores only, bit 4 = no global loads only, bit 6 = software-pipelined LDS-DMA loop (bits 0-2 ablate its reads / DMA / barriers).
template <int BM, int BN, int BK, int WM, int WN, bool A_KC, bool B_KC, int VEC, bool E, int CONV, int ABL>
__device__ __forceinline__ void gemm_mainloop(const GemmArgs& a, float* lds, f32x16 (&acc)[WM / 32][WN / 32], long m_blk,
                                              long n_blk, long k_begin, long k_end, int nk, int tid, int wm0, int wn0) {
  constexpr int NT = Geometry<BM, BN, WM, WN>::NT;
  constexpr int MI = WM / 32, NI = WN / 32;
  // VEC = 41: 16-byte loads for A, element loads for B (a bias-sized B whose rows of N = 10 floats
  // are not 16-byte aligned must not force the big operand onto 4-byte loads)
  constexpr int VA = VEC == 41 ? 4 : VEC, VB = VEC == 41 ? 1 : VEC;
  using LoadA = TileLoader<BM, BK, NT, A_KC, VA, E, CONV == 1>;
  using LoadB = TileLoader<BN, BK, NT, B_KC, VB, E, CONV == 2>;
  constexpr int SA = LoadA::STRIDE, SB = LoadB::STRIDE;
  constexpr int BUF = BK * (SA + SB);  // one stage: A tile then B tile
  const int lane = tid & 63;

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

  constexpr bool AIL = !E && Interleaved<A_KC, MI>::value, BIL = !E && Interleaved<B_KC, NI>::value;
  const int a_off = (lane >> 5) * SA + wm0 + (AIL ? MI * (lane & 31) : (lane & 31));
  const int b_off = (lane >> 5) * SB + wn0 + (BIL ? NI * (lane & 31) : (lane & 31));
  constexpr int A_STEP = AIL ? 1 : 32, B_STEP = BIL ? 1 : 32;  // distance between a lane's blocks in an LDS row

  // Edge tiles: 32x32 sub-blocks of the wave tile that lie completely outside the problem are
  // skipped (wave-uniform branch), so a ragged M or N costs matrix work at 32-row granularity
  // instead of tile granularity (M = 784 with 128-row tiles: 800 rows of work, not 896).
  unsigned live = 0xffffffffu;
  if (E) {
    const long m_w = __builtin_amdgcn_readfirstlane((int)min(a.M - m_blk - wm0, (long)BM));
    const long n_w = __builtin_amdgcn_readfirstlane((int)min(a.N - n_blk - wn0, (long)BN));
    live = 0;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j)
        if ((AIL ? i : i * 32) < m_w && (BIL ? j : j * 32) < n_w) live |= 1u << (i * NI + j);
  }

  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    const bool more = (ABL & 2) ? false : kt + 1 < nk;
    if (more && !(ABL & 16)) {
      const long k0 = k_begin + (long)(kt + 1) * BK;
      la.load(a, a.A, a.lda, m_blk, k0, a.M, k_end, tid);
      lb.load(a, a.B, a.ldb, n_blk, k0, a.N, k_end, tid);
    }
    const float* as = lds + cur * BUF + a_off;
    const float* bs = lds + cur * BUF + BK * SA + b_off;
    // software pipeline over the k-steps of this tile: fragments of step kk+1 are in flight
    // while the MFMAs of step kk issue
    float av[2][MI], bv[2][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i) av[0][i] = as[i * A_STEP];
#pragma unroll
    for (int j = 0; j < NI; ++j) bv[0][j] = bs[j * B_STEP];
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      const int c = kk & 1, n = c ^ 1;
      if ((ABL & 1) && kk + 1 < BK / 2) {
#pragma unroll
        for (int i = 0; i < MI; ++i) av[n][i] = av[c][i];
#pragma unroll
        for (int j = 0; j < NI; ++j) bv[n][j] = bv[c][j];
      } else if (kk + 1 < BK / 2) {
#pragma unroll
        for (int i = 0; i < MI; ++i) av[n][i] = as[(kk + 1) * 2 * SA + i * A_STEP];
#pragma unroll
        for (int j = 0; j < NI; ++j) bv[n][j] = bs[(kk + 1) * 2 * SB + j * B_STEP];
      }
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
          if (!E || (live >> (i * NI + j) & 1))
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c][i], bv[c][j], acc[i][j], 0, 0, 0);
    }
    if (more && !(ABL & 8)) {
      la.store(lds + (cur ^ 1) * BUF, tid);
      lb.store(lds + (cur ^ 1) * BUF + BK * SA, tid);
    }
    if (!(ABL & 4)) __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// LDS-DMA variant of the K loop (interior tiles, 16-byte aligned operands, BK = 16 or 32).
//
// Ablation of the register-staged loop (tools/gemm_tune.hip, 4096^3) shows the VGPR -> LDS stores
// are its largest single cost (128x128: 144.6 TF without them vs 133.4 with; 256x256: 143.6 vs
// 137.5): ds_write moves at most ~80 B/clk/CU and blocks the operand reads of the other waves
// meanwhile.  `global_load_lds_dwordx4` writes the tile into LDS from the memory pipeline instead:
// no staging registers, no ds_write.  Its LDS image is lane-linear (wave-uniform base + lane*16 B),
// so the layouts are chosen to be lane-linear:
//   * m|n-contiguous operand: [k][m|n] rows, one 1 KiB instruction covers 256 consecutive floats;
//     fragments are ds_read_b32 of 32 consecutive dwords (conflict free).
//   * k-contiguous operand: [m|n][16] rows of four 16-byte chunks; chunk c of row r is stored in
//     slot c ^ ((r >> 2) & 3) — the permutation is applied to the per-lane GLOBAL address, the LDS
//     side stays linear — and fragments are ds_read_b128: one read feeds four MFMA k-steps, and the
//     16 lanes of every ds_read_b128 service group fall on 16 distinct 16-byte slots.
// MFMA k assignment inside a 16-deep tile: step (pp, j), pp in {0,1}, j in 0..3, multiplies
// k = 8*pp + j (lanes 0-31) and k = 8*pp + 4 + j (lanes 32-63); any assignment is valid as long as
// both operands use the same one.
// BK = 32 (used by the convolution: more matrix work per barrier) has eight chunks per row and
// the slot permutation c ^ ((r >> 1) & 7), with the same conflict-free property.
// CONV: the k-contiguous A operand is the virtual im2col matrix, its rows gathered from the NHWC
// image (requires C % BK == 0 so that a k-tile lies inside one filter tap).
// CLAMP (ragged tiles): rows past the end of the operand re-read its last row (k-contiguous) or
// its last 16-byte column chunk (m|n-contiguous) — always valid memory; what they produce lands
// in accumulator rows / columns the epilogue never stores.  Past the end of K both operands re-read
// their last k, and the main loop zeroes the A side of those k in LDS.
// The barrier that publishes LDS-DMA data to the other waves of the block.  An LDS-DMA's bytes are
// ordered for a ds_read only by the ISSUING wave's vmcnt wait followed by a barrier the reader has
// passed: every wave waits for its own loads BEFORE the barrier.  __syncthreads() alone does not say
// so — offline hipcc happens to emit `s_waitcnt vmcnt(0)` in front of its s_barrier while an LDS-DMA
// is pending, the hiprtc build of the same source puts it AFTER the barrier, in front of the wave's
// first ds_read (own loads only).  A block then reads tiles other waves' loads have not delivered yet:
// the first k-tile of a generated-epilogue contraction came out with stale LDS bytes whenever nothing
// delayed the first read (1500 x 96 x 20 on 64 x 64 tiles: wrong 32-column stripes in most runs; with
// more k-tiles or larger tiles the address arithmetic of the next prefetch usually hid it — the rare
// wrong update of round 1).  The wait is spelled out, with a memory clobber so that no load moves across.
__device__ __forceinline__ void dma_publish_barrier() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}

// KCLAMP = false (with CLAMP, plain operands): the k-tiles this loader is asked for are whole — only rows / columns
// are clamped, and those do not change from k-tile to k-tile: the clamped element offset of every chunk is
// computed once (init) and issue() adds k0 like the interior loader does, so the compiler turns it into the
// same pointer bumps.  (Clamping inside issue() is ~60 VALU instructions with two 64-bit multiplies per
// k-tile, in FRONT of the fragment reads: a ragged tile that is alone on its CU ran ~25 % slower per k-tile.)
// A block-uniform pointer, in scalar registers for certain.  A buffer-resource operand that the compiler finds in vector
// registers — a uniform value that went through a 64-bit VALU compare, such as min(mn0, limit - 4) — is made scalar with
// a "waterfall" loop around EVERY load (v_readfirstlane x 4, two 64-bit compares, exec juggling: 14 instructions per 1 KiB
// piece; the ragged tiles of the cfg-5 weight gradient, which live on their loads, ran 10 % slower).
__device__ __forceinline__ const float* uniform_pointer(const float* p) {
  const unsigned long v = reinterpret_cast<unsigned long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return reinterpret_cast<const float*>(((unsigned long)hi << 32) | lo);
}

template <int BMN, int BK, int NT, bool KC, bool CONV, bool CLAMP = false, bool KCLAMP = CLAMP>
struct DmaLoader {
  static constexpr bool FIXED = CLAMP && !KCLAMP && !CONV;  // precomputed clamped offsets
  // Plain operands whose k-tiles are whole go through `buffer_load_dwordx4 ... lds` (MUBUF) instead of
  // `global_load_lds_dwordx4` (FLAT): the descriptor's base is the tile's origin (scalar, bumped per k-tile), the
  // per-lane part is a 32-bit byte offset computed ONCE (no 64-bit vector adds per k-tile), and — what the
  // software-pipelined loop needs — the compiler's wait-count pass treats a pending FLAT LDS-DMA as touching both
  // counters and turns every `s_waitcnt lgkmcnt(n)` into lgkmcnt(0) while one is in flight, i.e. a wave could never
  // wait for its older fragment reads only.  The host keeps leading dimensions below 2^21 floats on this path.
  static constexpr bool BUFD = !CONV && !(CLAMP && KCLAMP);
  unsigned virtual_row = 0;  // FIXED: bit t = chunk t of this lane is (the start of) the virtual row of ones
  unsigned voff[(BK * BMN / 256 + NT / 64 - 1) / (NT / 64)];  // BUFD: byte offset of chunk t from the tile origin
  long mn_base = 0;  // BUFD: row / column of the tile origin: mn0, or the last valid row / chunk when the tile starts past the
                     // end (the virtual row of ones as the first row of a tile of its own) — offsets must not be negative

  static constexpr int INSTRS = BK * BMN / 256;  // 1 KiB wave instructions per tile
  static constexpr int WAVES = NT / 64;
  static constexpr int PER_WAVE = (INSTRS + WAVES - 1) / WAVES;
  static constexpr int CHUNKS = BK / 4;  // 16-byte chunks per k-contiguous row

  long row_off[PER_WAVE];  // CONV: element offset of this lane's output pixel's top-left input pixel
  // CONV, filter-gradient operand, whole k-tiles: the k loop walks the output pixels in order, so the window origin of
  // this lane's pixel is carried from k-tile to k-tile (x += BK with carries into y and the image) instead of being
  // recomputed with three 32-bit divisions per 16-byte load (two loads per thread and k-tile next to 16 MFMAs per wave:
  // the divisions cost as many issue cycles as the matrix work).
  mutable long pix_off[PER_WAVE];  // element offset of that origin for the k-tile at pix_next
  mutable int pix_x[PER_WAVE], pix_y[PER_WAVE];
  mutable long pix_next = -1;

  __device__ __forceinline__ static int swizzle(int r) { return BK == 16 ? (r >> 2) & 3 : BK == 32 ? (r >> 1) & 7 : r & 15; }

  __device__ __forceinline__ void init(const GemmArgs& a, long mn0, int wave, int lane, long limit = 0, long ld = 0,
                                       bool ones = false) {
    if (BUFD) {
      mn_base = FIXED ? min(mn0, KC ? limit - 1 : limit - 4) : mn0;
#pragma unroll
      for (int t = 0; t < PER_WAVE; ++t) {
        const int q = (wave + t * WAVES) * 64 + lane;
        if (KC) {
          const int r = q / CHUNKS, slot = q % CHUNKS;
          const int c = slot ^ swizzle(r);
          const long row = FIXED ? min(mn0 + r, limit - 1) - mn_base : (long)r;  // rows past the end re-read the last one
          voff[t] = (unsigned)((row * ld + c * 4) * 4);
          if (FIXED && ones && mn0 + r == limit) virtual_row |= 1u << t;
        } else {
          constexpr int CPR = BMN / 4;
          const int k = q / CPR, col = (q % CPR) * 4;
          const long cc = FIXED ? min(mn0 + col, limit - 4) - mn_base : (long)col;  // columns past the end: the last chunk
          voff[t] = (unsigned)(((long)k * ld + cc) * 4);
          if (FIXED && ones && mn0 + col == limit) virtual_row |= 1u << t;
        }
      }
      return;
    }
    if (CONV && KC) {
#pragma unroll
      for (int t = 0; t < PER_WAVE; ++t) {
        const int q = (wave + t * WAVES) * 64 + lane;
        const long m = mn0 + q / CHUNKS;
        const long img = m / (a.cHo * a.cWo);
        const unsigned rem = (unsigned)(m - img * (a.cHo * a.cWo));
        const unsigned y = rem / (unsigned)a.cWo, x = rem % (unsigned)a.cWo;
        row_off[t] = ((img * a.cH + y) * a.cW + x) * a.cC;
      }
    } else if (CONV) {
      // filter-gradient operand (k = output pixel, n = (dy, dx, c)): this lane's n is fixed, so is
      // its offset inside the window
#pragma unroll
      for (int t = 0; t < PER_WAVE; ++t) {
        constexpr int CPR = BMN / 4;
        const int q = (wave + t * WAVES) * 64 + lane;
        long n = mn0 + (q % CPR) * 4;
        if (CLAMP) n = min(n, limit - 4);
        const unsigned C = (unsigned)a.cC, FW = (unsigned)a.cFW;
        const unsigned tap = (unsigned)n / C, c = (unsigned)n % C;
        row_off[t] = (long)((tap / FW) * (unsigned)a.cW + tap % FW) * C + c;
      }
    }
  }

  // BUFD: wave instruction t (0 .. PER_WAVE - 1) of this wave's share alone — the pipelined loop spreads a tile's loads
  // over an MFMA group instead of issuing them back to back (a wave that sits in the memory pipeline's issue queue
  // behind the other waves' loads cannot issue its MFMAs: every wave of the block did that right after the barrier).
  __device__ __forceinline__ void issue_one(int t, const float* __restrict__ base, long ld, long mn0, long k0, float* tile,
                                            int wave, const float* ones = nullptr) const {
#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC_RTC__)
    const int instr = wave + t * WAVES;
    if (INSTRS % WAVES != 0 && instr >= INSTRS) return;
    const float* origin = uniform_pointer(base + (KC ? mn_base * ld + k0 : k0 * ld + mn_base));
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(origin), (short)0, -1, 0x00020000);
    __attribute__((address_space(3))) void* dst = (__attribute__((address_space(3))) void*)(tile + instr * 256);
    if (FIXED && ones) {
      const float* src = reinterpret_cast<const float*>(reinterpret_cast<const char*>(origin) + voff[t]);
      if (virtual_row >> t & 1) src = KC ? ones : ones + 4;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, dst, 16, 0, 0);
    } else {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, dst, 16, voff[t], 0, 0, 0);
    }
#endif
  }

  // issue this wave's share of the tile whose origin is (mn0, k0) into `tile` (LDS, lane-linear)
  __device__ __forceinline__ void issue(const GemmArgs& a, const float* __restrict__ base, long ld, long mn0, long k0,
                                        float* tile, int wave, int lane, long limit = 0, long k_lim = 0,
                                        const float* ones = nullptr) const {
#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC_RTC__)  // (the host pass of an offline build knows no buffer-resource type)
    if constexpr (BUFD) {
      const float* origin = uniform_pointer(base + (KC ? mn_base * ld + k0 : k0 * ld + mn_base));  // block-uniform: scalar registers
      const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(origin), (short)0, -1, 0x00020000);
#pragma unroll
      for (int t = 0; t < PER_WAVE; ++t) {
        const int instr = wave + t * WAVES;
        if (INSTRS % WAVES != 0 && instr >= INSTRS) break;
        __attribute__((address_space(3))) void* dst = (__attribute__((address_space(3))) void*)(tile + instr * 256);
        if (FIXED && ones) {
          // (block-uniform) a tile with the virtual row of ones: those lanes fetch {1,1,1,1} / {1,0,0,0} — another base
          // address, i.e. a per-lane pointer: the FLAT form of the load (two exec-masked buffer loads per piece, one per
          // descriptor, cost the load-bound ragged tiles of the cfg-5 weight gradient 8 %)
          const float* src = reinterpret_cast<const float*>(reinterpret_cast<const char*>(origin) + voff[t]);
          if (virtual_row >> t & 1) src = KC ? ones : ones + 4;
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, dst, 16, 0, 0);
        } else {
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, dst, 16, voff[t], 0, 0, 0);
        }
      }
      return;
    }
#endif
    long tap_off = 0;
    if (CONV && KC) {  // block-uniform: scalar work
      const unsigned C = (unsigned)a.cC, FW = (unsigned)a.cFW;
      const unsigned tap = (unsigned)k0 / C, c0 = (unsigned)k0 % C;
      tap_off = (long)((tap / FW) * (unsigned)a.cW + tap % FW) * C + c0;
    }
    if constexpr (CONV && !KC && !CLAMP) {
      if (k0 != pix_next) {  // (block-uniform) the first k-tile of this block: one full decode
#pragma unroll
        for (int t = 0; t < PER_WAVE; ++t) {
          constexpr int CPR = BMN / 4;
          const int q = (wave + t * WAVES) * 64 + lane;
          const unsigned gk = (unsigned)(k0 + q / CPR);
          const unsigned hw = (unsigned)(a.cHo * a.cWo);
          const unsigned img = gk / hw, rem = gk % hw;
          pix_y[t] = (int)(rem / (unsigned)a.cWo);
          pix_x[t] = (int)(rem % (unsigned)a.cWo);
          pix_off[t] = (((long)img * a.cH + pix_y[t]) * a.cW + pix_x[t]) * a.cC;
        }
      }
    }
#pragma unroll
    for (int t = 0; t < PER_WAVE; ++t) {
      const int instr = wave + t * WAVES;
      if (INSTRS % WAVES != 0 && instr >= INSTRS) break;
      const int q = instr * 64 + lane;  // 16-byte chunk index inside the tile
      const float* src;
      if (FIXED) {
        src = base + row_off[t] + (KC ? k0 : k0 * ld);
        if (ones && (virtual_row >> t & 1)) src = KC ? ones : ones + 4;  // {1,1,1,1} / {1,0,0,0}: 1 for every k
      } else if (KC) {
        const int r = q / CHUNKS, slot = q % CHUNKS;
        const int c = slot ^ swizzle(r);
        if (CONV)
          src = base + row_off[t] + tap_off + c * 4;
        else
          src = base + (CLAMP ? min(mn0 + r, limit - 1) : mn0 + r) * ld + (CLAMP ? min(k0 + c * 4, k_lim - 4) : k0 + c * 4);
        if (CLAMP && !CONV && ones && mn0 + r == limit) src = ones;  // the virtual row: 1 for every k
      } else {
        constexpr int CPR = BMN / 4;
        const int k = q / CPR, col = (q % CPR) * 4;
        if (CONV && !CLAMP) {
          src = base + pix_off[t] + row_off[t];
        } else if (CONV) {  // k -> output pixel (n, y, x) -> its window's top-left input pixel
          const unsigned gk = (unsigned)(CLAMP ? min(k0 + k, k_lim - 1) : k0 + k);
          const unsigned hw = (unsigned)(a.cHo * a.cWo);
          const unsigned img = gk / hw, rem = gk % hw;
          const unsigned y = rem / (unsigned)a.cWo, x = rem % (unsigned)a.cWo;
          src = base + (((long)img * a.cH + y) * a.cW + x) * a.cC + row_off[t];
        } else {
          src = base + (CLAMP ? min(k0 + k, k_lim - 1) : k0 + k) * ld + (CLAMP ? min(mn0 + col, limit - 4) : mn0 + col);
          if (CLAMP && ones && mn0 + col == limit) src = ones + 4;  // {1, 0, 0, 0}: the virtual row starts this chunk
        }
      }
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(tile + instr * 256), 16, 0, 0);
    }
    if constexpr (CONV && !KC && !CLAMP) {  // on to the pixels of the next k-tile
      const int Wo = (int)a.cWo, Ho = (int)a.cHo;
      const long row_step = (a.cW - a.cWo) * a.cC, image_step = (a.cH - a.cHo) * a.cW * a.cC;
#pragma unroll
      for (int t = 0; t < PER_WAVE; ++t) {
        pix_x[t] += BK;
        pix_off[t] += (long)BK * a.cC;
        while (pix_x[t] >= Wo) {
          pix_x[t] -= Wo;
          pix_off[t] += row_step;
          if (++pix_y[t] >= Ho) {
            pix_y[t] = 0;
            pix_off[t] += image_step;
          }
        }
      }
      pix_next = k0 + BK;
    }
  }
};

template <int BM, int BN, int BK, int WM, int WN, bool A_KC, bool B_KC, int CONV, bool CL = false, bool IL = !CL, bool KCL = CL,
          int ABL = 0>
__device__ __forceinline__ void gemm_mainloop_dma(const GemmArgs& a, float* lds, f32x16 (&acc)[WM / 32][WN / 32],
                                                  long m_blk, long n_blk, long k_begin, int nk, int tid, int wm0,
                                                  int wn0, long k_end = 0) {
  static_assert(BK == 16 || BK == 32, "LDS-DMA loop: BK is 16 or 32");
  constexpr int NT = Geometry<BM, BN, WM, WN>::NT;
  constexpr int MI = WM / 32, NI = WN / 32;
  constexpr int BUF = BK * (BM + BN);
  // KCL: the k range may end inside a k-tile (clamped k, zeroed tail); CL && !KCL: whole k-tiles of a ragged tile
  using DmaA = DmaLoader<BM, BK, NT, A_KC, CONV == 1, CL, KCL>;
  using DmaB = DmaLoader<BN, BK, NT, B_KC, CONV == 2, CL, KCL>;
  const int lane = tid & 63, wave = tid >> 6;
  const int i = lane & 31, hi = lane >> 5;
  // IL: interleaved accumulator rows / columns (see Interleaved).  The clamped loop runs blocked for tiles
  // that are ragged in M or N, interleaved for the last k-tile of a tile that is ragged at the end of K only.
  constexpr bool AIL = IL && Interleaved<A_KC, MI>::value, BIL = IL && Interleaved<B_KC, NI>::value;

  // ragged tile: 32x32 sub-blocks of the wave tile that lie outside the problem are skipped
  unsigned live = 0xffffffffu;
  if (CL) {
    const int m_w = __builtin_amdgcn_readfirstlane((int)min(a.M - m_blk - wm0, (long)BM));
    const int n_w = __builtin_amdgcn_readfirstlane((int)min(a.N - n_blk - wn0, (long)BN));
    live = 0;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
        if ((AIL ? mi : mi * 32) < m_w && (BIL ? ni : ni * 32) < n_w) live |= 1u << (mi * NI + ni);
  }

  DmaA da;
  DmaB db;
  da.init(a, m_blk, wave, lane, a.a_rows, a.lda, CL && a.ones_row);
  db.init(a, n_blk, wave, lane, a.N, a.ldb);
  const float* ones = CL && a.ones_row ? a.ones : nullptr;  // (ragged loop only: a tile with the virtual row is never interior)
  if (nk > 0) {
    da.issue(a, a.A, a.lda, m_blk, k_begin, lds, wave, lane, a.a_rows, k_end, ones);
    db.issue(a, a.B, a.ldb, n_blk, k_begin, lds + BK * BM, wave, lane, a.N, k_end);
  }
  const int k_tail = KCL ? (int)((k_end - k_begin) % BK) : 0;  // valid k of a ragged last k-tile (0 = full)
  dma_publish_barrier();

  // Fragments of k-group pp (8 k) of the tile at As: av[mi][j] / bv[ni][j] = this lane's A / B value of block mi / ni for
  // MFMA k-step j (k = 8 pp + j in lanes 0-31, 8 pp + 4 + j in lanes 32-63).
  auto fragments = [&](const float* As, int pp, float (&av)[MI][4], float (&bv)[NI][4]) {
    const float* Bs = As + BK * BM;
    if constexpr (AIL) {  // one 8- or 16-byte read per k brings this lane's value for every block
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        typedef float vecA __attribute__((ext_vector_type(MI)));
        const vecA v = *reinterpret_cast<const vecA*>(As + (8 * pp + j + 4 * hi) * BM + wm0 + MI * i);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) av[mi][j] = v[mi];
      }
    }
    if constexpr (BIL) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        typedef float vecB __attribute__((ext_vector_type(NI)));
        const vecB v = *reinterpret_cast<const vecB*>(Bs + (8 * pp + j + 4 * hi) * BN + wn0 + NI * i);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) bv[ni][j] = v[ni];
      }
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      if (AIL) break;
      const int row = wm0 + mi * 32 + i;
      if (A_KC) {
        const int slot = (2 * pp + hi) ^ DmaA::swizzle(row);
        const f32x4 v = *reinterpret_cast<const f32x4*>(As + row * BK + slot * 4);
        av[mi][0] = v[0];
        av[mi][1] = v[1];
        av[mi][2] = v[2];
        av[mi][3] = v[3];
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) av[mi][j] = As[(8 * pp + j + 4 * hi) * BM + row];
      }
    }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      if (BIL) break;
      const int col = wn0 + ni * 32 + i;
      if (B_KC) {
        const int slot = (2 * pp + hi) ^ DmaB::swizzle(col);
        const f32x4 v = *reinterpret_cast<const f32x4*>(Bs + col * BK + slot * 4);
        bv[ni][0] = v[0];
        bv[ni][1] = v[1];
        bv[ni][2] = v[2];
        bv[ni][3] = v[3];
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) bv[ni][j] = Bs[(8 * pp + j + 4 * hi) * BN + col];
      }
    }
  };
  auto multiply_step = [&](const float (&av)[MI][4], const float (&bv)[NI][4], int j) {  // MFMA k-step j of a k-group
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
        if (!CL || MI * NI == 1 || (live >> (mi * NI + ni) & 1))
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mi][j], bv[ni][j], acc[mi][ni], 0, 0, 0);
  };
  auto multiply = [&](const float (&av)[MI][4], const float (&bv)[NI][4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          // A wave with ONE block (64 x 64, 128 x 32 tiles) multiplies unconditionally: a test per MFMA is a scalar
          // branch per MFMA, with the fragment reads stuck in front of it — a ragged tile's waves ran ~25 % slower
          // per k-tile than an interior tile's, and a launch is as slow as its slowest block.  What a wave outside
          // the problem accumulates (clamped re-reads of valid data) is never stored.
          if (!CL || MI * NI == 1 || (live >> (mi * NI + ni) & 1))
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mi][j], bv[ni][j], acc[mi][ni], 0, 0, 0);
  };

  if constexpr (!KCL && (ABL & 64)) {
    // ---- software-pipelined loop (round 3; tuning harness only, ABL bit 6 — measured EQUAL to the straight loop, see below).  The straight loop below reads the fragments of a k-group and then
    // multiplies them; the compiler keeps that order, so every wave of the block asks LDS for its fragments at
    // the same moments (right after the barrier, and once per k-group) and the matrix pipes wait while LDS
    // serves 8 waves x 5-6 KiB: measured 0.89-0.90 MFMA-busy at 4096^3 although the loop holds nothing but
    // MFMAs and reads.  Here the reads of k-group g + 1 are issued BEFORE the MFMAs of group g (two fragment
    // sets, +24 registers), also across the tile boundary: after the barrier that publishes tile kt + 1 a wave
    // issues the DMA of tile kt + 2, the reads of (kt + 1, group 0), and only then the MFMAs of tile kt's last
    // group.  Same MFMAs on the same accumulators in the same order: results are bit-identical.
    // One barrier per k-tile as before: when a wave passes the barrier inside iteration kt, every wave has
    // completed its reads of tile kt (s_waitcnt lgkmcnt(0) in front of s_barrier) — its stage may be refilled —
    // and its own DMA of tile kt + 1 has landed (dma_publish_barrier).
    constexpr int NPP = BK / 8;
    static_assert(NPP % 2 == 0, "fragment sets alternate per k-group");
    float av[2][MI][4], bv[2][NI][4];
    auto issue_tile = [&](int kt) {  // DMA of tile kt into its stage
      const long k0 = k_begin + (long)kt * BK;
      float* stage = lds + (kt & 1) * BUF;
      da.issue(a, a.A, a.lda, m_blk, k0, stage, wave, lane, a.a_rows, k_end, ones);
      db.issue(a, a.B, a.ldb, n_blk, k0, stage + BK * BM, wave, lane, a.N, k_end);
    };
    if (nk > 0) {
      if (nk > 1) issue_tile(1);
      fragments(lds, 0, av[0], bv[0]);
      __builtin_amdgcn_sched_barrier(0);
    }
    // Every tile but the last: ... reads of the tile's later k-groups, barrier, DMA of tile kt + 2 (if any), first reads
    // of tile kt + 1, last MFMA group.  The last tile is a second copy of the body without the barrier part — inside one
    // body the skipped barrier is a merging path on which the older reads are the youngest, and the wait-count pass then
    // waits for everything (lgkmcnt(0)) in front of the last MFMA group of EVERY tile.
    for (int kt = 0; kt + 1 < nk; ++kt) {
      const float* As = lds + (kt & 1) * BUF;
#pragma unroll
      for (int pp = 0; pp + 1 < NPP; ++pp) {
        if (!(ABL & 1)) fragments(As, pp + 1, av[(pp + 1) & 1], bv[(pp + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        multiply(av[(ABL & 1) ? 0 : (pp & 1)], bv[(ABL & 1) ? 0 : (pp & 1)]);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (!(ABL & 4)) dma_publish_barrier();
      if (!(ABL & 1)) fragments(lds + ((kt + 1) & 1) * BUF, 0, av[0], bv[0]);
      __builtin_amdgcn_sched_barrier(0);
      // the last MFMA group of tile kt with the DMA of tile kt + 2 spread over it: one wave instruction behind each
      // k-step's MFMAs (see DmaLoader::issue_one).  (The MFMAs stay outside the `more` condition: MFMA groups in two
      // branches made the register allocator copy and spill accumulators where the branches meet.)
      const bool more = (ABL & 2) ? false : kt + 2 < nk;
      if constexpr (DmaA::BUFD && DmaB::BUFD && DmaA::PER_WAVE + DmaB::PER_WAVE <= 4) {
        const long k0 = k_begin + (long)(kt + 2) * BK;
        float* stage = lds + (kt & 1) * BUF;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          multiply_step(av[(ABL & 1) ? 0 : ((NPP - 1) & 1)], bv[(ABL & 1) ? 0 : ((NPP - 1) & 1)], j);
          __builtin_amdgcn_sched_barrier(0);
          if (more) {
            if (j < DmaA::PER_WAVE)
              da.issue_one(j, a.A, a.lda, m_blk, k0, stage, wave, ones);
            else if (j - DmaA::PER_WAVE < DmaB::PER_WAVE)
              db.issue_one(j - DmaA::PER_WAVE, a.B, a.ldb, n_blk, k0, stage + BK * BM, wave);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
        if (more) issue_tile(kt + 2);
        __builtin_amdgcn_sched_barrier(0);
        multiply(av[(NPP - 1) & 1], bv[(NPP - 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (nk > 0) {
      const float* As = lds + ((nk - 1) & 1) * BUF;
#pragma unroll
      for (int pp = 0; pp + 1 < NPP; ++pp) {
        fragments(As, pp + 1, av[(pp + 1) & 1], bv[(pp + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        multiply(av[pp & 1], bv[pp & 1]);
        __builtin_amdgcn_sched_barrier(0);
      }
      multiply(av[(NPP - 1) & 1], bv[(NPP - 1) & 1]);
    }
    __syncthreads();  // every wave is done reading: the epilogue (or a following loop) may reuse the stages
    return;
  }

  if constexpr (!KCL && NT >= 512 && BK / 8 >= 2 && !(ABL & 256)) {
    // ---- skewed waves (round 4).  A wave that issues anything between two MFMAs idles the matrix pipe of its SIMD for
    // that long — measured on a one-wave-per-SIMD kernel (conv2_gradf_halo.hip): 56 cycles per LDS-DMA piece, 7 per LDS
    // read, whether or not the instruction stands behind an MFMA — and the two waves that share a SIMD here (2 s and
    // 2 s + 1) run the same code between the same barriers, so they issue their loads and fragment reads at the same
    // moments and both stop multiplying: the three costs of the k loop ADD (972 us at 4096^3 against 903 without them).
    // Here the ODD waves run one k-group late: they read the last k-group of a tile in front of the barrier and multiply
    // it behind it, while the even waves issue their loads and reads; then the odd waves load and read while the even
    // ones multiply.  Same MFMAs on the same accumulators in the same k order: bit-identical.  LDS hazards are
    // unchanged — every read of a stage still precedes the barrier that frees it.  4096^3: 965 -> 944 us (145.7 TFLOP/s,
    // 0.926 of peak); with the UPPER HALF of the block late instead (waves 4 - 7: not SIMD neighbours) nothing changes
    // (966 us) — which is how the wave-to-SIMD assignment was found (ABL bit 9 of the tuning harness selects that split).
    constexpr int NPP = BK / 8;
    if (!a.no_skew && ((ABL & 512) ? wave >= NT / 128 : (wave & 1) != 0)) {
      float avp[MI][4], bvp[NI][4];
      for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt > 0) multiply(avp, bvp);   // tile kt - 1, last k-group
        if (kt + 1 < nk) {
          const long k0 = k_begin + (long)(kt + 1) * BK;
          float* nxt = lds + (cur ^ 1) * BUF;
          da.issue(a, a.A, a.lda, m_blk, k0, nxt, wave, lane, a.a_rows, k_end, ones);
          db.issue(a, a.B, a.ldb, n_blk, k0, nxt + BK * BM, wave, lane, a.N, k_end);
        }
        const float* As = lds + cur * BUF;
#pragma unroll
        for (int pp = 0; pp + 1 < NPP; ++pp) {
          float av[MI][4], bv[NI][4];
          fragments(As, pp, av, bv);
          multiply(av, bv);
        }
        fragments(As, NPP - 1, avp, bvp);
        dma_publish_barrier();
      }
      if (nk > 0) multiply(avp, bvp);
      return;
    }
  }
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) {
      const long k0 = k_begin + (long)(kt + 1) * BK;
      float* nxt = lds + (cur ^ 1) * BUF;
      da.issue(a, a.A, a.lda, m_blk, k0, nxt, wave, lane, a.a_rows, k_end, ones);
      db.issue(a, a.B, a.ldb, n_blk, k0, nxt + BK * BM, wave, lane, a.N, k_end);
    }
    if (KCL && k_tail != 0 && kt == nk - 1) {
      // ragged end of K: the loaders re-read the last valid k for the missing ones; zero them on
      // the A side so they contribute nothing
      float* At = lds + cur * BUF;
      const int width = BK - k_tail;
      for (int e = tid; e < BM * width; e += NT) {
        if (A_KC) {
          const int r = e / width, k = k_tail + e % width;
          At[r * BK + (((k >> 2) ^ DmaA::swizzle(r)) << 2) + (k & 3)] = 0.f;
        } else {
          At[k_tail * BM + e] = 0.f;
        }
      }
      __syncthreads();
    }
    const float* As = lds + cur * BUF;
#pragma unroll
    for (int pp = 0; pp < BK / 8; ++pp) {
      float av[MI][4], bv[NI][4];
      fragments(As, pp, av, bv);
      multiply(av, bv);
    }
    dma_publish_barrier();  // the next k-tile's loads (issued above) have landed; this stage is free for the one after
  }
}

// K loop of a block that carries the extra rows (GemmArgs::x_rows): the interior LDS-DMA loop of the TN form plus a
// [k][32] strip of A (rows m_blk + BM .., clamped at a_rows; it may hold the virtual row of ones) and one more accumulator
// block per wave: wave w multiplies the strip with columns [32 w, 32 w + 32) of the B tile.
template <int BM, int BN, int BK, int WM, int WN>
__device__ __forceinline__ void gemm_mainloop_dma_x(const GemmArgs& a, float* lds, f32x16 (&acc)[WM / 32][WN / 32], f32x16& accx,
                                                    long m_blk, long n_blk, long k_begin, int nk, int tid, int wm0, int wn0) {
  constexpr int NT = Geometry<BM, BN, WM, WN>::NT;
  constexpr int MI = WM / 32, NI = WN / 32;
  constexpr int BUF = BK * (BM + BN);
  static_assert(Interleaved<false, MI>::value && Interleaved<false, NI>::value && BN / 32 == NT / 64,
                "extra rows: the 256 x 256 tile of the TN form (one 32-column block of the strip per wave)");
  using DmaA = DmaLoader<BM, BK, NT, false, false>;
  using DmaB = DmaLoader<BN, BK, NT, false, false>;
  using DmaX = DmaLoader<32, BK, NT, false, false, true, false>;
  const int lane = tid & 63, wave = tid >> 6;
  const int i = lane & 31, hi = lane >> 5;
  float* xs = lds + 2 * BUF;  // two stages of the strip, [BK][32] each
  DmaA da;
  DmaB db;
  DmaX dx;
  da.init(a, m_blk, wave, lane, a.a_rows, a.lda);
  db.init(a, n_blk, wave, lane, a.N, a.ldb);
  dx.init(a, m_blk + BM, wave, lane, a.a_rows, a.lda, a.ones_row != 0);
  const float* ones = a.ones_row ? a.ones : nullptr;
  auto issue = [&](int kt) {
    const long k0 = k_begin + (long)kt * BK;
    float* stage = lds + (kt & 1) * BUF;
    da.issue(a, a.A, a.lda, m_blk, k0, stage, wave, lane);
    db.issue(a, a.B, a.ldb, n_blk, k0, stage + BK * BM, wave, lane, a.N);
    dx.issue(a, a.A, a.lda, m_blk + BM, k0, xs + (kt & 1) * BK * 32, wave, lane, a.a_rows, 0, ones);
  };
  if (nk > 0) issue(0);
  dma_publish_barrier();
  // fragments of k-group pp of the tile at stage `st` (A / B interleaved: one read per k brings every block's value; the
  // strip's A value and this wave's 32 columns of B for the ninth block)
  auto fragments = [&](int st, int pp, float (&av)[MI][4], float (&bv)[NI][4], float (&ax)[4], float (&bx)[4]) {
    const float* As = lds + st * BUF;
    const float* Bs = As + BK * BM;
    const float* Xs = xs + st * BK * 32;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = 8 * pp + j + 4 * hi;
      typedef float vecA __attribute__((ext_vector_type(MI)));
      typedef float vecB __attribute__((ext_vector_type(NI)));
      const vecA va = *reinterpret_cast<const vecA*>(As + k * BM + wm0 + MI * i);
      const vecB vb = *reinterpret_cast<const vecB*>(Bs + k * BN + wn0 + NI * i);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) av[mi][j] = va[mi];
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) bv[ni][j] = vb[ni];
      ax[j] = Xs[k * 32 + i];
      bx[j] = Bs[k * BN + wave * 32 + i];
    }
  };
  auto multiply = [&](const float (&av)[MI][4], const float (&bv)[NI][4], const float (&ax)[4], const float (&bx)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mi][j], bv[ni][j], acc[mi][ni], 0, 0, 0);
      accx = __builtin_amdgcn_mfma_f32_32x32x2f32(ax[j], bx[j], accx, 0, 0, 0);
    }
  };
  constexpr int NPP = BK / 8;
  if (NPP >= 2 && (wave & 1) && !a.no_skew) {
    // the odd waves run one k-group late (gemm_mainloop_dma, "skewed waves"): same MFMAs in the same order
    float avp[MI][4], bvp[NI][4], axp[4], bxp[4];
    for (int kt = 0; kt < nk; ++kt) {
      if (kt > 0) multiply(avp, bvp, axp, bxp);
      if (kt + 1 < nk) issue(kt + 1);
#pragma unroll
      for (int pp = 0; pp + 1 < NPP; ++pp) {
        float av[MI][4], bv[NI][4], ax[4], bx[4];
        fragments(kt & 1, pp, av, bv, ax, bx);
        multiply(av, bv, ax, bx);
      }
      fragments(kt & 1, NPP - 1, avp, bvp, axp, bxp);
      dma_publish_barrier();
    }
    if (nk > 0) multiply(avp, bvp, axp, bxp);
    return;
  }
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) issue(kt + 1);
#pragma unroll
    for (int pp = 0; pp < NPP; ++pp) {
      float av[MI][4], bv[NI][4], ax[4], bx[4];
      fragments(kt & 1, pp, av, bv, ax, bx);
      multiply(av, bv, ax, bx);
    }
    dma_publish_barrier();
  }
}

// MINB: blocks per CU the register allocator must leave room for (waves/SIMD = MINB * WAVES / 4).
// EDGE kernels still run their interior tiles on the unpredicated loop.
// DMA: interior tiles use the LDS-DMA loop (requires VEC == 4, BK in {16, 32}; with CONV the host
// checks C % BK == 0).
// CONV: 0 = plain operands, 1 = A is the im2col matrix of an NHWC image (forward convolution),
// 2 = B is that matrix with k = output pixel, n = tap (filter-gradient contraction).
template <int BM, int BN, int BK, int WM, int WN, bool A_KC, bool B_KC, int VEC, bool EDGE, int CONV, int ABL, bool DMA,
          class Epi, bool XR = false>
__device__ __forceinline__ void gemm_block(const GemmArgs& a) {
  constexpr int WAVES_N = BN / WN;
  constexpr int MI = WM / 32, NI = WN / 32;
  constexpr int SA = LdsStride<BM, BK, A_KC>::value, SB = LdsStride<BN, BK, B_KC>::value;

  // (XR: + the strip's stages; a row product: two padded stages of parked rows + the 256 x 16 piece of W2 + one hand-over block)
  constexpr bool RD = Epi::RD_N > 0;
  constexpr int RD_FLOATS = RD ? 2 * (BM / WM) * 32 * (BN + 4) + BN * 16 + 4 * 64 * 4 : 0;
  constexpr int OPERAND_FLOATS = 2 * BK * (SA + SB) + (XR ? 2 * BK * 32 : 0);
  __shared__ __attribute__((aligned(16))) float lds[RD_FLOATS > OPERAND_FLOATS ? RD_FLOATS : OPERAND_FLOATS];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm0 = (wave / WAVES_N) * WM;
  const int wn0 = (wave % WAVES_N) * WN;
  if (a.prio) __builtin_amdgcn_s_setprio(3);
  // (nothing of the trace stays live across the k loop: the pointer is re-read from the kernel arguments at every stamp —
  // kept in registers it cost the extra-row kernel, which has none to spare, 10 - 15 us)
  auto stamp = [&](int k) {
    if (a.trace && lane == 0) a.trace[((long)blockIdx.x * (BM / WM) * WAVES_N + wave) * 4 + k] = __builtin_readcyclecounter();
  };
  stamp(0);

  // ---- tile coordinates: XCD-contiguous ids, then 8-row groups so co-resident tiles share
  //      A row-panels and B column-panels inside one L2.
  // Work items are (k-split, tile) pairs, split-major.  xcd_remap hands every XCD a contiguous
  // range of them, so the tiles of one k-split — which all stream the same rows of A and B — run
  // on one XCD at the same time and share its L2 (with the splits spread round-robin over the
  // XCDs a weight-gradient contraction fetched its operands 4.9x: 1.65 GB for 339 MB).
  const int rows_m = a.edge_splits > 0 ? a.tiles_m - 1 : a.tiles_m;  // tile rows cut into a.splits slices
  const int nwg = rows_m * a.tiles_n;
  const int nsplit = a.splits > 1 ? a.splits : 1;
  const int nfull = nwg * nsplit;
  const int tail_first = nwg - a.tail_tiles;  // (tail mode: a.splits <= 1, a.edge_splits == 0)
  // Tail mode: block ids below tail_first are the whole tiles, the rest the short tail slices, each
  // set spread over the XCDs on its own (one contiguous remap of both would hand some XCDs nothing but
  // whole tiles and others nothing but slices: measured 2.3 ms instead of 1.4 for 4100^3).
  const int work = a.tail_tiles > 0
                       ? ((int)blockIdx.x < tail_first
                              ? xcd_remap(blockIdx.x, tail_first)
                              : tail_first + xcd_remap(blockIdx.x - tail_first, a.tail_tiles * a.tail_splits))
                       : xcd_remap(blockIdx.x, nfull + a.tiles_n * a.edge_splits);
  int split;
  long m_blk, n_blk, k_slice;
  int tail_slab = -1;  // >= 0: this block writes a whole-tile slab
  if (a.tail_tiles > 0) {
    if (work < tail_first) {
      split = 0;
      tile_origin<BM, BN>(work, rows_m, a.tiles_n, m_blk, n_blk);
      k_slice = a.k_per_split;
    } else {
      const int w = work - tail_first;
      const int t = w / a.tail_splits;
      split = w - t * a.tail_splits;
      tile_origin<BM, BN>(tail_first + t, rows_m, a.tiles_n, m_blk, n_blk);
      k_slice = a.tail_k_per_split;
      tail_slab = w;
    }
  } else if (work < nfull) {
    split = work / nwg;
    tile_origin<BM, BN>(work - split * nwg, rows_m, a.tiles_n, m_blk, n_blk);
    k_slice = a.k_per_split;
  } else {
    const int w = work - nfull;
    split = w / a.tiles_n;
    m_blk = (long)rows_m * BM;
    n_blk = (long)(w - split * a.tiles_n) * BN;
    k_slice = a.k_per_split_edge;
  }

  const long k_begin = (long)split * k_slice;
  const long k_end = min(a.K, k_begin + k_slice);
  const int nk = (int)((k_end - k_begin + BK - 1) / BK);

  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if constexpr (RD) {
    // a row product's piece of W2 (rows n_blk .. n_blk + BN, 16 columns) goes to LDS behind the two stages of parked
    // rows — outside the operand buffers, so it is fetched here, under the k loop's prologue, and published by the
    // barriers of the k loop: [k / 4][16][k % 4] (one ds_read_b128 per 16-k window and lane)
    constexpr int RT_ = (BM / WM) * 32, NTHREADS = Geometry<BM, BN, WM, WN>::NT;
    float* w2s = lds + 2 * RT_ * (BN + 4);
    const float* w2 = static_cast<const float*>(a.epi[Epi::RD_W]);
    // (all loads first, then the stores: as one `w2s[...] = cond ? w2[...] : 0` loop the eight trips of a thread were eight
    // dependent round trips to L2 — EG_GEMM_TRACE, round 6: the k loop of the fused forward product began 7 800 cycles after
    // the wave's start, 1 600 in the plain kernel)
    constexpr int TRIPS = (BN * 16 + NTHREADS - 1) / NTHREADS;
    float wv[TRIPS];
#pragma unroll
    for (int t = 0; t < TRIPS; ++t) {
      const int e = tid + t * NTHREADS, k = (e >> 4) < BN ? (e >> 4) : BN - 1, c = e & 15;
      const float v = w2[(n_blk + k) * (long)Epi::RD_LDW + (c < Epi::RD_N ? c : 0)];
      wv[t] = c < Epi::RD_N ? v : 0.f;
    }
#pragma unroll
    for (int t = 0; t < TRIPS; ++t) {
      const int e = tid + t * NTHREADS, k = e >> 4, c = e & 15;
      if (e < BN * 16) w2s[(((k >> 2) * 16 + c) << 2) + (k & 3)] = wv[t];
    }
  }
  static_assert(!DMA || VEC == 4, "LDS-DMA loop: 16-byte aligned operands");
  const bool whole_k = (k_end - k_begin) % BK == 0;
  const bool interior = m_blk + BM <= a.a_rows && n_blk + BN <= a.N && whole_k;
  // A tile that is whole in M and N but ends inside a k-tile (K = 4100: every tile) runs its whole k-tiles
  // on the interior loop and only the last one on the clamped loop (4096 x 4096 x 4100 took 1060 us with
  // every k-tile clamped, against 973 us for K = 4112).
  const bool k_tail_only = EDGE && DMA && CONV == 0 && !whole_k && k_end > k_begin && m_blk + BM <= a.a_rows && n_blk + BN <= a.N;
  stamp(1);
  bool done = false;
  if constexpr (XR) {
    // extra rows: the blocks of the last tile row multiply and store the strip [tiles_m * BM, M) as well (split-K only:
    // the strip goes to this block's slab)
    if (a.x_rows > 0 && m_blk == (long)(a.tiles_m - 1) * BM) {
      f32x16 accx;
#pragma unroll
      for (int r = 0; r < 16; ++r) accx[r] = 0.f;
      gemm_mainloop_dma_x<BM, BN, BK, WM, WN>(a, lds, acc, accx, m_blk, n_blk, k_begin, nk, tid, wm0, wn0);
      float* slab = a.partial + (long)split * a.M * a.N;
      const long n = n_blk + wave * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long m = m_blk + BM + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m < a.M) slab[m * a.N + n] = accx[r];
      }
      done = true;
    }
  }
  if constexpr (DMA) {  // kernels without the DMA loop (tuning harness: BK = 8) never instantiate it
    if (done) {
    } else if (!EDGE || interior || k_tail_only) {
      const int n_main = (EDGE && k_tail_only) ? nk - 1 : nk;
      if (n_main > 0)
        gemm_mainloop_dma<BM, BN, BK, WM, WN, A_KC, B_KC, CONV, false, true, false, ABL>(a, lds, acc, m_blk, n_blk, k_begin, n_main, tid, wm0, wn0);
      if constexpr (EDGE && CONV == 0) {
        if (k_tail_only)
          gemm_mainloop_dma<BM, BN, BK, WM, WN, A_KC, B_KC, CONV, true, true>(a, lds, acc, m_blk, n_blk,
                                                                               k_begin + (long)n_main * BK, 1, tid, wm0, wn0, k_end);
      }
      done = true;
    } else if (CONV != 1) {
      // ragged in M or N (and maybe K): still the LDS-DMA loop, with clamped addresses and a zeroed K tail.
      // Plain operands: the whole k-tiles on the loader with precomputed clamped offsets, the last, partial
      // one (if any) on the loader that clamps k as well.
      if constexpr (CONV == 0) {
        const int n_whole = nk <= 0 ? 0 : (whole_k ? nk : nk - 1);
        if (n_whole > 0)
          gemm_mainloop_dma<BM, BN, BK, WM, WN, A_KC, B_KC, CONV, true, false, false, ABL>(a, lds, acc, m_blk, n_blk, k_begin, n_whole, tid,
                                                                                       wm0, wn0, k_end);
        if (n_whole < nk)
          gemm_mainloop_dma<BM, BN, BK, WM, WN, A_KC, B_KC, CONV, true, false, true>(a, lds, acc, m_blk, n_blk,
                                                                                      k_begin + (long)n_whole * BK, 1, tid, wm0, wn0, k_end);
      } else {
        gemm_mainloop_dma<BM, BN, BK, WM, WN, A_KC, B_KC, CONV, true>(a, lds, acc, m_blk, n_blk, k_begin, nk, tid, wm0, wn0,
                                                                        k_end);
      }
      done = true;
    }
  }
  if constexpr (!DMA || (EDGE && CONV == 1)) {  // register-staged loop: unaligned operands, ragged im2col tiles
    if (!done) {
      if (EDGE && !interior)
        gemm_mainloop<BM, BN, BK, WM, WN, A_KC, B_KC, VEC, true, CONV, ABL>(a, lds, acc, m_blk, n_blk, k_begin, k_end, nk,
                                                                           tid, wm0, wn0);
      else
        gemm_mainloop<BM, BN, BK, WM, WN, A_KC, B_KC, VEC, false, CONV, ABL>(a, lds, acc, m_blk, n_blk, k_begin, k_end, nk,
                                                                            tid, wm0, wn0);
    }
  }

  if constexpr ((ABL & 128) != 0) {
    // tuning harness only: no epilogue at all (the accumulators are folded into one value that is never equal to the
    // constant, so the matrix work stays) — what a step would cost if its tile stores were free
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    if (s == 1.2345678e-30f) a.C[0] = s;
    return;
  }
  stamp(2);
  // ---- epilogue.  32x32 accumulator block: register r of lane l holds
  //      row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31.
  const bool to_partial = a.tail_tiles > 0 ? tail_slab >= 0 : a.partial != nullptr;
  // a tail block's slab is addressed with global (m, n) like C: slab[(m - m_blk) * BN + (n - n_blk)]
  float* out = tail_slab >= 0 ? a.partial + (long)tail_slab * BM * BN - (m_blk * BN + n_blk)
                              : (to_partial ? a.partial + (long)split * a.M * a.N : a.C);
  const long ldo = tail_slab >= 0 ? BN : (to_partial ? a.N : a.ldc);
  const bool accumulate = !to_partial && a.accumulate;
  const bool has_bias = !to_partial && a.bias != nullptr;
  const bool whole_tile = m_blk + BM <= a.M && n_blk + BN <= a.N;
  // rows / columns of the wave sub-tile that block (i, j) register r of this lane holds (see Interleaved)
  const bool ail = Interleaved<A_KC, MI>::value && (!EDGE || interior || k_tail_only);  // block-uniform
  const bool bil = Interleaved<B_KC, NI>::value && (!EDGE || interior || k_tail_only);
  if (a.wide_store && (!EDGE || whole_tile)) {  // block-uniform
    // ---- whole tile, through LDS (GemmArgs::wide_store).  Pass i: every wave parks block row i of its
    // sub-tile (32 rows x WN columns) at [wave row * 32 + row][wn0 + col]; then all threads walk the
    // (BM / WM) * 32 staged rows in 16-byte chunks, one full tile row per wave instruction.
    constexpr int WAVES_M = BM / WM;
    constexpr int RT = WAVES_M * 32;            // staged rows per pass
    constexpr int C4 = BN / 4;                  // 16-byte chunks per staged row
    constexpr int NT_ = WAVES_M * WAVES_N * 64;
    static_assert(RD || RT * BN <= 2 * BK * (SA + SB), "the staged rows fit the operand buffers");
    static_assert(!RD || (BM == 256 && BN == 256 && WAVES_M * WAVES_N == 8), "row products ride on the 256 x 256 tile");
    constexpr int PS = RD ? BN + 4 : BN;        // row stride of the parked rows (padded: the row product reads them as MFMA fragments)
    float* w2s = lds + 2 * RT * PS;             // RD: W2[n_blk .. n_blk + BN) x 16 as [k / 4][16][k % 4], staged before the k loop
    const int wmi = wave / WAVES_N;
    static_assert(NT_ % C4 == 0, "a thread keeps its column chunk");
    constexpr int NQ = (RT * C4 + NT_ - 1) / NT_;  // chunks per thread and pass
    const int c4 = tid % C4;
    const long n = n_blk + c4 * 4;
    f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
    if (has_bias) b4 = *reinterpret_cast<const f32x4*>(a.bias + n);
    float rd_half[4] = {0.f, 0.f, 0.f, 0.f};  // RD: this wave's half of the row product of the previous pass
    float* rdx = w2s + BN * 16;                 // RD: [4 blocks][64 lanes][4]: the second column half, waves 4 .. 7 -> 0 .. 3
    // out2 += partial product of pass `pass` (+ bias from the first N-tile).  Lane l of wave w < 4 holds column l % 16 of
    // staged rows 16 w + 4 (l / 16) + v.  Float atomics without return — nothing waits for them (a compare-and-swap loop
    // costs a memory round trip behind the tile's own store burst: +14 us per tile).  gfx950's global_atomic_add_f32
    // honours the denormal mode (tests/test_gpu_epilogue.py holds a denormal result to the bit).
    // (the second layer's bias: loaded ONCE per tile here, not in front of the atomics of every pass — waves 0 .. 3 waited
    // a memory round trip per pass for it and the other four waited for them at the pass's barrier)
    float bias2 = 0.f;
    if constexpr (RD) {
      if (Epi::RD_BIAS >= 0 && n_blk == 0 && wave < 4 && (lane & 15) < Epi::RD_N)
        bias2 = static_cast<const float*>(a.epi[Epi::RD_BIAS >= 0 ? Epi::RD_BIAS : 0])[lane & 15];
    }
    auto rd_send = [&](int pass) {
      if constexpr (RD) {
        const int r = lane & 15, g = lane >> 4;
        if (wave < 4 && r < Epi::RD_N) {
          typedef float rd4 __attribute__((ext_vector_type(4)));
          const rd4 other = *reinterpret_cast<const rd4*>(rdx + (wave * 64 + lane) * 4);
          float* out2 = static_cast<float*>(a.epi[Epi::RD_OUT]);
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const int srow = wave * 16 + 4 * g + v;
            const long m = m_blk + (long)(srow >> 5) * WM + sub_index<MI>(ail, pass, srow & 31);
            atomicAdd(out2 + m * (long)Epi::RD_LDO + r, (rd_half[v] + other[v]) + bias2);
          }
        }
      }
    };
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      // (RD: passes alternate between two stages, so the row product of pass i reads its rows while pass i + 1 parks)
      float* park = lds + (RD ? (i & 1) * RT * PS : 0);
      if (bil) {  // the lane's NI columns are adjacent: one 8- / 16-byte LDS write per row
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          typedef float vecN __attribute__((ext_vector_type(NI)));
          vecN v;
#pragma unroll
          for (int j = 0; j < NI; ++j) v[j] = acc[i][j][r];
          *reinterpret_cast<vecN*>(&park[(wmi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * PS + wn0 + NI * (lane & 31)]) = v;
        }
      } else {
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            park[(wmi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * PS + wn0 + j * 32 + (lane & 31)] = acc[i][j][r];
      }
      __syncthreads();
      if (RD && i > 0) rd_send(i - 1);
      // A thread's chunks of one pass: q = q0 + tid, q0 a multiple of the block size — the same 16-byte column chunk c4 of
      // NQ different rows (C4 divides the block size), so the bias chunk is loaded once.  The loads of all NQ chunks (a
      // generated epilogue's operands, an accumulating launch's old values) are issued BEFORE the first store: loads and
      // stores share vmcnt on gfx9, so a load issued behind a store makes its consumer wait for that store as well —
      // chunk by chunk (load, wait, store, load, ...) a tile's stores went out one store latency apart.
      auto index_of = [&](int c) {  // flat output index of this thread's chunk c of pass i
        const int row = (c * NT_ + tid) / C4;
        return (m_blk + (long)(row >> 5) * WM + sub_index<MI>(ail, i, row & 31)) * ldo + n;
      };
      // (the 256 x 256 tile has the registers to hold a whole pass; tiles that run four waves per SIMD batch four chunks)
      constexpr int GROUP = BM * BN >= 256 * 256 ? NQ : (NQ < 4 ? NQ : 4);
      const bool packed = C4 % 8 == 0 && (ldo & 31) == 0;  // predicate bits: eight neighbouring lanes hold one 32-bit word
#pragma unroll
      for (int g0 = 0; g0 < NQ; g0 += GROUP) {
      f32x4 x4[GROUP][Epi::NX];
      f32x4 old[GROUP];
      unsigned nibs[GROUP];   // predicate bits of the group's chunks (packed words: folded behind the loop, all chunks together)
#pragma unroll
      for (int c = 0; c < GROUP; ++c) nibs[c] = 0;
#pragma unroll
      for (int c = g0; c < g0 + GROUP && c < NQ; ++c) {
        if ((RT * C4) % NT_ != 0 && c * NT_ + tid >= RT * C4) break;
        if (Epi::ACTIVE) Epi::prefetch4(a, index_of(c), x4[c - g0]);
        else if (accumulate) old[c - g0] = *reinterpret_cast<const f32x4*>(out + index_of(c));
      }
#pragma unroll
      for (int c = g0; c < g0 + GROUP && c < NQ; ++c) {
        if ((RT * C4) % NT_ != 0 && c * NT_ + tid >= RT * C4) break;
        const int row = (c * NT_ + tid) / C4;
        const long idx = index_of(c);
        f32x4 v = *reinterpret_cast<const f32x4*>(&park[row * PS + c4 * 4]);
        if (Epi::ACTIVE) {
          f32x4 res;
          unsigned nibble = 0;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float x[Epi::NX];
#pragma unroll
            for (int o = 0; o < Epi::NX; ++o) x[o] = x4[c - g0][o][e];
            v[e] = v[e] + b4[e];
            res[e] = Epi::compute(a, idx + e, v[e], x);
            if constexpr (Epi::PRED >= 0) nibble |= (Epi::predicate(v[e]) ? 1u : 0u) << e;
          }
          if constexpr (Epi::PRED >= 0) {
            unsigned* bits = static_cast<unsigned*>(a.epi[Epi::PRED]);
            if (packed) {
              nibs[c - g0] = nibble;   // (folded below)
            } else if (nibble) {
              atomicOr(bits + (idx >> 5), nibble << (idx & 31));  // (idx is a multiple of 4: a nibble never straddles words)
            }
          }
          if constexpr (RD) *reinterpret_cast<f32x4*>(&park[row * PS + c4 * 4]) = res;  // (this thread's own chunk)
          if (a.nt_store) {
            if (Epi::STORE_C) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(a.C + idx));
            __builtin_nontemporal_store(res, reinterpret_cast<f32x4*>(static_cast<float*>(a.epi[Epi::OUT]) + idx));
          } else {
            if (Epi::STORE_C) *reinterpret_cast<f32x4*>(a.C + idx) = v;
            *reinterpret_cast<f32x4*>(static_cast<float*>(a.epi[Epi::OUT]) + idx) = res;
          }
        } else {
          f32x4* p = reinterpret_cast<f32x4*>(out + idx);
          if (accumulate) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (old[c - g0][e] + v[e]) + b4[e];
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] + b4[e];
          }
          if (a.nt_store) __builtin_nontemporal_store(v, p);
          else *p = v;
        }
      }
      if constexpr (Epi::ACTIVE && Epi::PRED >= 0) {
        if (packed) {
          // (whole tiles: n_blk and the rows' starts are multiples of 32) eight neighbouring lanes hold the eight nibbles of a
          // 32-bit word: three butterfly steps, one lane stores the word.  The steps run over ALL chunks of the group together
          // (round 6: chunk by chunk, each behind its guarded store, every one of the 3 x 8 crossbar shuffles of a pass
          // waited out its own latency — the pattern EG_ROW_TRACE found in the row groups).
          unsigned* bits = static_cast<unsigned*>(a.epi[Epi::PRED >= 0 ? Epi::PRED : 0]);
          unsigned w[GROUP];
#pragma unroll
          for (int c = 0; c < GROUP; ++c) w[c] = nibs[c] << (4 * (tid & 7));
#pragma unroll
          for (int step = 1; step <= 4; step <<= 1)
#pragma unroll
            for (int c = 0; c < GROUP; ++c) w[c] |= __shfl_xor(w[c], step, 64);
          if ((tid & 7) == 0) {
#pragma unroll
            for (int c = g0; c < g0 + GROUP && c < NQ; ++c) {
              if ((RT * C4) % NT_ != 0 && c * NT_ + tid >= RT * C4) break;
              bits[index_of(c) >> 5] = w[c - g0];
            }
          }
        }
      }
      }
      __syncthreads();
      if constexpr (RD) {
        // Row product of the RT = 64 rows of this pass: wave w takes the 16 rows of block w % 4 and the column half w / 4,
        // v_mfma_f32_16x16x4_f32 over 16-wide windows — lane (r = l % 16, g = l / 16) reads res[row r][16 s + 4 g .. + 3]
        // and W2[16 s + 4 g .. + 3][column r] as one ds_read_b128 each (stride BN + 4: 4 r + g covers the 64 banks once);
        // MFMA j of a window multiplies k = 16 s + 4 g + j on both sides.  Two accumulators, windows alternating.
        // Waves 4 .. 7 hand their half over through LDS (rdx); waves 0 .. 3 pick it up behind the NEXT barrier every wave
        // passes anyway (the one that publishes the next pass's parked rows) and send the sums off there.
        typedef float rd4 __attribute__((ext_vector_type(4)));
        const int r = lane & 15, g = lane >> 4, mb = wave & 3, kh = wave >> 2;
        const float* arow = park + (mb * 16 + r) * PS + 4 * g + kh * (BN / 2);
        const float* brow = w2s + ((g * 16 + r) << 2) + kh * (BN / 2) * 16;
        // (four accumulators — two windows x even / odd k of a window: a chain of 8 dependent MFMAs each instead of 16; the
        // instruction's result is not ready for the next one of its chain when that is the next but one in line)
        rd4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f}, d2 = {0.f, 0.f, 0.f, 0.f}, d3 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s2 = 0; s2 < BN / 32; s2 += 2) {
          const rd4 a0 = *reinterpret_cast<const rd4*>(arow + 16 * s2);
          const rd4 b0 = *reinterpret_cast<const rd4*>(brow + 256 * s2);
          const rd4 a1 = *reinterpret_cast<const rd4*>(arow + 16 * s2 + 16);
          const rd4 b1 = *reinterpret_cast<const rd4*>(brow + 256 * s2 + 256);
#pragma unroll
          for (int j = 0; j < 4; j += 2) {
            d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[j], b0[j], d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[j], b1[j], d1, 0, 0, 0);
            d2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[j + 1], b0[j + 1], d2, 0, 0, 0);
            d3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[j + 1], b1[j + 1], d3, 0, 0, 0);
          }
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) rd_half[v] = (d0[v] + d2[v]) + (d1[v] + d3[v]);
        if (kh == 1) *reinterpret_cast<rd4*>(rdx + (mb * 64 + lane) * 4) = rd4{rd_half[0], rd_half[1], rd_half[2], rd_half[3]};
      }
    }
    if constexpr (RD) {
      __syncthreads();
      rd_send(MI - 1);
    }
    if (a.trace) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamp(3);
    return;
  }
  if constexpr (RD) __builtin_trap();  // the host launches a row product only where every tile takes the wide-store pass
#pragma unroll
  for (int i = 0; i < MI; ++i) {
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const long n = n_blk + wn0 + sub_index<NI>(bil, j, lane & 31);
      const bool n_ok = !EDGE || n < a.N;
      float bias = 0.f;
      if (has_bias && n_ok) bias = a.bias[n];
      // row of register r: m_base + RS * ((r & 3) + 8 * (r >> 2))
      const int RS = ail ? MI : 1;
      const long m_base = m_blk + wm0 + (ail ? i + MI * 4 * (lane >> 5) : i * 32 + 4 * (lane >> 5));
      float* col = out + n;
      if (Epi::ACTIVE) {
        float x[16][Epi::NX];
        if (!EDGE || whole_tile) {  // whole tile inside: branch-free, loads batched
#pragma unroll
          for (int r = 0; r < 16; ++r) Epi::prefetch(a, (m_base + RS * ((r & 3) + 8 * (r >> 2))) * ldo + n, x[r]);
#pragma unroll
          for (int r = 0; r < 16; ++r)
            epi_apply<Epi>(a, (m_base + RS * ((r & 3) + 8 * (r >> 2))) * ldo + n, acc[i][j][r] + bias, x[r]);
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const long m = m_base + RS * ((r & 3) + 8 * (r >> 2));
            if (m >= a.M || !n_ok) continue;
            Epi::prefetch(a, m * ldo + n, x[r]);
            epi_apply<Epi>(a, m * ldo + n, acc[i][j][r] + bias, x[r]);
          }
        }
      } else if (!EDGE || whole_tile) {  // branch-free: the 16 stores (and loads) of a block are issued back to back
        if (accumulate) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const long m = m_base + RS * ((r & 3) + 8 * (r >> 2));
            col[m * ldo] = (col[m * ldo] + acc[i][j][r]) + bias;
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) col[(m_base + RS * ((r & 3) + 8 * (r >> 2))) * ldo] = acc[i][j][r] + bias;
        }
      } else if (accumulate) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const long m = m_base + RS * ((r & 3) + 8 * (r >> 2));
          if (m >= a.M || !n_ok) continue;
          col[m * ldo] = (col[m * ldo] + acc[i][j][r]) + bias;
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const long m = m_base + RS * ((r & 3) + 8 * (r >> 2));
          if (m >= a.M || !n_ok) continue;
          col[m * ldo] = acc[i][j][r] + bias;
        }
      }
    }
  }
  if (a.trace) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  stamp(3);
}

// Waves per SIMD the register allocator must leave room for.  The ragged-tile variants of the
// 64-wide tiles need a few registers more than the 128 that four waves allow (they spilled 2-38
// VGPRs to scratch): three waves there.
// The unaligned (VEC = 1) variants stage element by element and need far more: two waves.
template <int BM, int BN, int WM, int WN, int MINB, bool EDGE, int VEC>
struct WavesPerSimd {
  static constexpr int plain = (MINB * Geometry<BM, BN, WM, WN>::WAVES + 3) / 4;
  static constexpr int value = (EDGE && VEC == 1 && BM * BN >= 128 * 64 && plain > 2) ? 2
                               : (EDGE && BN == 64 && BM >= 128 && plain == 4)        ? 3
                                                                                       : plain;
};

template <int BM, int BN, int BK, int WM, int WN, int MINB, bool A_KC, bool B_KC, int VEC, bool EDGE, int CONV,
          int ABL = 0, bool DMA = false, bool XR = false>
__global__ __launch_bounds__((Geometry<BM, BN, WM, WN>::NT), (WavesPerSimd<BM, BN, WM, WN, MINB, EDGE, VEC>::value)) void
gemm_f32_mfma_kernel(GemmArgs a) {
  gemm_block<BM, BN, BK, WM, WN, A_KC, B_KC, VEC, EDGE, CONV, ABL, DMA, EpiNone, XR>(a);
}

// ---- stream-K for 64 x 64 tiles (round 6) -------------------------------------------------------------------------------
// A mid-size product on 64 x 64 tiles leaves the CUs unevenly loaded: 1792^3 is 784 tiles on 256 CUs (3.06 blocks per CU: the
// launch is as long as the CUs that got four), 1152^3 324 tiles (1.27: as long as two) — 0.64 / 0.45 of peak where 2048^3,
// exactly four blocks per CU, runs at 0.83.  Here the launch is gridDim PERSISTENT blocks.  Block b first multiplies `rounds`
// whole tiles (b, b + gridDim, ...: stored as usual), then its share of the REMAINING tiles' (tile, k-tile) space, which the
// blocks divide evenly: units [b * per, (b + 1) * per) of it, walked tile by tile.  A piece of a tile goes to one of the
// block's two slabs — slot 0: the block's first piece, slot 1: its last — and gemm_streamk_fixup_kernel adds the pieces of
// every remaining tile in k order (block order), a fixed order: run-to-run identical.  (A first version shared ALL tiles'
// units: with about as many tiles as blocks nearly every tile was cut and the whole output travelled through the slabs —
// 1920^3 126.9 us against 121.8 for one block per tile.)  Whole 64 x 64 tiles, K a multiple of BK, 16-byte aligned operands,
// no generated epilogue (host: run_gemm).  rounds = a.splits, per = a.k_per_split (units of BK).
template <int BK, bool A_KC, bool B_KC>
__global__ __launch_bounds__(256, 4) void gemm_streamk_kernel(GemmArgs a) {
  constexpr int BM = 64, BN = 64, WM = 32, WN = 32;
  constexpr int SA = LdsStride<BM, BK, A_KC>::value, SB = LdsStride<BN, BK, B_KC>::value;
  __shared__ __attribute__((aligned(16))) float lds[2 * BK * (SA + SB)];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave >> 1) * WM, wn0 = (wave & 1) * WN;
  const int i = lane & 31, hi = lane >> 5;
  const int nk = (int)(a.K / BK);
  const int rounds = a.splits;
  const long tiles = (long)a.tiles_m * a.tiles_n, tiles_dp = (long)rounds * gridDim.x;
  // (blocks in XCD-contiguous order: neighbouring tiles / unit ranges share an L2)
  const long b = xcd_remap(blockIdx.x, gridDim.x);
  // register r of lane l: row (r & 3) + 8 (r >> 2) + 4 (l >> 5), column l & 31 of the wave's 32 x 32 block
  auto piece = [&](int tile, int k0, int cnt, float* slab) {
    long m_blk, n_blk;
    tile_origin<BM, BN>(tile, a.tiles_m, a.tiles_n, m_blk, n_blk);
    f32x16 acc[1][1];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.f;
    gemm_mainloop_dma<BM, BN, BK, WM, WN, A_KC, B_KC, 0, false, true, false, 0>(a, lds, acc, m_blk, n_blk, (long)k0 * BK, cnt, tid, wm0, wn0);
    if (slab == nullptr) {
      float* col = a.C + (m_blk + wm0 + 4 * hi) * a.ldc + n_blk + wn0 + i;
      const float bias = a.bias ? a.bias[n_blk + wn0 + i] : 0.f;
      if (a.accumulate) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float* p = col + (long)((r & 3) + 8 * (r >> 2)) * a.ldc;
          *p = (*p + acc[0][0][r]) + bias;
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) col[(long)((r & 3) + 8 * (r >> 2)) * a.ldc] = acc[0][0][r] + bias;
      }
    } else {
      float* dst = slab + (wm0 + 4 * hi) * BN + wn0 + i;
#pragma unroll
      for (int r = 0; r < 16; ++r) dst[((r & 3) + 8 * (r >> 2)) * BN] = acc[0][0][r];
    }
  };
  for (int r = 0; r < rounds; ++r) piece((int)((long)r * gridDim.x + b), 0, nk, nullptr);
  const long total = (tiles - tiles_dp) * nk;
  const long per = a.k_per_split;
  long u = b * per;
  const long u_end = total < u + per ? total : u + per;
  bool first = true;
  while (u < u_end) {
    const int t = (int)(u / nk);
    const int k0 = (int)(u - (long)t * nk);
    const int cnt = (int)((long)(nk - k0) < u_end - u ? nk - k0 : u_end - u);
    // (a remaining tile that one block owns completely is stored directly; the fix-up skips it)
    piece((int)(tiles_dp + t), k0, cnt, (k0 == 0 && cnt == nk) ? nullptr : a.partial + ((long)b * 2 + (first ? 0 : 1)) * (BM * BN));
    first = false;
    u += cnt;
  }
}

// One block per REMAINING tile: the sum of its pieces in block (= k) order.
__global__ __launch_bounds__(256) void gemm_streamk_fixup_kernel(const float* __restrict__ partial, float* C, const float* __restrict__ bias,
                                                                 long ldc, int tiles_m, int tiles_n, long tiles_dp, int nk, long per,
                                                                 int accumulate) {
  constexpr int BM = 64, BN = 64;
  const int t = blockIdx.x;                    // remaining tile t = tile tiles_dp + t
  const long u0 = (long)t * nk, u1 = u0 + nk;
  const long b_first = u0 / per, b_last = (u1 - 1) / per;
  if (b_first == b_last) return;   // one block owned the whole tile and stored it
  const int tile = (int)(tiles_dp + t);
  long m_blk, n_blk;
  {
    constexpr int GROUP = 8;
    const int per_group = GROUP * tiles_n, group = tile / per_group, first_m = group * GROUP;
    const int gsize = tiles_m - first_m < GROUP ? tiles_m - first_m : GROUP, in_group = tile % per_group;
    m_blk = (long)(first_m + in_group % gsize) * BM;
    n_blk = (long)(in_group / gsize) * BN;
  }
  typedef float v4 __attribute__((ext_vector_type(4)));
  for (int e = threadIdx.x; e < BM * BN / 4; e += 256) {
    v4 s = {0.f, 0.f, 0.f, 0.f};
    for (long bb = b_first; bb <= b_last; ++bb) {
      const int slot = (bb * per) / nk == t ? 0 : 1;   // the block's first piece, or its last
      s += *reinterpret_cast<const v4*>(partial + (bb * 2 + slot) * (BM * BN) + e * 4);
    }
    const int row = (e * 4) / BN, c = (e * 4) % BN;
    float* dst = C + (m_blk + row) * ldc + n_blk + c;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float bv = bias ? bias[n_blk + c + j] : 0.f;
      dst[j] = accumulate ? (dst[j] + s[j]) + bv : s[j] + bv;
    }
  }
}

// ---- contraction with a TINY K and a generated epilogue as a streaming kernel on the vector ALUs ----------------------
// The activation-gradient product of a classifier's last layer, ga[y, j] = sum_c gz[y, c] * W2[j, c] with 10 classes
// (dense, dnn.nim:19-24, differentiated: passes.nim:519-549), followed by relu's gradient in the epilogue: 65 536 x 512 x 10
// is 0.67 GFLOP and 134 MB of output — a store stream.  On the matrix tile (128 x 128, whole tiles through LDS) it writes
// at 3.1 TB/s (43 us).  Here a thread owns FOUR consecutive columns of a row: the K values of its columns of B sit in
// registers for the whole launch, a row's K values of A are the same for every thread of the row (scalar loads when a
// wave works on one row), the 4 x K multiply-adds run in k order (an fmaf chain, as the matrix core evaluates it), and
// the row leaves as nontemporal 16-byte stores, the epilogue's operands read the same way.  A first version (round 4:
// rows of a step a whole grid apart, eight blocks per CU) wrote at 3.1 TB/s like the tile and was dropped; what the
// round-6 stream probe (tools/hbm_probe.hip, PROBE_GH) showed to matter is the shape of the stream: a block owns ONE
// CONTIGUOUS run of rows, two rows' loads (U = 2 steps) are in flight before the first dependent store, up to 32 blocks
// per CU — 23.5 us = 6.0 TB/s for the same 134 MB.  Requirements (host, plan_fused): K <= 16, N % 4 = 0, 256 % (N / 4) = 0
// (a thread keeps its columns from row to row), 16-byte aligned C / epilogue operands, ldc % 4 = 0; no row product.
// Predicate bits of the result: eight neighbouring threads hold one word (N % 32 = 0: whole words), else atomic OR.
// TPR = N / 4 (threads per row) is a template argument and the block's run of rows arrives in a.k_per_split (the host
// divides): with both computed in the kernel — two 64-bit and three 32-bit divisions per block in front of its first
// load — the launch got SLOWER with more, shorter blocks (2 048 / 8 192 / 16 384 blocks: 27.7 / 41.6 / 57.2 us) where the
// probe with literal extents got faster (27.6 / 23.7 us).
template <int V>
struct NarrowTrip {
  static constexpr int value = V;
};
template <int K, int TPR, bool A_KC, bool B_KC, class Epi>
__device__ __forceinline__ void gemm_narrow_k_block(const GemmArgs& a) {
  const int tid = threadIdx.x;
  constexpr int tpr = TPR;                          // threads per row (divides 256)
  constexpr int rpb = 256 / tpr;                    // rows per block and step
  const int c4 = tid % tpr;
  const long n = (long)c4 * 4;
  float w[4][K];
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int k = 0; k < K; ++k) w[e][k] = B_KC ? a.B[(n + e) * a.ldb + k] : a.B[(long)k * a.ldb + n + e];
  f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
  if (a.bias) b4 = *reinterpret_cast<const f32x4*>(a.bias + n);
  const bool packed = (tpr & 7) == 0 && (a.ldc & 31) == 0;
  // A through the CONSTANT address space: a row's K values are one address for the whole wave, and only loads the compiler
  // knows to be invariant go through the scalar cache into SGPRs (a pointer out of the argument struct carries no such
  // promise; as vector loads the rows cost U x K VGPRs, the kernel needed 161 registers and three blocks shared a CU
  // instead of eight).  A is written by an earlier launch, never by this one.
  typedef const __attribute__((address_space(4))) float* ConstF;
  const ConstF Ap = (ConstF)(unsigned long)a.A;
  float* __restrict__ const Cp = a.C;
  float* __restrict__ const Op = static_cast<float*>(a.epi[Epi::OUT]);
  constexpr int U = 4;
  // the block's run of rows: a multiple of what it takes per trip (computed by the host, plan_fused)
  const long per = a.k_per_split;
  const long lo = (long)blockIdx.x * per;
  const long hi = a.M < lo + per ? a.M : lo + per;
  // UU rows per trip, every one of them inside [lo, hi): NO condition between the loads of a trip and its stores (with a
  // bounds test per row the compiler sank the later rows' loads behind the earlier rows' stores: one load in flight)
  auto trip = [&](long m0, auto uu_tag) {
    constexpr int UU = decltype(uu_tag)::value;
    float ar[UU][K];
    f32x4 x4[UU][Epi::NX];
#pragma unroll
    for (int u = 0; u < UU; ++u) {
      const long m = m0 + (long)u * rpb;
      if (tpr >= 64) {  // a wave works on ONE row: its K values of A arrive as scalar / uniform loads
        const long mu = ((long)__builtin_amdgcn_readfirstlane((int)(m >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)m);
#pragma unroll
        for (int k = 0; k < K; ++k) ar[u][k] = A_KC ? Ap[mu * a.lda + k] : Ap[(long)k * a.lda + mu];
      } else {
#pragma unroll
        for (int k = 0; k < K; ++k) ar[u][k] = A_KC ? Ap[m * a.lda + k] : Ap[(long)k * a.lda + m];
      }
      Epi::prefetch4(a, m * a.ldc + n, x4[u]);
    }
#pragma unroll
    for (int u = 0; u < UU; ++u) {
      const long m = m0 + (long)u * rpb;
      const long idx = m * a.ldc + n;
      f32x4 v, res;
      unsigned nibble = 0;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < K; ++k) acc = __builtin_fmaf(ar[u][k], w[e][k], acc);
        v[e] = acc + b4[e];
        float x[Epi::NX];
#pragma unroll
        for (int o = 0; o < Epi::NX; ++o) x[o] = x4[u][o][e];
        res[e] = Epi::compute(a, idx + e, v[e], x);
        if constexpr (Epi::PRED >= 0) nibble |= (Epi::predicate(v[e]) ? 1u : 0u) << e;
      }
      if constexpr (Epi::PRED >= 0) {
        unsigned* bits = static_cast<unsigned*>(a.epi[Epi::PRED >= 0 ? Epi::PRED : 0]);
        if (packed) {
          unsigned word = nibble << (4 * (tid & 7));
          word |= __shfl_xor(word, 1, 64);
          word |= __shfl_xor(word, 2, 64);
          word |= __shfl_xor(word, 4, 64);
          if ((tid & 7) == 0) bits[idx >> 5] = word;
        } else if (nibble) {
          atomicOr(bits + (idx >> 5), nibble << (idx & 31));
        }
      }
      if (Epi::STORE_C) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(Cp + idx));
      __builtin_nontemporal_store(res, reinterpret_cast<f32x4*>(Op + idx));
    }
  };
  long m0 = lo + tid / tpr;
  for (; m0 + (long)(U - 1) * rpb < hi; m0 += (long)rpb * U) trip(m0, NarrowTrip<U>());
  for (; m0 < hi; m0 += rpb) trip(m0, NarrowTrip<1>());   // (the ragged end of the last block)
}

// Second pass of split-K: C[m,n] = (accumulate ? C : 0) + sum_z partial[z][m][n] + bias[n],
// slabs added in increasing z (fixed order => run-to-run deterministic).
__global__ __launch_bounds__(256) void gemm_splitk_reduce_kernel(const float* __restrict__ partial, float* C,
                                                                 const float* __restrict__ bias, long M, long N,
                                                                 long ldc, int splits, int accumulate,
                                                                 long edge_row, int edge_splits) {
  // rows >= edge_row were cut into edge_splits slices only (GemmArgs::edge_splits)
  const long total = M * N;
  const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x, nthreads = (long)gridDim.x * blockDim.x;
  if ((N & 3) == 0 && (ldc & 3) == 0 && (reinterpret_cast<unsigned long>(C) & 15) == 0 &&
      (reinterpret_cast<unsigned long>(partial) & 15) == 0) {
    // 16 bytes per lane: a 4-wide group never straddles a row
    for (long i4 = tid; i4 < (total >> 2); i4 += nthreads) {
      const long i = i4 << 2, m = i / N, n = i % N;
      // four interleaved chains (slabs z = c, c + 4, ...), combined ((s0 + s1) + (s2 + s3)): eight loads in flight per
      // thread instead of a serial walk (67 MB of slabs for the cfg-5 weight gradient: 16.5 us = 4.1 TB/s)
      f32x4 c4[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) c4[c] = f32x4{0.f, 0.f, 0.f, 0.f};
      const int nz = m >= edge_row ? edge_splits : splits;
      int z = 0;
      for (; z + 8 <= nz; z += 8) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(partial + (long)(z + u) * total + i);
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
          for (int j = 0; j < 4; ++j) c4[u & 3][j] += v[u][j];
      }
      for (; z < nz; ++z) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(partial + (long)z * total + i);
#pragma unroll
        for (int j = 0; j < 4; ++j) c4[z & 3][j] += v[j];
      }
      f32x4 s;
#pragma unroll
      for (int j = 0; j < 4; ++j) s[j] = (c4[0][j] + c4[1][j]) + (c4[2][j] + c4[3][j]);
      f32x4* p = reinterpret_cast<f32x4*>(C + m * ldc + n);
      if (accumulate) {
        const f32x4 o = *p;
#pragma unroll
        for (int j = 0; j < 4; ++j) s[j] = o[j] + s[j];
      }
      if (bias) {
#pragma unroll
        for (int j = 0; j < 4; ++j) s[j] += bias[n + j];
      }
      *p = s;
    }
    return;
  }
  for (long i = tid; i < total; i += nthreads) {
    const long m = i / N, n = i % N;
    float s = 0.f;
    const int nz = m >= edge_row ? edge_splits : splits;
    for (int z = 0; z < nz; ++z) s += partial[(long)z * total + i];
    float* p = C + m * ldc + n;
    if (accumulate) s = *p + s;
    if (bias) s += bias[n];
    *p = s;
  }
}

// Second pass of the tail mode: C tile = (accumulate ? C : 0) + sum over the tile's slices + bias, slices
// added in increasing order (deterministic).  One block per (tail tile, 32-row band).
template <int BM, int BN>
__global__ __launch_bounds__(256) void gemm_tail_reduce_kernel(const float* __restrict__ partial, float* C,
                                                               const float* __restrict__ bias, long M, long N, long ldc,
                                                               int tiles_m, int tiles_n, int tail_tiles, int tail_splits,
                                                               int accumulate) {
  constexpr int BANDS = BM / 32;
  const int t = blockIdx.x / BANDS, band = blockIdx.x % BANDS;
  long m_blk, n_blk;
  tile_origin<BM, BN>(tiles_m * tiles_n - tail_tiles + t, tiles_m, tiles_n, m_blk, n_blk);
  const float* slabs = partial + (long)t * tail_splits * BM * BN;
  for (int e = threadIdx.x; e < 32 * BN; e += 256) {
    const int r = band * 32 + e / BN, c = e % BN;
    const long m = m_blk + r, n = n_blk + c;
    if (m >= M || n >= N) continue;
    float s = 0.f;
    for (int z = 0; z < tail_splits; ++z) s += slabs[(long)z * BM * BN + r * BN + c];
    float* p = C + m * ldc + n;
    if (accumulate) s = *p + s;
    if (bias) s += bias[n];
    *p = s;
  }
}

}  // namespace gemm
}  // namespace eg
