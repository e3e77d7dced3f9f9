// Wave-pair contraction kernel (round 4): 64 x 64 tiles of a plain f32 product whose tiles fill the chip at most
// once — 1024^3 is 256 blocks on 256 CUs.  The four-wave kernel of gemm_f32_mfma.hpp runs ONE wave per SIMD there, and a
// k-tile costs it 1.36x its matrix time.  Where that goes was measured on this kernel with parts of the loop removed
// (tools/gemm_pipe.hip, 1024 x 1024 x K, per 32 k): MFMAs alone 1042 cycles (= the matrix rate); + the barrier 132; + the
// fragment reads 175; both 365; the LDS-DMA loads on top of that: nothing (a loop without any load is as slow as the
// full loop, so deeper load pipelines buy nothing: three and five stages measured equal or slower than two).  Here
//   * every 32 x 32-blocked sub-tile of the output belongs to a PAIR of waves, 2 s and 2 s + 1 — the two waves that share
//     SIMD s — which split every k-tile between them (first half of its k-groups / second half);
//   * the odd wave runs one k-group late (it reads its last k-group of a tile in front of the barrier and multiplies it
//     behind it), so the pair is out of phase where both would otherwise issue their reads right behind the barrier
//     (in phase the pair is no faster than one wave);
//   * k-tiles are 64 deep (KB): half the barriers per MFMA; two stages (ST) of 32 KB;
//   * the barrier is the bare instruction behind an explicit s_waitcnt (see publish below);
//   * the pair's two accumulator sets meet on their way out: both waves park their blocks in LDS (two copies of the
//     staged rows), and the 16-byte row walk of the wide-store pass adds them, even wave's value + odd wave's value.
// 1024^3: 22.1 -> 20.9 us (NN, NT), 23.1 -> 19.8 (TN) with sustained clocks = 0.65 - 0.69 of the MFMA peak; 512^3 12.3 ->
// 11.6.  Ragged tiles and a K that ends inside a k-tile: the EDGE form (template parameter; 1000^3 28.1 -> 23.1 us).  A first
// version handled the last k-tile inside the loop (other loaders, a zero fill and a barrier under a block-uniform condition):
// whole problems ran 14 % slower on it (1024 x 1024 x 4096: 83.3 against 73.3 us) although the condition was never true —
// the tile now has a stage of its own, is loaded first and multiplied last, and the loop is the whole-tile loop (73.9).  The same design on 128 x 128 tiles: 2048^3 127.2 us against 129.8 for the four-wave 64 x 64 kernel — not taken.
// Results are deterministic (fixed order) but not bit-identical to the four-wave kernels: an output element is the sum of
// KW f32 chains (one per wave of its sub-tile, added in wave order) instead of one.
// ABL (tuning harness only; the library instantiates 0): bit 0 no fragment reads / MFMAs, bit 1 no loads behind the
// prologue, bit 2 no barriers, bit 3 fragments read once, bit 8 in phase.
// Reference semantics: c[y, x] ++= a[y, it] * b[it, x] (base.nim:27-28), like every variant of the contraction kernel.
#pragma once
#include "gemm_f32_mfma.hpp"

namespace eg {
namespace gemm {

// KW: waves per sub-tile (2: a pair; 8: one 32 x 32 tile per block, every k-tile split eight ways — small outputs, where 64 x 64
// tiles would leave three quarters of the chip idle: 512^3 is 64 of them, and 256 of 32 x 32)
template <int BM, int BN, int WM, int WN, int ST = 3, int KB = 32, bool EDGE = false, int KW = 2>
struct PairGeometry {
  static constexpr int SUB = (BM / WM) * (BN / WN);  // sub-tiles = wave groups
  static constexpr int NT = SUB * 64 * KW;
  static constexpr int BK = KB, STAGES = ST;
  static constexpr int BUF = BK * (BM + BN);
  static constexpr int RT = (BM / WM) * 32;  // staged rows per wide-store pass
  static constexpr int ALL_STAGES = STAGES + (EDGE ? 1 : 0);   // (EDGE: one more stage for the k-tile K ends in)
  static constexpr int LDS_FLOATS = ALL_STAGES * BUF > KW * RT * BN ? ALL_STAGES * BUF : KW * RT * BN;
  // two blocks per CU where LDS and registers allow it (64 x 64: 48 KB, 16 accumulators)
  static constexpr int WAVES_PER_SIMD = LDS_FLOATS * 4 * 2 <= 160 * 1024 && (WM / 32) * (WN / 32) <= 2 ? 4 : 2;
};

// EDGE: tiles may be ragged in M and N (row / column offsets clamped once per block, stores masked; N a multiple of 4) and K
// may end inside a k-tile (the last k-tile on the loaders that clamp k as well, its missing k zeroed on the A side in LDS).
template <int BM, int BN, int WM, int WN, bool A_KC, bool B_KC, int ABL = 0, int ST = 3, int KB = 32, bool EDGE = false, int KW = 2>
__global__ __launch_bounds__((PairGeometry<BM, BN, WM, WN, ST, KB, EDGE, KW>::NT), (PairGeometry<BM, BN, WM, WN, ST, KB, EDGE, KW>::WAVES_PER_SIMD)) void
gemm_pair_kernel(GemmArgs a) {
  using G = PairGeometry<BM, BN, WM, WN, ST, KB, EDGE, KW>;
  constexpr int GW = KB / 8 / KW;  // k-groups per wave and k-tile
  static_assert(GW >= 1 && GW * KW * 8 == KB, "the k-groups of a k-tile divide among the waves of a sub-tile");
  constexpr int BK = G::BK, NT = G::NT, BUF = G::BUF, S = G::STAGES;
  static_assert(S >= 2 && S <= 5, "two to five stages");
  constexpr int WAVES_N = BN / WN;
  constexpr int MI = WM / 32, NI = WN / 32;
  constexpr bool AIL = Interleaved<A_KC, MI>::value, BIL = Interleaved<B_KC, NI>::value;
  using DmaA = DmaLoader<BM, BK, NT, A_KC, false, EDGE, false>;
  using DmaB = DmaLoader<BN, BK, NT, B_KC, false, EDGE, false>;
  using DmaAT = DmaLoader<BM, BK, NT, A_KC, false, true, true>;   // (EDGE) the k-tile K ends in
  using DmaBT = DmaLoader<BN, BK, NT, B_KC, false, true, true>;
  static_assert(!EDGE || (WM == 32 && WN == 32), "ragged tiles: one block per wave (no interleaved rows)");
  static_assert(DmaA::INSTRS % DmaA::WAVES == 0 && DmaB::INSTRS % DmaB::WAVES == 0, "every wave issues the same number of loads per k-tile");
  constexpr int LOADS = DmaA::PER_WAVE + DmaB::PER_WAVE;  // wave instructions per wave and k-tile
  __shared__ __attribute__((aligned(16))) float lds[G::LDS_FLOATS];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int sub = wave / KW, kw = wave % KW;
  const int wm0 = (sub / WAVES_N) * WM, wn0 = (sub % WAVES_N) * WN;
  const int i = lane & 31, hi = lane >> 5;
  if (a.prio) __builtin_amdgcn_s_setprio(3);

  long m_blk, n_blk;
  tile_origin<BM, BN>(xcd_remap(blockIdx.x, a.tiles_m * a.tiles_n), a.tiles_m, a.tiles_n, m_blk, n_blk);
  const int k_tail = EDGE ? (int)(a.K % BK) : 0;          // valid k of a ragged last k-tile (0 = none): it runs behind the loop
  const int nk = (int)(a.K / BK);                         // whole k-tiles: the loop

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  DmaA da;
  DmaB db;
  da.init(a, m_blk, wave, lane, a.a_rows, a.lda);
  db.init(a, n_blk, wave, lane, a.N, a.ldb);
  auto issue_tile = [&](int kt, int stage) {
    float* s = lds + stage * BUF;
    da.issue(a, a.A, a.lda, m_blk, (long)kt * BK, s, wave, lane, a.a_rows, a.K);
    db.issue(a, a.B, a.ldb, n_blk, (long)kt * BK, s + BK * BM, wave, lane, a.N, a.K);
  };
  // fragments of k-group pp (8 k) of the tile at As, as in gemm_mainloop_dma: av[mi][j] / bv[ni][j] = this lane's A / B
  // value of block mi / ni for MFMA k-step j (k = 8 pp + j in lanes 0 - 31, 8 pp + 4 + j in lanes 32 - 63)
  auto fragments = [&](const float* As, int pp, float (&av)[MI][4], float (&bv)[NI][4]) {
    const float* Bs = As + BK * BM;
    if constexpr (AIL) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        typedef float vecA __attribute__((ext_vector_type(MI)));
        const vecA v = *reinterpret_cast<const vecA*>(As + (8 * pp + j + 4 * hi) * BM + wm0 + MI * i);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) av[mi][j] = v[mi];
      }
    } else {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const int row = wm0 + mi * 32 + i;
        if constexpr (A_KC) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(As + row * BK + (((2 * pp + hi) ^ DmaA::swizzle(row)) << 2));
#pragma unroll
          for (int j = 0; j < 4; ++j) av[mi][j] = v[j];
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) av[mi][j] = As[(8 * pp + j + 4 * hi) * BM + row];
        }
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
    } else {
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int col = wn0 + ni * 32 + i;
        if constexpr (B_KC) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(Bs + col * BK + (((2 * pp + hi) ^ DmaB::swizzle(col)) << 2));
#pragma unroll
          for (int j = 0; j < 4; ++j) bv[ni][j] = v[j];
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) bv[ni][j] = Bs[(8 * pp + j + 4 * hi) * BN + col];
        }
      }
    }
  };
  auto multiply = [&](const float (&av)[MI][4], const float (&bv)[NI][4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mi][j], bv[ni][j], acc[mi][ni], 0, 0, 0);
  };
  // tile kt has landed for THIS wave (its loads of tile kt + 1, if any, may still be in flight), then the barrier: tile
  // kt is published, and every wave is done reading tile kt - 1 — the stage the loads of tile kt + 2 go to
  // (the barrier is the bare instruction: __syncthreads() carries a fence that the offline compiler turns into
  // s_waitcnt vmcnt(0) while an LDS-DMA is pending — the wave would wait for tile kt + 1 as well.  The wave's own LDS
  // reads are waited for explicitly: the odd wave's last fragment reads of tile kt - 1 must have returned before another
  // wave's loads may overwrite that stage.)
  auto publish = [&](int ahead) {  // ahead = tiles behind kt whose loads are in flight: min(S - 2, nk - 1 - kt)
    if (ahead >= 3) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(3 * LOADS) : "memory");
    else if (ahead == 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(2 * LOADS) : "memory");
    else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(LOADS) : "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  };
  auto in_flight = [&](int kt) { return nk - 1 - kt < S - 2 ? nk - 1 - kt : S - 2; };

  // (EDGE) the k-tile K ends in goes to its own stage before anything else, on the loaders that clamp k as well (FLAT
  // loads); the first barrier of the loop — or the one below — waits for it together with tile 0, so its round trip is not
  // seen, and the loop itself is the loop of the whole-tile kernel.  Its missing k are re-reads of the last valid ones:
  // zeroed on the A side behind the loop.
  if (EDGE && k_tail != 0) {
    DmaAT dat;
    DmaBT dbt;
    float* s = lds + S * BUF;
    dat.issue(a, a.A, a.lda, m_blk, (long)nk * BK, s, wave, lane, a.a_rows, a.K);
    dbt.issue(a, a.B, a.ldb, n_blk, (long)nk * BK, s + BK * BM, wave, lane, a.N, a.K);
  }
#pragma unroll
  for (int t = 0; t < S - 1; ++t)
    if (t < nk) issue_tile(t, t);
  if (EDGE && k_tail != 0) __builtin_amdgcn_s_waitcnt(0x0F70);   // s_waitcnt vmcnt(0), as an instruction the wait-count pass sees
  const bool late = (wave & 1) != 0 && !a.no_skew && !(ABL & 256);   // (waves 2 s and 2 s + 1 share SIMD s)
  if (late) {
    float avp[MI][4], bvp[NI][4];
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
      publish(in_flight(kt));
      if (kt > 0) multiply(avp, bvp);  // tile kt - 1, this wave's last k-group
      if (kt + S - 1 < nk) issue_tile(kt + S - 1, cur >= 1 ? cur - 1 : S - 1);
      const float* As = lds + cur * BUF;
#pragma unroll
      for (int g = 0; g + 1 < GW; ++g) {
        float av[MI][4], bv[NI][4];
        fragments(As, GW * kw + g, av, bv);
        multiply(av, bv);
      }
      fragments(As, GW * kw + GW - 1, avp, bvp);
      cur = cur == S - 1 ? 0 : cur + 1;
    }
    if (nk > 0) multiply(avp, bvp);
  } else {
    int cur = 0;
    float av0[MI][4], bv0[NI][4];  // (ABL bit 3: the fragments are read once)
    if (ABL & 8) {
      publish(in_flight(0));
      fragments(lds, 0, av0, bv0);
    }
    for (int kt = 0; kt < nk; ++kt) {
      if (!(ABL & 4)) publish(in_flight(kt));
      if (kt + S - 1 < nk && !(ABL & 2)) issue_tile(kt + S - 1, cur >= 1 ? cur - 1 : S - 1);
      const float* As = lds + cur * BUF;
#pragma unroll
      for (int g = 0; g < GW && !(ABL & 1); ++g) {
        if (ABL & 8) {
          multiply(av0, bv0);
          asm volatile("" ::: "memory");
          continue;
        }
        float av[MI][4], bv[NI][4];
        fragments(As, GW * kw + g, av, bv);
        multiply(av, bv);
      }
      cur = cur == S - 1 ? 0 : cur + 1;
    }
  }
  if (EDGE && k_tail != 0) {   // (block-uniform) the last, partial k-tile: landed long ago (prologue)
    float* At = lds + S * BUF;
    __syncthreads();   // (K < BK: nothing has waited for the tile yet)
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
#pragma unroll
    for (int g = 0; g < GW; ++g) {
      if (8 * (GW * kw + g) < k_tail) {   // (wave-uniform) a k-group past the end holds zeros only
        float av[MI][4], bv[NI][4];
        fragments(At, GW * kw + g, av, bv);
        multiply(av, bv);
      }
    }
  }
  __syncthreads();  // every wave is done reading the stages: the staged rows may overwrite them

  // ---- epilogue: whole tiles only.  Pass mi: every wave parks block row mi of its sub-tile (32 rows x WN columns) in its
  // copy of the staged rows; then all threads walk the RT staged rows in 16-byte chunks and add the two copies.
  constexpr int RT = G::RT, C4 = BN / 4;
  static_assert(NT % C4 == 0, "a thread keeps its column chunk");
  constexpr int NQ = (RT * C4 + NT - 1) / NT;   // (fewer chunks than threads: the upper threads have none)
  float* park = lds + kw * RT * BN;
  const int wmi = sub / WAVES_N;
  const int c4 = tid % C4;
  const long n = n_blk + c4 * 4;
  f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
  const bool n_ok = !EDGE || n + 4 <= a.N;   // (EDGE: N is a multiple of 4: a chunk is inside or outside)
  if (a.bias && n_ok) b4 = *reinterpret_cast<const f32x4*>(a.bias + n);
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    if (mi > 0) __syncthreads();
    if (BIL) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        typedef float vecN __attribute__((ext_vector_type(NI)));
        vecN v;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) v[ni] = acc[mi][ni][r];
        *reinterpret_cast<vecN*>(&park[(wmi * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi) * BN + wn0 + NI * i]) = v;
      }
    } else {
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) park[(wmi * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi) * BN + wn0 + ni * 32 + i] = acc[mi][ni][r];
    }
    __syncthreads();
    f32x4 old[NQ];
    long idx[NQ];
#pragma unroll
    for (int c = 0; c < NQ; ++c) {
      const int row = (c * NT + tid) / C4;
      const bool mine = (RT * C4) % NT == 0 || c * NT + tid < RT * C4;
      const long m = m_blk + (long)(row >> 5) * WM + sub_index<MI>(AIL, mi, row & 31);
      idx[c] = (mine && (!EDGE || (n_ok && m < a.M))) ? m * a.ldc + n : -1;
      if (a.accumulate && idx[c] >= 0) old[c] = *reinterpret_cast<const f32x4*>(a.C + idx[c]);
    }
#pragma unroll
    for (int c = 0; c < NQ; ++c) {
      const int row = (c * NT + tid) / C4;
      if (idx[c] < 0) continue;
      f32x4 v = *reinterpret_cast<const f32x4*>(&lds[row * BN + c4 * 4]);
#pragma unroll
      for (int w = 1; w < KW; ++w) {   // the waves' copies in wave order
        const f32x4 other = *reinterpret_cast<const f32x4*>(&lds[w * RT * BN + row * BN + c4 * 4]);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] + other[e];
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = a.accumulate ? (old[c][e] + v[e]) + b4[e] : v[e] + b4[e];
      f32x4* p = reinterpret_cast<f32x4*>(a.C + idx[c]);
      if (a.nt_store) __builtin_nontemporal_store(v, p);
      else *p = v;
    }
  }
}

}  // namespace gemm
}  // namespace eg
