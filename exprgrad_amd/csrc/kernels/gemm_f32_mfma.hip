// f32 contraction  C[m,n] (+)= sum_k opA(m,k) * opB(k,n) (+ bias[n])  on CDNA4 matrix cores, and
// the direct (im2col-free) convolution built on the same kernel.
//
// Replaces what the reference emits for `c[y,x] ++= a[y,it] * b[it,x]` on its GPU target
// (tests/cache/matmul_basic.ir: one work-item per output, global-memory RMW per k; or the
// user-scheduled 16x16x16 LDS tiling of tests/cache/matmul_schedule_tiled16.ir), for the two
// gradient contractions passes.nim:519-549 derives from it, and for conv2 (dnn.nim:45-49).
// The kernel itself is in gemm_f32_mfma.hpp; this file is the host-side planning:
// tile shape, split-K, vector/edge variant, launch, deterministic second pass.
#include "gemm_skinny.hpp"
#include "gemm_f32_pair.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "../eg_internal.hpp"

namespace eg {
namespace gemm {
long long* trace_begin(eg_ctx* ctx, unsigned blocks, unsigned waves);   // EG_GEMM_TRACE (defined below, declared in gemm_fused.hpp)
void trace_end(eg_ctx* ctx, long long* buffer, unsigned blocks, unsigned waves, const char* what);
}  // namespace gemm
}  // namespace eg

namespace {

using namespace eg::gemm;

constexpr int BK = 16;

struct TileCfg {
  int bm, bn, blocks_per_cu;
};

template <int BM, int BN, int WM, int WN, int MINB, int KB = BK>
int launch_config(eg_ctx* ctx, bool a_kc, bool b_kc, const GemmArgs& args, int splits, bool vec, bool edge,
                  int conv, bool a_vec_only = false) {
  constexpr int NT = Geometry<BM, BN, WM, WN>::NT;
  const long rows_m = args.edge_splits > 0 ? args.tiles_m - 1 : args.tiles_m;
  dim3 grid((unsigned)(rows_m * args.tiles_n * splits + (long)args.tiles_n * args.edge_splits), 1, 1);
  if (args.tail_tiles > 0)
    grid.x = (unsigned)((long)args.tiles_m * args.tiles_n - args.tail_tiles + (long)args.tail_tiles * args.tail_splits);
  dim3 block(NT);
  hipStream_t s = ctx->stream;
  // 16-byte aligned operands: interior tiles run the LDS-DMA loop (gemm_f32_mfma.hpp)
#define EG_GEMM_LAUNCH(AKC, BKC, V, E, CV)                                                                        \
  hipLaunchKernelGGL((gemm_f32_mfma_kernel<BM, BN, KB, WM, WN, MINB, AKC, BKC, V, E, CV, 0, (V == 4 && CV != 1)>), grid, \
                     block, 0, s, args)
#define EG_GEMM_LAYOUT(AKC, BKC)                  \
  do {                                            \
    if (!edge)                                    \
      EG_GEMM_LAUNCH(AKC, BKC, 4, false, 0);  \
    else if (vec)                                 \
      EG_GEMM_LAUNCH(AKC, BKC, 4, true, 0);   \
    else if (BN == 32 && a_vec_only) {            \
      if constexpr (BN == 32) EG_GEMM_LAUNCH(AKC, BKC, 41, true, 0); \
    } else                                        \
      EG_GEMM_LAUNCH(AKC, BKC, 1, true, 0);   \
  } while (0)
  if (conv == 2) {  // filter gradient: A = gOut [pixels][F], B = im2col gathered from the image
    if (vec)
      EG_GEMM_LAUNCH(false, false, 4, true, 2);
    else
      EG_GEMM_LAUNCH(false, false, 1, true, 2);
  } else if (conv) {
    if (vec)
      EG_GEMM_LAUNCH(true, true, 4, true, 1);
    else
      EG_GEMM_LAUNCH(true, true, 1, true, 1);
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

// Convolution variants: deeper k-tiles (BK = 32) — with F = 64 filters a block has little matrix
// work per barrier, so the prefetch distance of one k-tile must cover the L2 latency.
template <int BM, int BN, int CBK, int WM, int WN, int MINB>
int launch_conv(eg_ctx* ctx, const GemmArgs& args, bool vec) {
  constexpr int NT = Geometry<BM, BN, WM, WN>::NT;
  dim3 grid((unsigned)(args.tiles_m * args.tiles_n), 1, 1);
  // channels a multiple of the k-tile: interior tiles gather with LDS-DMA (a k-tile lies inside one tap)
  if (vec && args.cC % CBK == 0)
    hipLaunchKernelGGL((gemm_f32_mfma_kernel<BM, BN, CBK, WM, WN, MINB, true, true, 4, true, true, 0, true>), grid,
                       dim3(NT), 0, ctx->stream, args);
  else if (vec)
    hipLaunchKernelGGL((gemm_f32_mfma_kernel<BM, BN, CBK, WM, WN, MINB, true, true, 4, true, true>), grid, dim3(NT), 0,
                       ctx->stream, args);
  else
    hipLaunchKernelGGL((gemm_f32_mfma_kernel<BM, BN, CBK, WM, WN, MINB, true, true, 1, true, true>), grid, dim3(NT), 0,
                       ctx->stream, args);
  EG_HIP_CHECK(hipGetLastError());
  return EG_OK;
}

int run_conv(eg_ctx* ctx, GemmArgs args, bool vec) {
  args.a_rows = args.M;
  int v = 0;
  args.partial = nullptr;
  auto tiles = [&](int bm, int bn, int bk) {
    args.tiles_m = (int)((args.M + bm - 1) / bm);
    args.tiles_n = (int)((args.N + bn - 1) / bn);
    args.k_per_split = ((args.K + bk - 1) / bk) * bk;
  };
  if (args.N > 64 || v == 9) return -1;  // wide filter banks: the generic tile choice
  // narrow filter banks (the second layer of the fashion_mnist network: 8 -> 16 channels, 5 x 5; its image gradient:
  // 16 -> 8): a 64-column tile multiplies 48 .. 56 columns of padding; 128 x 32 tiles (EG_CONV_VARIANT=5 forces them)
  if ((v == 0 && args.N <= 32 && args.M >= 128L * 4 * ctx->compute_units) || v == 5) {
    tiles(128, 32, 16);
    return launch_conv<128, 32, 16, 32, 32, 4>(ctx, args, vec);
  }
  switch (v) {
    // measured on cfg 4 (256x256x64 -> 64, 3x3): 64x64x32 66 TF, 128x64x32 64, 256x64x32 63, 128x64x16 62
    case 1: tiles(128, 64, 32); return launch_conv<128, 64, 32, 64, 32, 2>(ctx, args, vec);
    case 2: tiles(256, 64, 32); return launch_conv<256, 64, 32, 64, 32, 1>(ctx, args, vec);
    case 4: tiles(128, 64, 16); return launch_conv<128, 64, 16, 64, 32, 4>(ctx, args, vec);
    default: tiles(64, 64, 32); return launch_conv<64, 64, 32, 32, 32, 4>(ctx, args, vec);
  }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// Time of a ragged last-row tile relative to a full one per k-tile (it skips its empty 32x32
// sub-blocks and loads only its valid rows, but stages the whole B tile).  Calibrated on
// 784 x 512 x 65536 (TN): 128-row tiles 0.55, 256-row tiles 0.36 (a ragged 256-row tile's k-tile takes 1.2 us
// against 3.8 us; 0.40 / 0.36 / 0.33: dense step 1.111 / 1.101 / 1.103 ms).
double ragged_tile_share(int bm, long m_rest) {
  const double live = (double)((m_rest + 31) / 32 * 32) / bm;
  const double floor = bm >= 256 ? 0.36 : 0.55;
  return live > floor ? live : floor;
}

// Relative cost of running the problem with a given tile: (block rounds on the chip) x (work of
// the co-resident blocks of one CU), slightly favouring the larger tile whose measured
// efficiency is higher (tools/gemm_tune.hip: 135 vs 128 TFLOP/s at 4096^3).
double tile_cost(const TileCfg& t, long M, long N, long k_tiles, int cus, int& splits_out) {
  const long tm = (M + t.bm - 1) / t.bm, tn = (N + t.bn - 1) / t.bn;
  const long tiles = tm * tn;
  const long slots = (long)cus * t.blocks_per_cu;
  // split K when the output alone cannot fill the chip and K is long (weight gradients:
  // K = batch); every slice keeps at least 8 k-tiles
  int splits = 1;
  // bias-sized outputs (N <= 32) with at least one tile per CU stream their big operand once whatever the
  // split: slabs and a second pass only add traffic (65536 x 10 x 512: 35 us as one pass of A)
  const bool skinny = N <= 32 && tiles >= cus;
  if (tiles < slots && k_tiles >= 32 && !skinny) {
    long want = slots / tiles;  // floor: one more slice would spill a few blocks into a second round
    long max_by_k = k_tiles / 8;
    splits = (int)(want < max_by_k ? want : max_by_k);
    if (splits < 1) splits = 1;
    if (splits > 1024) splits = 1024;
  }
  long per = (k_tiles + splits - 1) / splits;
  if (per < 1) per = 1;
  splits = (int)((k_tiles + per - 1) / per);
  if (splits < 1) splits = 1;
  splits_out = splits;
  const long blocks = tiles * splits;
  const long rounds = (blocks + slots - 1) / slots;
  // short K: the launch is bound by writing the output, which wants many waves in flight rather
  // than the 8-wave 256x256 block (one per CU)
  const double big = k_tiles >= 8 ? 1.05 : 0.8;
  const double eff = t.bm * t.bn >= 256 * 256 ? big : (t.bm * t.bn >= 128 * 128 ? 1.0 : 0.9);
  // edge tiles skip their empty 32x32 sub-blocks; co-resident blocks of a CU share the matrix
  // pipe, so with several blocks per CU the saved work shortens the round
  double fill = 1.0;
  if (t.blocks_per_cu > 1) {
    const double m32 = (double)((M + 31) / 32 * 32), n32 = (double)((N + 31) / 32 * 32);
    // only the matrix work shrinks (operand staging does not): credit half of it
    fill = 0.5 + 0.5 * (m32 * n32) / ((double)tm * t.bm * (double)tn * t.bn);
  }
  // a partial last round: its blocks have their CU (almost) to themselves and finish sooner than a
  // full round of co-resident blocks — but never faster than about 1.3 / blocks_per_cu of it
  const long tail = blocks % slots;
  double eff_rounds = (double)(blocks / slots);
  if (tail) {
    const double alone = 1.3 / t.blocks_per_cu < 1.0 ? 1.3 / t.blocks_per_cu : 1.0;
    const double share = (double)tail / (double)slots;
    eff_rounds += share > alone ? share : alone;
  }
  double cost = eff_rounds * t.blocks_per_cu * t.bm * t.bn * (double)per * fill / eff;
  // split-K with a ragged last tile row: run_gemm cuts those tiles into fewer slices, the k-slices
  // of the full tiles shrink accordingly
  const long m_rest = M % t.bm;
  if (splits > 1 && rounds == 1 && m_rest != 0 && m_rest * 2 <= t.bm && tm >= 2) {
    const double share = ((double)(tm - 1) * tn + tn * ragged_tile_share(t.bm, m_rest)) / (double)tiles;
    cost = cost / fill * share;
  }
  if (splits > 1) cost += (double)M * N * splits * 0.02;  // second pass traffic
  return cost;
}

// ---- outputs wider than one narrow tile in both directions (M, N > 64): a time estimate per (tile, split).
//
// Calibrated on tools/sweep_mid.py (square problems 256 .. 4096, every tile x split, round 2).  A block
// needs `mfma` us of matrix-core time per 16-deep k-tile; alone on its CU it cannot go faster than `alone`
// us per k-tile (one wave per SIMD: the LDS-DMA round trip of the next k-tile is not hidden by the little
// matrix work of a small tile).  The busiest CU runs ceil(blocks / CUs) blocks, co-resident up to `blocks_per_cu`:
//     T = k-tiles per block x max(mfma x blocks on the busiest CU, alone x rounds) + fixed x rounds + second pass
// 1024^3: 64 x 64 tiles, one per CU, no split: 23.9 us (the old choice, 256 x 256 x 16 splits: 38.5 us);
// 3072^3: 64 x 64: 484 us (144 tiles of 256 x 256 leave 112 CUs idle: 584 us); 4096^3 keeps 256 x 256.
struct WideTile {
  int bm, bn, wm, wn, blocks_per_cu;
  double mfma, alone, alone_k32, fixed;
};
const WideTile kWideTiles[] = {
    {256, 256, 128, 64, 1, 3.80, 1.30, 1.30, 8.0},
    {128, 128, 64, 64, 4, 1.05, 0.70, 0.70, 8.0},
    {64, 64, 32, 32, 4, 0.25, 0.44, 0.27, 4.5},   // (0.275 / 0.30 until round 4; re-measured with sustained clocks: 2048^3 131.6 us, 3072^3 432; one block per CU = the wave-pair kernel: 1024 x 1024 x 4096 73.4)
};

// Matrix time of a tile with `rows` x `cols` valid outputs relative to a whole tile.  A ragged tile skips
// its empty 32 x 32 blocks, but the block is as slow as its busiest SIMD: wave w runs on SIMD w % 4, so
// 128 valid columns of a 256 x 256 tile (wave columns 2 and 3 idle) leave two SIMDs with the work of a whole
// tile (8192 x 128 x 8192 on 256 x 256 tiles: 278 us, as long as N = 256), while 128 valid ROWS halve it.
double ragged_tile_factor(const WideTile& t, long rows, long cols) {
  const int waves_m = t.bm / t.wm, waves_n = t.bn / t.wn, mi = t.wm / 32, ni = t.wn / 32;
  long load[4] = {0, 0, 0, 0};
  for (int wr = 0; wr < waves_m; ++wr)
    for (int wc = 0; wc < waves_n; ++wc) {
      long lm = (rows - (long)wr * t.wm + 31) / 32, ln = (cols - (long)wc * t.wn + 31) / 32;
      lm = lm < 0 ? 0 : (lm > mi ? mi : lm);
      ln = ln < 0 ? 0 : (ln > ni ? ni : ln);
      load[(wr * waves_n + wc) % 4] += lm * ln;
    }
  long worst = 0;
  for (long l : load) worst = l > worst ? l : worst;
  const long whole = (long)((waves_m * waves_n + 3) / 4) * mi * ni;
  return (double)worst / (double)whole;
}

double wide_tile_time(const WideTile& t, long M, long N, long k_tiles, int cus, bool vec, int& splits_out, bool k64 = false) {
  const long tm = (M + t.bm - 1) / t.bm, tn = (N + t.bn - 1) / t.bn;
  const long tiles = tm * tn;
  const long slots = (long)cus * t.blocks_per_cu;
  // whole tiles, the ragged last row / column / corner (clamped loop: ~10 % slower per k-tile)
  const long m_rest = M % t.bm, n_rest = N % t.bn;
  const long full_m = M / t.bm, full_n = N / t.bn;
  const double f_m = m_rest ? 1.1 * ragged_tile_factor(t, m_rest, t.bn) : 0, f_n = n_rest ? 1.1 * ragged_tile_factor(t, t.bm, n_rest) : 0;
  const double f_mn = m_rest && n_rest ? 1.1 * ragged_tile_factor(t, m_rest, n_rest) : 0;
  const double mean = ((double)full_m * full_n + f_m * full_n + f_n * full_m + f_mn) / (double)tiles;
  double worst = full_m && full_n ? 1.0 : 0;
  if (full_n && f_m > worst) worst = f_m;
  if (full_m && f_n > worst) worst = f_n;
  if (f_mn > worst) worst = f_mn;
  const long max_by_k = k_tiles / 8 > 1 ? k_tiles / 8 : 1;  // every slice keeps at least 8 k-tiles
  double best = 0;
  splits_out = 1;
  long last = 0;
  // candidate slice counts: a geometric ladder plus the counts that fill the CUs / the block slots exactly
  long cand[40];
  int ncand = 0;
  for (long want = 1; want <= 1024 && ncand < 32; want = want < 4 ? want + 1 : want + want / 2) cand[ncand++] = want;
  for (long fillers : {(long)cus / tiles, slots / tiles, 2 * (long)cus / tiles, (long)cus / tiles + 1})
    if (fillers > 1) cand[ncand++] = fillers;
  std::sort(cand, cand + ncand);
  for (int ci = 0; ci < ncand; ++ci) {
    const long want = cand[ci];
    if (want > max_by_k) break;
    const long per = (k_tiles + want - 1) / want;
    const long s = (k_tiles + per - 1) / per;
    if (s == last) continue;
    last = s;
    const long blocks = tiles * s;
    if (s > 1 && blocks > 2 * slots) break;  // more slices than the chip can hold at once only add slabs
    const long on_cu = (blocks + cus - 1) / cus, rounds = (blocks + slots - 1) / slots;
    // (one unsliced block per CU of 64 x 64 tiles: the wave-pair kernel, 0.27 for whole tiles, 0.285 ragged; otherwise the
    // four-wave kernel with 32-deep k-tiles, 0.30)
    const bool pair = s == 1 && on_cu == 1;
    const bool pair_whole = pair && k64 && m_rest == 0 && n_rest == 0;
    const double alone = (vec && t.bm == 64 && on_cu <= 2) ? (pair_whole ? t.alone_k32 : pair ? 0.285 : 0.30) : t.alone;
    // the busiest CU: its blocks are a sample of the tiles, never faster than one of the slowest kind.  More blocks than
    // slots of a tile that shares its CU four ways: the CUs pick up blocks as slots free up, so the busiest one carries
    // the average plus about half a block, not the next whole number (2304^3 on 64 x 64 tiles, 5.06 blocks per CU:
    // 202 us measured; "6 blocks" predicted 225 and lost to a sliced 256 x 256 launch that takes 230)
    double load = (double)on_cu;
    if (blocks > slots && t.blocks_per_cu >= 4) {
      const double avg = (double)blocks / (double)cus;
      load = blocks % cus == 0 ? avg : avg + 0.5;
    }
    double matrix = t.mfma * (load * mean > worst ? load * mean : worst);
    if (s > 1 && rounds == 1 && m_rest != 0 && m_rest * 2 <= t.bm && tm >= 2)  // run_gemm: ragged rows get fewer slices
      matrix = t.mfma * (double)on_cu * ((double)(tm - 1) * tn + tn * ragged_tile_share(t.bm, m_rest)) / (double)tiles;
    const double step = matrix > alone * rounds ? matrix : alone * rounds;
    double time = (double)per * step + t.fixed * rounds;
    if (s > 1) {
      const double mb = (double)M * N * 4e-6;           // one slab, MB
      time += 4.5 + mb * (double)s / 4.0 + mb * (double)(s + 1) / 4.5;  // second launch (3.0 until round 4: 384^3 and 512^3 sliced 11.0 / 13.1 us, unsliced 9.7 / 11.9) + slabs out at ~4 TB/s, second pass at ~4.5
    }
    if (best == 0 || time < best) {
      best = time;
      splits_out = (int)s;
    }
  }
  return best;
}

// Tile shape and split count for an M x N x K contraction on this device.
void choose_tile(eg_ctx* ctx, long M, long N, long K, int& bm, int& bn, int& splits, bool vec = true, bool plain = true) {
  const long k_tiles = (K + BK - 1) / BK;
  static const bool old_model = eg::sw::raw("EG_GEMM_OLD_TILE_MODEL") != nullptr;
  // (convolutions keep their measured choices; with fewer than 8 k-tiles a launch is bound by writing its output, which
  // the time model does not describe: 65536 x 512 x 10 with a generated epilogue, 67 us on the tile the older rule picks, 79 us)
  if (plain && M > 64 && N > 64 && k_tiles >= 8 && !old_model && eg::sw::raw("EG_GEMM_FORCE_TILE") == nullptr) {
    static const bool debug_tile = eg::sw::raw("EG_DEBUG_TILE") != nullptr;
    double best = 0;
    for (const WideTile& t : kWideTiles) {
      int sp;
      const double time = wide_tile_time(t, M, N, k_tiles, ctx->compute_units, vec, sp, K % 64 == 0);
      if (debug_tile) fprintf(stderr, "[eg] tile model %ld x %ld x %ld: %d x %d, %d slices: %.1f us\n", M, N, K, t.bm, t.bn, sp, time);
      if (best == 0 || time < best * 0.97) {  // larger tiles listed first: a smaller one has to win by 3 %
        best = time;
        bm = t.bm;
        bn = t.bn;
        splits = sp;
      }
    }
    if (const char* f = eg::sw::raw("EG_GEMM_FORCE_SPLITS")) {  // tuning aid
      const int want = atoi(f);
      if (want >= 1 && want <= k_tiles) {
        const long per = (k_tiles + want - 1) / want;
        splits = (int)((k_tiles + per - 1) / per);
      }
    }
    return;
  }
  // Candidates: 256x256 (16 waves, 1 block/CU) for large outputs, 128x128 (4 waves, 4 blocks/CU),
  // and narrow tiles for bias-sized N (the N = 1/4/10 layers of the XOR and dense nets, F = 64
  // filter banks) so the padding wasted in the matrix core stays small.
  static const TileCfg cfgs[] = {{256, 256, 1}, {128, 128, 4}, {128, 64, 4}, {128, 32, 4}, {256, 64, 2}, {64, 64, 4}};
  constexpr int NCFG = 6;
  int forced_bm = 0, forced_bn = 0;
  if (const char* f = eg::sw::raw("EG_GEMM_FORCE_TILE")) sscanf(f, "%d,%d", &forced_bm, &forced_bn);  // tuning aid
  int best = 1, best_splits = 1;
  double best_cost = 0;
  for (int c = 0; c < NCFG; ++c) {
    if (forced_bm && (cfgs[c].bm != forced_bm || cfgs[c].bn != forced_bn)) continue;
    if (!forced_bm) {
      // 64-wide tiles: narrow outputs, or a single tile row (M <= BM: the filter gradient of a
      // convolution, M = F)
      if (cfgs[c].bn == 64 && N > 64 && M > cfgs[c].bm) continue;
      if (cfgs[c].bn == 32 && N > 32) continue;
      if (cfgs[c].bn >= 128 && N <= 64) continue;
    }
    int sp;
    const double cost = tile_cost(cfgs[c], M, N, k_tiles, ctx->compute_units, sp);
    if (best_cost == 0 || cost < best_cost) {
      best = c;
      best_cost = cost;
      best_splits = sp;
    }
  }
  bm = cfgs[best].bm;
  bn = cfgs[best].bn;
  splits = best_splits;
  if (const char* f = eg::sw::raw("EG_GEMM_FORCE_SPLITS")) {  // tuning aid
    const int want = atoi(f);
    if (want >= 1 && want <= k_tiles) {
      const long per = (k_tiles + want - 1) / want;
      splits = (int)((k_tiles + per - 1) / per);
    }
  }
}

// Wide stores of whole tiles as nontemporal stores: the output of a contraction is not read again by the
// same launch (4096^3: 137.1 -> 139.0 TFLOP/s, 65536x512x784: 447 -> 438 us; EG_GEMM_NO_NT_STORE=1 switches it off).
int nt_store_enabled() {
  constexpr bool off = false;
  return off ? 0 : 1;
}

int side_priority(const eg_ctx* ctx) {
  constexpr bool off = false;
  return ctx->on_side_lane && !off ? 1 : 0;
}

// Whole tiles leave through LDS as 16-byte stores (GemmArgs::wide_store) when every address the
// epilogue touches is 16-byte aligned: the output (or the split-K slabs, which come from the
// workspace), the bias, and whole rows of four.
bool wide_store_ok(const GemmArgs& a, bool to_partial, bool fused = false) {
  static const bool off = eg::sw::raw("EG_GEMM_NO_WIDE_STORE") != nullptr;
  if (off || a.N % 4 != 0) return false;
  // measured: +2.5 % at 4096^3, -10 % on a 65536 x 512 x 10 product (two barriers per block row against
  // almost no k loop): plain contractions with fewer than 8 k-tiles keep the direct stores
  if (!fused && a.K < 8 * BK) return false;
  if (to_partial) return (a.M * a.N) % 4 == 0;   // slabs are [split][M][N] in the 256-byte aligned workspace
  return a.ldc % 4 == 0 && aligned16(a.C) && (a.bias == nullptr || aligned16(a.bias));
}

// Shared host-side planning: tile shape, split-K, vector/edge variant, launch, second pass.
int run_gemm(eg_ctx* ctx, bool a_kc, bool b_kc, GemmArgs args, int conv, bool vec_ok, bool a_vec_only = false) {
  if (!args.ones_row) args.a_rows = args.M;
  const long M = args.M, N = args.N, K = args.K;
  int BM, BN, splits;
  choose_tile(ctx, M, N, K, BM, BN, splits, vec_ok, conv == 0);
  // Small outputs — between half a chip and three chips of 32 x 32 tiles (512 x 512: 256 of them, 64 of 64 x 64): one
  // 32 x 32 tile per block, eight waves that split every 128-deep k-tile (gemm_f32_pair.hpp, KW = 8), unsliced whatever K
  // is, up to K = 4096: 512^3 13.1 -> 8.0 us NN, 13.4 -> 6.1 TN; 384^3 10.7 -> 6.4; 500 x 500 x 1000 17.3 -> 9.3; 512 x 512 x 2048
  // 20.1 -> 13.3; equal at K = 4096 (25.3 / 26.4); a long K is bound by the tile's loads (512 x 512 x 65536: 337 us against 282
  // for sliced 64 x 64 tiles).  EG_GEMM_NO_PAIR=1 (or a forced tile / slice count) keeps the choice below.
  {
    const long t32 = ((M + 31) / 32) * ((N + 31) / 32);
    const bool kw8_on = eg::sw::raw("EG_GEMM_NO_PAIR") == nullptr && eg::sw::raw("EG_GEMM_FORCE_TILE") == nullptr &&
                        eg::sw::raw("EG_GEMM_FORCE_SPLITS") == nullptr;
    // (whole tiles: 64 KB of LDS, two blocks share a CU — up to three blocks per CU pay: 640^3 14.4 -> 9.4 us, 768^3 16.7 -> 14.9,
    // 768 x 768 x 2048 35.9 -> 31.6; 896^3 and 1024^3 do not.  Ragged: 96 KB, one block per CU: up to two per CU, 576^3 13.5 -> 12.8)
    // (fewer tiles than half a chip: still better than slices with their second launch while K is short — 256 x 256 x 512
    // 12.0 -> 6.7 us, 256 x 256 x 1024 14.9 -> 8.1, 320 x 320 x 1024 14.7 -> 7.8, 256^3 7.9 -> 6.4; at K = 2048 the slices win, 10.6
    // against 12.6.)
    constexpr long kw8_small_k = 1024;
    const bool kw8_ragged = M % 32 != 0 || N % 32 != 0 || K % 128 != 0;
    if (kw8_on && !conv && vec_ok && !a_vec_only && !args.ones_row && t32 <= (kw8_ragged ? 2L : 3L) * ctx->compute_units && (2 * t32 >= ctx->compute_units || (K <= kw8_small_k && t32 >= 4)) &&
        K >= 256 && K <= 4096 && N % 4 == 0 && args.ldc % 4 == 0 && aligned16(args.C) && (args.bias == nullptr || aligned16(args.bias))) {
      args.tiles_m = (int)((M + 31) / 32);
      args.tiles_n = (int)((N + 31) / 32);
      args.partial = nullptr;
      args.splits = 1;
      args.k_per_split = K;
      args.prio = side_priority(ctx);
      args.nt_store = nt_store_enabled();
      args.no_skew = eg::sw::raw("EG_GEMM_NO_SKEW") != nullptr;
      const bool ragged = kw8_ragged;
      // (four stages — three 128-deep k-tiles in flight, 128 KB of LDS — measured equal: 512^3 5.7 / 5.7 us back to back, 512 x 512 x 2048 13.3 / 12.9)
      dim3 grid((unsigned)t32), block(512);
#define EG_KW8(AKC, BKC)                                                                                                       \
  do {                                                                                                                         \
    if (ragged) hipLaunchKernelGGL((gemm_pair_kernel<32, 32, 32, 32, AKC, BKC, 0, 2, 128, true, 8>), grid, block, 0, ctx->stream, args);  \
    else hipLaunchKernelGGL((gemm_pair_kernel<32, 32, 32, 32, AKC, BKC, 0, 2, 128, false, 8>), grid, block, 0, ctx->stream, args);        \
  } while (0)
      if (a_kc && !b_kc) EG_KW8(true, false);
      else if (a_kc && b_kc) EG_KW8(true, true);
      else if (!a_kc && !b_kc) EG_KW8(false, false);
      else EG_KW8(false, true);
#undef EG_KW8
      EG_HIP_CHECK(hipGetLastError());
      return EG_OK;
    }
  }
  // 96 x 96 tiles (round 5): an output that is ONE round of them — 1536^2 = 256 tiles on 256 CUs — is 2.25 rounds of 64 x 64
  // tiles (576 blocks: three on some CUs, two on others: 0.59 of peak) and a quarter of a round of 256 x 256.  Same kernel as
  // the wave pairs (gemm_f32_pair.hpp), three 96 x 32 sub-tiles per block, each shared by FOUR waves that split every
  // 64-deep k-tile (12 waves = three per SIMD, 24 matrix instructions per wave and k-tile).  EG_GEMM_NO_PAIR=1 keeps 64 x 64.
  {
    const long t96 = (M / 96) * (N / 96);
    const bool on = eg::sw::raw("EG_GEMM_NO_PAIR") == nullptr && eg::sw::raw("EG_GEMM_NO_T96") == nullptr && eg::sw::raw("EG_GEMM_FORCE_TILE") == nullptr &&
                    eg::sw::raw("EG_GEMM_FORCE_SPLITS") == nullptr;
    if (on && !conv && vec_ok && !a_vec_only && !args.ones_row && M % 96 == 0 && N % 96 == 0 && K % 64 == 0 && K >= 512 &&
        t96 <= ctx->compute_units && 4 * t96 > 3L * ctx->compute_units && args.ldc % 4 == 0 && aligned16(args.C) &&
        (args.bias == nullptr || aligned16(args.bias))) {
      args.tiles_m = (int)(M / 96);
      args.tiles_n = (int)(N / 96);
      args.partial = nullptr;
      args.splits = 1;
      args.k_per_split = K;
      args.prio = side_priority(ctx);
      args.nt_store = nt_store_enabled();
      args.no_skew = eg::sw::raw("EG_GEMM_NO_SKEW") != nullptr;
      dim3 grid((unsigned)t96), block(768);
#define EG_T96(AKC, BKC) hipLaunchKernelGGL((gemm_pair_kernel<96, 96, 96, 32, AKC, BKC, 0, 2, 64, false, 4>), grid, block, 0, ctx->stream, args)
      if (a_kc && !b_kc) EG_T96(true, false);
      else if (a_kc && b_kc) EG_T96(true, true);
      else if (!a_kc && !b_kc) EG_T96(false, false);
      else EG_T96(false, true);
#undef EG_T96
      EG_HIP_CHECK(hipGetLastError());
      return EG_OK;
    }
  }
  // Stream-K on 64 x 64 tiles (round 6; gemm_streamk_kernel): the planner's choice is unsliced 64 x 64 tiles, there are more
  // tiles than CUs, and they do not divide evenly over the four block slots of a CU — the launch is as long as its busiest
  // CU (1280^3: 400 tiles, 1.56 per CU; 1792^3: 784 tiles, 3.06 per CU).  Persistent blocks (four per CU) share the
  // (tile, k-tile) space evenly instead; the tiles they cut are folded in k order by one more launch.  EG_GEMM_NO_STREAMK=1 off.
  if (BM == 64 && BN == 64 && splits == 1 && !conv && vec_ok && !a_vec_only && !args.ones_row && M % 64 == 0 && N % 64 == 0 && K % 32 == 0 &&
      K >= 256 && args.ldc % 4 == 0 && aligned16(args.C) && (args.bias == nullptr || aligned16(args.bias)) &&
      !eg::sw::present("EG_GEMM_NO_STREAMK") && !eg::sw::present("EG_GEMM_FORCE_TILE") && !eg::sw::present("EG_GEMM_FORCE_SPLITS")) {
    const long tiles = (M / 64) * (N / 64), cus = ctx->compute_units, slots = 4 * cus;
    const long nk = K / 32;
    const long busiest = (tiles + cus - 1) / cus;
    const double even = (double)tiles / (double)cus;
    // Every tile's units are shared (rounds = 0) by four blocks per CU (two when there are fewer than two tiles per CU).  The
    // hybrid form — `rounds` whole tiles per block first, only the remaining tiles shared — is kept behind the tuning aid
    // EG_STREAMK_BLOCKS_PER_CU: it wins at 2560^3 (278 -> 265 us) and loses at 1792^3 (three blocks per CU: 118 against 105).
    long g = tiles > 2 * cus ? 4 : 2, rounds = 0;
    // More tiles than block slots: whole rounds of tiles first (one per block and round, stored directly), only the tiles of
    // the partial last round shared — worth it while that round is at most 0.6 full and K is long (2432^3 238 -> 233 us, 2560^3
    // 280 -> 269, 2560 x 2560 x 4096 446 -> 425, 3584^3 733 -> 720; 2688^3 / 2816^3, last round 0.72 / 0.89 full: 3 % slower;
    // K = 1024: slower).
    bool hybrid = false;
    if (tiles >= slots && nk >= 64) {
      const long last = tiles % slots;
      if (last > 0 && 10 * last <= 6 * slots) {
        g = 4;
        rounds = tiles / slots;
        hybrid = true;
      }
    }
    if (const char* e = eg::sw::raw("EG_STREAMK_BLOCKS_PER_CU")) {   // tuning aid
      g = atol(e);
      rounds = tiles / (g * cus);
      hybrid = false;
    }
    const long grid = g * cus;
    const long rest = tiles - rounds * grid;          // tiles the blocks share unit by unit
    // a block's share of the units: even, but at least four k-tiles (a piece pays a prologue and a slab)
    long per = (rest * nk + grid - 1) / grid;
    if (per < 4) per = 4;
    // What it buys: (1 - even / busiest) of the one-block-per-tile launch, whose length is about busiest x nk x 0.51 us
    // (2048^3: four blocks per CU, 64 k-tiles, 131 us); with more tiles than block slots the dispatcher refills slots as they
    // free up and the busiest CU carries about even + 0.5.  What it costs: nearly every tile is cut, so the output travels
    // through the slabs and a second launch — 14 us at 1792^3.  Measured (tools/streamk_ab.py): 1792^3 115 -> 105 us, 1280 x
    // 1280 x 4096 139 -> 122, 1152^3 42.6 -> 39.3; equal at 1280^3; 3 - 5 % slower at 1664^3 / 1920^3 / 2304^3 — hence the bar.
    const double busiest_eff = tiles >= slots ? even + 0.5 : (double)busiest;
    const double saved_us = (1.0 - even / busiest_eff) * busiest_eff * (double)nk * 0.51;
    const double min_saved = eg::sw::real("EG_STREAMK_MIN_RATIO", 24.0);   // tuning aid: microseconds the model must promise
    if (tiles > cus && tiles < 6 * slots && rest > 0 && (hybrid || saved_us >= min_saved)) {
      int rc = eg::ensure_workspace(ctx, (size_t)grid * 2 * 64 * 64 * sizeof(float));
      if (rc) return rc;
      args.tiles_m = (int)(M / 64);
      args.tiles_n = (int)(N / 64);
      args.partial = static_cast<float*>(ctx->workspace);
      args.splits = (int)rounds;
      args.k_per_split = per;
      args.a_rows = M;
      dim3 gd((unsigned)grid), block(256);
      if (a_kc && !b_kc) hipLaunchKernelGGL((gemm_streamk_kernel<32, true, false>), gd, block, 0, ctx->stream, args);
      else if (a_kc && b_kc) hipLaunchKernelGGL((gemm_streamk_kernel<32, true, true>), gd, block, 0, ctx->stream, args);
      else if (!a_kc && !b_kc) hipLaunchKernelGGL((gemm_streamk_kernel<32, false, false>), gd, block, 0, ctx->stream, args);
      else hipLaunchKernelGGL((gemm_streamk_kernel<32, false, true>), gd, block, 0, ctx->stream, args);
      hipLaunchKernelGGL(gemm_streamk_fixup_kernel, dim3((unsigned)rest), dim3(256), 0, ctx->stream, args.partial, args.C, args.bias,
                         args.ldc, args.tiles_m, args.tiles_n, rounds * grid, (int)nk, per, args.accumulate);
      EG_HIP_CHECK(hipGetLastError());
      return EG_OK;
    }
  }
  // (the same design on one round of 128 x 128 tiles — four 64 x 64 sub-tiles x four waves — measured equal to what the model
  //  picks: 2048^3 129.1 against 130.7 us, 1792^3 113.1 / 113.9, 2048 x 2048 x 512 40.8 / 39.2: the gain above is the whole round, not the wave count)
  // A few rows / columns beyond whole 256 x 256 tiles of a large output (4100 = 16 x 256 + 4): the ragged
  // tile row and column stage whole operand tiles for 1/64 of the matrix work and push the launch into another
  // round of blocks (4100 x 4096 x 4096: +76 us, 4096 x 4100 x 4096: +154 us over 969 us).  As contractions
  // of their own they are one pass over the other operand (4 x 4100 x 4100: 30 us), so the output is cut into
  // whole tiles + remainder rows + remainder columns when the ragged tiles would cost a round
  // (EG_GEMM_NO_REMAINDER=1: one launch).  Split-K launches keep their ragged tiles: those get fewer,
  // longer k-slices next to the whole tiles at about the same cost.
  constexpr bool rem_split = true;
  if (rem_split && !conv && vec_ok && !a_vec_only && splits == 1 && !args.ones_row) {
    const long m_rem = M % 256, n_rem = N % 256;
    const long m0 = M - (m_rem <= 32 ? m_rem : 0), n0 = N - (n_rem <= 32 ? n_rem : 0);
    const long slots = ctx->compute_units;
    const long tiles_all = ((M + 255) / 256) * ((N + 255) / 256), tiles_main = ((m0 + 255) / 256) * ((n0 + 255) / 256);
    const bool saves_round = (tiles_all + slots - 1) / slots > (tiles_main + slots - 1) / slots;
    int bm_main = 0, bn_main = 0, splits_main = 0;
    if ((m0 < M || n0 < N) && m0 >= 256 && n0 >= 256 && n0 % 4 == 0 && m0 % 4 == 0 && saves_round)
      choose_tile(ctx, m0, n0, K, bm_main, bn_main, splits_main, vec_ok);  // the whole-tile part on its own: 256 x 256 tiles, no split-K?
    if (bm_main == 256 && bn_main == 256 && splits_main == 1) {
      GemmArgs part = args;
      part.M = part.a_rows = m0;
      part.N = n0;
      int rc = run_gemm(ctx, a_kc, b_kc, part, conv, vec_ok, a_vec_only);
      if (rc) return rc;
      if (m0 < M) {  // remainder rows, every column
        part = args;
        part.A = a_kc ? args.A + m0 * args.lda : args.A + m0;
        part.C = args.C + m0 * args.ldc;
        part.M = part.a_rows = M - m0;
        rc = run_gemm(ctx, a_kc, b_kc, part, conv, vec_ok, a_vec_only);
        if (rc) return rc;
      }
      if (n0 < N) {  // remainder columns of the whole-tile rows
        part = args;
        part.B = b_kc ? args.B + n0 * args.ldb : args.B + n0;
        part.C = args.C + n0;
        if (args.bias) part.bias = args.bias + n0;
        part.M = part.a_rows = m0;
        part.N = N - n0;
        rc = run_gemm(ctx, a_kc, b_kc, part, conv, vec_ok, a_vec_only);
        if (rc) return rc;
      }
      return EG_OK;
    }
  }
  // Extra rows (GemmArgs::x_rows): a TN product with a long K (a weight gradient: K = the batch) whose M is a few rows
  // beyond whole 256-row tiles and whose tiles cannot fill the chip by themselves.  The last tile row's blocks carry the
  // extra rows as a ninth accumulator block (+ 1/8 matrix work) and get proportionally more, shorter k-slices, so every
  // block finishes together; no ragged tile row exists.  EG_GEMM_NO_XROW=1: the ragged tile row of round 2.
  static const bool xrow_on = eg::sw::raw("EG_GEMM_NO_XROW") == nullptr;
  if (xrow_on && !conv && vec_ok && !a_vec_only && !a_kc && !b_kc && M > 256 && M % 256 > 0 && M % 256 <= 32 && N % 256 == 0 &&
      K % BK == 0 && args.ldc % 4 == 0 && eg::sw::raw("EG_GEMM_FORCE_TILE") == nullptr) {
    const long tm = M / 256, tn = N / 256, k_tiles = K / BK, slots = ctx->compute_units;
    const long full = (tm - 1) * tn;
    long best_s1 = 0, best_s2 = 0;
    double best_t = 0;
    // k-tile of a strip-carrying block relative to a plain one
    constexpr double xw = 1.2;  // measured on 784 x 512 x 65536: 1.0 449 us, 1.125 441, 1.2 428, 1.3 446, 1.4 452 (the strip adds 2 DMA pieces and 16 LDS reads per k-tile to its 4 MFMAs)
    for (long s1 = 2; full * s1 + tn * 2 <= slots && s1 <= k_tiles / 8; ++s1) {
      long s2 = (slots - full * s1) / tn;
      if (s2 > k_tiles / 8) s2 = k_tiles / 8;
      if (s2 < 2) continue;
      const long per1 = (k_tiles + s1 - 1) / s1, per2 = (k_tiles + s2 - 1) / s2;
      const double t = std::max((double)per1, xw * (double)per2);  // k-tiles of the slowest block, in whole-tile units
      if (best_s1 == 0 || t < best_t) {
        best_t = t;
        best_s1 = s1;
        best_s2 = s2;
      }
    }
    if (full == 0) {  // a single tile row: every block carries the strip
      long s2 = slots / tn;
      if (s2 > k_tiles / 8) s2 = k_tiles / 8;
      if (s2 >= 2) best_s1 = best_s2 = s2;
    }
    if (best_s1 >= 2 && best_s2 >= 2) {
      const long per1 = (k_tiles + best_s1 - 1) / best_s1, per2 = (k_tiles + best_s2 - 1) / best_s2;
      const long s1 = (k_tiles + per1 - 1) / per1, s2 = (k_tiles + per2 - 1) / per2;
      args.tiles_m = (int)tm;
      args.tiles_n = (int)tn;
      args.x_rows = (int)(M % 256);
      args.splits = (int)s1;
      args.k_per_split = per1 * BK;
      args.edge_splits = (int)s2;
      args.k_per_split_edge = per2 * BK;
      args.wide_store = wide_store_ok(args, true);
      args.prio = side_priority(ctx);
      args.nt_store = nt_store_enabled();
  args.no_skew = eg::sw::raw("EG_GEMM_NO_SKEW") != nullptr;
      const long total = M * N, slabs = std::max(s1, s2);
      int rc = eg::ensure_workspace(ctx, (((size_t)slabs * total + 3) & ~(size_t)3) * sizeof(float));
      if (rc) return rc;
      args.partial = static_cast<float*>(ctx->workspace);
      const unsigned grid = (unsigned)(full * s1 + tn * s2);
      static const bool trace_on = eg::sw::raw("EG_GEMM_TRACE") != nullptr;
      if (trace_on) args.trace = trace_begin(ctx, grid, 8);
      hipLaunchKernelGGL((gemm_f32_mfma_kernel<256, 256, BK, 128, 64, 1, false, false, 4, true, 0, 0, true, true>), dim3(grid), dim3(512),
                         0, ctx->stream, args);
      EG_HIP_CHECK(hipGetLastError());
      if (trace_on) trace_end(ctx, args.trace, grid, 8, "TN 256 x 256 with extra rows, k-sliced");
      long blocks = (total + 255) / 256;
      if (blocks > 2048) blocks = 2048;
      // rows of the last tile row and the extra rows were cut into s2 slices, the others into s1
      hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, args.partial, args.C, args.bias,
                         M, N, args.ldc, (int)s1, args.accumulate, (tm - 1) * 256, (int)s2);
      EG_HIP_CHECK(hipGetLastError());
      return EG_OK;
    }
  }
  // (32-deep k-tiles for the 256x256 tile were measured in round 2: +1 % at 4096^3, -7 % at K = 784, 0 elsewhere)
  // the convolution's filter gradient (M = F = 64 rows, 64 x 64 tiles, K = every output pixel): a block has
  // little matrix work per barrier, so its k-tiles are 32 deep like the forward gather's (EG_CONVGF_BK16=1: 16)
  constexpr bool gf16 = false;
  int KB = (conv == 2 && BM == 64 && BN == 64 && vec_ok && !gf16) ? 32 : BK;
  // 64 x 64 tiles with at most two blocks per CU are bound by the LDS-DMA round trip of the next k-tile, not
  // by matrix work: 32-deep k-tiles halve the round trips (1024^3: 28.1 -> 23.9 us; four blocks per CU hide
  // it by themselves: 2048^3 142.5 vs 146.5 us).  EG_GEMM_SMALL_BK16=1: 16.
  // Round 4, sustained clocks: 32 also wins with up to four blocks per CU (1536^3 81.4 -> 77.0 us, 1792^3 120.8 -> 114.4,
  // 2048^3 137.9 -> 131.6); beyond that the two are equal within 1 % (2304^3 202 / 206, 3072^3 432 / 439): 16 stays there.
  if (!conv && BM == 64 && BN == 64 && vec_ok && K >= 256 &&
      (M + 63) / 64 * ((N + 63) / 64) * splits <= 4L * ctx->compute_units)
    KB = 32;
  if (eg::sw::raw("EG_GEMM_SMALL_BK32") != nullptr && !conv && BM == 64 && BN == 64 && vec_ok && K >= 256) KB = 32;  // tuning aid
  const long k_tiles = (K + KB - 1) / KB;
  args.tiles_m = (int)((M + BM - 1) / BM);
  args.tiles_n = (int)((N + BN - 1) / BN);
  args.partial = nullptr;
  long tiles_per_split = (k_tiles + splits - 1) / splits;
  if (tiles_per_split < 1) tiles_per_split = 1;
  args.k_per_split = tiles_per_split * KB;
  splits = (int)((k_tiles + tiles_per_split - 1) / tiles_per_split);  // no empty slice (the count was planned in 16-deep k-tiles)
  if (splits < 1) splits = 1;
  args.splits = splits;

  // Tiny outputs split many ways (the XOR net's [2,4] and [4,1] weight gradients): a
  // per-element serial walk over hundreds of slabs is latency bound, so the slabs are folded
  // with the tree column-sum instead of the serial second pass.
  const long total = M * N;
  const bool tree_reduce = splits > 1 && (total <= 4096 || (splits >= 64 && total <= 65536)) && args.ldc == N && args.bias == nullptr;
  // Ragged last tile row (M = 784 with 128-row tiles: 16 rows): its blocks run a fraction of the
  // matrix work but, cut like the others, would occupy their CU slots just as long.  Give them
  // fewer, longer slices so every block carries about the same work; the freed slots go to the
  // full tiles.  Needs the LDS-DMA loop (cheap ragged tiles) and the serial second pass.
  long edge_row = M;
  int edge_splits = 0;
  const long m_rest = M % BM;
  if (splits > 1 && !tree_reduce && !conv && vec_ok && m_rest != 0 && m_rest * 2 <= BM && args.tiles_m >= 2 &&
      true) {
    const double frac = ragged_tile_share(BM, m_rest);
    const long full_tiles = (long)(args.tiles_m - 1) * args.tiles_n;
    const long slots = (long)ctx->compute_units * (BM * BN >= 256 * 256 ? 1 : (BM == 256 ? 2 : 4));
    long s_full = (long)((double)slots / ((double)full_tiles + args.tiles_n * frac));
    const long max_by_k = k_tiles / 8;
    if (s_full > max_by_k) s_full = max_by_k;
    long s_edge = (long)(s_full * frac + 0.5);
    if (s_full >= 2 && s_edge >= 1 && s_edge < s_full) {
      long per_full = (k_tiles + s_full - 1) / s_full;
      s_full = (k_tiles + per_full - 1) / per_full;
      long per_edge = (k_tiles + s_edge - 1) / s_edge;
      s_edge = (k_tiles + per_edge - 1) / per_edge;
      args.k_per_split = per_full * KB;
      args.splits = (int)s_full;
      args.edge_splits = edge_splits = (int)s_edge;
      args.k_per_split_edge = per_edge * KB;
      edge_row = (long)(args.tiles_m - 1) * BM;
    }
  }
  // Tail tiles: more tiles than block slots and a short last round -> cut the last round's tiles along K.
  long tail_slab_floats = 0;
  if (splits == 1 && !conv) {
    const long tiles = (long)args.tiles_m * args.tiles_n;
    const long slots = (long)ctx->compute_units * (BM * BN >= 256 * 256 ? 1 : (BM == 256 ? 2 : 4));
    const long tail = tiles % slots;
    if (tiles > slots && tail > 0 && tail * 2 <= slots && k_tiles >= 16) {
      long ts = slots / tail;
      if (ts > k_tiles / 8) ts = k_tiles / 8;
      if (ts > 16) ts = 16;
      if (ts >= 2) {
        const long per = (k_tiles + ts - 1) / ts;
        ts = (k_tiles + per - 1) / per;
        args.tail_tiles = (int)tail;
        args.tail_splits = (int)ts;
        args.tail_k_per_split = per * KB;
        tail_slab_floats = tail * ts * (long)BM * BN;
      }
    }
  }
  const int launch_splits = args.edge_splits > 0 ? args.splits : splits;
  args.wide_store = wide_store_ok(args, splits > 1);
  args.prio = side_priority(ctx);
  args.nt_store = nt_store_enabled();
  args.no_skew = eg::sw::raw("EG_GEMM_NO_SKEW") != nullptr;
  float* scratch = nullptr;
  if (args.tail_tiles > 0) {
    int rc = eg::ensure_workspace(ctx, (size_t)tail_slab_floats * sizeof(float));
    if (rc) return rc;
    args.partial = static_cast<float*>(ctx->workspace);
  }
  if (splits > 1) {
    const size_t slab_floats = ((size_t)launch_splits * total + 3) & ~(size_t)3;
    const size_t scratch_floats = tree_reduce ? (size_t)eg::colsum_scratch_floats(ctx, splits, total) : 0;
    int rc = eg::ensure_workspace(ctx, (slab_floats + scratch_floats) * sizeof(float));
    if (rc) return rc;
    args.partial = static_cast<float*>(ctx->workspace);
    scratch = args.partial + slab_floats;
  }

  const bool vec = vec_ok;
  const bool edge = conv || !(vec && M % BM == 0 && N % BN == 0 && K % KB == 0 && K > 0);

  int rc;
  // Whole 256 x 256 tiles of a long, unsliced product: 32-deep k-tiles (half the barriers and half the load issues per
  // MFMA; 128 KB of LDS, still one block per CU).  With the skewed waves of round 4 on top: 4096^3 949 -> 941 us in the
  // harness (+0.8 %); short products keep 16 (K = 784: round 2 measured -7 % with 32).  EG_GEMM_NO_BK32=1: 16 everywhere.
  const bool bk32_on = eg::sw::raw("EG_GEMM_NO_BK32") == nullptr;   // (read per call: a test compares the two loops)
  if (bk32_on && BM == 256 && BN == 256 && !edge && !conv && splits <= 1 && args.tail_tiles == 0 && args.edge_splits == 0 &&
      K % 32 == 0 && K >= 2048 && args.k_per_split == K) {
    dim3 grid((unsigned)((long)args.tiles_m * args.tiles_n)), block(512);
#define EG_BK32(AKC, BKC) \
  hipLaunchKernelGGL((gemm_f32_mfma_kernel<256, 256, 32, 128, 64, 1, AKC, BKC, 4, false, 0, 0, true>), grid, block, 0, ctx->stream, args)
    if (a_kc && !b_kc) EG_BK32(true, false);
    else if (a_kc && b_kc) EG_BK32(true, true);
    else if (!a_kc && !b_kc) EG_BK32(false, false);
    else EG_BK32(false, true);
#undef EG_BK32
    EG_HIP_CHECK(hipGetLastError());
    return EG_OK;
  }
  // Wave pairs (gemm_f32_pair.hpp): whole 64 x 64 tiles of an unsliced product with at most one block per CU — one wave
  // per SIMD on the four-wave kernel, where a k-tile costs 1.36x its matrix time (barrier + fragment reads, measured with
  // the loads removed).  Two waves per sub-tile split every 64-deep k-tile, the odd one a k-group late: 1024^3 22.1 ->
  // 20.9 us (NN / NT), 23.1 -> 19.8 (TN), 512^3 12.3 -> 11.6.  With two or more blocks per CU the four-wave kernel is
  // as fast or faster (1536^3, 3072^3), so those keep it.  Not bit-identical to it (two f32 chains per element instead
  // of one); EG_GEMM_NO_PAIR=1 (read per call) keeps the four-wave kernel.
  // Ragged tiles and a K that ends inside a k-tile take the EDGE form of the same kernel (clamped row / column offsets,
  // masked stores, the k-tile K ends in loaded in the prologue and multiplied behind the loop): 1000^3 28.1 -> 23.1 us (NN),
  // 27.5 -> 22.0 (TN), 1000 x 1024 x 4096 90.8 -> 74.6.
  const bool pair_on = eg::sw::raw("EG_GEMM_NO_PAIR") == nullptr;
  if (pair_on && BM == 64 && BN == 64 && vec && !conv && splits <= 1 && args.tail_tiles == 0 && args.edge_splits == 0 &&
      args.wide_store && !args.ones_row && (long)args.tiles_m * args.tiles_n <= ctx->compute_units) {
    const bool ragged = edge || K % 64 != 0;
    // (four waves per sub-tile, 16 per block, 128-deep k-tiles: 1024^3 21.9 us either way, 1024 x 1024 x 4096 76.1 against 77.2 — not taken)
    dim3 grid((unsigned)((long)args.tiles_m * args.tiles_n)), block(512);
#define EG_PAIR(AKC, BKC)                                                                                                  \
  do {                                                                                                                     \
    if (ragged) hipLaunchKernelGGL((gemm_pair_kernel<64, 64, 32, 32, AKC, BKC, 0, 2, 64, true>), grid, block, 0, ctx->stream, args);  \
    else hipLaunchKernelGGL((gemm_pair_kernel<64, 64, 32, 32, AKC, BKC, 0, 2, 64, false>), grid, block, 0, ctx->stream, args);        \
  } while (0)
    if (a_kc && !b_kc) EG_PAIR(true, false);
    else if (a_kc && b_kc) EG_PAIR(true, true);
    else if (!a_kc && !b_kc) EG_PAIR(false, false);
    else EG_PAIR(false, true);
#undef EG_PAIR
    EG_HIP_CHECK(hipGetLastError());
    return EG_OK;
  }
  if (BN == 32)
    rc = launch_config<128, 32, 32, 32, 4>(ctx, a_kc, b_kc, args, launch_splits, vec, edge, conv, a_vec_only && !vec);
  else if (BN == 64 && BM == 256)
    rc = launch_config<256, 64, 64, 32, 2>(ctx, a_kc, b_kc, args, launch_splits, vec, edge, conv);
  else if (BN == 64 && BM == 64 && KB == 32)
    rc = launch_config<64, 64, 32, 32, 4, 32>(ctx, a_kc, b_kc, args, launch_splits, vec, edge, conv);
  else if (BN == 64 && BM == 64)
    rc = launch_config<64, 64, 32, 32, 4>(ctx, a_kc, b_kc, args, launch_splits, vec, edge, conv);
  else if (BN == 64)
    rc = launch_config<128, 64, 64, 32, 4>(ctx, a_kc, b_kc, args, launch_splits, vec, edge, conv);
  else if (BN == 128)
    rc = launch_config<128, 128, 64, 64, 4>(ctx, a_kc, b_kc, args, launch_splits, vec, edge, conv);
  else
    rc = launch_config<256, 256, 128, 64, 1>(ctx, a_kc, b_kc, args, launch_splits, vec, edge, conv);
  if (rc) return rc;

  if (args.tail_tiles > 0) {
    const unsigned blocks = (unsigned)(args.tail_tiles * (BM / 32));
#define EG_TAIL_REDUCE(TM, TN)                                                                                          \
  hipLaunchKernelGGL((gemm_tail_reduce_kernel<TM, TN>), dim3(blocks), dim3(256), 0, ctx->stream, args.partial, args.C,    \
                     args.bias, M, N, args.ldc, args.tiles_m, args.tiles_n, args.tail_tiles, args.tail_splits, args.accumulate)
    if (BM == 256 && BN == 256) EG_TAIL_REDUCE(256, 256);
    else if (BM == 128 && BN == 128) EG_TAIL_REDUCE(128, 128);
    else if (BM == 128 && BN == 64) EG_TAIL_REDUCE(128, 64);
    else if (BM == 128 && BN == 32) EG_TAIL_REDUCE(128, 32);
    else if (BM == 256 && BN == 64) EG_TAIL_REDUCE(256, 64);
    else EG_TAIL_REDUCE(64, 64);
#undef EG_TAIL_REDUCE
    EG_HIP_CHECK(hipGetLastError());
    return EG_OK;
  }
  if (splits > 1) {
    if (tree_reduce) {
      if (eg::slab_sum_supported(total, args.partial, args.C)) return eg::slab_sum(ctx, splits, total, args.partial, args.C, args.accumulate);
      return eg::colsum_with_scratch(ctx, splits, total, args.partial, args.C, args.accumulate, scratch);
    }
    long blocks = (total + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, args.partial,
                       args.C, args.bias, M, N, args.ldc, launch_splits, args.accumulate, edge_row, edge_splits);
    EG_HIP_CHECK(hipGetLastError());
  }
  return EG_OK;
}

}  // namespace

namespace {

// Tiny contractions (a 32-sample batch through dense(400, 10): 320 outputs, K = 400): one matrix-core
// block would walk K alone for ~20 us.  One WAVE per output element instead: lane l sums
// k = l, l + 64, ... and a shuffle tree folds the 64 partial sums (fixed order: deterministic).
__device__ __forceinline__ void gemm_small_body(long block, const float* __restrict__ A, const float* __restrict__ B, float* C,
                                                const float* __restrict__ bias, long M, long N, long K, long a_sm, long a_sk,
                                                long b_sk, long b_sn, long ldc, int accumulate) {
  const long idx = block * 4 + (threadIdx.x >> 6);  // 4 waves per block, one output each
  if (idx >= M * N) return;
  const int lane = threadIdx.x & 63;
  const long m = idx / N, n = idx - m * N;
  const float* a = A + m * a_sm;
  const float* b = B + n * b_sn;
  float s = 0.f;
  for (long k = lane; k < K; k += 64) s = s + a[k * a_sk] * b[k * b_sk];
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
  if (lane == 0) {
    float* c = C + m * ldc + n;
    float v = s;
    if (accumulate) v = *c + v;
    if (bias) v = v + bias[n];
    *c = v;
  }
}

__global__ __launch_bounds__(256) void gemm_small_kernel(const float* __restrict__ A, const float* __restrict__ B, float* C,
                                                         const float* __restrict__ bias, long M, long N, long K,
                                                         long a_sm, long a_sk, long b_sk, long b_sn, long ldc,
                                                         int accumulate) {
  gemm_small_body(blockIdx.x, A, B, C, bias, M, N, K, a_sm, a_sk, b_sk, b_sn, ldc, accumulate);
}

// Two independent tiny contractions in one launch (a dense layer's weight gradient and input gradient at a small batch:
// in a captured graph each launch costs ~4.5 us whatever it does).  The first blocks0 blocks run the first, the rest the
// second; every output element is computed exactly as by gemm_small_kernel.
__global__ __launch_bounds__(256) void gemm_small_pair_kernel(eg::SmallGemm g0, eg::SmallGemm g1, long blocks0) {
  const bool second = (long)blockIdx.x >= blocks0;   // block-uniform
  const eg::SmallGemm& g = second ? g1 : g0;
  gemm_small_body(second ? (long)blockIdx.x - blocks0 : (long)blockIdx.x, g.A, g.B, g.C, g.bias, g.M, g.N, g.K, g.a_sm, g.a_sk, g.b_sk,
                  g.b_sn, g.ldc, g.accumulate);
}

bool small_gemm_enabled() {
  static const bool on = [] {
    const char* e = eg::sw::raw("EG_NO_SMALL_GEMM");
    return !(e && e[0] && e[0] != '0');
  }();
  return on;
}


bool skinny_enabled() {
  static const bool on = [] {
    const char* e = eg::sw::raw("EG_NO_SKINNY_GEMM");
    return !(e && e[0] && e[0] != '0');
  }();
  return on;
}

}  // namespace

namespace eg {
bool gemm_small_suits(long M, long N, long K) {
  // (2.6 M multiply-adds: above that the matrix tiles are faster since round 4 — 128 x 128 x 256 10.1 us here, 5.2 on eight-wave
  // 32 x 32 tiles; 96 x 96 x 400 10.8 / 6.8; 80 x 160 x 300 10.7 / 5.7; equal at 100 x 128 x 200 and below)
  return M * N <= 16384 && K <= 2048 && M * N * K <= (5L << 19) && K > 0 && M > 0 && N > 0 && small_gemm_enabled();
}

SmallGemm small_gemm(int trans_a, int trans_b, long M, long N, long K, const float* A, long lda, const float* B, long ldb, float* C,
                     long ldc, int accumulate, const float* bias) {
  SmallGemm g = {A, B, C, bias, M, N, K, trans_a ? 1 : lda, trans_a ? lda : 1, trans_b ? 1 : ldb, trans_b ? ldb : 1, ldc, accumulate};
  return g;
}

int gemm_small_pair(eg_ctx* ctx, const SmallGemm& g0, const SmallGemm& g1) {
  int rc = set_device(ctx);
  if (rc) return rc;
  const long blocks0 = (g0.M * g0.N + 3) / 4, blocks1 = (g1.M * g1.N + 3) / 4;
  hipLaunchKernelGGL(gemm_small_pair_kernel, dim3((unsigned)(blocks0 + blocks1)), dim3(256), 0, ctx->stream, g0, g1, blocks0);
  EG_HIP_CHECK(hipGetLastError());
  return EG_OK;
}
}  // namespace eg

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

  if (eg::gemm_small_suits(M, N, K)) {
    const long a_sm = trans_a ? 1 : lda, a_sk = trans_a ? lda : 1;  // A(m, k)
    const long b_sk = trans_b ? 1 : ldb, b_sn = trans_b ? ldb : 1;  // B(k, n)
    const long blocks = (M * N + 3) / 4;
    hipLaunchKernelGGL(gemm_small_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, A, B, C, bias, (long)M, (long)N,
                       (long)K, a_sm, a_sk, b_sk, b_sn, (long)ldc, accumulate);
    EG_HIP_CHECK(hipGetLastError());
    return EG_OK;
  }
  // tall and skinny: stream A once with B resident in LDS
  if (!trans_a && !trans_b && N <= 16 && K >= 64 && K <= 1024 && K % 16 == 0 && M >= 4096 && lda % 4 == 0 && aligned16(A) &&
      skinny_enabled()) {
    const long groups = (M + 15) / 16;
    long blocks = (groups + 3) / 4;
    const long cap = 8L * ctx->compute_units;  // (LDS: K x 64 bytes per block; eight blocks of four waves per CU)
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL((eg_skinny::gemm_skinny_nn_kernel<4>), dim3((unsigned)blocks), dim3(256), (size_t)K * 64, ctx->stream, A, B, C, bias, (long)M,
                       (int)N, (int)K, (long)lda, (long)ldb, (long)ldc, accumulate);
    EG_HIP_CHECK(hipGetLastError());
    return EG_OK;
  }
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
  // (the LDS-DMA loaders address a tile with 32-bit byte offsets from its origin: 256 rows x ld x 4 bytes < 2^31)
  const bool vec_a = (lda % 4 == 0) && (a_contig % 4 == 0) && (A == nullptr || aligned16(A)) && lda < (1L << 21);
  const bool vec_b = (ldb % 4 == 0) && (b_contig % 4 == 0) && (B == nullptr || aligned16(B)) && ldb < (1L << 21);
  constexpr bool no_mixed = false;
  return run_gemm(ctx, a_kc, b_kc, args, /*conv=*/0, vec_a && vec_b, vec_a && !vec_b && !no_mixed);
}

// C[0..M) = op(A) * op(B) and C[M] = column sums of op(B) in ONE contraction: A gets a virtual last row
// of ones (GemmArgs::ones_row).  Used by the model layer to let a bias gradient ride along with the
// weight gradient that reduces over the same batch; C must have room for M + 1 rows.  Returns
// EG_ERR_UNSUPPORTED when the operands do not qualify for the LDS-DMA loop (the caller then runs the
// two reductions separately).
namespace eg {
namespace gemm {
bool ones_row_supported(int trans_a, int trans_b, long M, long N, long K, const float* A, long lda, const float* B, long ldb) {
  const bool a_kc = !trans_a, b_kc = trans_b != 0;
  const long a_contig = a_kc ? K : M, b_contig = b_kc ? K : N;
  const bool vec_a = (lda % 4 == 0) && (a_contig % 4 == 0) && aligned16(A) && lda < (1L << 21);
  const bool vec_b = (ldb % 4 == 0) && (b_contig % 4 == 0) && aligned16(B) && ldb < (1L << 21);
  // large enough for the matrix-core path (not the one-wave-per-output kernel) and at least one k-tile
  const char* off = eg::sw::raw("EG_NO_ONES_ROW");
  return vec_a && vec_b && K >= 16 && !((M + 1) * N <= 16384 && K <= 2048) && !(off && off[0] && off[0] != '0');
}

int sgemm_ones_row(eg_ctx* ctx, int trans_a, int trans_b, long M, long N, long K, const float* A, long lda, const float* B,
                   long ldb, float* C, long ldc, int accumulate) {
  EG_REQUIRE(ctx, EG_ERR_INVALID, "sgemm_ones_row: ctx is NULL");
  // an empty batch (K = 0, possibly NULL operands), a tiny problem, unaligned operands: the caller's two-call form
  if (!A || !B || !C || M <= 0 || N <= 0 || !ones_row_supported(trans_a, trans_b, M, N, K, A, lda, B, ldb)) return EG_ERR_UNSUPPORTED;
  int rc = eg::set_device(ctx);
  if (rc) return rc;
  GemmArgs args = {};
  args.A = A;
  args.B = B;
  args.C = C;
  args.M = M + 1;
  args.a_rows = M;
  args.ones_row = 1;
  if (!ctx->ones) {
    static const float values[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 0.f, 0.f, 0.f};
    EG_HIP_CHECK(hipMalloc((void**)&ctx->ones, sizeof(values)));
    EG_HIP_CHECK(hipMemcpy(ctx->ones, values, sizeof(values), hipMemcpyHostToDevice));
  }
  args.ones = ctx->ones;
  args.N = N;
  args.K = K;
  args.lda = lda;
  args.ldb = ldb;
  args.ldc = ldc;
  args.accumulate = accumulate;
  return run_gemm(ctx, !trans_a, trans_b != 0, args, /*conv=*/0, /*vec_ok=*/true);
}
}  // namespace gemm
}  // namespace eg

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
  if (FH == 1 && FW == 1 && C > 0)  // a 1x1 filter bank is a plain contraction over the channels: out[P,F] = img[P,C] * flt[F,C]^T
    return eg_sgemm(ctx, 0, 1, N * H * W, F, C, img, C, flt, C, out, F, accumulate, nullptr);
  if (C > 0) {  // a few million multiply-adds in all (a batch-32 step of a small network): one thread per output element
    bool launched = false;
    rc = eg::conv2_tiny_forward_try(ctx, N, H, W, C, F, FH, FW, img, flt, out, accumulate, &launched);
    if (rc || launched) return rc;
  }
  if (C > 0 && C <= 16 && F <= 16) {  // few channels and few filters: the filter bank as fragments of 16 x 16 x 4 matrix instructions (conv2_band.cpp)
    bool launched = false;
    rc = eg::conv2_band_forward_try(ctx, false, N, H, W, C, F, FH, FW, img, flt, out, accumulate, &launched);
    if (rc || launched) return rc;
  }
  if (C > 0 && C <= 16) {  // few channels (an image network's first layers): per-pixel kernel specialised for the filter geometry
    bool launched = false;
    rc = eg::conv2_direct_try(ctx, N, H, W, C, F, FH, FW, img, flt, out, accumulate, &launched);
    if (rc || launched) return rc;
  }
  if (C > 0) {  // 3x3-class filters on full-sized images: the LDS-halo kernel
    bool launched = false;
    rc = eg::conv2_halo_try(ctx, N, H, W, C, F, FH, FW, img, flt, out, accumulate, &launched);
    if (rc || launched) return rc;
  }
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
  rc = run_conv(ctx, args, vec);
  if (rc >= 0) return rc;
  return run_gemm(ctx, true, true, args, /*conv=*/1, vec);
}

// ---- convolution gradients ------------------------------------------------------------------------
// What derive (passes.nim:383-549) produces for conv2 (dnn.nim:45-49):
//   gFlt[f,dy,dx,c]     ++= gOut[n,y,x,f] * img[n,y+dy,x+dx,c]
//   gImg[n,y+dy,x+dx,c] ++= gOut[n,y,x,f] * flt[f,dy,dx,c]
// The reference runs both as the same 7-deep loop nest as the forward pass (the second one as a
// scatter).  Here both are contractions on the matrix cores.

namespace {

// Operands of the image gradient in ONE launch:
//  - blocks [0, pad_blocks): gOut [N,Ho,Wo,F] -> zero-bordered [N, Ho + 2(FH-1), Wo + 2(FW-1), F], one
//    thread per VEC floats of one padded pixel; 32-bit index arithmetic (the host checks the sizes);
//  - the remaining blocks: flt [F,FH,FW,C] -> [C,FH,FW,F] with both spatial axes reversed.
template <int VEC>
__global__ __launch_bounds__(256) void grad_image_operands_kernel(const float* __restrict__ g, float* __restrict__ out,
                                                                  unsigned pixels, unsigned Hp, unsigned Wp, unsigned Ho,
                                                                  unsigned Wo, unsigned F, unsigned ph, unsigned pw,
                                                                  unsigned pad_blocks, const float* __restrict__ flt,
                                                                  float* __restrict__ flipped, unsigned FH, unsigned FW,
                                                                  unsigned C) {
  if (blockIdx.x >= pad_blocks) {
    const unsigned total = F * FH * FW * C;
    const unsigned stride = (gridDim.x - pad_blocks) * blockDim.x;
    for (unsigned i = (blockIdx.x - pad_blocks) * blockDim.x + threadIdx.x; i < total; i += stride) {
      const unsigned f = i % F, p = i / F;
      const unsigned dx = p % FW, q = p / FW;
      const unsigned dy = q % FH, c = q / FH;
      flipped[i] = flt[((size_t)(f * FH + (FH - 1 - dy)) * FW + (FW - 1 - dx)) * C + c];
    }
    return;
  }
  const unsigned per_pixel = F / VEC;
  const unsigned total = pixels * per_pixel;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += pad_blocks * blockDim.x) {
    const unsigned p = i / per_pixel, f = (i - p * per_pixel) * VEC;
    const unsigned xp = p % Wp, q = p / Wp;
    const unsigned yp = q % Hp, n = q / Hp;
    const unsigned y = yp - ph, x = xp - pw;  // wraps to a huge value left of / above the image
    const bool inside = y < Ho && x < Wo;
    const size_t src = ((size_t)(n * Ho + y) * Wo + x) * F + f;
    if (VEC == 4) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (inside) v = *reinterpret_cast<const f32x4*>(g + src);
      *reinterpret_cast<f32x4*>(out + (size_t)p * F + f) = v;
    } else {
      out[(size_t)p * F + f] = inside ? g[src] : 0.f;
    }
  }
}

}  // namespace

// Filter gradient as ONE contraction over all output pixels:
//   gFlt[F, FH*FW*C] (+)= gOut^T [F, P] * im2col(img) [P, FH*FW*C],  P = N*Ho*Wo
// A = gOut is a plain [P][F] matrix (m-contiguous); B is gathered from the image inside the tile
// loader (CONV = 2), never materialised.  K = P is long and the output small: split-K.
extern "C" int eg_conv2_nhwc_grad_filter(eg_ctx* ctx, int64_t N, int64_t H, int64_t W, int64_t C, int64_t F, int64_t FH,
                                         int64_t FW, const float* img, const float* gout, float* gflt,
                                         int accumulate) {
  EG_REQUIRE(ctx, EG_ERR_INVALID, "eg_conv2_nhwc_grad_filter: ctx is NULL");
  EG_REQUIRE(N >= 0 && H >= 0 && W >= 0 && C >= 0 && F >= 0 && FH >= 1 && FW >= 1, EG_ERR_INVALID,
             "eg_conv2_nhwc_grad_filter: bad extent");
  const long Ho = H - FH + 1, Wo = W - FW + 1;
  EG_REQUIRE(Ho >= 0 && Wo >= 0, EG_ERR_SHAPE, "eg_conv2_nhwc_grad_filter: filter larger than image");
  if (F == 0 || C == 0) return EG_OK;
  EG_REQUIRE(gflt, EG_ERR_INVALID, "eg_conv2_nhwc_grad_filter: NULL tensor");
  int rc = eg::set_device(ctx);
  if (rc) return rc;
  const long P = N * Ho * Wo;
  if (P == 0) {
    if (!accumulate) return eg_fill_f32(ctx, F * FH * FW * C, 0.f, gflt);
    return EG_OK;
  }
  EG_REQUIRE(img && gout, EG_ERR_INVALID, "eg_conv2_nhwc_grad_filter: NULL tensor");
  if (FH == 1 && FW == 1)  // plain contraction: gflt[F,C] = gout[P,F]^T * img[P,C]
    return eg_sgemm(ctx, 1, 0, F, C, P, gout, F, img, C, gflt, C, accumulate, nullptr);
  {  // a few million multiply-adds in all: blocks of pixels, every output element per block, slabs folded in a fixed order
    bool launched = false;
    rc = eg::conv2_tiny_grad_filter_try(ctx, N, H, W, C, F, FH, FW, img, gout, gflt, accumulate, &launched);
    if (rc || launched) return rc;
  }
  if (C <= 16 && F <= 16) {  // few channels and few filters: 16 filter rows x 4 pixels x 16 taps per matrix instruction (conv2_band.cpp)
    bool launched = false;
    rc = eg::conv2_band_grad_filter_try(ctx, false, N, H, W, C, F, FH, FW, img, gout, gflt, accumulate, &launched);
    if (rc || launched) return rc;
  }
  if (C <= 4) {
    bool launched = false;
    rc = eg::conv2_direct_grad_filter_try(ctx, N, H, W, C, F, FH, FW, img, gout, gflt, accumulate, &launched);
    if (rc || launched) return rc;
  }
  EG_REQUIRE(P < (1L << 31) && FH * FW * C < (1L << 31), EG_ERR_INVALID,
             "eg_conv2_nhwc_grad_filter: more than 2^31 output pixels or taps");
  if (FH == 3 && FW == 3 && C % 32 == 0 && F % 32 == 0) {  // the halo form: every image pixel staged once per row step, not once per tap
    bool launched = false;
    rc = eg::conv2_gradf_halo_try(ctx, N, H, W, C, F, FH, FW, img, gout, gflt, accumulate, &launched);
    if (rc || launched) return rc;
  }
  GemmArgs args = {};
  args.A = gout;
  args.B = img;
  args.C = gflt;
  args.M = F;
  args.N = FH * FW * C;
  args.K = P;
  args.lda = F;
  args.ldb = 0;
  args.ldc = args.N;
  args.accumulate = accumulate;
  args.cH = H;
  args.cW = W;
  args.cC = C;
  args.cFW = FW;
  args.cHo = Ho;
  args.cWo = Wo;
  const bool vec = (C % 4 == 0) && (F % 4 == 0) && aligned16(img) && aligned16(gout);
  return run_gemm(ctx, false, false, args, /*conv=*/2, vec);
}

// Image gradient = "full" correlation of gOut with the flipped, channel-transposed filters:
//   gImg[n,yy,xx,c] = sum_{dy,dx,f} pad(gOut)[n, yy+dy', xx+dx', f] * fltT[c, dy', dx', f],
//   dy' = FH-1-dy, dx' = FW-1-dx, pad = (FH-1, FW-1) zeros on every side.
// That is exactly a forward convolution with F and C exchanged, so it runs on eg_conv2_nhwc
// (LDS-DMA gather and all); the padded gradient and the flipped bank live in the context's
// auxiliary scratch.
extern "C" int eg_conv2_nhwc_grad_image(eg_ctx* ctx, int64_t N, int64_t H, int64_t W, int64_t C, int64_t F, int64_t FH,
                                        int64_t FW, const float* flt, const float* gout, float* gimg, int accumulate) {
  EG_REQUIRE(ctx, EG_ERR_INVALID, "eg_conv2_nhwc_grad_image: ctx is NULL");
  EG_REQUIRE(N >= 0 && H >= 0 && W >= 0 && C >= 0 && F >= 0 && FH >= 1 && FW >= 1, EG_ERR_INVALID,
             "eg_conv2_nhwc_grad_image: bad extent");
  const long Ho = H - FH + 1, Wo = W - FW + 1;
  EG_REQUIRE(Ho >= 0 && Wo >= 0, EG_ERR_SHAPE, "eg_conv2_nhwc_grad_image: filter larger than image");
  if (N == 0 || H == 0 || W == 0 || C == 0) return EG_OK;
  EG_REQUIRE(gimg, EG_ERR_INVALID, "eg_conv2_nhwc_grad_image: NULL tensor");
  int rc = eg::set_device(ctx);
  if (rc) return rc;
  if (F == 0 || Ho == 0 || Wo == 0) {
    if (!accumulate) return eg_fill_f32(ctx, N * H * W * C, 0.f, gimg);
    return EG_OK;
  }
  EG_REQUIRE(flt && gout, EG_ERR_INVALID, "eg_conv2_nhwc_grad_image: NULL tensor");
  if (FH == 1 && FW == 1)  // plain contraction: gimg[P,C] = gout[P,F] * flt[F,C]
    return eg_sgemm(ctx, 0, 0, N * H * W, C, F, gout, F, flt, C, gimg, C, accumulate, nullptr);
  {  // a few million multiply-adds in all: one thread per image element, no flipped bank, no padded gradient
    bool launched = false;
    rc = eg::conv2_tiny_grad_image_try(ctx, N, H, W, C, F, FH, FW, flt, gout, gimg, accumulate, &launched);
    if (rc || launched) return rc;
  }
  if (C <= 16 && F <= 16) {  // few channels and few filters: the forward band kernel on the gradient with a virtual zero border and the bank read flipped
    bool launched = false;
    rc = eg::conv2_band_grad_image_try(ctx, false, N, H, W, C, F, FH, FW, flt, gout, gimg, accumulate, &launched);
    if (rc || launched) return rc;
  }
  const long Hp = Ho + 2 * (FH - 1), Wp = Wo + 2 * (FW - 1);
  const size_t flt_floats = (size_t)(C * FH * FW * F);
  EG_REQUIRE(flt_floats < (1UL << 32), EG_ERR_INVALID, "eg_conv2_nhwc_grad_image: filter bank exceeds 2^32 elements");
  // The halo kernel pads by itself (pixels outside the gradient come from a block of zeros): only the flipped bank is
  // prepared — no padded copy of the gradient is written and read back (cfg 4: 2 x 17 MB, 8 -> 3 us of preparation).
  // EG_CONV_NO_VIRTUAL_PAD=1: the padded copy of rounds 1 and 2.
  const bool virtual_pad = eg::sw::raw("EG_CONV_NO_VIRTUAL_PAD") == nullptr;  // (read per call: a test compares the two routes)
  if (virtual_pad && FH <= 3 && FW <= 3 && F % 16 == 0 && aligned16(gout) &&
      eg::conv2_halo_suits(ctx, N, Ho, Wo, F, C, FH, FW, FH - 1, FW - 1, gout, true)) {   // (the bank goes to ctx->aux: aligned)
    rc = eg::ensure_aux(ctx, flt_floats * sizeof(float));
    if (rc) return rc;
    float* flipped_only = static_cast<float*>(ctx->aux);
    const long fb = std::min<long>(((long)flt_floats + 255) / 256, 2L * ctx->compute_units);
    hipLaunchKernelGGL(grad_image_operands_kernel<1>, dim3((unsigned)fb), dim3(256), 0, ctx->stream, gout, flipped_only, 0u,
                       (unsigned)Hp, (unsigned)Wp, (unsigned)Ho, (unsigned)Wo, (unsigned)F, (unsigned)(FH - 1), (unsigned)(FW - 1),
                       0u, flt, flipped_only, (unsigned)FH, (unsigned)FW, (unsigned)C);
    EG_HIP_CHECK(hipGetLastError());
    bool launched = false;
    rc = eg::conv2_halo_try_padded(ctx, N, Ho, Wo, F, C, FH, FW, FH - 1, FW - 1, gout, flipped_only, gimg, accumulate, &launched);
    if (rc || launched) return rc;
  }
  const size_t pad_floats = ((size_t)(N * Hp * Wp * F) + 3) & ~(size_t)3;
  rc = eg::ensure_aux(ctx, (pad_floats + flt_floats) * sizeof(float));
  if (rc) return rc;
  float* padded = static_cast<float*>(ctx->aux);
  float* flipped = padded + pad_floats;
  EG_REQUIRE(N * Hp * Wp * F < (1L << 32), EG_ERR_INVALID, "eg_conv2_nhwc_grad_image: padded gradient exceeds 2^32 elements");
  const bool vec4 = F % 4 == 0 && aligned16(gout);
  const long work = N * Hp * Wp * (vec4 ? F / 4 : F);
  const long blocks = std::min<long>((work + 255) / 256, 16L * ctx->compute_units);
  const long fblocks = std::min<long>(((long)flt_floats + 255) / 256, 2L * ctx->compute_units);
  if (vec4)
    hipLaunchKernelGGL(grad_image_operands_kernel<4>, dim3((unsigned)(blocks + fblocks)), dim3(256), 0, ctx->stream, gout,
                       padded, (unsigned)(N * Hp * Wp), (unsigned)Hp, (unsigned)Wp, (unsigned)Ho, (unsigned)Wo,
                       (unsigned)F, (unsigned)(FH - 1), (unsigned)(FW - 1), (unsigned)blocks, flt, flipped, (unsigned)FH,
                       (unsigned)FW, (unsigned)C);
  else
    hipLaunchKernelGGL(grad_image_operands_kernel<1>, dim3((unsigned)(blocks + fblocks)), dim3(256), 0, ctx->stream, gout,
                       padded, (unsigned)(N * Hp * Wp), (unsigned)Hp, (unsigned)Wp, (unsigned)Ho, (unsigned)Wo,
                       (unsigned)F, (unsigned)(FH - 1), (unsigned)(FW - 1), (unsigned)blocks, flt, flipped, (unsigned)FH,
                       (unsigned)FW, (unsigned)C);
  EG_HIP_CHECK(hipGetLastError());
  return eg_conv2_nhwc(ctx, N, Hp, Wp, F, C, FH, FW, padded, flipped, gimg, accumulate);
}

// ---- contraction with a generated epilogue (gemm_fused.hpp) ----------------------------------------
#include "gemm_fused.hpp"

namespace {
const char* const kGemmHeaderText =
#include "gemm_src.inc"
    ;
}

namespace eg {
namespace gemm {

static_assert(sizeof(GemmArgs) <= sizeof(FusedLaunch::args), "FusedLaunch::args too small");

long long* trace_begin(eg_ctx* ctx, unsigned blocks, unsigned waves) {
  long long* buffer = nullptr;
  const size_t bytes = (size_t)blocks * waves * 4 * sizeof(long long);
  if (hipMalloc((void**)&buffer, bytes) != hipSuccess) return nullptr;
  if (hipMemsetAsync(buffer, 0, bytes, ctx->stream) != hipSuccess) {
    (void)hipFree(buffer);
    return nullptr;
  }
  return buffer;
}

void fused_set_trace(FusedLaunch& f, long long* buffer) {
  GemmArgs a;
  memcpy(&a, f.args, sizeof(a));
  a.trace = buffer;
  memcpy(f.args, &a, sizeof(a));
}

void trace_end(eg_ctx* ctx, long long* buffer, unsigned blocks, unsigned waves, const char* what) {
  if (!buffer) return;
  std::vector<long long> h((size_t)blocks * waves * 4);
  if (hipStreamSynchronize(ctx->stream) == hipSuccess &&
      hipMemcpy(h.data(), buffer, h.size() * sizeof(long long), hipMemcpyDeviceToHost) == hipSuccess) {
    double sum[4] = {0, 0, 0, 0}, mx[4] = {0, 0, 0, 0};
    long cnt = 0;
    for (size_t w = 0; w < (size_t)blocks * waves; ++w) {
      if (h[w * 4] == 0 || h[w * 4 + 3] == 0) continue;
      ++cnt;
      for (int k = 1; k < 4; ++k) {
        const double d = (double)(h[w * 4 + k] - h[w * 4]);
        sum[k] += d;
        mx[k] = d > mx[k] ? d : mx[k];
      }
    }
    if (cnt)
      fprintf(stderr, "[eg] gemm trace %s (%u blocks x %u waves; cycles since wave start, mean / max): k loop begins %.0f / %.0f, ends %.0f / %.0f, "
                      "epilogue done %.0f / %.0f\n", what, blocks, waves, sum[1] / cnt, mx[1], sum[2] / cnt, mx[2], sum[3] / cnt, mx[3]);
  }
  (void)hipFree(buffer);
}
static_assert(MAX_EPILOGUE_OPERANDS == sizeof(GemmArgs::epi) / sizeof(void*), "epilogue operand count");

int plan_fused(eg_ctx* ctx, int trans_a, int trans_b, long M, long N, long K, const float* A, long lda, const float* B,
               long ldb, float* C, long ldc, const float* bias, FusedLaunch& out) {
  EG_REQUIRE(ctx && M > 0 && N > 0 && K >= 0 && C, EG_ERR_INVALID, "plan_fused: bad problem");
  const bool a_kc = !trans_a, b_kc = trans_b != 0;
  int bm, bn, splits;
  choose_tile(ctx, M, N, K, bm, bn, splits);
  out = FusedLaunch();
  out.bm = bm;
  out.bn = bn;
  out.bk = BK;
  out.splits = splits;
  if (bn == 32) { out.wm = 32; out.wn = 32; out.minb = 4; }
  else if (bn == 64 && bm == 256) { out.wm = 64; out.wn = 32; out.minb = 2; }
  else if (bn == 64 && bm == 64) { out.wm = 32; out.wn = 32; out.minb = 4; }
  else if (bn == 64) { out.wm = 64; out.wn = 32; out.minb = 4; }
  else if (bn == 128) { out.wm = 64; out.wn = 64; out.minb = 4; }
  else { out.wm = 128; out.wn = 64; out.minb = 1; }
  out.nt = (bm / out.wm) * (bn / out.wn) * 64;
  out.a_kc = a_kc;
  out.b_kc = b_kc;
  const long a_contig = a_kc ? K : M, b_contig = b_kc ? K : N;
  const bool vec = (lda % 4 == 0) && (ldb % 4 == 0) && (a_contig % 4 == 0) && (b_contig % 4 == 0) && aligned16(A) &&
                   aligned16(B) && lda < (1L << 21) && ldb < (1L << 21);
  out.edge = !(vec && M % bm == 0 && N % bn == 0 && K % BK == 0 && K > 0);
  out.vec = (!out.edge || vec) ? 4 : 1;
  out.dma = out.vec == 4;
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
  args.accumulate = 0;
  args.a_rows = M;
  args.tiles_m = (int)((M + bm - 1) / bm);
  args.tiles_n = (int)((N + bn - 1) / bn);
  args.partial = nullptr;
  args.k_per_split = ((K + BK - 1) / BK) * BK;
  if (args.k_per_split < BK) args.k_per_split = BK;
  out.grid = (unsigned)(args.tiles_m * args.tiles_n);
  args.wide_store = wide_store_ok(args, false, true);   // set_epilogue_operands withdraws it for unaligned operands
  args.prio = side_priority(ctx);
  args.nt_store = nt_store_enabled();
  args.no_skew = eg::sw::raw("EG_GEMM_NO_SKEW") != nullptr;
  memcpy(out.args, &args, sizeof(args));
  out.args_size = sizeof(args);
  {  // tiny K: a store stream, not matrix work (gemm_narrow_k_block)
    const bool off = eg::sw::raw("EG_NO_NARROW_K") != nullptr;   // (read per call: a test compares the two routes)
    const long tpr = N / 4;
    if (!off && K >= 1 && K <= 16 && N % 4 == 0 && tpr >= 1 && tpr <= 256 && 256 % tpr == 0 && ldc % 4 == 0 && aligned16(C) &&
        (!bias || aligned16(bias)) && M * N >= (1L << 16)) {
      out.narrow = true;
      out.narrow_k = (int)K;
      const long rows_per_trip = 4 * (256 / tpr), blocks = (M + rows_per_trip - 1) / rows_per_trip;
      // eight blocks per CU (all resident at once): every block first loads its threads' 4 x K values of B — 40 KB per block
      // at K = 10 — so more, shorter blocks cost more than they gain here (65 536 x 512 x 10, 1 024 / 2 048 / 4 096 / 8 192
      // blocks: 28.8 / 24.3 / 28.4 / 37.2 us standalone)
      const long cap = 8L * ctx->compute_units;
      long grid = blocks < cap ? blocks : cap;
      // a block's run of rows: a multiple of what it takes per trip; the kernel reads it from k_per_split (no use for it there)
      const long per = ((M + grid - 1) / grid + rows_per_trip - 1) / rows_per_trip * rows_per_trip;
      grid = (M + per - 1) / per;
      out.narrow_grid = (unsigned)grid;
      out.matrix_k_per_split = args.k_per_split;
      args.k_per_split = per;
      memcpy(out.args, &args, sizeof(args));
    }
  }
  return EG_OK;
}

void set_epilogue_operands(FusedLaunch& f, void* const* ptrs, int count, float grad_scale, long epoch) {
  GemmArgs* a = reinterpret_cast<GemmArgs*>(f.args);
  for (int i = 0; i < MAX_EPILOGUE_OPERANDS; ++i) a->epi[i] = i < count ? ptrs[i] : nullptr;
  for (int i = 0; i < count; ++i)
    if (!aligned16(ptrs[i])) {
      a->wide_store = 0;
      fused_withdraw_narrow(f);
    }
  a->epi_gs = grad_scale;
  a->epi_ep = epoch;
}

void fused_withdraw_narrow(FusedLaunch& f) {
  if (!f.narrow) return;
  f.narrow = false;
  reinterpret_cast<GemmArgs*>(f.args)->k_per_split = f.matrix_k_per_split;
}

bool fused_wide_store(const FusedLaunch& f) {
  // (the streaming kernel covers every column of a row: predicate words are stored whole when a word is eight threads)
  if (f.narrow) return reinterpret_cast<const GemmArgs*>(f.args)->ldc % 32 == 0 && (reinterpret_cast<const GemmArgs*>(f.args)->N / 4) % 8 == 0;
  return reinterpret_cast<const GemmArgs*>(f.args)->wide_store != 0 && !f.edge;
}

std::string fused_variant(const FusedLaunch& f) {
  char buf[96];
  if (f.narrow) {
    snprintf(buf, sizeof(buf), "narrow_k%d_n%ld_%c%c", f.narrow_k, (long)reinterpret_cast<const GemmArgs*>(f.args)->N, f.a_kc ? 'k' : 'm',
             f.b_kc ? 'k' : 'n');
    return buf;
  }
  snprintf(buf, sizeof(buf), "%dx%dx%d_%dx%d_%d_%c%c_v%d%s%s", f.bm, f.bn, f.bk, f.wm, f.wn, f.minb, f.a_kc ? 'k' : 'm',
           f.b_kc ? 'k' : 'n', f.vec, f.edge ? "_edge" : "", f.dma ? "_dma" : "");
  return buf;
}

std::string fused_source(const FusedLaunch& f, const std::string& epi_struct, const std::string& epi_name,
                         const std::string& kernel_name) {
  std::string s = kGemmHeaderText;
  s += "\n" + epi_struct + "\n";
  char buf[512];
  if (f.narrow) {
    snprintf(buf, sizeof(buf),
             "extern \"C\" __global__ __launch_bounds__(256) void %s(eg::gemm::GemmArgs a) {\n"
             "  eg::gemm::gemm_narrow_k_block<%d, %d, %s, %s, %s>(a);\n}\n",
             kernel_name.c_str(), f.narrow_k, (int)(reinterpret_cast<const GemmArgs*>(f.args)->N / 4), f.a_kc ? "true" : "false",
             f.b_kc ? "true" : "false", epi_name.c_str());
    s += buf;
    return s;
  }
  const int waves = f.nt / 64;
  // a ragged tile with a generated epilogue needs a few registers more than four waves per SIMD leave
  // (48 bytes of scratch at 128 VGPRs): three there
  int per_simd = (f.minb * waves + 3) / 4;
  if (f.edge && per_simd >= 4) per_simd = 3;
  snprintf(buf, sizeof(buf),
           "extern \"C\" __global__ __launch_bounds__(%d, %d) void %s(eg::gemm::GemmArgs a) {\n"
           "  eg::gemm::gemm_block<%d, %d, %d, %d, %d, %s, %s, %d, %s, false, 0, %s, %s>(a);\n}\n",
           f.nt, per_simd, kernel_name.c_str(), f.bm, f.bn, f.bk, f.wm, f.wn, f.a_kc ? "true" : "false",
           f.b_kc ? "true" : "false", f.vec, f.edge ? "true" : "false", f.dma ? "true" : "false", epi_name.c_str());
  s += buf;
  return s;
}

}  // namespace gemm
}  // namespace eg
