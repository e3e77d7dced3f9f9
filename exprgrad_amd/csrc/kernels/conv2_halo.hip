// Direct convolution with an LDS-resident input halo (small filters, channels a multiple of 16).
//
//   out[n,y,x,f] (+)= sum_{dy,dx,c} img[n,y+dy,x+dx,c] * flt[f,dy,dx,c]        dnn.nim:45-49
//
// The implicit-GEMM path (gemm_f32_mfma.hpp, CONV = 1) gathers the im2col rows of a 64-pixel tile
// once per filter tap: every input pixel travels L2 -> LDS FH*FW times and a block synchronises
// after every 32 values of k.  For 3x3-class filters this kernel turns the loop nest around:
//   * a block owns an 8 x 32 patch of output pixels of one image and 64 filters (8 waves; wave w
//     owns output row w: a 32-pixel x 64-filter accumulator tile);
//   * the channels are walked in chunks of 16; for a chunk the (8+FH-1) x (32+FW-1) input halo
//     and the FH*FW x 64 x 16 filter slab are brought into LDS by LDS-DMA
//     (`global_load_lds_dwordx4`, no staging registers), double buffered against the matrix work;
//   * inside a chunk all FH*FW taps are multiplied straight out of the halo — the A fragment of
//     tap (dy,dx) is the halo shifted by (dy,dx), read with one conflict-free ds_read_b128 per four
//     MFMA k-steps — with no barrier between taps: one barrier per 16 channels instead of one per
//     32 values of k, and every input pixel crosses L2 -> LDS once per block instead of FH*FW times.
// LDS images are lane-linear (DMA), so both tiles use the 16-float row layout of the GEMM's
// k-contiguous operands: chunk c of row r sits in slot c ^ ((r >> 2) & 3), the permutation applied
// to the global address.  With 32 consecutive halo pixels per wave the four 16-lane service groups of
// ds_read_b128 ({0-3,12-15,20-27}, ...) touch 16 distinct 16-byte slots for every tap shift (a
// 16 x 16 patch with two rows per wave measured 25 % bank-conflict cycles).
//
// Out-of-range rows/columns of edge patches read clamped (valid) addresses and are not stored.
#include <hip/hip_runtime.h>

#include <cstdlib>

// the run-time-sized variant keeps its tap loops rolled
#pragma clang diagnostic ignored "-Wpass-failed"

#include "../eg_internal.hpp"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int TH = 8, TW = 32;    // output patch: one row of 32 pixels per wave
constexpr int FB = 64;            // filters per block
constexpr int CK = 16;            // channels per chunk
constexpr int NT = 512;           // 8 waves
constexpr int MAX_TAPS = 9;

struct HaloArgs {
  const float* img;
  const float* flt;
  float* out;
  long N, H, W, C, F, FH, FW, Ho, Wo;
  // Virtual zero padding (the image gradient: a "full" correlation of the output gradient, gemm_f32_mfma.hip
  // eg_conv2_nhwc_grad_image): output pixel (y, x) reads image rows y - py .. and columns x - px ..; pixels outside the
  // image come from `zeros` (>= C floats of zeros) instead of a padded copy of the image.  Ho = H + 2 py - FH + 1.
  long py, px;
  const float* zeros;
  int tiles_x, tiles_y, tiles_f;
  int accumulate;
  int wide_store;  // whole patches leave through LDS as 16-byte stores (needs F % 4 == 0 and a 16-byte aligned output)
  long items;  // N * tiles_y * tiles_x * tiles_f
};

__device__ __forceinline__ int swz(int r) { return (r >> 2) & 3; }

__device__ __forceinline__ void dma16(const float* src, float* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// HALO_MAX: pixels of the largest halo (10 x 34); the kernel is compiled for FH, FW <= 3.
template <int TAPS_MAX>
struct HaloLds {
  static constexpr int HALO_MAX = (TH + 2) * (TW + 2);
  // every wave issues the same number of 1 KiB DMA instructions per stage, unconditionally (no
  // branches inside a chunk): the regions are rounded up to 8 instructions
  static constexpr int HALO_PER_WAVE = ((HALO_MAX + 15) / 16 + 7) / 8;
  static constexpr int FLT_PER_WAVE = (TAPS_MAX * 4 + 7) / 8;
  static constexpr int HALO_FLOATS = HALO_PER_WAVE * 8 * 256;
  static constexpr int FLT_FLOATS = FLT_PER_WAVE * 8 * 256;
  static constexpr int STAGE = HALO_FLOATS + FLT_FLOATS;
  static constexpr int BYTES = 2 * STAGE * (int)sizeof(float);  // 128 KiB of the CU's 160 KiB
};

// CFH x CFW: compile-time filter size (the tap loops unroll completely and the compiler moves the
// fragment reads of the next taps above the MFMAs of the current one); 0 = run-time size <= 3 x 3.
template <int TAPS_MAX, int CFH, int CFW>
__global__ __launch_bounds__(NT, 2) void conv2_halo_kernel(HaloArgs a) {
  constexpr int HALO_FLOATS = HaloLds<TAPS_MAX>::HALO_FLOATS;
  constexpr int STAGE = HaloLds<TAPS_MAX>::STAGE;
  extern __shared__ __attribute__((aligned(16))) float lds[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 31, hi = lane >> 5;

  const int FH = CFH ? CFH : (int)a.FH, FW = CFW ? CFW : (int)a.FW, taps = FH * FW;
  const int HW = TW + FW - 1, HH = TH + FH - 1;
  const int halo = HH * HW;
  constexpr int HALO_PER_WAVE = HaloLds<TAPS_MAX>::HALO_PER_WAVE;
  constexpr int FLT_PER_WAVE = HaloLds<TAPS_MAX>::FLT_PER_WAVE;

  // Work item -> (image, patch, filter group).  Blocks are persistent: block b runs items b,
  // b + gridDim.x, ... and the LDS double buffer keeps rolling across items, so the first chunk
  // of the next patch is already in flight while the last chunk of this one is multiplied.
  struct Item {
    long n, y0, x0, f0;
  };
  auto decode = [&](long w) {
    Item it;
    it.f0 = (w % a.tiles_f) * FB;
    w /= a.tiles_f;
    it.x0 = (w % a.tiles_x) * TW;
    w /= a.tiles_x;
    it.y0 = (w % a.tiles_y) * TH;
    it.n = w / a.tiles_y;
    return it;
  };

  // ---- per-lane DMA sources of the item being loaded (fixed across its channel chunks)
  // halo: instruction t covers halo pixels 16t .. 16t+15, lane l -> pixel 16t + l/4, slot l%4
  // filters: per tap 64 rows x 4 chunks = 4 instructions; instruction u = tap * 4 + v covers rows
  // 16v .. 16v+15
  const float* halo_src[HALO_PER_WAVE];
  long flt_src[FLT_PER_WAVE];
  auto sources = [&](const Item& it) {
#pragma unroll
    for (int t = 0; t < HALO_PER_WAVE; ++t) {
      const int instr = wave + t * 8;
      int q = instr * 16 + (lane >> 2);
      if (q >= halo) q = halo - 1;
      const int qy = q / HW, qx = q - qy * HW;
      long yy = it.y0 + qy - a.py, xx = it.x0 + qx - a.px;
      const bool outside = yy < 0 || xx < 0 || yy > a.H - 1 || xx > a.W - 1;
      if (yy > a.H - 1) yy = a.H - 1;  // (without padding: the rows / columns of a ragged patch, never stored)
      if (xx > a.W - 1) xx = a.W - 1;
      if (yy < 0) yy = 0;
      if (xx < 0) xx = 0;
      const int chunk = (lane & 3) ^ swz(instr * 16 + (lane >> 2));
      halo_src[t] = a.img + ((it.n * a.H + yy) * a.W + xx) * a.C + chunk * 4;
      if (outside && a.zeros) halo_src[t] = a.zeros + chunk * 4;
    }
#pragma unroll
    for (int t = 0; t < FLT_PER_WAVE; ++t) {
      const int instr = wave + t * 8;
      const int tap = instr >> 2, r = (instr & 3) * 16 + (lane >> 2);
      long f = it.f0 + r;
      if (f > a.F - 1) f = a.F - 1;
      const int chunk = (lane & 3) ^ swz(r);
      flt_src[t] = (f * taps + (tap < taps ? tap : taps - 1)) * a.C + chunk * 4;
    }
  };
  // One DMA instruction of this wave's share of a stage; pieces 0 .. PIECES-1.
  constexpr int PIECES = HALO_PER_WAVE + FLT_PER_WAVE;
  auto issue_piece = [&](int piece, int c0, float* stage) {
    if (piece < HALO_PER_WAVE) {
      dma16(halo_src[piece] + c0, stage + (wave + piece * 8) * 256);
    } else {
      const int t = piece - HALO_PER_WAVE;
      dma16(a.flt + flt_src[t] + c0, stage + HALO_FLOATS + (wave + t * 8) * 256);
    }
  };
  auto issue = [&](int c0, float* stage) {
#pragma unroll
    for (int piece = 0; piece < PIECES; ++piece) issue_piece(piece, c0, stage);
  };

  // this lane's output pixel inside the patch (A-fragment row of the 32-pixel wave tile)
  const int q_base = wave * HW + i;
  const int nchunks = (int)(a.C / CK);

  long w = blockIdx.x;
  if (w >= a.items) return;
  Item cur_item = decode(w);
  sources(cur_item);
  issue(0, lds);
  int stage = 0;
  // every wave waits for ITS OWN LDS-DMA loads before the barrier that publishes them (gemm_f32_mfma.hpp:
  // dma_publish_barrier — the compiler is free to put that wait behind the barrier)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  while (true) {
    // four independent accumulator chains per wave (filters 0-31 / 32-63 x even / odd k-step):
    // with only two, every MFMA waited on the one issued two before it
    f32x16 acc[2], acc_odd[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc[j][r] = 0.f;
        acc_odd[j][r] = 0.f;
      }
    const long w_next = w + gridDim.x;
    for (int ch = 0; ch < nchunks; ++ch) {
      const float* Hs = lds + stage * STAGE;
      const float* Fs = Hs + HALO_FLOATS;
      // The loads of the next stage (next chunk, or chunk 0 of the next item; after the last item a
      // harmless reload of this item's chunk 0) are issued behind the first fragment reads, in the
      // shadow of the MFMAs.  No branch inside the chunk.
      const int c_next = ch + 1 < nchunks ? (ch + 1) * CK : 0;
      float* nxt = lds + (stage ^ 1) * STAGE;
      if (ch + 1 == nchunks && w_next < a.items) sources(decode(w_next));
      auto fragments = [&](int tp, int pp, f32x4& av, f32x4 (&bv)[2]) {
        const int dy = tp / FW, dx = tp - dy * FW;
        const int q = q_base + dy * HW + dx;
        av = *reinterpret_cast<const f32x4*>(Hs + q * CK + (((2 * pp + hi) ^ swz(q)) << 2));
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int f = j * 32 + i;
          bv[j] = *reinterpret_cast<const f32x4*>(Fs + tp * (FB * CK) + f * CK + (((2 * pp + hi) ^ swz(f)) << 2));
        }
      };
      // One filter row (FW taps x 16 channels) per batch: all its fragments are requested first
      // (up to 18 ds_read_b128), then its 8 * FW * 2 MFMAs run while the waits retire in order —
      // a wave waits for LDS once per 48 MFMAs instead of once per 8.
      constexpr int PPS = CK / 8;
      constexpr int ROW_MAX = 3 * PPS;
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        if (dy >= FH) break;
        f32x4 av[ROW_MAX], bv[ROW_MAX][2];
#pragma unroll
        for (int u = 0; u < ROW_MAX; ++u)
          if (u < FW * PPS) fragments(dy * FW + u / PPS, u % PPS, av[u], bv[u]);
#pragma unroll
        for (int piece = 0; piece < PIECES; ++piece)
          if (dy == 0) issue_piece(piece, c_next, nxt);
#pragma unroll
        for (int u = 0; u < ROW_MAX; ++u) {
          if (u >= FW * PPS) break;
#pragma unroll
          for (int k = 0; k < 4; k += 2)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][k], bv[u][j][k], acc[j], 0, 0, 0);
              acc_odd[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][k + 1], bv[u][j][k + 1], acc_odd[j], 0, 0, 0);
            }
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the next chunk's loads have landed ...
      __syncthreads();                                  // ... for every wave; this stage is free for the one after
      stage ^= 1;
    }

    // ---- epilogue: register r of lane l holds pixel p = (r & 3) + 8 * (r >> 2) + 4 * hi of the
    //      wave's row and filter l & 31 (+ 32 j)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] += acc_odd[j][r];
    const bool whole = cur_item.y0 + TH <= a.Ho && cur_item.x0 + TW <= a.Wo && cur_item.f0 + FB <= a.F;
    if (whole && !a.accumulate && a.wide_store) {
      // The wave's 32 pixels x 64 filters leave through LDS: parked as [pixel][filter] in the stage the last chunk has
      // just released (this wave's own 8 KiB of it — no other wave touches them, so no block barrier on the way in),
      // then written as 16 bytes per lane: four whole pixels (1 KiB, contiguous when F = 64) per wave instruction instead
      // of 128-byte pieces of two pixels.  The barrier behind it keeps the next item's DMA (which every wave issues into
      // this stage) away from a slower wave's parked values.
      float* park = lds + (stage ^ 1) * STAGE + wave * (TW * FB);
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int p = (r & 3) + 8 * (r >> 2) + 4 * hi;
          park[p * FB + j * 32 + i] = acc[j][r];
        }
      float* base = a.out + ((cur_item.n * a.Ho + cur_item.y0 + wave) * a.Wo + cur_item.x0) * a.F + cur_item.f0;
#pragma unroll
      for (int it = 0; it < TW * FB / 256; ++it) {
        const int e = it * 256 + lane * 4;  // element of the wave's [32][64] tile
        const f32x4 v = *reinterpret_cast<const f32x4*>(park + e);
        __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(base + (long)(e / FB) * a.F + (e % FB)));
      }
      if (w_next < a.items) __syncthreads();
    } else if (whole && !a.accumulate) {  // branch-free: the 32 stores of a lane are issued back to back
      float* base = a.out + ((cur_item.n * a.Ho + cur_item.y0 + wave) * a.Wo + cur_item.x0) * a.F + cur_item.f0 + i;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int p = (r & 3) + 8 * (r >> 2) + 4 * hi;
          base[(long)p * a.F + j * 32] = acc[j][r];
        }
    } else {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const long f = cur_item.f0 + j * 32 + i;
        if (f >= a.F) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int p = (r & 3) + 8 * (r >> 2) + 4 * hi;
          const long y = cur_item.y0 + wave, x = cur_item.x0 + p;
          if (y >= a.Ho || x >= a.Wo) continue;
          float* dst = a.out + ((cur_item.n * a.Ho + y) * a.Wo + x) * a.F + f;
          *dst = a.accumulate ? *dst + acc[j][r] : acc[j][r];
        }
      }
    }
    if (w_next >= a.items) break;
    w = w_next;
    cur_item = decode(w);
  }
}

}  // namespace

namespace eg {

// Launches the halo kernel if the problem suits it; *launched tells the caller whether it did.
int conv2_halo_try(eg_ctx* ctx, long N, long H, long W, long C, long F, long FH, long FW, const float* img,
                   const float* flt, float* out, int accumulate, bool* launched) {
  return conv2_halo_try_padded(ctx, N, H, W, C, F, FH, FW, 0, 0, img, flt, out, accumulate, launched);
}

// Whether conv2_halo_try_padded would launch for this problem (the image-gradient route asks before it prepares the
// flipped filter bank: a batch-32 `fit` step paid a 2 us launch for a bank the declining halo kernel never read).
// flt_aligned: the filter bank the caller will pass is 16-byte aligned.
bool conv2_halo_suits(eg_ctx* ctx, long N, long H, long W, long C, long F, long FH, long FW, long py, long px, const float* img,
                      bool flt_aligned) {
  static const bool off = [] {
    const char* e = eg::sw::raw("EG_CONV_NO_HALO");
    return e && e[0] && e[0] != '0';
  }();
  if (off) return false;
  const long Ho = H + 2 * py - FH + 1, Wo = W + 2 * px - FW + 1;
  if (FH > 3 || FW > 3 || C % CK != 0 || C < CK || Ho <= 0 || Wo <= 0) return false;
  if ((reinterpret_cast<uintptr_t>(img) & 15) || !flt_aligned) return false;
  const long tiles_x = (Wo + TW - 1) / TW, tiles_y = (Ho + TH - 1) / TH, tiles_f = (F + FB - 1) / FB;
  const long blocks = N * tiles_y * tiles_x * tiles_f;
  // worth it when the patches are reasonably full and the chip is busy
  const double fill = (double)(Ho * Wo) / (double)(tiles_y * TH * tiles_x * TW) * (double)F / (double)(tiles_f * FB);
  return !(fill < 0.7 || blocks < ctx->compute_units / 2 || blocks > (1L << 30));
}

int conv2_halo_try_padded(eg_ctx* ctx, long N, long H, long W, long C, long F, long FH, long FW, long py, long px,
                          const float* img, const float* flt, float* out, int accumulate, bool* launched) {
  *launched = false;
  if (!conv2_halo_suits(ctx, N, H, W, C, F, FH, FW, py, px, img, (reinterpret_cast<uintptr_t>(flt) & 15) == 0)) return EG_OK;
  const long Ho = H + 2 * py - FH + 1, Wo = W + 2 * px - FW + 1;
  const long tiles_x = (Wo + TW - 1) / TW, tiles_y = (Ho + TH - 1) / TH, tiles_f = (F + FB - 1) / FB;
  const long blocks = N * tiles_y * tiles_x * tiles_f;
  HaloArgs a = {};
  a.img = img;
  a.flt = flt;
  a.out = out;
  a.N = N;
  a.H = H;
  a.W = W;
  a.C = C;
  a.F = F;
  a.FH = FH;
  a.FW = FW;
  a.Ho = Ho;
  a.Wo = Wo;
  a.tiles_x = (int)tiles_x;
  a.tiles_y = (int)tiles_y;
  a.tiles_f = (int)tiles_f;
  a.accumulate = accumulate;
  a.py = py;
  a.px = px;
  a.zeros = nullptr;
  if (py > 0 || px > 0) {
    if (ctx->zeros_floats < (size_t)C) {  // once per context: a block of zeros the padding reads
      // An outgrown block stays allocated until the context goes: launch sequences captured into HIP graphs hold its
      // address in their kernel arguments, and hipFree is not allowed while a stream capture is in progress.  The first
      // block is generous (64 KiB), so in practice there is exactly one.
      if (ctx->zeros) ctx->zeros_retired.push_back(ctx->zeros);
      ctx->zeros = nullptr;
      ctx->zeros_floats = 0;
      const size_t want = std::max<size_t>(16384, ((size_t)C + 1023) & ~(size_t)1023);
      EG_HIP_CHECK(hipMalloc((void**)&ctx->zeros, want * sizeof(float)));
      // on the context's stream: a legacy-stream hipMemset returns before the fill has run and is not ordered against
      // the kernel launched below (the first padded launch of a context read its border from unfilled memory)
      EG_HIP_CHECK(hipMemsetAsync(ctx->zeros, 0, want * sizeof(float), ctx->stream));
      ctx->zeros_floats = want;
    }
    a.zeros = ctx->zeros;
  }
  const void* kernel = FH == 3 && FW == 3   ? reinterpret_cast<const void*>(&conv2_halo_kernel<MAX_TAPS, 3, 3>)
                       : FH == 1 && FW == 1 ? reinterpret_cast<const void*>(&conv2_halo_kernel<MAX_TAPS, 1, 1>)
                                            : reinterpret_cast<const void*>(&conv2_halo_kernel<MAX_TAPS, 0, 0>);
  static const bool lds_ok = [] {
    hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv2_halo_kernel<MAX_TAPS, 3, 3>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, HaloLds<MAX_TAPS>::BYTES);
    hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv2_halo_kernel<MAX_TAPS, 1, 1>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, HaloLds<MAX_TAPS>::BYTES);
    hipError_t e3 = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv2_halo_kernel<MAX_TAPS, 0, 0>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, HaloLds<MAX_TAPS>::BYTES);
    return e1 == hipSuccess && e2 == hipSuccess && e3 == hipSuccess;
  }();
  if (!lds_ok) {  // the device does not grant this much LDS to one block: implicit GEMM instead
    (void)hipGetLastError();
    return EG_OK;
  }
  a.items = blocks;
  static const bool wide_off = eg::sw::raw("EG_CONV_NO_WIDE_STORE") != nullptr;
  a.wide_store = !wide_off && F % 4 == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0;
  // one block per CU (LDS); several rounds of work run as persistent blocks
  const long grid = blocks < (long)ctx->compute_units ? blocks : (long)ctx->compute_units;
  void* params[] = {&a};
  EG_HIP_CHECK(hipLaunchKernel(kernel, dim3((unsigned)grid), dim3(NT), params, HaloLds<MAX_TAPS>::BYTES, ctx->stream));
  EG_HIP_CHECK(hipGetLastError());
  *launched = true;
  return EG_OK;
}

}  // namespace eg
