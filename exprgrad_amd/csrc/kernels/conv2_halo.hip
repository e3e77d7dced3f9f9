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

#include <cstdio>
#include <cstdlib>
#include <vector>

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
  // image are loaded from an offset past the end of the image's buffer descriptor (img_bytes = its num_records), which
  // a buffer load answers with zeros — no padded copy of the image, no block of zeros.  Ho = H + 2 py - FH + 1.
  long py, px;
  unsigned img_bytes;  // N * H * W * C * 4 when there is padding, 0 = no range check
  int tiles_x, tiles_y, tiles_f;
  int accumulate;
  int wide_store;  // whole patches leave through LDS as 16-byte stores (needs F % 4 == 0 and a 16-byte aligned output)
  long items;  // N * tiles_y * tiles_x * tiles_f
  long long* trace;  // EG_HALO_TRACE=1 (debugging aid): cycle stamps of every wave, 32 per wave
};

__device__ __forceinline__ int swz(int r) { return (r >> 2) & 3; }

// LDS-DMA through a buffer descriptor (`buffer_load_dwordx4 ... lds`), not `global_load_lds_dwordx4`: the latter is a FLAT
// instruction, and while one is outstanding the compiler's wait-count pass turns every `s_waitcnt lgkmcnt(n)` into
// lgkmcnt(0) (gemm_f32_mfma.hpp, DmaLoader::BUFD) — from the issue of the next stage to the end of the chunk every batch
// of fragment reads was waited for in full before its first MFMA (round 6: the ISA showed read 4 - wait 0 - 8 MFMAs).
// The base is block-uniform, the per-lane part a 32-bit byte offset: the host keeps image and filter bank below 4 GiB.
__device__ __forceinline__ const float* uniform_pointer(const float* p) {
  const unsigned long v = reinterpret_cast<unsigned long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return reinterpret_cast<const float*>(((unsigned long)hi << 32) | lo);
}
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_off, float* lds_dst) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_dst, 16, byte_off, 0, 0, 0);
}
constexpr unsigned OUTSIDE = 0xFFFF0000u;  // byte offset of a padding pixel: past any image the host admits

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

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 31, hi = lane >> 5;
  const __amdgpu_buffer_rsrc_t img_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(uniform_pointer(a.img)), (short)0, a.img_bytes ? (int)a.img_bytes : -1, 0x00020000);
  const __amdgpu_buffer_rsrc_t flt_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(uniform_pointer(a.flt)), (short)0, -1, 0x00020000);

  long long* tr = a.trace ? a.trace + ((long)blockIdx.x * (NT / 64) + wave) * 32 : nullptr;
  int trn = 0;
  auto stamp = [&]() {
    if (tr && lane == 0 && trn < 32) tr[trn] = __builtin_readcyclecounter();
    ++trn;
  };
  stamp();  // 0: start
  const int FH = CFH ? CFH : (int)a.FH, FW = CFW ? CFW : (int)a.FW, taps = FH * FW;
  const int HW = TW + FW - 1, HH = TH + FH - 1;
  const int halo = HH * HW;
  constexpr int HALO_PER_WAVE = HaloLds<TAPS_MAX>::HALO_PER_WAVE;
  constexpr int FLT_PER_WAVE = HaloLds<TAPS_MAX>::FLT_PER_WAVE;

  // Work item -> (image, patch, filter group).  Blocks are persistent: block b runs items b,
  // b + gridDim.x, ... and the LDS double buffer keeps rolling across items, so the first chunk
  // of the next patch is already in flight while the last chunk of this one is multiplied.
  // 32-bit arithmetic throughout (the host admits images and filter banks below 4 GiB and fewer than 2^30 items): the
  // cycle stamps showed 2 850 cycles between a wave's start and its first DMA instruction with 64-bit divisions and
  // products here — 1.2 us of a 40 us launch in front of everything else.
  struct Item {
    int n, y0, x0, f0;
  };
  auto decode = [&](long w64) {
    unsigned w = (unsigned)w64;
    Item it;
    const unsigned tf = (unsigned)a.tiles_f, tx = (unsigned)a.tiles_x, ty = (unsigned)a.tiles_y;
    unsigned q = w / tf;
    it.f0 = (int)(w - q * tf) * FB;
    w = q;
    q = w / tx;
    it.x0 = (int)(w - q * tx) * TW;
    w = q;
    q = w / ty;
    it.y0 = (int)(w - q * ty) * TH;
    it.n = (int)q;
    return it;
  };
  const int iH = (int)a.H, iW = (int)a.W, iC = (int)a.C, iF = (int)a.F, ipy = (int)a.py, ipx = (int)a.px;

  // ---- per-lane DMA sources of the item being loaded (fixed across its channel chunks)
  // halo: instruction t covers halo pixels 16t .. 16t+15, lane l -> pixel 16t + l/4, slot l%4
  // filters: per tap 64 rows x 4 chunks = 4 instructions; instruction u = tap * 4 + v covers rows
  // 16v .. 16v+15
  unsigned halo_src[HALO_PER_WAVE], flt_src[FLT_PER_WAVE];  // byte offsets
  auto sources = [&](const Item& it) {
#pragma unroll
    for (int t = 0; t < HALO_PER_WAVE; ++t) {
      const int instr = wave + t * 8;
      int q = instr * 16 + (lane >> 2);
      if (q >= halo) q = halo - 1;
      const int qy = q / HW, qx = q - qy * HW;
      int yy = it.y0 + qy - ipy, xx = it.x0 + qx - ipx;
      const bool outside = yy < 0 || xx < 0 || yy > iH - 1 || xx > iW - 1;
      if (yy > iH - 1) yy = iH - 1;  // (without padding: the rows / columns of a ragged patch, never stored)
      if (xx > iW - 1) xx = iW - 1;
      if (yy < 0) yy = 0;
      if (xx < 0) xx = 0;
      const int chunk = (lane & 3) ^ swz(instr * 16 + (lane >> 2));
      halo_src[t] = ((((unsigned)(it.n * iH + yy) * (unsigned)iW + (unsigned)xx) * (unsigned)iC + (unsigned)chunk * 4u) * 4u);
      if (outside && a.img_bytes) halo_src[t] = OUTSIDE;
    }
#pragma unroll
    for (int t = 0; t < FLT_PER_WAVE; ++t) {
      const int instr = wave + t * 8;
      const int tap = instr >> 2, r = (instr & 3) * 16 + (lane >> 2);
      int f = it.f0 + r;
      if (f > iF - 1) f = iF - 1;
      const int chunk = (lane & 3) ^ swz(r);
      flt_src[t] = ((unsigned)(f * taps + (tap < taps ? tap : taps - 1)) * (unsigned)iC + (unsigned)chunk * 4u) * 4u;
    }
  };
  // One DMA instruction of this wave's share of a stage; pieces 0 .. PIECES-1.
  constexpr int PIECES = HALO_PER_WAVE + FLT_PER_WAVE;
  auto issue_piece = [&](int piece, int c0, float* stage) {
    if (piece < HALO_PER_WAVE) {
      dma16(img_rsrc, halo_src[piece] + (unsigned)c0 * 4u, stage + (wave + piece * 8) * 256);
    } else {
      const int t = piece - HALO_PER_WAVE;
      dma16(flt_rsrc, flt_src[t] + (unsigned)c0 * 4u, stage + HALO_FLOATS + (wave + t * 8) * 256);
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
  stamp();  // 1: first stage issued
  int stage = 0;
  // every wave waits for ITS OWN LDS-DMA loads before the barrier that publishes them (gemm_f32_mfma.hpp:
  // dma_publish_barrier — the compiler is free to put that wait behind the barrier)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  stamp();  // 2: first stage landed
  auto fragments = [&](const float* Hs, int tp, int pp, f32x4& av, f32x4 (&bv)[2]) {
    const float* Fs = Hs + HALO_FLOATS;
    const int dy = tp / FW, dx = tp - dy * FW;
    const int q = q_base + dy * HW + dx;
    av = *reinterpret_cast<const f32x4*>(Hs + q * CK + (((2 * pp + hi) ^ swz(q)) << 2));
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int f = j * 32 + i;
      bv[j] = *reinterpret_cast<const f32x4*>(Fs + tp * (FB * CK) + f * CK + (((2 * pp + hi) ^ swz(f)) << 2));
    }
  };
  // the two fragment buffers of the compile-time-size pipeline (they live across chunks and items)
  f32x4 fa[2], fb[2][2];
  if constexpr (CFH > 0) fragments(lds, 0, 0, fa[0], fb[0]);
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
      // The loads of the next stage (next chunk, or chunk 0 of the next item; after the last item a
      // harmless reload of this item's chunk 0) are issued behind the first fragment reads, in the
      // shadow of the MFMAs.  No branch inside the chunk.
      const int c_next = ch + 1 < nchunks ? (ch + 1) * CK : 0;
      float* nxt = lds + (stage ^ 1) * STAGE;
      if (ch + 1 == nchunks && w_next < a.items) sources(decode(w_next));
      constexpr int PPS = CK / 8;
      if constexpr (CFH > 0) {
        // Compile-time filter size: an explicit two-deep pipeline over the FH * FW * 2 units (a tap's 8 channels: one A and
        // two B fragments, 8 MFMAs) that runs ACROSS the chunks.  The fragments of unit u + 1 are requested among the MFMAs
        // of unit u and the scheduler is told not to move anything across units (left to itself it sinks every read to just
        // above its first use: read 4 - wait - 8 MFMAs, the LDS latency of every unit exposed to the other wave of the SIMD
        // alone).  The last unit's MFMAs are issued BEHIND the chunk's barrier, after the first reads of the new stage, so
        // the matrix pipe has work while the barrier releases and those reads travel.  Cycle stamps (EG_HALO_TRACE), per
        // chunk of 18 432 MFMA cycles: 19 530 before, 19 440 with the pipeline, 19 230 with the reads spread.  The next
        // stage's DMA instructions ride along, spread over the first units.
        constexpr int UNITS = CFH * CFW * PPS;
        static_assert(UNITS % 2 == 0, "unit 0 lives in buffer 0 and the last unit in buffer 1");
        constexpr int PER_U = (PIECES + UNITS - 1) / UNITS;
        auto multiply = [&](int b) {
#pragma unroll
          for (int k = 0; k < 4; k += 2)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[b][k], fb[b][j][k], acc[j], 0, 0, 0);
              acc_odd[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[b][k + 1], fb[b][j][k + 1], acc_odd[j], 0, 0, 0);
            }
        };
#pragma unroll
        for (int u = 0; u + 1 < UNITS; ++u) {
          fragments(Hs, (u + 1) / PPS, (u + 1) % PPS, fa[(u + 1) & 1], fb[(u + 1) & 1]);
#pragma unroll
          for (int pc = 0; pc < PER_U; ++pc)
            if (u * PER_U + pc < PIECES) issue_piece(u * PER_U + pc, c_next, nxt);
          multiply(u & 1);
          // issue order inside the unit: two MFMAs, one fragment read of unit u + 1, three times, then the last two MFMAs
          // (all three reads in front of the eight MFMAs: 19 450 cycles per chunk; spread like this: 19 230; the MFMAs
          // alone are 18 432.  s_setprio(1) around the MFMAs: 19 490)
#pragma unroll
          for (int g = 0; g < 3; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int piece = (UNITS - 1) * PER_U; piece < PIECES; ++piece) issue_piece(piece, c_next, nxt);  // (1 x 1: the rest)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // the next chunk's loads have landed, this stage's reads too ...
        __syncthreads();                                             // ... for every wave; this stage is free for the one after
        stamp();  // 3 ..: a chunk done
        stage ^= 1;
        fragments(lds + stage * STAGE, 0, 0, fa[0], fb[0]);   // unit 0 of the next chunk (next item; or never used)
        __builtin_amdgcn_sched_barrier(0);
        multiply(1);
        __builtin_amdgcn_sched_barrier(0);
      } else {
        // Run-time filter size: one filter row (FW taps x 16 channels) per batch: all its fragments are requested first
        // (up to 18 ds_read_b128), then its 8 * FW * 2 MFMAs run while the waits retire in order —
        // a wave waits for LDS once per 48 MFMAs instead of once per 8.
        constexpr int ROW_MAX = 3 * PPS;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
          if (dy >= FH) break;
          f32x4 av[ROW_MAX], bv[ROW_MAX][2];
#pragma unroll
          for (int u = 0; u < ROW_MAX; ++u)
            if (u < FW * PPS) fragments(Hs, dy * FW + u / PPS, u % PPS, av[u], bv[u]);
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
        stamp();  // 3 ..: a chunk done
        stage ^= 1;
      }
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
    stamp();  // the item's stores issued
    if (w_next >= a.items) break;
    w = w_next;
    cur_item = decode(w);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  stamp();  // the last stores acknowledged
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
  // 32-bit byte offsets in the buffer loads; the padding's offset (OUTSIDE) lies past the image and C floats behind it do not wrap
  if (N * H * W * C * 4 >= (long)OUTSIDE || F * FH * FW * C * 4 >= (1L << 32) || C * 4 >= 0x10000) return false;
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
  a.img_bytes = (py > 0 || px > 0) ? (unsigned)(N * H * W * C * 4) : 0u;
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
  static const bool trace_on = eg::sw::raw("EG_HALO_TRACE") != nullptr;
  const long nwaves = grid * (NT / 64);
  if (trace_on) {
    EG_HIP_CHECK(hipMalloc((void**)&a.trace, (size_t)nwaves * 32 * sizeof(long long)));
    EG_HIP_CHECK(hipMemsetAsync(a.trace, 0, (size_t)nwaves * 32 * sizeof(long long), ctx->stream));
  }
  void* params[] = {&a};
  EG_HIP_CHECK(hipLaunchKernel(kernel, dim3((unsigned)grid), dim3(NT), params, HaloLds<MAX_TAPS>::BYTES, ctx->stream));
  EG_HIP_CHECK(hipGetLastError());
  *launched = true;
  if (trace_on) {  // debugging aid: per-wave cycle stamps of this launch (mean / max over the waves, cycles since wave start)
    std::vector<long long> h((size_t)nwaves * 32);
    EG_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    EG_HIP_CHECK(hipMemcpy(h.data(), a.trace, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
    EG_HIP_CHECK(hipFree(a.trace));
    // (the cycle counter is not comparable between CUs: every stamp is relative to its own wave's start)
    fprintf(stderr, "[eg] halo trace (cycles since wave start, mean / max over %ld waves):", nwaves);
    for (int k = 1; k < 32; ++k) {
      double sum = 0, mx = 0;
      long cnt = 0;
      for (long w = 0; w < nwaves; ++w) {
        const long long d = h[w * 32 + k] - h[w * 32];
        if (h[w * 32 + k] == 0 || d < 0) continue;
        sum += (double)d;
        mx = d > mx ? (double)d : mx;
        ++cnt;
      }
      if (cnt) fprintf(stderr, " %d:%.0f/%.0f", k, sum / cnt, mx);
    }
    fprintf(stderr, "\n");
  }
  return EG_OK;
}

}  // namespace eg
