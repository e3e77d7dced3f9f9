// Filter gradient of a 3 x 3 convolution with the input halo in LDS (channels and filters multiples of 32).
//
//   gflt[f,dy,dx,c] (+)= sum_{n,y,x} gout[n,y,x,f] * img[n,y+dy,x+dx,c]        derive of dnn.nim:45-49 (passes.nim:519-549)
//
// As ONE contraction over the output pixels (gemm_f32_mfma.hpp, CONV = 2) the operand gathered from the image is staged
// once per filter tap — nine times — by 64 x 64 tiles whose waves hold one accumulator block each (two LDS reads per
// MFMA), and 85 - 113 k-slices of a 147 KB output leave 12 - 17 MB of slabs: 50.7 + 4.9 us at cfg 4 for 30.2 us of matrix
// work.  Here the loop nest is turned around like in conv2_halo.hip:
//   * the output [F, 9, C] is cut into 32 x 32 (filter, channel) quadrants; a block owns ONE quadrant for all nine taps
//     and a range of output pixels; its eight waves split the range, and every wave holds the quadrant's nine 32 x 32
//     accumulator blocks (144 registers): one A fragment (gout) and nine B fragments (the image at the nine tap shifts)
//     feed nine MFMAs — 1.1 LDS reads per MFMA, every image pixel fetched once per row step instead of once per tap;
//   * a wave walks its pixels in segments of 16 consecutive pixels of one output row: per segment the 3 x 18 halo of its
//     32 channels and the 16 x 32 piece of gout go to the wave's OWN stage of LDS by LDS-DMA, double buffered — no
//     block barrier anywhere in the loop (a wave waits for its own loads: s_waitcnt vmcnt(<loads of the next stage>));
//   * the fragment reads of k-step j + 1 are issued in front of the MFMAs of k-step j, the loads of the next segment
//     behind the first MFMAs of the first k-steps.  Two waves per SIMD: with ONE (segments of 32 pixels, 34 KB of LDS per
//     wave) every instruction a wave issues between two MFMAs idles the matrix pipe — cycle stamps gave 78 cycles per
//     MFMA instead of 64, 56 of them per LDS-DMA piece and 7 per LDS read whether or not they stood behind an MFMA;
//   * at the end the eight waves of a block add their accumulators through LDS in a fixed order and the block writes its
//     quadrant of slab `range`; eg::slab_sum folds the ranges (64 slabs at cfg 4: 9.4 MB instead of 12 - 17) in a fixed
//     order: run-to-run identical, no float atomics.
// Pixels past the end of a row segment multiply a zero gradient (their A fragment is masked in registers).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../eg_internal.hpp"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int SEG = 16;                      // output pixels per segment (one k-step = 2 pixels: 8 k-steps)
constexpr int HW = SEG + 2;                  // halo pixels per row
constexpr int QB = 32;                       // quadrant: 32 filters x 32 channels
constexpr int TAPS = 9;
constexpr int HALO_PX = 3 * HW;              // 54 halo pixels
constexpr int HALO_INSTR = (HALO_PX * 8 + 63) / 64;   // 1 KiB DMA instructions per halo (8 lanes of 16 B per pixel): 7
constexpr int GOUT_INSTR = SEG * 8 / 64;              // 2
constexpr int PER_STAGE = HALO_INSTR + GOUT_INSTR;    // 9 loads per stage and wave
constexpr int HALO_FLOATS = HALO_INSTR * 256;         // 1792 (the last instruction's tail is never read)
constexpr int STAGE_FLOATS = HALO_FLOATS + GOUT_INSTR * 256;   // 2304
constexpr int WAVES = 8;                                       // two per SIMD: a wave's LDS reads and loads issue under the other's MFMAs
constexpr int LOOP_FLOATS = WAVES * 2 * STAGE_FLOATS;          // 36 864 floats = 144 KiB
constexpr int FOLD_FLOATS = 4 * TAPS * QB * QB;                // 36 864 floats = 144 KiB (the eight waves meet in two rounds)
constexpr int LDS_BYTES = (LOOP_FLOATS > FOLD_FLOATS ? LOOP_FLOATS : FOLD_FLOATS) * (int)sizeof(float);

struct GradFArgs {
  const float* img;    // [N, H, W, C]
  const float* gout;   // [N, Ho, Wo, F]
  float* slabs;        // [ranges][F * 9 * C]
  long N, H, W, C, F, Ho, Wo;
  int nx;              // segments per output row
  long segs;           // N * nx * Ho, ordered (n, x segment, y) with y fastest: a wave walks down a column of segments
  int qc;              // quadrants along the channels (C / 32); quadrant q = (f block) * qc + (c block)
  int ranges;          // pixel ranges (blocks per quadrant); wave slot = range * 4 + wave
  long long* trace;    // EG_GRADF_TRACE=1 (debugging aid): cycle stamps of every wave, 64 per wave
};

// LDS-DMA through a buffer descriptor (`buffer_load_dwordx4 ... lds`), not `global_load_lds`: a pending FLAT LDS-DMA makes
// the compiler's wait-count pass turn every `s_waitcnt lgkmcnt(n)` into lgkmcnt(0) (gemm_f32_mfma.hpp, DmaLoader::BUFD) —
// the fragment reads issued ahead of the MFMAs would be waited for at once.  The base is block-uniform (scalar registers),
// the per-lane part a 32-bit byte offset: the host keeps both tensors below 4 GiB.
__device__ __forceinline__ const float* uniform_pointer(const float* p) {
  const unsigned long v = reinterpret_cast<unsigned long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return reinterpret_cast<const float*>(((unsigned long)hi << 32) | lo);
}

__global__ __launch_bounds__(WAVES * 64, 1) void conv2_gradf_halo_kernel(GradFArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 31, hi = lane >> 5;
  // Block -> (quadrant, pixel range).  The quadrants of ONE range read the same pixels: they should share an L2.  Block b
  // runs on XCD b % 8, so with a multiple of 8 ranges the blocks of an XCD are numbered range-major among themselves.
  const int nq = (int)((a.F / QB) * a.qc);
  int quadrant, range;
  if (a.ranges % 8 == 0) {
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    quadrant = local % nq;
    range = (local / nq) * 8 + xcd;
  } else {
    quadrant = blockIdx.x % nq;
    range = blockIdx.x / nq;
  }
  const long f0 = (long)(quadrant / a.qc) * QB, c0 = (long)(quadrant % a.qc) * QB;

  // this wave's run of segments: slots share the segments as evenly as possible
  const long slots = (long)a.ranges * WAVES, slot = (long)range * WAVES + wave;
  const long base = a.segs / slots, extra = a.segs % slots;
  const long seg_begin = slot * base + (slot < extra ? slot : extra);
  const int nseg = (int)(base + (slot < extra ? 1 : 0));

  float* stage0 = lds + wave * 2 * STAGE_FLOATS;
  long long* tr = a.trace ? a.trace + ((long)blockIdx.x * WAVES + wave) * 64 : nullptr;
  int trn = 0;
  auto stamp = [&]() {
    if (tr && lane == 0 && trn < 64) tr[trn] = __builtin_readcyclecounter();
    ++trn;
  };
  stamp();   // 0: start

  // One descriptor per SEGMENT and tensor (scalar arithmetic: base + the segment's first byte, num_records = the bytes
  // from there to the tensor's end), so that the per-lane offsets of the 9 loads are constants of the kernel: a ragged
  // last segment reads past its row into the next one (finite values that meet a zeroed gradient row) and, in the last
  // rows of the tensor, past the end — which the range check answers with zeros.  Before round 6 every load clamped its
  // pixel first: a v_min and a 64-bit multiply-add per load and segment in front of the MFMAs.
  const char* const img_u = reinterpret_cast<const char*>(uniform_pointer(a.img));
  const char* const gout_u = reinterpret_cast<const char*>(uniform_pointer(a.gout));
  const unsigned img_bytes = __builtin_amdgcn_readfirstlane((unsigned)(a.N * a.H * a.W * a.C * 4));
  const unsigned gout_bytes = __builtin_amdgcn_readfirstlane((unsigned)(a.N * a.Ho * a.Wo * a.F * 4));
  // Per-lane byte offsets of the 9 loads, fixed for the whole kernel: halo instruction t covers halo pixels 8 t .. 8 t + 7
  // (dy, hx), eight lanes x 16 bytes = the pixel's 32 channels; gout instruction t covers pixels 8 t .. 8 t + 7.  What
  // changes from segment to segment is the descriptor (above).
  const unsigned chunk_bytes = (unsigned)(lane & 7) * 16u;
  const unsigned row_bytes = (unsigned)(a.W * a.C * 4), px_bytes = (unsigned)(a.C * 4), gpx_bytes = (unsigned)(a.F * 4);
  unsigned halo_off[HALO_INSTR];
#pragma unroll
  for (int t = 0; t < HALO_INSTR; ++t) {
    int q = t * 8 + (lane >> 3);
    if (q > HALO_PX - 1) q = HALO_PX - 1;
    const int dy = q / HW, hx = q - dy * HW;
    halo_off[t] = (unsigned)dy * row_bytes + (unsigned)hx * px_bytes + chunk_bytes;
  }
  // the segment the next issue() loads, as (image, segment column, row): decoded once, then stepped (three 64-bit
  // divisions per segment are several hundred scalar instructions in front of the MFMAs of a wave that has its SIMD to itself)
  long next_n, next_col, next_y;
  {
    const long col = seg_begin / a.Ho;
    next_y = seg_begin - col * a.Ho;
    next_n = col / a.nx;
    next_col = col - next_n * a.nx;
  }
  // A segment's block-uniform part of the loads: byte offsets of its first pixel in the image and in gout, the last
  // valid halo column / pixel (for the clamps).  next_segment() describes (next_n, next_col, next_y) and steps on.
  struct Seg {
    __amdgpu_buffer_rsrc_t img, gout;
    int pmax;
  };
  auto next_segment = [&]() {
    const long n = next_n, y = next_y, x0 = next_col * SEG;
    if (++next_y == a.Ho) {
      next_y = 0;
      if (++next_col == a.nx) {
        next_col = 0;
        ++next_n;
      }
    }
    Seg g;
    g.pmax = __builtin_amdgcn_readfirstlane((int)(a.Wo - 1 - x0));
    const unsigned img_base = __builtin_amdgcn_readfirstlane((unsigned)((((n * a.H + y) * a.W + x0) * a.C + c0) * 4));
    const unsigned gout_base = __builtin_amdgcn_readfirstlane((unsigned)((((n * a.Ho + y) * a.Wo + x0) * a.F + f0) * 4));
    g.img = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(img_u + img_base), (short)0, (int)(img_bytes - img_base), 0x00020000);
    g.gout = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(gout_u + gout_base), (short)0, (int)(gout_bytes - gout_base), 0x00020000);
    return g;
  };
  // load piece t (0 .. 16) of segment g into `stage`: 13 halo pieces, then 4 gout pieces
  auto piece = [&](int t, const Seg& g, float* stage) {
    if (t < HALO_INSTR) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(g.img, (__attribute__((address_space(3))) void*)(stage + t * 256), 16, halo_off[t], 0, 0, 0);
    } else {                                    // (pixels past the end of a row: their rows are zeroed in LDS before they are used)
      const int u = t - HALO_INSTR, p0 = u * 8 + (lane >> 3);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(g.gout, (__attribute__((address_space(3))) void*)(stage + HALO_FLOATS + u * 256), 16,
                                               (unsigned)p0 * gpx_bytes + chunk_bytes, 0, 0, 0);
    }
  };
  f32x16 acc[TAPS];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  Seg seg_next = {};
  if (nseg > 0) {
    seg_next = next_segment();
#pragma unroll
    for (int t = 0; t < PER_STAGE; ++t) piece(t, seg_next, stage0);
  }
  stamp();   // 1: first stage issued
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  stamp();   // 2: first stage landed
  for (int s = 0; s < nseg; ++s) {
    float* cur = stage0 + (s & 1) * STAGE_FLOATS;
    float* nxt = stage0 + ((s + 1) & 1) * STAGE_FLOATS;
    const int nvalid = seg_next.pmax + 1 < SEG ? seg_next.pmax + 1 : SEG;
    // the segment after this one goes to the other stage WHILE this one is multiplied (after the last segment: a
    // harmless reload of it — no branch inside the step)
    if (s + 1 < nseg) seg_next = next_segment();
    // k-step j multiplies pixels 2 j (lanes 0 - 31) and 2 j + 1 (lanes 32 - 63): A = gout[pixel][f0 + i],
    // B of tap (dy, dx) = halo[dy][pixel + dx][c0 + i]
    const float* gs = cur + HALO_FLOATS + hi * QB + i;
    const float* hs = cur + hi * QB + i;
    float av[2], bv[2][TAPS];
    auto fragments = [&](int j, float& A, float (&B)[TAPS]) {
      A = gs[2 * j * QB];
#pragma unroll
      for (int t = 0; t < TAPS; ++t) B[t] = hs[((t / 3) * HW + 2 * j + t % 3) * QB];
    };
    // A ragged last segment of a row: the wave zeroes the gradient rows of the pixels past the end in its own stage (they
    // were loaded from the row's last pixel) — once per row instead of a compare and a select in front of the MFMAs of
    // every k-step (round 6: every instruction a wave issues between two MFMAs costs the matrix pipe about four cycles).
    if (nvalid < SEG) {
      float* gz = cur + HALO_FLOATS + i;
      for (int p = nvalid + hi; p < SEG; p += 2) gz[p * QB] = 0.f;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    fragments(0, av[0], bv[0]);
#pragma unroll
    for (int j = 0; j < SEG / 2; ++j) {
      if (j + 1 < SEG / 2) fragments(j + 1, av[(j + 1) & 1], bv[(j + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
      // (one explicit `s_waitcnt lgkmcnt(5)` here instead of the three or four the compiler spreads between the MFMAs:
      // measured slower, 41.3 against 40.7 us — the first MFMAs then wait for the step's last fragment)
      const float A = av[j & 1];
      // The nine loads of the next stage ride behind the first two MFMAs of the first five k-steps (three k-steps of
      // MFMAs, twice that with the SIMD's other wave, separate the last one from the wait at the end of the segment).
#pragma unroll
      for (int t = 0; t < TAPS; ++t) {
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(A, bv[j & 1][t], acc[t], 0, 0, 0);
        if (t < 2 && 2 * j + t < PER_STAGE) piece(2 * j + t, seg_next, nxt);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    stamp();   // 3 + 2 s: multiplied
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the next stage has landed
    stamp();   // 4 + 2 s: next stage landed
  }

  // ---- the eight waves' accumulators meet in LDS in a fixed order, four at a time ([wave & 3][tap][f 32][c 32] fills the
  //      144 KiB): waves 0 - 3 park theirs and every thread sums its elements of the four, ((0 + 1) + (2 + 3)), into
  //      registers; then waves 4 - 7 park theirs, ((4 + 5) + (6 + 7)) is added, and the thread stores its 16-byte groups.
  //      (Waves 4 - 7 adding onto the parked values in place — a read and a write per element — took twice as long.)
  __syncthreads();   // every wave is done with its stages
  stamp();   // after the loop's barrier
  float* fold = lds + (wave & 3) * (TAPS * QB * QB);
  constexpr int GROUPS = (TAPS * QB * QB / 4 + WAVES * 64 - 1) / (WAVES * 64);   // 16-byte groups per thread: 5 (the last one half used)
  f32x4 sum[GROUPS];
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    if ((wave >> 2) == half) {
#pragma unroll
      for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) fold[t * (QB * QB) + ((r & 3) + 8 * (r >> 2) + 4 * hi) * QB + i] = acc[t][r];
    }
    __syncthreads();
#pragma unroll
    for (int g = 0; g < GROUPS; ++g) {
      const int e = (g * WAVES * 64 + tid) * 4;
      if (e >= TAPS * QB * QB) break;
      const f32x4 w0 = *reinterpret_cast<const f32x4*>(lds + e), w1 = *reinterpret_cast<const f32x4*>(lds + TAPS * QB * QB + e);
      const f32x4 w2 = *reinterpret_cast<const f32x4*>(lds + 2 * TAPS * QB * QB + e),
                  w3 = *reinterpret_cast<const f32x4*>(lds + 3 * TAPS * QB * QB + e);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float v = (w0[k] + w1[k]) + (w2[k] + w3[k]);
        sum[g][k] = half == 0 ? v : sum[g][k] + v;
      }
    }
    if (half == 0) __syncthreads();   // the parked values are read: the second half may overwrite them
  }
  float* slab = a.slabs + (long)range * (a.F * TAPS * a.C);
#pragma unroll
  for (int g = 0; g < GROUPS; ++g) {
    const int e = (g * WAVES * 64 + tid) * 4;
    if (e >= TAPS * QB * QB) break;
    const int t = e / (QB * QB), f = (e / QB) % QB, c = e % QB;
    *reinterpret_cast<f32x4*>(slab + ((f0 + f) * TAPS + t) * a.C + c0 + c) = sum[g];
  }
  stamp();   // end
}

}  // namespace

namespace eg {

// Launches the halo form of the filter gradient if the problem suits it; *launched tells the caller whether it did.
int conv2_gradf_halo_try(eg_ctx* ctx, long N, long H, long W, long C, long F, long FH, long FW, const float* img, const float* gout,
                         float* gflt, int accumulate, bool* launched) {
  *launched = false;
  const char* e = eg::sw::raw("EG_CONV_NO_GRADF_HALO");   // (read per call: a test compares the two routes)
  const bool off = e && e[0] && e[0] != '0';
  if (off || FH != 3 || FW != 3 || C % QB != 0 || F % QB != 0 || C < QB || F < QB) return EG_OK;
  const long Ho = H - 2, Wo = W - 2;
  if (Ho <= 0 || Wo <= 0) return EG_OK;
  if ((reinterpret_cast<uintptr_t>(img) & 15) || (reinterpret_cast<uintptr_t>(gout) & 15) || (reinterpret_cast<uintptr_t>(gflt) & 15))
    return EG_OK;
  if (N * H * W * C >= (1L << 30) || N * Ho * Wo * F >= (1L << 30)) return EG_OK;  // 32-bit byte offsets in the loaders
  const long quadrants = (F / QB) * (C / QB);
  if (quadrants > ctx->compute_units) return EG_OK;
  const long nx = (Wo + SEG - 1) / SEG, segs = N * nx * Ho;
  long ranges = ctx->compute_units / quadrants;
  // every wave wants a few segments, or its prologue and the fold at the end outweigh its matrix work
  while (ranges > 1 && segs < ranges * WAVES * 4) ranges >>= 1;
  if (segs < ranges * WAVES * 4 || (double)(Wo) / (double)(nx * SEG) < 0.7) return EG_OK;
  const long total = F * TAPS * C;
  int rc = set_device(ctx);
  if (rc) return rc;
  rc = ensure_workspace(ctx, (size_t)(ranges * total) * sizeof(float));
  if (rc) return rc;
  if (!slab_sum_supported(total, static_cast<const float*>(ctx->workspace), gflt)) return EG_OK;
  static const bool lds_ok = [] {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&conv2_gradf_halo_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               LDS_BYTES) == hipSuccess;
  }();
  if (!lds_ok) {
    (void)hipGetLastError();
    return EG_OK;
  }
  GradFArgs a = {};
  a.img = img;
  a.gout = gout;
  a.slabs = static_cast<float*>(ctx->workspace);
  a.N = N; a.H = H; a.W = W; a.C = C; a.F = F; a.Ho = Ho; a.Wo = Wo;
  a.nx = (int)nx;
  a.segs = segs;
  a.qc = (int)(C / QB);
  a.ranges = (int)ranges;
  static const bool trace_on = eg::sw::raw("EG_GRADF_TRACE") != nullptr;
  const long nwaves = quadrants * ranges * WAVES;
  if (trace_on) EG_HIP_CHECK(hipMalloc((void**)&a.trace, (size_t)nwaves * 64 * sizeof(long long)));
  void* params[] = {&a};
  EG_HIP_CHECK(hipLaunchKernel(reinterpret_cast<const void*>(&conv2_gradf_halo_kernel), dim3((unsigned)(quadrants * ranges)),
                               dim3(WAVES * 64), params, LDS_BYTES, ctx->stream));
  EG_HIP_CHECK(hipGetLastError());
  *launched = true;
  if (trace_on) {  // debugging aid: per-wave cycle stamps (mean over the waves) of this launch
    std::vector<long long> h((size_t)nwaves * 64);
    EG_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    EG_HIP_CHECK(hipMemcpy(h.data(), a.trace, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
    EG_HIP_CHECK(hipFree(a.trace));
    const int stamps = 3 + 2 * (int)((segs + ranges * WAVES - 1) / (ranges * WAVES)) + 2;
    fprintf(stderr, "[eg] gradf trace (cycles since wave start, mean / max over %ld waves):", nwaves);
    for (int k = 1; k < stamps && k < 64; ++k) {
      double sum = 0, mx = 0;
      long cnt = 0;
      for (long w = 0; w < nwaves; ++w) {
        const long long d = h[w * 64 + k] - h[w * 64];
        if (h[w * 64 + k] == 0 || d < 0) continue;
        sum += (double)d;
        mx = d > mx ? (double)d : mx;
        ++cnt;
      }
      fprintf(stderr, " %d:%.0f/%.0f", k, cnt ? sum / cnt : 0.0, mx);
    }
    fprintf(stderr, "\n");
  }
  return slab_sum(ctx, ranges, total, a.slabs, gflt, accumulate);
}

}  // namespace eg
