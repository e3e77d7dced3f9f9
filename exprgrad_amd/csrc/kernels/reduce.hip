// Reductions of the compiled-tensor hot path: column sums (bias gradients), row sums (softmax
// denominators), full sums (scalar losses).
//
// The reference's GPU target has no usable lowering for these: a kernel whose write index
// lacks an iterator keeps that loop serial inside every work-item (`gb[x] += g[y,x]` walks the
// whole batch in one work-item) and a kernel with no independent loop at
// all (`loss[0] += ...`) gets no InstrGpu (passes.nim:2411-2524).  Here every reduction is a
// two-stage tree: wave-level shuffles -> LDS across the 4 waves of a block -> one partial per
// block in the context workspace -> a single-block second pass in fixed order.  No float
// atomics, so results are run-to-run deterministic.
//
// Round 6 (tools/hbm_probe.hip, 134 MB operands in rotation so that the 256 MB Infinity Cache cannot serve them): the
// 4-byte-per-lane forms read at 2.0 (column sum) / 3.4 (full sum) / 5.4 TB/s (row sum).  Aligned operands now move as
// nontemporal 16-byte loads: full sum 6.5 TB/s, row sum 6.5, column sum 6.0 (a thread owns one 16-byte column group
// and walks rows with four loads in flight); the column sum's second pass is the fixed-order slab sum (one launch
// for any partial count) instead of one thread per column adding a thousand partials one after the other.
#include "../eg_internal.hpp"

namespace {

constexpr int NT = 256;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// ---- column sum ---------------------------------------------------------------------------
// in [rows, cols].  A wave covers RPW = 64 / colsP consecutive rows x colsP columns (colsP =
// cols rounded up to a power of two, at most 64), so narrow matrices still issue nearly
// contiguous loads.  blockIdx.y walks column tiles of 64 when cols > 64.
// partial[blockIdx.x][cols] holds one row of sums per row-chunk.
__global__ __launch_bounds__(NT) void colsum_partial_kernel(const float* __restrict__ in,
                                                            float* __restrict__ partial, long rows, long cols,
                                                            int colsP, long rows_per_block) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rpw = 64 / colsP;
  const int r_in_wave = lane / colsP;
  const long c = (long)blockIdx.y * 64 + (lane % colsP);
  const long row_begin = (long)blockIdx.x * rows_per_block;
  const long row_end = min(rows, row_begin + rows_per_block);
  float acc = 0.f;
  if (c < cols) {
    for (long r = row_begin + wave * rpw + r_in_wave; r < row_end; r += 4 * rpw) acc += in[r * cols + c];
  }
  // fold the rpw row phases held in different lane groups
  for (int off = 32; off >= colsP; off >>= 1) acc += __shfl_xor(acc, off, 64);
  red[wave][lane] = acc;
  __syncthreads();
  if (wave == 0 && lane < colsP && c < cols) {
    const float s = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
    partial[(long)blockIdx.x * cols + c] = s;
  }
}

template <bool ACC>
__global__ __launch_bounds__(NT) void colsum_final_kernel(const float* __restrict__ partial, float* __restrict__ out,
                                                          long cols, int nparts) {
  const long c = (long)blockIdx.x * NT + threadIdx.x;
  if (c >= cols) return;
  float s = 0.f;
  for (int p = 0; p < nparts; ++p) s += partial[(long)p * cols + c];
  out[c] = ACC ? out[c] + s : s;
}

// ---- row sum --------------------------------------------------------------------------------
template <bool ACC>
__global__ __launch_bounds__(NT) void rowsum_thread_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                           long rows, long cols) {
  for (long r = (long)blockIdx.x * NT + threadIdx.x; r < rows; r += (long)gridDim.x * NT) {
    float s = 0.f;
    for (long c = 0; c < cols; ++c) s += in[r * cols + c];
    out[r] = ACC ? out[r] + s : s;
  }
}

template <bool ACC>
__global__ __launch_bounds__(NT) void rowsum_wave_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                         long rows, long cols) {
  const int lane = threadIdx.x & 63;
  const long wave = ((long)blockIdx.x * NT + threadIdx.x) >> 6;
  const long nwaves = ((long)gridDim.x * NT) >> 6;
  for (long r = wave; r < rows; r += nwaves) {
    float s = 0.f;
    for (long c = lane; c < cols; c += 64) s += in[r * cols + c];
    s = wave_sum(s);
    if (lane == 0) out[r] = ACC ? out[r] + s : s;
  }
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 ldnt(const f32x4* p) { return __builtin_nontemporal_load(p); }
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// ---- column sum, cols % 4 == 0, 16-byte aligned -------------------------------------------------------------------
// A block covers cg = min(cols / 4, 256) column groups (blockIdx.y walks further ones) x 256 / cg row phases; a thread
// adds rows r0 + ph, + phases, ... of its group with CU_ loads in flight (CU_ accumulators, folded in index order), the
// phases meet in LDS in phase order.  partial[blockIdx.x][cols].
constexpr int CU_ = 4;
__global__ __launch_bounds__(NT) void colsum_vec_kernel(const float* __restrict__ in, float* __restrict__ partial, long rows,
                                                        long cols, long rows_per_block) {
  __shared__ f32x4 red[NT];
  const int cg = (int)(cols / 4 < NT ? cols / 4 : NT);
  const int phases = NT / cg;
  const int gi = threadIdx.x % cg, ph = threadIdx.x / cg;
  const long c4 = (long)blockIdx.y * cg + gi;
  const long r0 = (long)blockIdx.x * rows_per_block;
  const long r1 = rows < r0 + rows_per_block ? rows : r0 + rows_per_block;
  const bool live = ph < phases && c4 * 4 < cols;
  f32x4 acc[CU_];
#pragma unroll
  for (int u = 0; u < CU_; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (live) {
    const f32x4* p = reinterpret_cast<const f32x4*>(in) + c4;
    const long ld4 = cols / 4;
    long r = r0 + ph;
    for (; r + (long)(CU_ - 1) * phases < r1; r += (long)CU_ * phases) {
      f32x4 x[CU_];
#pragma unroll
      for (int u = 0; u < CU_; ++u) x[u] = ldnt(p + (r + (long)u * phases) * ld4);
#pragma unroll
      for (int u = 0; u < CU_; ++u) acc[u] += x[u];
    }
    for (; r < r1; r += phases) acc[0] += ldnt(p + r * ld4);
  }
  red[threadIdx.x] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
  __syncthreads();
  if (ph == 0 && c4 * 4 < cols) {
    f32x4 s = red[gi];
    for (int q = 1; q < phases; ++q) s += red[q * cg + gi];
    reinterpret_cast<f32x4*>(partial + (long)blockIdx.x * cols)[c4] = s;
  }
}

// second pass for narrow matrices the slab sum cannot take (cols % 4 != 0 or an unaligned destination): one block per
// column, the partials over its threads, the tree of sum_final_kernel
template <bool ACC>
__global__ __launch_bounds__(NT) void colsum_final_tree_kernel(const float* __restrict__ partial, float* __restrict__ out,
                                                               long cols, int nparts) {
  __shared__ float red[4];
  const long c = blockIdx.x;
  float acc = 0.f;
  for (int p = threadIdx.x; p < nparts; p += NT) acc += partial[(long)p * cols + c];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float s = ((red[0] + red[1]) + red[2]) + red[3];
    out[c] = ACC ? out[c] + s : s;
  }
}

// ---- row sum, cols % 4 == 0, 16-byte aligned: a wave per row, 16 bytes per lane ---------------------------------------
template <bool ACC>
__global__ __launch_bounds__(NT) void rowsum_vec_kernel(const float* __restrict__ in, float* __restrict__ out, long rows, long cols) {
  const int lane = threadIdx.x & 63;
  const long wave = ((long)blockIdx.x * NT + threadIdx.x) >> 6;
  const long nwaves = ((long)gridDim.x * NT) >> 6;
  const long ld4 = cols / 4;
  for (long r = wave; r < rows; r += nwaves) {
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    const f32x4* p = reinterpret_cast<const f32x4*>(in) + r * ld4;
    for (long c = lane; c < ld4; c += 64) a += ldnt(p + c);
    const float s = wave_sum((a[0] + a[1]) + (a[2] + a[3]));
    if (lane == 0) out[r] = ACC ? out[r] + s : s;
  }
}

// ---- full sum, 16-byte aligned ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void sum_partial_vec_kernel(const float* __restrict__ in, float* __restrict__ partial, long n) {
  __shared__ float red[4];
  const long n4 = n >> 2;
  f32x4 a = {0.f, 0.f, 0.f, 0.f};
  for (long i = (long)blockIdx.x * NT + threadIdx.x; i < n4; i += (long)gridDim.x * NT) a += ldnt(reinterpret_cast<const f32x4*>(in) + i);
  float acc = (a[0] + a[1]) + (a[2] + a[3]);
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) acc += in[(n4 << 2) + threadIdx.x];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
}

// ---- full sum -------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void sum_partial_kernel(const float* __restrict__ in, float* __restrict__ partial,
                                                         long n) {
  __shared__ float red[4];
  float acc = 0.f;
  for (long i = (long)blockIdx.x * NT + threadIdx.x; i < n; i += (long)gridDim.x * NT) acc += in[i];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
}

template <bool ACC>
__global__ __launch_bounds__(NT) void sum_final_kernel(const float* __restrict__ partial, float* __restrict__ out,
                                                       int nparts) {
  __shared__ float red[4];
  float acc = 0.f;
  for (int i = threadIdx.x; i < nparts; i += NT) acc += partial[i];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float s = ((red[0] + red[1]) + red[2]) + red[3];
    out[0] = ACC ? out[0] + s : s;
  }
}

}  // namespace

namespace eg {

static bool colsum_vec(long rows, long cols, const float* in) {
  return cols >= 4 && cols % 4 == 0 && rows >= 64 && (in == nullptr || aligned16(in));
}

static void colsum_geometry(const eg_ctx* ctx, long rows, long cols, long& nparts, long& rows_per_block,
                            long& col_tiles, bool vec) {
  if (vec) {
    // 16-byte column groups, up to 256 per block; ~8 blocks per CU in total; a block's row range a multiple of what its
    // threads take per trip (phases x CU_ rows)
    const long cg = cols / 4 < NT ? cols / 4 : NT;
    const long phases = NT / cg;
    col_tiles = (cols / 4 + cg - 1) / cg;
    nparts = (8L * ctx->compute_units + col_tiles - 1) / col_tiles;
    const long trip = phases * CU_;
    rows_per_block = ((rows + nparts - 1) / nparts + trip - 1) / trip * trip;
    if (rows_per_block < trip) rows_per_block = trip;
    nparts = (rows + rows_per_block - 1) / rows_per_block;
    if (nparts < 1) nparts = 1;
    return;
  }
  col_tiles = (cols + 63) / 64;
  // ~4 blocks per CU in total, at least 64 rows per block.
  nparts = (4L * ctx->compute_units + col_tiles - 1) / col_tiles;
  const long max_parts = (rows + 63) / 64;
  if (nparts > max_parts) nparts = max_parts;
  if (nparts < 1) nparts = 1;
  rows_per_block = (rows + nparts - 1) / nparts;
  nparts = rows_per_block > 0 ? (rows + rows_per_block - 1) / rows_per_block : 1;
  if (nparts < 1) nparts = 1;
}

long colsum_scratch_floats(const eg_ctx* ctx, long rows, long cols) {
  // (the larger of the two geometries: whether the operand is aligned is only known at the call)
  long nparts, rpb, tiles, most = 0;
  for (int v = 0; v < 2; ++v) {
    if (v && !colsum_vec(rows, cols, nullptr)) continue;
    colsum_geometry(ctx, rows, cols, nparts, rpb, tiles, v != 0);
    most = nparts * cols > most ? nparts * cols : most;
  }
  return most + 4;   // (+ 4: the scratch block is handed on 16-byte aligned)
}

bool slab_sum_supported(long total, const float* slab, const float* out);
int slab_sum(eg_ctx* ctx, long slabs, long total, const float* slab, float* out, int accumulate);

int colsum_with_scratch(eg_ctx* ctx, long rows, long cols, const float* in, float* out, int accumulate,
                        float* scratch) {
  if (cols == 0) return EG_OK;
  int rc = set_device(ctx);
  if (rc) return rc;
  long nparts, rows_per_block, col_tiles;
  const bool vec = colsum_vec(rows, cols, in) && aligned16(scratch);
  colsum_geometry(ctx, rows, cols, nparts, rows_per_block, col_tiles, vec);
  if (vec) {
    hipLaunchKernelGGL(colsum_vec_kernel, dim3((unsigned)nparts, (unsigned)col_tiles), dim3(NT), 0, ctx->stream, in, scratch, rows,
                       cols, rows_per_block);
  } else {
    int colsP = 1;
    while (colsP < cols && colsP < 64) colsP <<= 1;
    hipLaunchKernelGGL(colsum_partial_kernel, dim3((unsigned)nparts, (unsigned)col_tiles), dim3(NT), 0, ctx->stream, in,
                       scratch, rows, cols, colsP, rows_per_block);
  }
  EG_HIP_CHECK(hipGetLastError());
  // second pass, fixed order: the slab sum (16-byte groups over lanes, LDS tree) where it applies; narrow matrices one
  // block per column; otherwise one thread per column
  if (nparts > 1 && slab_sum_supported(cols, scratch, out)) return slab_sum(ctx, nparts, cols, scratch, out, accumulate);
  if (cols <= 256 && nparts > 64) {
    if (accumulate)
      hipLaunchKernelGGL((colsum_final_tree_kernel<true>), dim3((unsigned)cols), dim3(NT), 0, ctx->stream, scratch, out, cols, (int)nparts);
    else
      hipLaunchKernelGGL((colsum_final_tree_kernel<false>), dim3((unsigned)cols), dim3(NT), 0, ctx->stream, scratch, out, cols, (int)nparts);
    EG_HIP_CHECK(hipGetLastError());
    return EG_OK;
  }
  const unsigned fgrid = (unsigned)((cols + NT - 1) / NT);
  if (accumulate)
    hipLaunchKernelGGL((colsum_final_kernel<true>), dim3(fgrid), dim3(NT), 0, ctx->stream, scratch, out, cols,
                       (int)nparts);
  else
    hipLaunchKernelGGL((colsum_final_kernel<false>), dim3(fgrid), dim3(NT), 0, ctx->stream, scratch, out, cols,
                       (int)nparts);
  EG_HIP_CHECK(hipGetLastError());
  return EG_OK;
}

// Sum of `slabs` slabs of `total` floats (the k-slices of a contraction with a small output: a classifier's 512 x 10
// weight gradient in 41 slices, a convolution's 36 864-float filter gradient in 113) in ONE launch: the column sum above
// takes two (partials, then their sum: 7 + 5 us for 0.8 MB).  A block owns G 16-byte groups of the output and splits the
// slabs over 256 / G lanes each: lane s of a group adds slabs s, s + L, s + 2 L, ... in that order, a fixed tree in LDS
// folds the L partial sums — run-to-run identical.  G groups x 16 bytes are contiguous in every slab.
template <bool ACC>
__global__ __launch_bounds__(NT) void slab_sum_kernel(const float* __restrict__ slab, float* out, long total, int slabs, int G, int prio) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  __shared__ f4 part[NT];
  // on the side lane, next to a long contraction whose waves are always ready to issue: raise the issue priority like the
  // side lane's contractions do (GemmArgs::prio) — without it this 6 us fold took 36 us there
  if (prio) __builtin_amdgcn_s_setprio(3);
  const int L = NT / G;                       // slab lanes per group
  const int gi = threadIdx.x % G, sl = threadIdx.x / G;
  const long group = (long)blockIdx.x * G + gi;
  const bool live = group * 4 < total;
  f4 s = {0.f, 0.f, 0.f, 0.f};
  if (live)
    for (int z = sl; z < slabs; z += L) {
      const f4 v = *reinterpret_cast<const f4*>(slab + (long)z * total + group * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) s[j] += v[j];
    }
  part[threadIdx.x] = s;
  __syncthreads();
  for (int h = L >> 1; h >= 1; h >>= 1) {
    if (sl < h) {
      const f4 o = part[(sl + h) * G + gi];
      f4 m = part[sl * G + gi];
#pragma unroll
      for (int j = 0; j < 4; ++j) m[j] += o[j];
      part[sl * G + gi] = m;
    }
    __syncthreads();
  }
  if (sl == 0 && live) {
    f4 r = part[gi];
    f4* p = reinterpret_cast<f4*>(out + group * 4);
    if (ACC) {
      const f4 o = *p;
#pragma unroll
      for (int j = 0; j < 4; ++j) r[j] = o[j] + r[j];
    }
    *p = r;
  }
}

bool slab_sum_supported(long total, const float* slab, const float* out) {
  static const bool off = [] {
    const char* e = eg::sw::raw("EG_NO_SLAB_SUM");
    return e && e[0] && e[0] != '0';
  }();
  return !off && total > 0 && total % 4 == 0 && (reinterpret_cast<uintptr_t>(slab) & 15) == 0 &&
         (reinterpret_cast<uintptr_t>(out) & 15) == 0;
}

int slab_sum(eg_ctx* ctx, long slabs, long total, const float* slab, float* out, int accumulate) {
  int rc = set_device(ctx);
  if (rc) return rc;
  const long groups = total / 4;
  // enough blocks to fill the chip, enough slab lanes to keep a lane's chain short
  int G = 16;
  while (G > 1 && (groups / G < 2L * ctx->compute_units || slabs / (NT / G) > 16)) G >>= 1;
  const long blocks = (groups + G - 1) / G;
  const int prio = ctx->on_side_lane ? 1 : 0;
  if (accumulate)
    hipLaunchKernelGGL((slab_sum_kernel<true>), dim3((unsigned)blocks), dim3(NT), 0, ctx->stream, slab, out, total, (int)slabs, G, prio);
  else
    hipLaunchKernelGGL((slab_sum_kernel<false>), dim3((unsigned)blocks), dim3(NT), 0, ctx->stream, slab, out, total, (int)slabs, G, prio);
  EG_HIP_CHECK(hipGetLastError());
  return EG_OK;
}

__global__ __launch_bounds__(NT) void row_finalize_kernel(const float* __restrict__ partial, int nblocks, int E, int stride,
                                                          RowFinalizeArgs a) {
  __shared__ float red[4];
  const int e = blockIdx.x;
  float acc = 0.f;
  for (int b = threadIdx.x; b < nblocks; b += NT) acc += partial[(long)b * stride + e];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float s = ((red[0] + red[1]) + red[2]) + red[3];
    int seg = 0;
    while (seg + 1 < a.nseg && e >= a.offset[seg + 1]) ++seg;
    float* p = a.dst[seg] + (e - a.offset[seg]);
    *p = a.accumulate[seg] ? *p + s : s;
  }
}

int row_finalize(eg_ctx* ctx, const float* partial, int nblocks, int E, int stride, const RowFinalizeArgs& args) {
  if (E <= 0) return EG_OK;
  int rc = set_device(ctx);
  if (rc) return rc;
  hipLaunchKernelGGL(row_finalize_kernel, dim3((unsigned)E), dim3(NT), 0, ctx->stream, partial, nblocks, E, stride, args);
  EG_HIP_CHECK(hipGetLastError());
  return EG_OK;
}

}  // namespace eg

extern "C" {

int eg_colsum(eg_ctx* ctx, int64_t rows, int64_t cols, const float* in, float* out, int accumulate) {
  EG_REQUIRE(ctx, EG_ERR_INVALID, "eg_colsum: ctx is NULL");
  EG_REQUIRE(rows >= 0 && cols >= 0, EG_ERR_INVALID, "eg_colsum: negative extent");
  if (cols == 0) return EG_OK;
  EG_REQUIRE(out && (rows == 0 || in), EG_ERR_INVALID, "eg_colsum: NULL tensor");
  int rc = eg::ensure_workspace(ctx, (size_t)eg::colsum_scratch_floats(ctx, rows, cols) * sizeof(float));
  if (rc) return rc;
  return eg::colsum_with_scratch(ctx, rows, cols, in, out, accumulate, static_cast<float*>(ctx->workspace));
}

int eg_rowsum(eg_ctx* ctx, int64_t rows, int64_t cols, const float* in, float* out, int accumulate) {
  EG_REQUIRE(ctx, EG_ERR_INVALID, "eg_rowsum: ctx is NULL");
  EG_REQUIRE(rows >= 0 && cols >= 0, EG_ERR_INVALID, "eg_rowsum: negative extent");
  if (rows == 0) return EG_OK;
  EG_REQUIRE(out && (cols == 0 || in), EG_ERR_INVALID, "eg_rowsum: NULL tensor");
  int rc = eg::set_device(ctx);
  if (rc) return rc;
  const long cap = 8L * ctx->compute_units;
  if (cols <= 32) {
    long blocks = (rows + NT - 1) / NT;
    if (blocks > cap) blocks = cap;
    if (accumulate)
      hipLaunchKernelGGL((rowsum_thread_kernel<true>), dim3((unsigned)blocks), dim3(NT), 0, ctx->stream, in, out,
                         (long)rows, (long)cols);
    else
      hipLaunchKernelGGL((rowsum_thread_kernel<false>), dim3((unsigned)blocks), dim3(NT), 0, ctx->stream, in, out,
                         (long)rows, (long)cols);
  } else if (cols % 4 == 0 && aligned16(in)) {
    long blocks = (rows + 3) / 4;
    if (blocks > cap) blocks = cap;
    if (accumulate)
      hipLaunchKernelGGL((rowsum_vec_kernel<true>), dim3((unsigned)blocks), dim3(NT), 0, ctx->stream, in, out,
                         (long)rows, (long)cols);
    else
      hipLaunchKernelGGL((rowsum_vec_kernel<false>), dim3((unsigned)blocks), dim3(NT), 0, ctx->stream, in, out,
                         (long)rows, (long)cols);
  } else {
    long blocks = (rows + 3) / 4;
    if (blocks > cap) blocks = cap;
    if (accumulate)
      hipLaunchKernelGGL((rowsum_wave_kernel<true>), dim3((unsigned)blocks), dim3(NT), 0, ctx->stream, in, out,
                         (long)rows, (long)cols);
    else
      hipLaunchKernelGGL((rowsum_wave_kernel<false>), dim3((unsigned)blocks), dim3(NT), 0, ctx->stream, in, out,
                         (long)rows, (long)cols);
  }
  EG_HIP_CHECK(hipGetLastError());
  return EG_OK;
}

int eg_sum(eg_ctx* ctx, int64_t n, const float* in, float* out, int accumulate) {
  EG_REQUIRE(ctx, EG_ERR_INVALID, "eg_sum: ctx is NULL");
  EG_REQUIRE(n >= 0, EG_ERR_INVALID, "eg_sum: negative length");
  EG_REQUIRE(out && (n == 0 || in), EG_ERR_INVALID, "eg_sum: NULL tensor");
  int rc = eg::set_device(ctx);
  if (rc) return rc;
  const bool vec = n >= 4 && aligned16(in);
  long blocks = (n + NT * 4 - 1) / (NT * 4);
  const long cap = (vec ? 8L : 4L) * ctx->compute_units;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  rc = eg::ensure_workspace(ctx, (size_t)blocks * sizeof(float));
  if (rc) return rc;
  float* partial = static_cast<float*>(ctx->workspace);
  if (vec)
    hipLaunchKernelGGL(sum_partial_vec_kernel, dim3((unsigned)blocks), dim3(NT), 0, ctx->stream, in, partial, (long)n);
  else
    hipLaunchKernelGGL(sum_partial_kernel, dim3((unsigned)blocks), dim3(NT), 0, ctx->stream, in, partial, (long)n);
  if (accumulate)
    hipLaunchKernelGGL((sum_final_kernel<true>), dim3(1), dim3(NT), 0, ctx->stream, partial, out, (int)blocks);
  else
    hipLaunchKernelGGL((sum_final_kernel<false>), dim3(1), dim3(NT), 0, ctx->stream, partial, out, (int)blocks);
  EG_HIP_CHECK(hipGetLastError());
  return EG_OK;
}

}  // extern "C"
