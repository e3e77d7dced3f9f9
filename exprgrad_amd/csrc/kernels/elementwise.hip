// HBM-bound kernels of the compiled-tensor hot path: raw-indexed maps and their derived
// gradients, broadcast bias add, axpy (gradientDescent), fill.
//
// Reference lowering being replaced: tests/cache/relu_basic.ir — one work-item per element in
// work-groups of 16 (ir.nim:283), i.e. a quarter of a wavefront per group.  Here: 256-thread
// blocks, 16-byte accesses per lane (1 KiB per wave instruction).  Round 6 (tools/hbm_probe.hip, operands past the
// 256 MB Infinity Cache): a block owns ONE CONTIGUOUS chunk and a thread has U = 4 sixteen-byte loads per stream in
// flight before the first dependent store, every access nontemporal (nothing is read twice), at most 32 blocks per
// CU — a two-stream map moves 268 MB at 6.27 TB/s and a three-stream gradient 403 MB at 6.29 (the guide's float4-copy
// ceiling is 6.29); the grid-stride form with one load in flight it replaces: 5.40 / 5.46 TB/s.
//
// Arithmetic follows llvmgen.nim:212-301 operation by operation (fadd/fsub/fmul/fdiv, ordered
// compares, select, libm-class exp/sin/cos); this file is built with -ffp-contract=off so no
// multiply-add is fused that the reference's no-fast-math JIT (llvm.nim:486-491) keeps apart.
#include "../eg_internal.hpp"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int NT = 256;
constexpr int UV = 4;   // 16-byte pieces per thread and stream in flight (vector path)

// vector path: blocks of a chunked launch over n4 sixteen-byte groups, and the chunk a block owns (a multiple of NT * UV)
inline unsigned chunk_grid(const eg_ctx* ctx, long n4) {
  long blocks = (n4 + (long)NT * UV - 1) / ((long)NT * UV);
  const long cap = 32L * ctx->compute_units;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  return (unsigned)blocks;
}
__device__ __forceinline__ void chunk_range(long n4, long& lo, long& hi) {
  const long per = ((n4 + gridDim.x - 1) / gridDim.x + (long)NT * UV - 1) / ((long)NT * UV) * ((long)NT * UV);
  lo = (long)blockIdx.x * per;
  hi = lo + per < n4 ? lo + per : n4;
}
__device__ __forceinline__ f32x4 ldnt(const f32x4* p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ void stnt(f32x4 v, f32x4* p) { __builtin_nontemporal_store(v, p); }

inline unsigned grid_for(const eg_ctx* ctx, long work_items) {
  long blocks = (work_items + NT - 1) / NT;
  long cap = 8L * ctx->compute_units;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  return (unsigned)blocks;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// ---- forward maps -----------------------------------------------------------------------
template <int OP>
__device__ __forceinline__ float map_fwd(float x, float p) {
  switch (OP) {
    case EG_MAP_IDENTITY: return x;
    case EG_MAP_RELU: return x >= 0.0f ? x : 0.0f;                 // dnn.nim:26-27 (0 <= x)
    case EG_MAP_LEAKY_RELU: return (x >= 0.0f ? 1.0f : p) * x;     // dnn.nim:29-30
    case EG_MAP_SIGMOID: return 1.0f / (1.0f + expf(-x));          // dnn.nim:32-33
    case EG_MAP_TANH: {                                            // dnn.nim:35-40 (naive form)
      const float a = expf(x), b = expf(-x);
      return (a - b) / (a + b);
    }
    case EG_MAP_SCALE: return x * p;                               // base.nim:24
    case EG_MAP_SIN: return sinf(x);                               // dnn.nim:42-43
    case EG_MAP_XOR_LEAKY: return x <= 0.0f ? p * x : x;           // xor_from_scratch.nim:22
    case EG_MAP_EXP: return expf(x);
  }
  return x;
}

// ---- derived gradients: exactly the instruction sequences passes.nim:392-505 emits ---------
template <int OP>
__device__ __forceinline__ float map_bwd(float x, float g, float p) {
  switch (OP) {
    case EG_MAP_IDENTITY: return g;
    case EG_MAP_RELU: return x >= 0.0f ? g : 0.0f;  // select rule 471-476: select(c, g, 0) (+ nothing: 0.0 literal)
    case EG_MAP_LEAKY_RELU: {
      // out = s * x with s = select(c, 1, p): mul rule 399-403 -> grad_x = g * s (s has no tensor input).
      const float s = x >= 0.0f ? 1.0f : p;
      return g * s;
    }
    case EG_MAP_SIGMOID: {
      // out = 1 / s, s = 1 + e, e = exp(n), n = -x.
      // div rule 404-415: grad_s = (-1) * (g / (s*s)); add: grad_e = grad_s;
      // exp rule 456-459: grad_n = grad_e * e; neg rule 416-419: grad_x = -grad_n.
      const float e = expf(-x);
      const float s = 1.0f + e;
      const float gs = (-1.0f) * (g / (s * s));
      return -(gs * e);
    }
    case EG_MAP_TANH: {
      // out = d / t, d = a - b, t = a + b, a = exp(x), b = exp(n), n = -x.
      const float a = expf(x), b = expf(-x);
      const float d = a - b, t = a + b;
      const float gd = g / t;                    // div: grad of numerator
      const float gt = (-d) * (g / (t * t));     // div: grad of denominator
      // visit order is reverse program order: t = a + b first, then d = a - b.
      float ga = gt, gb = gt;                    // add rule
      ga = ga + gd;                              // sub rule: (g, -g), accumulated with + (510-517)
      gb = gb + (-gd);
      const float gn = gb * b;                   // exp(n)
      const float gx_from_n = -gn;               // neg
      const float gx_from_a = ga * a;            // exp(x)
      // exp(-x) is visited before exp(x) (later instruction first), so -gn is the first term.
      return gx_from_n + gx_from_a;
    }
    case EG_MAP_SCALE: return g * p;
    case EG_MAP_SIN: return cosf(x) * g;  // 460-464
    case EG_MAP_XOR_LEAKY: {
      // out = select(c, p*x, x): select rule gives select(c,g,0) for arg1 and select(c,0,g) for arg2;
      // mul rule on p*x gives g1 * p.  Reverse visit: select first, then the mul.  x's gradient
      // accumulates arg2's term first (it reaches x directly), then the mul's term.
      const bool c = x <= 0.0f;
      const float g1 = c ? g : 0.0f, g2 = c ? 0.0f : g;
      return g2 + g1 * p;
    }
    case EG_MAP_EXP: return g * expf(x);
  }
  return g;
}

template <int OP, bool ACC>
__global__ __launch_bounds__(NT) void map_kernel(const float* __restrict__ in, float* __restrict__ out, long n,
                                                 float p, int vec) {
  if (vec) {
    const long n4 = n >> 2;
    long lo, hi;
    chunk_range(n4, lo, hi);
    const f32x4* in4 = reinterpret_cast<const f32x4*>(in);
    f32x4* out4 = reinterpret_cast<f32x4*>(out);
    for (long i = lo + threadIdx.x; i < hi; i += NT * UV) {
      f32x4 x[UV], o[UV];
#pragma unroll
      for (int u = 0; u < UV; ++u)
        if (i + u * NT < hi) {
          x[u] = ldnt(in4 + i + u * NT);
          if (ACC) o[u] = ldnt(out4 + i + u * NT);
        }
#pragma unroll
      for (int u = 0; u < UV; ++u)
        if (i + u * NT < hi) {
          f32x4 y;
#pragma unroll
          for (int j = 0; j < 4; ++j) y[j] = ACC ? o[u][j] + map_fwd<OP>(x[u][j], p) : map_fwd<OP>(x[u][j], p);
          stnt(y, out4 + i + u * NT);
        }
    }
    if (blockIdx.x == 0)
      for (long i = (n4 << 2) + threadIdx.x; i < n; i += NT) {
        float y = map_fwd<OP>(in[i], p);
        out[i] = ACC ? out[i] + y : y;
      }
  } else {
    const long tid = (long)blockIdx.x * NT + threadIdx.x;
    const long nthreads = (long)gridDim.x * NT;
    for (long i = tid; i < n; i += nthreads) {
      float y = map_fwd<OP>(in[i], p);
      out[i] = ACC ? out[i] + y : y;
    }
  }
}

template <int OP, bool ACC>
__global__ __launch_bounds__(NT) void map_grad_kernel(const float* __restrict__ in, const float* __restrict__ gout,
                                                      float* __restrict__ gin, long n, float p, int vec) {
  if (vec) {
    const long n4 = n >> 2;
    long lo, hi;
    chunk_range(n4, lo, hi);
    const f32x4* in4 = reinterpret_cast<const f32x4*>(in);
    const f32x4* g4 = reinterpret_cast<const f32x4*>(gout);
    f32x4* out4 = reinterpret_cast<f32x4*>(gin);
    for (long i = lo + threadIdx.x; i < hi; i += NT * UV) {
      f32x4 x[UV], g[UV], o[UV];
#pragma unroll
      for (int u = 0; u < UV; ++u)
        if (i + u * NT < hi) {
          x[u] = ldnt(in4 + i + u * NT);
          g[u] = ldnt(g4 + i + u * NT);
          if (ACC) o[u] = ldnt(out4 + i + u * NT);
        }
#pragma unroll
      for (int u = 0; u < UV; ++u)
        if (i + u * NT < hi) {
          f32x4 y;
#pragma unroll
          for (int j = 0; j < 4; ++j) y[j] = ACC ? o[u][j] + map_bwd<OP>(x[u][j], g[u][j], p) : map_bwd<OP>(x[u][j], g[u][j], p);
          stnt(y, out4 + i + u * NT);
        }
    }
    if (blockIdx.x == 0)
      for (long i = (n4 << 2) + threadIdx.x; i < n; i += NT) {
        float y = map_bwd<OP>(in[i], gout[i], p);
        gin[i] = ACC ? gin[i] + y : y;
      }
  } else {
    const long tid = (long)blockIdx.x * NT + threadIdx.x;
    const long nthreads = (long)gridDim.x * NT;
    for (long i = tid; i < n; i += nthreads) {
      float y = map_bwd<OP>(in[i], gout[i], p);
      gin[i] = ACC ? gin[i] + y : y;
    }
  }
}

// out[y,x] (+)= bias[x]
template <bool ACC>
__global__ __launch_bounds__(NT) void bias_add_kernel(const float* __restrict__ bias, float* __restrict__ out,
                                                      long rows, long cols, int vec) {
  const long n = rows * cols;
  if (vec) {  // cols % 4 == 0: a 16-byte chunk never straddles a row
    const long n4 = n >> 2;
    const long c4 = cols >> 2;
    long lo, hi;
    chunk_range(n4, lo, hi);
    f32x4* out4 = reinterpret_cast<f32x4*>(out);
    for (long i = lo + threadIdx.x; i < hi; i += NT * UV) {
      f32x4 b[UV], o[UV];
#pragma unroll
      for (int u = 0; u < UV; ++u)
        if (i + u * NT < hi) {
          b[u] = *reinterpret_cast<const f32x4*>(bias + (((i + u * NT) % c4) << 2));   // (the bias row is re-read: cached)
          if (ACC) o[u] = ldnt(out4 + i + u * NT);
        }
#pragma unroll
      for (int u = 0; u < UV; ++u)
        if (i + u * NT < hi) {
          f32x4 y = b[u];
          if (ACC) {
#pragma unroll
            for (int j = 0; j < 4; ++j) y[j] = o[u][j] + b[u][j];
          }
          stnt(y, out4 + i + u * NT);
        }
    }
  } else {
    const long tid = (long)blockIdx.x * NT + threadIdx.x;
    const long nthreads = (long)gridDim.x * NT;
    for (long i = tid; i < n; i += nthreads) {
      const float b = bias[i % cols];
      out[i] = ACC ? out[i] + b : b;
    }
  }
}

__global__ __launch_bounds__(NT) void axpy_kernel(float alpha, const float* __restrict__ x, float* __restrict__ y,
                                                  long n, int vec) {
  if (vec) {
    const long n4 = n >> 2;
    long lo, hi;
    chunk_range(n4, lo, hi);
    const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
    f32x4* y4 = reinterpret_cast<f32x4*>(y);
    for (long i = lo + threadIdx.x; i < hi; i += NT * UV) {
      f32x4 xv[UV], yv[UV];
#pragma unroll
      for (int u = 0; u < UV; ++u)
        if (i + u * NT < hi) {
          xv[u] = ldnt(x4 + i + u * NT);
          yv[u] = ldnt(y4 + i + u * NT);
        }
#pragma unroll
      for (int u = 0; u < UV; ++u)
        if (i + u * NT < hi) {
#pragma unroll
          for (int j = 0; j < 4; ++j) yv[u][j] = yv[u][j] + alpha * xv[u][j];
          stnt(yv[u], y4 + i + u * NT);
        }
    }
    if (blockIdx.x == 0)
      for (long i = (n4 << 2) + threadIdx.x; i < n; i += NT) y[i] = y[i] + alpha * x[i];
  } else {
    const long tid = (long)blockIdx.x * NT + threadIdx.x;
    const long nthreads = (long)gridDim.x * NT;
    for (long i = tid; i < n; i += nthreads) y[i] = y[i] + alpha * x[i];
  }
}

__global__ __launch_bounds__(NT) void fill_kernel(float value, float* __restrict__ out, long n, int vec) {
  if (vec) {
    const long n4 = n >> 2;
    long lo, hi;
    chunk_range(n4, lo, hi);
    const f32x4 v = {value, value, value, value};
    f32x4* out4 = reinterpret_cast<f32x4*>(out);
    for (long i = lo + threadIdx.x; i < hi; i += NT) stnt(v, out4 + i);
    if (blockIdx.x == 0)
      for (long i = (n4 << 2) + threadIdx.x; i < n; i += NT) out[i] = value;
  } else {
    const long tid = (long)blockIdx.x * NT + threadIdx.x;
    const long nthreads = (long)gridDim.x * NT;
    for (long i = tid; i < n; i += nthreads) out[i] = value;
  }
}

#define EG_MAP_CASES(LAUNCH)                     \
  switch (op) {                                  \
    case EG_MAP_IDENTITY: LAUNCH(EG_MAP_IDENTITY); break;     \
    case EG_MAP_RELU: LAUNCH(EG_MAP_RELU); break;             \
    case EG_MAP_LEAKY_RELU: LAUNCH(EG_MAP_LEAKY_RELU); break; \
    case EG_MAP_SIGMOID: LAUNCH(EG_MAP_SIGMOID); break;       \
    case EG_MAP_TANH: LAUNCH(EG_MAP_TANH); break;             \
    case EG_MAP_SCALE: LAUNCH(EG_MAP_SCALE); break;           \
    case EG_MAP_SIN: LAUNCH(EG_MAP_SIN); break;               \
    case EG_MAP_XOR_LEAKY: LAUNCH(EG_MAP_XOR_LEAKY); break;   \
    case EG_MAP_EXP: LAUNCH(EG_MAP_EXP); break;               \
    default:                                                  \
      eg::set_error("unknown map op %d", op);                 \
      return EG_ERR_INVALID;                                  \
  }

}  // namespace

// splitmix64 finaliser: a bijective 64-bit mix with full avalanche
__device__ __forceinline__ unsigned long long mix64(unsigned long long z) {
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
  return z ^ (z >> 31);
}

// state[0] = seed, state[1] = number of fills drawn so far; `stream` separates the random tensors
// filled from the same state in one run
__global__ __launch_bounds__(NT) void fill_uniform_kernel(float lo, float hi, const uint64_t* __restrict__ state,
                                                          unsigned long long stream, float* __restrict__ out, long n) {
  const unsigned long long key = mix64(state[0] + 0x9e3779b97f4a7c15ULL * (state[1] + 1)) ^ mix64(stream + 0x632be59bd9b4e019ULL);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const unsigned long long bits = mix64(key + 0x9e3779b97f4a7c15ULL * (unsigned long long)(i + 1));
    const float u = (float)(bits >> 40) * 0x1.0p-24f;  // 24 random bits: [0, 1), exactly representable
    out[i] = lo + (hi - lo) * u;
  }
}

__global__ void rng_advance_kernel(uint64_t* state) { state[1] += 1; }

// Copies up to 8 contiguous float ranges in one launch: Model.fit's mini-batch views of the
// device-resident data set -> the inputs' fixed staging buffers (eg_model_fit, host/model.cpp).
__global__ __launch_bounds__(NT) void copy_segments_kernel(eg::CopySegments seg) {
  const int s = blockIdx.y;
  const float* __restrict__ src = seg.src[s];
  float* __restrict__ dst = seg.dst[s];
  const long n = seg.count[s];
  const long stride = (long)gridDim.x * NT;
  long i = (long)blockIdx.x * NT + threadIdx.x;
  if ((((unsigned long)src | (unsigned long)dst) & 15) == 0) {
    const long n4 = n >> 2;
    for (long j = i; j < n4; j += stride) reinterpret_cast<float4*>(dst)[j] = reinterpret_cast<const float4*>(src)[j];
    for (long j = (n4 << 2) + i; j < n; j += stride) dst[j] = src[j];
  } else {
    for (; i < n; i += stride) dst[i] = src[i];
  }
}

namespace eg {
// The launch copy_segments would make, as graph kernel-node parameters (host/fit.cpp re-points the copy nodes of a captured
// group of batches): `arg` must stay alive until the parameters have been handed over.
bool copy_segments_node_params(eg_ctx* ctx, const CopySegments& seg, hipKernelNodeParams* out, void** arg) {
  long most = 0;
  for (int i = 0; i < seg.n; ++i) most = seg.count[i] > most ? seg.count[i] : most;
  if (seg.n <= 0 || most == 0) return false;
  *out = hipKernelNodeParams{};
  out->func = reinterpret_cast<void*>(copy_segments_kernel);
  out->gridDim = dim3(grid_for(ctx, (most + 3) / 4), seg.n);
  out->blockDim = dim3(NT);
  out->sharedMemBytes = 0;
  arg[0] = const_cast<CopySegments*>(&seg);
  out->kernelParams = arg;
  out->extra = nullptr;
  return true;
}
const void* copy_segments_function() { return reinterpret_cast<const void*>(copy_segments_kernel); }

int copy_segments(eg_ctx* ctx, const CopySegments& seg) {
  if (seg.n <= 0) return EG_OK;
  EG_REQUIRE(seg.n <= 8, EG_ERR_INVALID, "copy_segments: more than 8 ranges");
  long most = 0;
  for (int i = 0; i < seg.n; ++i) most = seg.count[i] > most ? seg.count[i] : most;
  if (most == 0) return EG_OK;
  hipLaunchKernelGGL(copy_segments_kernel, dim3(grid_for(ctx, (most + 3) / 4), seg.n), dim3(NT), 0, ctx->stream, seg);
  EG_HIP_CHECK(hipGetLastError());
  return EG_OK;
}
}  // namespace eg

// Results start from zero on every call (allocShapes / zeroResultTensor, model.nim:318, 383): the arena's zero prefix and
// the accumulated gradients of the bucket, all in ONE launch.  (hipMemsetAsync becomes a runtime kernel with a 10 - 17 us
// wait in front of it, captured or not: the XOR step is 25 us long.)
namespace {
__global__ __launch_bounds__(256) void zero_ranges_kernel(eg::ZeroRanges r) {
  int which = 0;
  long block = blockIdx.x;
  while (which < r.count - 1 && block >= r.blocks[which]) block -= r.blocks[which++];
  float* p = r.ptr[which];
  const long n = r.floats[which];
  // whole 16-byte groups of an aligned range as one store, the rest one by one
  const long head = ((reinterpret_cast<uintptr_t>(p) & 15) == 0) ? n / 4 : 0;
  const long stride = (long)r.blocks[which] * 256;
  for (long i = block * 256 + threadIdx.x; i < head; i += stride) reinterpret_cast<float4*>(p)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (long i = head * 4 + block * 256 + threadIdx.x; i < n; i += stride) p[i] = 0.f;
}
}  // namespace

int eg::zero_ranges(eg_ctx* ctx, const std::vector<std::pair<float*, long>>& ranges) {
  int rc = eg::set_device(ctx);
  if (rc) return rc;
  size_t at = 0;
  while (at < ranges.size()) {
    eg::ZeroRanges r = {};
    long total = 0;
    for (; at < ranges.size() && r.count < eg::ZeroRanges::MAX; ++at) {
      if (ranges[at].second <= 0) continue;
      const long groups = (ranges[at].second + 3) / 4;
      long blocks = (groups + 255) / 256;
      const long cap = 8L * ctx->compute_units;
      if (blocks > cap) blocks = cap;
      r.ptr[r.count] = ranges[at].first;
      r.floats[r.count] = ranges[at].second;
      r.blocks[r.count] = (int)blocks;
      total += blocks;
      ++r.count;
    }
    if (r.count == 0) break;
    hipLaunchKernelGGL(zero_ranges_kernel, dim3((unsigned)total), dim3(256), 0, ctx->stream, r);
    EG_HIP_CHECK(hipGetLastError());
  }
  return EG_OK;
}

extern "C" {

int eg_map(eg_ctx* ctx, int op, int64_t n, const float* in, float* out, float param, int accumulate) {
  EG_REQUIRE(ctx, EG_ERR_INVALID, "eg_map: ctx is NULL");
  EG_REQUIRE(n >= 0, EG_ERR_INVALID, "eg_map: negative length");
  if (n == 0) return EG_OK;
  EG_REQUIRE(in && out, EG_ERR_INVALID, "eg_map: NULL tensor");
  int rc = eg::set_device(ctx);
  if (rc) return rc;
  const int vec = aligned16(in) && aligned16(out) && n >= 4;
  const unsigned grid = vec ? chunk_grid(ctx, n >> 2) : grid_for(ctx, n);
#define LAUNCH(OP)                                                                                          \
  do {                                                                                                      \
    if (accumulate)                                                                                         \
      hipLaunchKernelGGL((map_kernel<OP, true>), dim3(grid), dim3(NT), 0, ctx->stream, in, out, (long)n, param, vec); \
    else                                                                                                    \
      hipLaunchKernelGGL((map_kernel<OP, false>), dim3(grid), dim3(NT), 0, ctx->stream, in, out, (long)n, param, vec); \
  } while (0)
  EG_MAP_CASES(LAUNCH)
#undef LAUNCH
  EG_HIP_CHECK(hipGetLastError());
  return EG_OK;
}

int eg_map_grad(eg_ctx* ctx, int op, int64_t n, const float* in, const float* gout, float* gin, float param,
                int accumulate) {
  EG_REQUIRE(ctx, EG_ERR_INVALID, "eg_map_grad: ctx is NULL");
  EG_REQUIRE(n >= 0, EG_ERR_INVALID, "eg_map_grad: negative length");
  if (n == 0) return EG_OK;
  EG_REQUIRE(in && gout && gin, EG_ERR_INVALID, "eg_map_grad: NULL tensor");
  int rc = eg::set_device(ctx);
  if (rc) return rc;
  const int vec = aligned16(in) && aligned16(gout) && aligned16(gin) && n >= 4;
  const unsigned grid = vec ? chunk_grid(ctx, n >> 2) : grid_for(ctx, n);
#define LAUNCH(OP)                                                                                           \
  do {                                                                                                       \
    if (accumulate)                                                                                          \
      hipLaunchKernelGGL((map_grad_kernel<OP, true>), dim3(grid), dim3(NT), 0, ctx->stream, in, gout, gin,   \
                         (long)n, param, vec);                                                               \
    else                                                                                                     \
      hipLaunchKernelGGL((map_grad_kernel<OP, false>), dim3(grid), dim3(NT), 0, ctx->stream, in, gout, gin,  \
                         (long)n, param, vec);                                                               \
  } while (0)
  EG_MAP_CASES(LAUNCH)
#undef LAUNCH
  EG_HIP_CHECK(hipGetLastError());
  return EG_OK;
}

int eg_bias_add(eg_ctx* ctx, int64_t rows, int64_t cols, const float* bias, float* out, int accumulate) {
  EG_REQUIRE(ctx, EG_ERR_INVALID, "eg_bias_add: ctx is NULL");
  EG_REQUIRE(rows >= 0 && cols >= 0, EG_ERR_INVALID, "eg_bias_add: negative extent");
  if (rows == 0 || cols == 0) return EG_OK;
  EG_REQUIRE(bias && out, EG_ERR_INVALID, "eg_bias_add: NULL tensor");
  int rc = eg::set_device(ctx);
  if (rc) return rc;
  const int vec = (cols % 4 == 0) && aligned16(bias) && aligned16(out);
  const long n = rows * cols;
  const unsigned grid = vec ? chunk_grid(ctx, n / 4) : grid_for(ctx, n);
  if (accumulate)
    hipLaunchKernelGGL((bias_add_kernel<true>), dim3(grid), dim3(NT), 0, ctx->stream, bias, out, (long)rows,
                       (long)cols, vec);
  else
    hipLaunchKernelGGL((bias_add_kernel<false>), dim3(grid), dim3(NT), 0, ctx->stream, bias, out, (long)rows,
                       (long)cols, vec);
  EG_HIP_CHECK(hipGetLastError());
  return EG_OK;
}

int eg_axpy(eg_ctx* ctx, int64_t n, float alpha, const float* x, float* y) {
  EG_REQUIRE(ctx, EG_ERR_INVALID, "eg_axpy: ctx is NULL");
  EG_REQUIRE(n >= 0, EG_ERR_INVALID, "eg_axpy: negative length");
  if (n == 0) return EG_OK;
  EG_REQUIRE(x && y, EG_ERR_INVALID, "eg_axpy: NULL tensor");
  int rc = eg::set_device(ctx);
  if (rc) return rc;
  const int vec = aligned16(x) && aligned16(y) && n >= 4;
  hipLaunchKernelGGL(axpy_kernel, dim3(vec ? chunk_grid(ctx, n >> 2) : grid_for(ctx, n)), dim3(NT), 0, ctx->stream, alpha, x, y,
                     (long)n, vec);
  EG_HIP_CHECK(hipGetLastError());
  return EG_OK;
}

int eg_fill_f32(eg_ctx* ctx, int64_t n, float value, float* out) {
  EG_REQUIRE(ctx, EG_ERR_INVALID, "eg_fill_f32: ctx is NULL");
  EG_REQUIRE(n >= 0, EG_ERR_INVALID, "eg_fill_f32: negative length");
  if (n == 0) return EG_OK;
  EG_REQUIRE(out, EG_ERR_INVALID, "eg_fill_f32: NULL tensor");
  int rc = eg::set_device(ctx);
  if (rc) return rc;
  const int vec = aligned16(out) && n >= 4;
  hipLaunchKernelGGL(fill_kernel, dim3(vec ? chunk_grid(ctx, n >> 2) : grid_for(ctx, n)), dim3(NT), 0, ctx->stream, value, out,
                     (long)n, vec);
  EG_HIP_CHECK(hipGetLastError());
  return EG_OK;
}

// Uniform random fill of TensorRandom tensors (`rand`, parser.nim:732-736; dropout's mask,
// dnn.nim:96-100).  The reference fills them on the host with Nim's RNG and uploads them on every
// call (model.nim:310-314); here a counter-based generator runs on the device: element i of
// fill number c of a stream seeded s is hash(s, c, i) — reproducible, order independent, no state
// per element.  The fill counter lives in device memory (state[1]) so that a captured launch
// sequence draws fresh numbers on every replay: eg_rng_advance bumps it.
int eg_fill_uniform(eg_ctx* ctx, int64_t n, float lo, float hi, const uint64_t* state, uint64_t stream, float* out) {
  EG_REQUIRE(ctx, EG_ERR_INVALID, "eg_fill_uniform: ctx is NULL");
  EG_REQUIRE(n >= 0, EG_ERR_INVALID, "eg_fill_uniform: negative length");
  if (n == 0) return EG_OK;
  EG_REQUIRE(out && state, EG_ERR_INVALID, "eg_fill_uniform: NULL pointer");
  int rc = eg::set_device(ctx);
  if (rc) return rc;
  hipLaunchKernelGGL(fill_uniform_kernel, dim3(grid_for(ctx, n)), dim3(NT), 0, ctx->stream, lo, hi, state,
                     (unsigned long long)stream, out, (long)n);
  EG_HIP_CHECK(hipGetLastError());
  return EG_OK;
}

int eg_rng_advance(eg_ctx* ctx, uint64_t* state) {
  EG_REQUIRE(ctx && state, EG_ERR_INVALID, "eg_rng_advance: NULL argument");
  int rc = eg::set_device(ctx);
  if (rc) return rc;
  hipLaunchKernelGGL(rng_advance_kernel, dim3(1), dim3(1), 0, ctx->stream, state);
  EG_HIP_CHECK(hipGetLastError());
  return EG_OK;
}

}  // extern "C"
