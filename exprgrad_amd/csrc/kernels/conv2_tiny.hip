// Tiny convolutions (round 4): conv2 and its two gradients when the whole problem is a few million multiply-adds — one
// batch-32 step of the reference's fashion_mnist network (examples/fashion_mnist/fashion_mnist.nim:39-57: conv2(8,3,3,16) on
// 32 x 12 x 12 x 8 images is 3.7 M of them).  The contraction kernels spend ~10 us on such a launch whatever its size (three
// dependent k-tiles from a cold L2, ragged 64 x 64 tiles of which a quarter of the columns exist, a k-sliced filter gradient
// with its second launch), the per-pixel kernels of conv2_direct.cpp put a handful of blocks on the chip.  Here one thread
// owns one output element (forward, image gradient: the filter bank sits in LDS, one launch), and the filter gradient is
// blocks of pixels that each accumulate every output element over their pixels, folded by eg::slab_sum (two launches near
// the launch floor instead of a k-sliced contraction and its sum):
//   forward          out[n,y,x,f]  (+)= sum_{dy,dx,c} img[n,y+dy,x+dx,c] * flt[f,dy,dx,c]      (dnn.nim:45-53)
//   image gradient   gimg[n,y,x,c] (+)= sum_{dy,dx,f} gout[n,y-dy,x-dx,f] * flt[f,dy,dx,c]     (derive of the above,
//   filter gradient  gflt[f,dy,dx,c] (+)= sum_{n,y,x} gout[n,y,x,f] * img[n,y+dy,x+dx,c]         passes.nim:519-549)
// Sums run in a fixed order (taps in storage order; pixel ranges in order, folded by slab_sum's fixed tree): run-to-run
// deterministic, within rounding of the contraction route (different order).  EG_CONV_NO_TINY=1 (read per call) keeps the
// contraction route.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>

#include "../eg_internal.hpp"

namespace {

constexpr long kMaxMacs = 6L << 20;        // beyond this the contraction kernels win (measured: fit at batch 32 .. 128)
constexpr long kMaxBankFloats = 12288;     // 48 KiB of LDS for the filter bank

struct TinyArgs {
  const float* a;   // img (forward, filter gradient) / gout (image gradient)
  const float* b;   // flt (forward, image gradient) / gout (filter gradient)
  float* out;
  int N, H, W, C, F, FH, FW, Ho, Wo;
  int accumulate;
  int unused;
};

// forward: thread = (pixel, f), f fastest; the bank in LDS as [k][f] (k = (dy, dx, c) in storage order)
__global__ __launch_bounds__(256) void conv2_tiny_forward_kernel(TinyArgs a) {
  extern __shared__ float bank[];
  const int K = a.FH * a.FW * a.C, F = a.F;
  for (int i = threadIdx.x; i < K * F; i += 256) {
    const int f = i % F, k = i / F;
    bank[i] = a.b[(long)f * K + k];
  }
  __syncthreads();
  const long total = (long)a.N * a.Ho * a.Wo * F;
  const long o = (long)blockIdx.x * 256 + threadIdx.x;
  if (o >= total) return;
  const int f = (int)(o % F);
  long p = o / F;
  const int x = (int)(p % a.Wo);
  p /= a.Wo;
  const int y = (int)(p % a.Ho), n = (int)(p / a.Ho);
  const float* src = a.a + (((long)n * a.H + y) * a.W + x) * a.C;
  const int row = a.FW * a.C;   // one filter row is contiguous in the image: FW pixels x C channels
  float acc = 0.f;
  for (int dy = 0; dy < a.FH; ++dy) {
    const float* s = src + (long)dy * a.W * a.C;
    const float* w = bank + (long)dy * row * F + f;
#pragma unroll 8
    for (int j = 0; j < row; ++j) acc = fmaf(s[j], w[(long)j * F], acc);
  }
  a.out[o] = a.accumulate ? a.out[o] + acc : acc;
}

// image gradient: thread = (pixel of the image, c), c fastest; the bank in LDS as [dy][dx][f][c]
__global__ __launch_bounds__(256) void conv2_tiny_grad_image_kernel(TinyArgs a) {
  extern __shared__ float bank[];
  const int C = a.C, F = a.F, taps = a.FH * a.FW;
  for (int i = threadIdx.x; i < taps * F * C; i += 256) {
    const int c = i % C, q = i / C;
    const int f = q % F, tap = q / F;
    bank[i] = a.b[((long)f * taps + tap) * C + c];
  }
  __syncthreads();
  const long total = (long)a.N * a.H * a.W * C;
  const long o = (long)blockIdx.x * 256 + threadIdx.x;
  if (o >= total) return;
  const int c = (int)(o % C);
  long p = o / C;
  const int x = (int)(p % a.W);
  p /= a.W;
  const int y = (int)(p % a.H), n = (int)(p / a.H);
  float acc = 0.f;
  for (int dy = 0; dy < a.FH; ++dy) {
    const int yy = y - dy;
    if (yy < 0 || yy >= a.Ho) continue;
    for (int dx = 0; dx < a.FW; ++dx) {
      const int xx = x - dx;
      if (xx < 0 || xx >= a.Wo) continue;
      const float* g = a.a + (((long)n * a.Ho + yy) * a.Wo + xx) * F;
      const float* w = bank + (long)(dy * a.FW + dx) * F * C + c;
#pragma unroll 8
      for (int f = 0; f < F; ++f) acc = fmaf(g[f], w[(long)f * C], acc);
    }
  }
  a.out[o] = a.accumulate ? a.out[o] + acc : acc;
}

// filter gradient, first pass: block b owns the pixels [b * R, b * R + R) and ALL outputs; thread t accumulates outputs
// t, t + 256, ... (at most ACC of them) over those pixels and writes them to slab b.  Consecutive threads are consecutive
// (dx, c) of a filter row — a wave reads runs of the image row and one or two gradient values per pixel (broadcast).
// eg::slab_sum folds the slabs in its fixed order (second launch).
constexpr int ACC = 8;
template <int NACC>
__global__ __launch_bounds__(256) void conv2_tiny_grad_filter_kernel(TinyArgs a, float* __restrict__ slabs, int R) {
  const int Kf = a.FH * a.FW * a.C;
  const int total = a.F * Kf;
  int off[NACC], fo[NACC];   // image offset of the output's tap and channel; its filter
  float acc[NACC];
#pragma unroll
  for (int u = 0; u < NACC; ++u) {
    const int o = threadIdx.x + u * 256;
    const int oo = o < total ? o : total - 1;   // (threads past the end redo the last output: loads stay unconditional)
    const int f = oo / Kf, k = oo % Kf;
    const int c = k % a.C, tap = k / a.C;
    off[u] = ((tap / a.FW) * a.W + tap % a.FW) * a.C + c;
    fo[u] = f;
    acc[u] = 0.f;
  }
  const long P = (long)a.N * a.Ho * a.Wo;
  const long p0 = (long)blockIdx.x * R;
  const int count = (int)(p0 + R < P ? R : P - p0);
  int x = (int)(p0 % a.Wo);
  long q = p0 / a.Wo;
  int y = (int)(q % a.Ho), n = (int)(q / a.Ho);
  const float* g = a.b + p0 * a.F;
  // (block-uniform walk; four pixels' loads are issued before their multiply-adds)
  constexpr int UP = 4;
  for (int i = 0; i < count; i += UP) {
    float gv[UP][NACC], iv[UP][NACC];
#pragma unroll
    for (int v = 0; v < UP; ++v) {
      const bool live = i + v < count;   // block-uniform
      const float* im = a.a + (((long)n * a.H + y) * a.W + x) * a.C;
#pragma unroll
      for (int u = 0; u < NACC; ++u) {
        gv[v][u] = live ? g[(long)(i + v) * a.F + fo[u]] : 0.f;
        iv[v][u] = live ? im[off[u]] : 0.f;
      }
      if (live && ++x == a.Wo) {
        x = 0;
        if (++y == a.Ho) {
          y = 0;
          ++n;
        }
      }
    }
#pragma unroll
    for (int v = 0; v < UP; ++v)
#pragma unroll
      for (int u = 0; u < NACC; ++u) acc[u] = fmaf(gv[v][u], iv[v][u], acc[u]);
  }
  float* slab = slabs + (long)blockIdx.x * total;
#pragma unroll
  for (int u = 0; u < NACC; ++u)
    if (threadIdx.x + u * 256 < total) slab[threadIdx.x + u * 256] = acc[u];
}

bool tiny_on() { return eg::sw::raw("EG_CONV_NO_TINY") == nullptr; }   // (read per call: tests compare the two routes)

bool fits_int(long v) { return v >= 0 && v < (1L << 31); }

}  // namespace

namespace eg {

int conv2_tiny_forward_try(eg_ctx* ctx, long N, long H, long W, long C, long F, long FH, long FW, const float* img,
                           const float* flt, float* out, int accumulate, bool* launched) {
  *launched = false;
  const long Ho = H - FH + 1, Wo = W - FW + 1, K = FH * FW * C;
  const long outputs = N * Ho * Wo * F;
  if (!tiny_on() || C <= 0 || outputs <= 0 || K * F > kMaxBankFloats || outputs > (1L << 24) || outputs * K > kMaxMacs) return EG_OK;
  if (!fits_int(N * H * W * C)) return EG_OK;
  TinyArgs a = {img, flt, out, (int)N, (int)H, (int)W, (int)C, (int)F, (int)FH, (int)FW, (int)Ho, (int)Wo, accumulate, 0};
  hipLaunchKernelGGL(conv2_tiny_forward_kernel, dim3((unsigned)((outputs + 255) / 256)), dim3(256), (size_t)(K * F) * sizeof(float),
                     ctx->stream, a);
  EG_HIP_CHECK(hipGetLastError());
  *launched = true;
  return EG_OK;
}

int conv2_tiny_grad_image_try(eg_ctx* ctx, long N, long H, long W, long C, long F, long FH, long FW, const float* flt,
                              const float* gout, float* gimg, int accumulate, bool* launched) {
  *launched = false;
  const long Ho = H - FH + 1, Wo = W - FW + 1, bank = FH * FW * F * C;
  const long outputs = N * H * W * C;
  if (!tiny_on() || outputs <= 0 || F <= 0 || bank > kMaxBankFloats || outputs > (1L << 24) || outputs * FH * FW * F > kMaxMacs) return EG_OK;
  if (!fits_int(N * Ho * Wo * F)) return EG_OK;
  TinyArgs a = {gout, flt, gimg, (int)N, (int)H, (int)W, (int)C, (int)F, (int)FH, (int)FW, (int)Ho, (int)Wo, accumulate, 0};
  hipLaunchKernelGGL(conv2_tiny_grad_image_kernel, dim3((unsigned)((outputs + 255) / 256)), dim3(256), (size_t)bank * sizeof(float),
                     ctx->stream, a);
  EG_HIP_CHECK(hipGetLastError());
  *launched = true;
  return EG_OK;
}

int conv2_tiny_grad_filter_try(eg_ctx* ctx, long N, long H, long W, long C, long F, long FH, long FW, const float* img,
                               const float* gout, float* gflt, int accumulate, bool* launched) {
  *launched = false;
  const long Ho = H - FH + 1, Wo = W - FW + 1, P = N * Ho * Wo;
  const long outputs = F * FH * FW * C;
  if (!tiny_on() || outputs <= 0 || P <= 0 || outputs > ACC * 256 || outputs * P > kMaxMacs) return EG_OK;
  if (!fits_int(N * H * W * C) || !fits_int(P * F)) return EG_OK;
  // pixel ranges: two blocks per CU when there are pixels for it, at least 8 pixels each
  long blocks = 2L * ctx->compute_units;
  if (blocks > P / 8) blocks = P / 8 > 0 ? P / 8 : 1;
  const long R = (P + blocks - 1) / blocks;
  blocks = (P + R - 1) / R;
  int rc = eg::ensure_workspace(ctx, (size_t)blocks * outputs * sizeof(float));
  if (rc) return rc;
  float* slabs = static_cast<float*>(ctx->workspace);
  if (!slab_sum_supported(outputs, slabs, gflt)) return EG_OK;
  TinyArgs a = {img, gout, gflt, (int)N, (int)H, (int)W, (int)C, (int)F, (int)FH, (int)FW, (int)Ho, (int)Wo, accumulate, 0};
  const long per_thread = (outputs + 255) / 256;
  if (per_thread <= 1) hipLaunchKernelGGL(conv2_tiny_grad_filter_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, a, slabs, (int)R);
  else if (per_thread <= 2) hipLaunchKernelGGL(conv2_tiny_grad_filter_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, a, slabs, (int)R);
  else if (per_thread <= 5) hipLaunchKernelGGL(conv2_tiny_grad_filter_kernel<5>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, a, slabs, (int)R);
  else hipLaunchKernelGGL(conv2_tiny_grad_filter_kernel<ACC>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, a, slabs, (int)R);
  EG_HIP_CHECK(hipGetLastError());
  *launched = true;
  return slab_sum(ctx, blocks, outputs, slabs, gflt, accumulate);
}

}  // namespace eg
