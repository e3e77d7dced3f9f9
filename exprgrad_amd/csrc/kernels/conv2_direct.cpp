// Direct convolution for few input channels (the first layer of an image network: C = 1 or 3).
//
//   out[n,y,x,f]    (+)= sum_{dy,dx,c} img[n,y+dy,x+dx,c] * flt[f,dy,dx,c]          dnn.nim:45-49
//   gflt[f,dy,dx,c] (+)= sum_{n,y,x}   gout[n,y,x,f]     * img[n,y+dy,x+dx,c]       (derived)
//
// With C = 1 the contraction length of the implicit GEMM is FH*FW (25 for the 5x5 first layer of
// the reference's fashion_mnist network) and its width F = 8: a 64x64x32 MFMA tile is 90 % padding
// and the unaligned rows force the element-wise staging path (4096 x 28 x 28 x 1 -> 8: 325 us
// forward, 242 us filter gradient).  These shapes are bandwidth problems, so they get plain
// per-pixel kernels specialised (hiprtc) for the filter geometry — every extent a literal, every
// loop unrolled, the whole filter bank / the whole filter gradient in registers:
//   forward          one thread per output pixel: its FH*FW*C window against all F filters
//                    (summation order dy, dx, c as in the reference's loop nest)
//   filter gradient  a thread walks pixels grid-stride with the F*FH*FW*C partial gradient in
//                    registers; wave shuffle tree -> per-block partial row -> the library's
//                    fixed-order column sum (deterministic, no float atomics)
#include <cstdio>
#include <map>
#include <string>

#include "../eg_internal.hpp"

namespace eg {
namespace {

eg_kernel* get_or_build(eg_ctx* ctx, const std::string& name, const std::string& source) {
  auto it = ctx->jit.find(name);
  if (it != ctx->jit.end()) return it->second;
  eg_kernel* k = nullptr;
  if (eg_kernel_compile(ctx, name.c_str(), source.c_str(), &k) != EG_OK) return nullptr;
  ctx->jit[name] = k;
  return k;
}

bool disabled() {
  static const bool off = [] {
    const char* e = getenv("EG_CONV_NO_DIRECT");
    return e && e[0] && e[0] != '0';
  }();
  return off;
}

std::string L(long v) { return std::to_string(v) + "L"; }

}  // namespace

int conv2_direct_try(eg_ctx* ctx, long N, long H, long W, long C, long F, long FH, long FW, const float* img,
                     const float* flt, float* out, int accumulate, bool* launched) {
  *launched = false;
  const long taps = FH * FW * C;
  // F * taps multiply-adds per pixel on the vector pipe: only while that stays small (beyond it the
  // matrix cores win even with their padding), and only when there are enough pixels to fill the chip
  long max_c = 4, max_work = 512;
  if (const char* e = getenv("EG_CONV_DIRECT_LIMITS")) sscanf(e, "%ld,%ld", &max_c, &max_work);  // tuning aid
  if (disabled() || C > max_c || F < 1 || F * taps > max_work) return EG_OK;
  const long Ho = H - FH + 1, Wo = W - FW + 1, P = N * Ho * Wo;
  if (P < 8192) return EG_OK;
  const bool vec_out = F % 4 == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0;
  const std::string name = "eg_conv_direct_c" + std::to_string(C) + "_f" + std::to_string(F) + "_" + std::to_string(FH) +
                           "x" + std::to_string(FW) + (vec_out ? "_v4" : "");
  std::string s = "extern \"C\" __global__ void __launch_bounds__(256) " + name +
                  "(const float* __restrict__ img, const float* __restrict__ flt, float* __restrict__ out, long P, long Ho, "
                  "long Wo, long H, long W, int accumulate) {\n";
  // the filter values are wave-uniform: literal indices into the kernel argument become scalar loads
  s += "  for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < P; p += (long)gridDim.x * 256) {\n";
  s += "    const long n = p / (Ho * Wo), rem = p - n * (Ho * Wo), y = rem / Wo, x = rem - y * Wo;\n";
  s += "    const float* w = img + ((n * H + y) * W + x) * " + L(C) + ";\n";
  s += "    float acc[" + std::to_string(F) + "];\n";
  s += "    _Pragma(\"unroll\") for (int f = 0; f < " + std::to_string(F) + "; ++f) acc[f] = 0.0f;\n";
  s += "    _Pragma(\"unroll\") for (int dy = 0; dy < " + std::to_string(FH) + "; ++dy)\n";
  s += "      _Pragma(\"unroll\") for (int dc = 0; dc < " + std::to_string(FW * C) + "; ++dc) {\n";
  s += "        const float v = w[dy * W * " + L(C) + " + dc];\n";
  s += "        _Pragma(\"unroll\") for (int f = 0; f < " + std::to_string(F) + "; ++f)\n";
  s += "          acc[f] = acc[f] + v * flt[f * " + std::to_string(taps) + " + dy * " + std::to_string(FW * C) + " + dc];\n      }\n";
  s += "    float* o = out + p * " + L(F) + ";\n";
  if (vec_out) {  // 16-byte stores: F floats per pixel are contiguous
    s += "    typedef float f4 __attribute__((ext_vector_type(4)));\n";
    s += "    _Pragma(\"unroll\") for (int f = 0; f < " + std::to_string(F) + "; f += 4) {\n";
    s += "      f4 v = {acc[f], acc[f + 1], acc[f + 2], acc[f + 3]};\n";
    s += "      if (accumulate) { const f4 old = *(const f4*)(o + f); v = old + v; }\n";
    s += "      *(f4*)(o + f) = v;\n    }\n";
  } else {
    s += "    _Pragma(\"unroll\") for (int f = 0; f < " + std::to_string(F) + "; ++f) o[f] = accumulate ? o[f] + acc[f] : acc[f];\n";
  }
  s += "  }\n}\n";
  eg_kernel* k = get_or_build(ctx, name, s);
  if (!k) return EG_ERR_COMPILE;
  long blocks = (P + 255) / 256;
  const long cap = 16L * ctx->compute_units;
  if (blocks > cap) blocks = cap;
  long Hl = H, Wl = W, Hol = Ho, Wol = Wo, Pl = P;
  void* args[] = {(void*)&img, (void*)&flt, (void*)&out, &Pl, &Hol, &Wol, &Hl, &Wl, &accumulate};
  int rc = kernel_launch_raw(k, (unsigned)blocks, 1, 1, 256, args);
  if (rc) return rc;
  *launched = true;
  return EG_OK;
}

int conv2_direct_grad_filter_try(eg_ctx* ctx, long N, long H, long W, long C, long F, long FH, long FW, const float* img,
                                 const float* gout, float* gflt, int accumulate, bool* launched) {
  *launched = false;
  const long taps = FH * FW * C;
  if (disabled() || C > 4 || F < 1 || F * taps > 224) return EG_OK;
  const long Ho = H - FH + 1, Wo = W - FW + 1, P = N * Ho * Wo;
  if (P < 131072) return EG_OK;  // the per-block shuffle reduction of F*taps values needs pixels to amortise
  const long E = F * taps;  // length of the gradient = of a partial row
  const std::string name = "eg_conv_direct_gf_c" + std::to_string(C) + "_f" + std::to_string(F) + "_" + std::to_string(FH) +
                           "x" + std::to_string(FW);
  std::string s = "extern \"C\" __global__ void __launch_bounds__(256) " + name +
                  "(const float* __restrict__ img, const float* __restrict__ gout, float* __restrict__ partial, long P, "
                  "long Ho, long Wo, long H, long W) {\n";
  s += "  float acc[" + std::to_string(E) + "];\n";
  s += "  _Pragma(\"unroll\") for (int e = 0; e < " + std::to_string(E) + "; ++e) acc[e] = 0.0f;\n";
  s += "  for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < P; p += (long)gridDim.x * 256) {\n";
  s += "    const long n = p / (Ho * Wo), rem = p - n * (Ho * Wo), y = rem / Wo, x = rem - y * Wo;\n";
  s += "    const float* w = img + ((n * H + y) * W + x) * " + L(C) + ";\n";
  s += "    float win[" + std::to_string(taps) + "], g[" + std::to_string(F) + "];\n";
  s += "    _Pragma(\"unroll\") for (int dy = 0; dy < " + std::to_string(FH) + "; ++dy)\n";
  s += "      _Pragma(\"unroll\") for (int dc = 0; dc < " + std::to_string(FW * C) + "; ++dc) win[dy * " +
       std::to_string(FW * C) + " + dc] = w[dy * W * " + L(C) + " + dc];\n";
  s += "    _Pragma(\"unroll\") for (int f = 0; f < " + std::to_string(F) + "; ++f) g[f] = gout[p * " + L(F) + " + f];\n";
  s += "    _Pragma(\"unroll\") for (int f = 0; f < " + std::to_string(F) + "; ++f)\n";
  s += "      _Pragma(\"unroll\") for (int t = 0; t < " + std::to_string(taps) + "; ++t) acc[f * " + std::to_string(taps) +
       " + t] = acc[f * " + std::to_string(taps) + " + t] + g[f] * win[t];\n  }\n";
  s += "  __shared__ float red[4 * " + std::to_string(E) + "];\n";
  s += "  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;\n";
  s += "  _Pragma(\"unroll\") for (int e = 0; e < " + std::to_string(E) + "; ++e) {\n    float v = acc[e];\n";
  s += "    _Pragma(\"unroll\") for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);\n";
  s += "    if (lane == 0) red[wave * " + std::to_string(E) + " + e] = v;\n  }\n  __syncthreads();\n";
  s += "  for (int e = threadIdx.x; e < " + std::to_string(E) + "; e += 256)\n";
  s += "    partial[(long)blockIdx.x * " + std::to_string(E) + " + e] = (red[e] + red[" + std::to_string(E) + " + e]) + (red[2 * " +
       std::to_string(E) + " + e] + red[3 * " + std::to_string(E) + " + e]);\n}\n";
  eg_kernel* k = get_or_build(ctx, name, s);
  if (!k) return EG_ERR_COMPILE;
  long blocks = (P + 256 * 8 - 1) / (256 * 8);  // at least 8 pixels per thread where the problem allows
  long cap = ctx->compute_units;  // few blocks: the shuffle reduction of E values per block is the fixed cost
  if (const char* e = getenv("EG_CONV_DIRECT_BLOCKS")) cap = atol(e);
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  const size_t pfloats = ((size_t)blocks * E + 3) & ~(size_t)3;
  const size_t sfloats = (size_t)colsum_scratch_floats(ctx, blocks, E);
  int rc = ensure_workspace(ctx, (pfloats + sfloats) * sizeof(float));
  if (rc) return rc;
  float* partial = static_cast<float*>(ctx->workspace);
  long Hl = H, Wl = W, Hol = Ho, Wol = Wo, Pl = P;
  void* args[] = {(void*)&img, (void*)&gout, (void*)&partial, &Pl, &Hol, &Wol, &Hl, &Wl};
  rc = kernel_launch_raw(k, (unsigned)blocks, 1, 1, 256, args);
  if (rc) return rc;
  rc = colsum_with_scratch(ctx, blocks, E, partial, gflt, accumulate, partial + pfloats);
  if (rc) return rc;
  *launched = true;
  return EG_OK;
}

}  // namespace eg
