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
    const char* e = eg::sw::raw("EG_CONV_NO_DIRECT");
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

// compile[float64] (model.nim:253-260): the reference's own conv2 benchmark runs in float64 with 8 channels and 8 or 16
// filters of 3 x 3 (benchmarks/conv2/conv2.nim:134-138, 330-364: a 960 x 1280 x 8 image) — 157 MB in and out against
// 1.4 GFLOP.  Built three ways and measured at that shape (8 filters): a thread per pixel with the filter values as scalar
// loads, windows from memory 121 us, windows from LDS 132 us — the compiler hoists all F * FH * FW * C scalar loads and spills
// the scalar registers into vector lanes (1 444 lane moves around 576 multiply-adds); four pixels per thread 159 us.  So the
// filter bank lives in VECTOR registers instead, as the B fragments of the float64 matrix instruction: a wave multiplies
// 16 pixels x (FH * FW * C taps, four per instruction) x 16 filter columns (F <= 16; with 8 filters half of the columns
// idle, the instruction count is what the vector form would need at full rate), A fragments = the pixels' windows read
// from the FH input row segments a block stages in LDS (a window is a contiguous run of FW * C doubles per row).
// Summation order per output: taps in the order dy, dx, c, four at a time (the instruction's own order inside a group).
int conv2_direct_f64_try(eg_ctx* ctx, long N, long H, long W, long C, long F, long FH, long FW, const double* img, const double* flt,
                         double* out, int accumulate, bool* launched) {
  *launched = false;
  const long taps = FH * FW * C, run = FW * C;
  if (disabled() || F < 1 || F > 16 || taps > 144 || FH > 7) return EG_OK;
  const long Ho = H - FH + 1, Wo = W - FW + 1, P = N * Ho * Wo;
  if (Ho < 1 || Wo < 1 || P < 8192) return EG_OK;
  const long KS = (taps + 3) / 4;                 // matrix instructions per 16 pixels
  const long SEG = 128;                           // pixels of an output row per block: 4 waves x 2 groups of 16
  const long PIX = SEG + FW - 1;
  const long STR = C | 1;                         // doubles per staged pixel: odd, so that the 16 pixels of a fragment read fall on 16 different bank pairs
  const long ROWG = PIX * C;                      // doubles of a row segment in memory
  const long ROW = PIX * STR;                     // ... and in its LDS slot
  const long RING = FH + 1;                       // row slots: FH under the output row being computed + the one being fetched
  const long PRE = (ROWG + 511) / 512;            // 16-byte pieces of a row per thread
  if ((RING * ROW + 2) * 8 > 60 * 1024) return EG_OK;
  const bool vec_in = C % 2 == 0 && (reinterpret_cast<uintptr_t>(img) & 15) == 0;
  if (!vec_in) return EG_OK;                      // (odd channel counts / unaligned images: the generated kernel)
  const std::string name = "eg_conv_mfma64_c" + std::to_string(C) + "_f" + std::to_string(F) + "_" + std::to_string(FH) + "x" + std::to_string(FW);
  const std::string sSTR = std::to_string(STR), sROWG = std::to_string(ROWG);
  const std::string sF = std::to_string(F), sC = std::to_string(C), sROW = std::to_string(ROW), sKS = std::to_string(KS),
                    sTAPS = std::to_string(taps), sRUN = std::to_string(run), sRING = std::to_string(RING), sPRE = std::to_string(PRE),
                    sZERO = std::to_string(RING * ROW), sFH = std::to_string(FH);
  // A block walks `rpb` output rows of one 128-pixel column strip downwards: per output row ONE new input row segment is
  // fetched (a ring of FH + 1 row slots in LDS: the fetch of the next row is in flight while this row is multiplied), the
  // filter fragments are loaded once per block.  One block barrier per output row.  (One block per row segment, FH rows
  // staged each: 88 us at the benchmark shape, 34 us of which were block start-up — 9 580 blocks, each waiting for its
  // own filter fragments and rows.)
  std::string s = "typedef double d2 __attribute__((ext_vector_type(2)));\ntypedef double d4 __attribute__((ext_vector_type(4)));\n";
  s += "extern \"C\" __global__ void __launch_bounds__(256) " + name +
       "(const double* __restrict__ img, const double* __restrict__ flt, double* __restrict__ out, long segs, long yblocks, long rpb, long Ho, "
       "long Wo, long H, long W, int accumulate) {\n";
  s += "  __shared__ __attribute__((aligned(16))) double rows[" + std::to_string(RING * ROW + 2) + "];  // [RING][ROW], then a zero\n";
  // XCD-aware order: the 8 XCDs take blocks round robin; each gets a contiguous range of (n, row block, strip) order, so the
  // blocks that share halo rows and neighbouring strips share an L2
  s += "  long b = blockIdx.x;\n  { const long per = gridDim.x >> 3, full = per << 3; if (b < full) b = (b & 7) * per + (b >> 3); }\n";
  s += "  const long n = b / (yblocks * segs), rem = b - n * (yblocks * segs), yb = rem / segs, xb = (rem - yb * segs) * " + L(SEG) + ";\n";
  s += "  const long y0 = yb * rpb, y1 = y0 + rpb < Ho ? y0 + rpb : Ho;\n";
  s += "  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fk = lane >> 4;\n";
  s += "  const long avail = (W - xb < " + L(PIX) + " ? W - xb : " + L(PIX) + ") * " + L(C) + ";  // doubles of a segment that exist\n";
  s += "  const double* base = img + (n * H * W + xb) * " + L(C) + ";\n";
  s += "  d2 pre[" + sPRE + "];\n";
  s += "  auto fetch = [&](long r) {  // input row r of the strip -> registers\n";
  s += "    const double* src = base + r * W * " + L(C) + ";\n";
  s += "    _Pragma(\"unroll\") for (int i = 0; i < " + sPRE + "; ++i) {\n";
  s += "      const int e = tid * 2 + 512 * i;\n      pre[i] = d2{0.0, 0.0};\n      if (e < avail && e < " + sROWG + ") pre[i] = *(const d2*)(src + e);\n    }\n  };\n";
  s += "  auto stash = [&](long r) {  // registers -> the ring slot of input row r\n";
  s += "    double* dst = rows + (r % " + sRING + ") * " + sROW + ";\n";
  s += "    _Pragma(\"unroll\") for (int i = 0; i < " + sPRE + "; ++i) {\n";
  s += "      const int e = tid * 2 + 512 * i;\n      if (e < " + sROWG + ") { double* d = dst + (e / " + sC + ") * " + sSTR + " + e % " + sC + "; d[0] = pre[i][0]; d[1] = pre[i][1]; }\n    }\n  };\n";
  s += "  for (long r = y0; r < y0 + " + std::to_string(FH - 1) + "; ++r) { fetch(r); stash(r); }\n";
  s += "  fetch(y0 + " + std::to_string(FH - 1) + ");\n";
  s += "  if (tid == 0) { rows[" + sZERO + "] = 0.0; rows[" + std::to_string(RING * ROW + 1) + "] = 0.0; }\n";
  // B fragments: lane (k = fk, column = fr) of step s holds flt[fr][4 s + fk]; (dy, position in the row's window run) of that tap
  s += "  double bf[" + sKS + "];\n  int tdy[" + sKS + "], tj[" + sKS + "];\n";
  s += "  _Pragma(\"unroll\") for (int s = 0; s < " + sKS + "; ++s) {\n";
  s += "    const int kk = 4 * s + fk;\n";
  s += "    bf[s] = (fr < " + sF + " && kk < " + sTAPS + ") ? flt[fr * " + sTAPS + " + kk] : 0.0;\n";
  s += "    tdy[s] = kk < " + sTAPS + " ? kk / " + sRUN + " : -1;\n";
  s += "    tj[s] = ((kk % " + sRUN + ") / " + sC + " + wave * 32 + fr) * " + sSTR + " + kk % " + sC + ";\n  }\n";
  s += "  const bool active = xb + wave * 32 < Wo;\n";
  s += "  for (long y = y0; y < y1; ++y) {\n";
  s += "    stash(y + " + std::to_string(FH - 1) + ");\n";
  s += "    __syncthreads();\n";
  s += "    if (y + 1 < y1) fetch(y + " + sFH + ");\n";
  s += "    if (!active) continue;\n";
  s += "    d4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};\n";
  s += "    _Pragma(\"unroll\") for (int s = 0; s < " + sKS + "; ++s) {\n";
  s += "      const int o = tdy[s] < 0 ? " + sZERO + " : (int)((y + tdy[s]) % " + sRING + ") * " + sROW + " + tj[s];\n";
  s += "      const double a0 = rows[o], a1 = rows[tdy[s] < 0 ? o : o + 16 * " + sSTR + "];\n";
  s += "      acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, bf[s], acc0, 0, 0, 0);\n";
  s += "      acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, bf[s], acc1, 0, 0, 0);\n    }\n";
  // D: column fr (filter), row fk + 4 r (pixel of the group)
  s += "    if (fr < " + sF + ") {\n";
  s += "      _Pragma(\"unroll\") for (int g = 0; g < 2; ++g)\n";
  s += "        _Pragma(\"unroll\") for (int r = 0; r < 4; ++r) {\n";
  s += "          const long x = xb + wave * 32 + 16 * g + fk + 4 * r;\n";
  s += "          if (x < Wo) {\n";
  s += "            double* o = out + ((n * Ho + y) * Wo + x) * " + L(F) + " + fr;\n";
  s += "            const double v = g ? acc1[r] : acc0[r];\n";
  s += "            *o = accumulate ? *o + v : v;\n          }\n        }\n    }\n  }\n}\n";
  eg_kernel* k = get_or_build(ctx, name, s);
  if (!k) return EG_ERR_COMPILE;
  long segs = (Wo + SEG - 1) / SEG;
  long rpb = (N * Ho * segs + 5L * ctx->compute_units - 1) / (5L * ctx->compute_units);  // ~5 blocks per CU
  if (rpb < 4) rpb = 4;
  if (rpb > 64) rpb = 64;
  if (rpb > Ho) rpb = Ho;
  long yblocks = (Ho + rpb - 1) / rpb;
  const long blocks = N * yblocks * segs;
  if (blocks > 0x7fffffffL) return EG_OK;
  long Hl = H, Wl = W, Hol = Ho, Wol = Wo;
  void* args[] = {(void*)&img, (void*)&flt, (void*)&out, &segs, &yblocks, &rpb, &Hol, &Wol, &Hl, &Wl, &accumulate};
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
  // (the butterfly over all values of a step together, not value by value: rowfuse.cpp, round 6)
  s += "  _Pragma(\"unroll\") for (int off = 32; off >= 1; off >>= 1)\n";
  s += "    _Pragma(\"unroll\") for (int e = 0; e < " + std::to_string(E) + "; ++e) acc[e] += __shfl_xor(acc[e], off, 64);\n";
  s += "  if (lane == 0) {\n    _Pragma(\"unroll\") for (int e = 0; e < " + std::to_string(E) + "; ++e) red[wave * " + std::to_string(E) + " + e] = acc[e];\n  }\n  __syncthreads();\n";
  s += "  for (int e = threadIdx.x; e < " + std::to_string(E) + "; e += 256)\n";
  s += "    partial[(long)blockIdx.x * " + std::to_string(E) + " + e] = (red[e] + red[" + std::to_string(E) + " + e]) + (red[2 * " +
       std::to_string(E) + " + e] + red[3 * " + std::to_string(E) + " + e]);\n}\n";
  eg_kernel* k = get_or_build(ctx, name, s);
  if (!k) return EG_ERR_COMPILE;
  long blocks = (P + 256 * 8 - 1) / (256 * 8);  // at least 8 pixels per thread where the problem allows
  long cap = ctx->compute_units;  // few blocks: the shuffle reduction of E values per block is the fixed cost
  if (const char* e = eg::sw::raw("EG_CONV_DIRECT_BLOCKS")) cap = atol(e);
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
