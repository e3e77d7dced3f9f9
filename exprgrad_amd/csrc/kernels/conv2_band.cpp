// Convolutions with few channels AND few filters on the matrix cores, float32 and float64 (round 5).
//
//   out[n,y,x,f]        (+)= sum_{dy,dx,c} img[n,y+dy,x+dx,c] * flt[f,dy,dx,c]            dnn.nim:45-49
//   gimg[n,y+dy,x+dx,c] (+)= sum_f         gout[n,y,x,f]      * flt[f,dy,dx,c]            derive (passes.nim:519-549)
//   gflt[f,dy,dx,c]     (+)= sum_{n,y,x}   gout[n,y,x,f]      * img[n,y+dy,x+dx,c]        derive
//
// The layers in front of an image network (fashion_mnist.nim:39-57: 1 -> 8 filters of 5 x 5 on 28 x 28 images, 8 -> 16 of
// 3 x 3 on 12 x 12) have neither the 16-channel chunks of the LDS-halo kernels nor enough columns for a contraction tile:
// at batch 4096 the five convolution launches of a training step were 57 + 91 + 42 + 82 + 77 us (per-pixel kernels with
// the filter values as scalar loads — the compiler spills the scalar registers into vector lanes — and 128 x 32 / 64 x 64
// implicit-GEMM tiles that are mostly padding).  Here the small operand is the B (or A) fragment of a 16 x 16 x 4 matrix
// instruction and lives in registers:
//   forward / image gradient   16 pixels x 4 taps x 16 filter columns per instruction; the image gradient is the same
//                              kernel on the output gradient with a virtual zero border and the filter bank read
//                              flipped and with channels and filters exchanged (a full correlation)
//   filter gradient            16 filter rows x 4 pixels x 16 taps per instruction, accumulated per wave, folded per block
//                              in a fixed order, per-block partial rows summed by the library's fixed-order column sum
// A block stages a band of input rows (of one image, or several whole small images) in LDS with an odd pixel stride (the 16
// pixels of a fragment read then fall on 16 different banks) and walks the band's pixels FLATTENED, 16 at a time: a
// window element of pixel p is at base(p) + offset(tap), so a group of 16 pixels may straddle rows and images.
// Kernels are generated per shape (every extent a literal) and built with hiprtc, like kernels/conv2_direct.cpp.
#include <cstdio>
#include <cstdlib>
#include <map>
#include <string>

#include "../eg_internal.hpp"

namespace eg {
namespace {

eg_kernel* get_or_build(eg_ctx* ctx, const std::string& name, const std::string& source) {
  auto it = ctx->jit.find(name);
  if (it != ctx->jit.end()) return it->second;
  if (const char* dump = eg::sw::raw("EG_DUMP_BAND")) {  // debugging aid: the generated translation unit
    if (FILE* fp = fopen((std::string(dump) + "/" + name + ".hip").c_str(), "w")) {
      fputs(source.c_str(), fp);
      fclose(fp);
    }
  }
  eg_kernel* k = nullptr;
  if (eg_kernel_compile(ctx, name.c_str(), source.c_str(), &k) != EG_OK) return nullptr;
  ctx->jit[name] = k;
  return k;
}

bool disabled() {
  const char* e = eg::sw::raw("EG_CONV_NO_BAND");  // read per call: a test compares the routes
  return e && e[0] && e[0] != '0';
}

std::string S(long v) { return std::to_string(v); }

struct Band {
  long NB = 1, R = 1;       // images and output rows per block
  long RR = 0, WP = 0;      // staged rows per image, staged pixels per row
  long STR = 1;             // elements per staged pixel (odd)
  long PT = 0;              // output pixels per block
  long lds_elems = 0;
  bool ok = false;
};

// in: [N, HI, WI, CI] seen with a zero border of (PY, PX); out rows Ho = HI + 2 PY - FH + 1, columns Wo likewise
Band plan_band(long HI, long WI, long CI, long FH, long FW, long PY, long PX, long elem_bytes, long per_pixel = 0) {  // per_pixel: further LDS elements per output pixel
  Band b;
  const long Ho = HI + 2 * PY - FH + 1, Wo = WI + 2 * PX - FW + 1;
  if (Ho < 1 || Wo < 1) return b;
  b.WP = WI + 2 * PX;
  b.STR = CI | 1;
  const long budget = 40 * 1024 / elem_bytes;
  long target = 1024;
  if (const char* e = eg::sw::raw("EG_CONV_BAND_PIXELS")) target = atol(e) > 0 ? atol(e) : target;  // tuning aid
  b.R = std::min(Ho, std::max(1L, target / Wo));
  b.NB = b.R == Ho ? std::max(1L, std::min(16L, target / (Ho * Wo))) : 1;
  auto elems = [&](long nb, long r) { return nb * (r + FH - 1) * b.WP * b.STR; };
  auto all = [&](long nb, long r) { return elems(nb, r) + nb * r * Wo * per_pixel; };
  while (b.NB > 1 && all(b.NB, b.R) > budget) --b.NB;
  while (b.R > 1 && all(b.NB, b.R) > budget) --b.R;
  if (all(b.NB, b.R) > budget) return b;
  b.RR = b.R + FH - 1;
  b.PT = b.NB * b.R * Wo;
  b.lds_elems = elems(b.NB, b.R);
  b.ok = true;
  return b;
}

struct Ty {
  bool f64;
  const char* T;
  const char* acc;      // accumulator vector type
  const char* mfma;     // builtin
  const char* zero;
  const char* sfx;
  // pixel (row of D) that register r of a lane with fk = lane >> 4 holds
  std::string drow(const std::string& r) const { return f64 ? "(fk + 4 * " + r + ")" : "(4 * fk + " + r + ")"; }
};
const Ty kF32 = {false, "float", "f4", "__builtin_amdgcn_mfma_f32_16x16x4f32", "0.0f", "f32"};
const Ty kF64 = {true, "double", "d4", "__builtin_amdgcn_mfma_f64_16x16x4f64", "0.0", "f64"};

std::string prelude() {
  return "typedef float f4 __attribute__((ext_vector_type(4)));\ntypedef double d4 __attribute__((ext_vector_type(4)));\ntypedef double d2v __attribute__((ext_vector_type(2)));\n";
}

// Stage the block's band: rows [y0 - PY, y0 - PY + RR) of images n0 .. n0 + NB - 1, columns [-PX, WI + PX), zeros outside.
std::string stage_code(const Ty& ty, const Band& b, long HI, long WI, long CI, long PY, long PX, const char* src) {
  std::string s;
  // A wave takes whole staged rows (image i, row ry: wave-uniform), its lanes walk the row's elements.  The loads of a CHUNK of
  // rows are all issued before the first of them is stored: a loop that loads, waits and stores element by element keeps one
  // load in flight per wave, and the waves then spend their life waiting for memory (measured: 14 500 cycles per wave of the
  // image-gradient kernel for 2 300 cycles of matrix work, two thirds of the SIMDs empty).
  const long row_elems = b.WP * CI, rows = b.NB * b.RR;
  const long per_row = (row_elems + 63) / 64;                       // loads per lane and row
  const long rows_per_wave = (rows + 3) / 4;
  long chunk = std::max(1L, 32 / per_row);                          // rows per chunk: at most ~32 loads in flight per lane
  if (chunk > rows_per_wave) chunk = rows_per_wave;
  s += "  for (int rbase = 0; rbase < " + S(rows_per_wave) + "; rbase += " + S(chunk) + ") {\n";
  s += "    " + std::string(ty.T) + " stg[" + S(chunk) + "][" + S(per_row) + "];\n";
  s += "    _Pragma(\"unroll\") for (int rr = 0; rr < " + S(chunk) + "; ++rr) {\n";
  s += "      const int row = (rbase + rr) * 4 + (tid >> 6), i = row / " + S(b.RR) + ", ry = row % " + S(b.RR) + ";\n";
  s += "      const long gn = n0 + i, gy = y0 + ry - " + S(PY) + ";\n";
  s += "      const bool row_ok = rbase + rr < " + S(rows_per_wave) + " && row < " + S(rows) + " && gn < N && gy >= 0 && gy < " + S(HI) + ";\n";
  s += "      const " + std::string(ty.T) + "* src_row = " + src + " + (gn * " + S(HI) + " + gy) * " + S(WI * CI) + " - " + S(PX * CI) + ";\n";
  s += "      _Pragma(\"unroll\") for (int u = 0; u < " + S(per_row) + "; ++u) {\n";
  s += "        const int e = (tid & 63) + 64 * u, px = e / " + S(CI) + ";\n";
  s += "        stg[rr][u] = (row_ok && e < " + S(row_elems) + " && px >= " + S(PX) + " && px < " + S(PX + WI) + ") ? src_row[e] : " + ty.zero + ";\n      }\n    }\n";
  s += "    _Pragma(\"unroll\") for (int rr = 0; rr < " + S(chunk) + "; ++rr) {\n";
  s += "      const int row = (rbase + rr) * 4 + (tid >> 6);\n";
  s += "      if (rbase + rr < " + S(rows_per_wave) + " && row < " + S(rows) + ") {\n";
  s += "        _Pragma(\"unroll\") for (int u = 0; u < " + S(per_row) + "; ++u) {\n";
  s += "          const int e = (tid & 63) + 64 * u, px = e / " + S(CI) + ", ci = e % " + S(CI) + ";\n";
  s += "          if (e < " + S(row_elems) + ") band[row * " + S(b.WP * b.STR) + " + px * " + S(b.STR) + " + ci] = stg[rr][u];\n        }\n      }\n    }\n  }\n";
  return s;
}

// forward (flip = false) / image gradient (flip = true: `in` is the output gradient, CI = F of the layer, FO = C of the layer)
int launch_forward(eg_ctx* ctx, const Ty& ty, bool flip, long N, long HI, long WI, long CI, long FO, long FH, long FW, long PY, long PX,
                   const void* in, const void* flt, void* out, int accumulate, bool* launched) {
  *launched = false;
  const long taps = FH * FW * CI;
  const long Ho = HI + 2 * PY - FH + 1, Wo = WI + 2 * PX - FW + 1;
  if (disabled() || CI < 1 || FO < 1 || FO > 16 || CI > 16 || taps > 256 || N * Ho * Wo < 4096) return EG_OK;
  const Band b = plan_band(HI, WI, CI, FH, FW, PY, PX, ty.f64 ? 8 : 4);
  if (!b.ok) return EG_OK;
  const long KS = (taps + 3) / 4, G = (b.PT + 15) / 16;
  // static LDS of the kernel: the band, the origin table, the waves' parking rows — 64 KB is the limit of a block's static allocation
  if (b.lds_elems * (ty.f64 ? 8 : 4) + G * 64 + 4 * 32 * FO * (ty.f64 ? 8 : 4) > 60 * 1024) return EG_OK;
  const std::string name = std::string("eg_conv_band_") + ty.sfx + (flip ? "_gi" : "_fw") + "_c" + S(CI) + "_f" + S(FO) + "_" + S(FH) + "x" + S(FW) + "_" +
                           S(HI) + "x" + S(WI) + "_b" + S(b.NB) + "r" + S(b.R) + ((FO % (ty.f64 ? 2 : 4) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0) ? "_w" : "");
  std::string s = prelude();
  s += "extern \"C\" __global__ void __launch_bounds__(256) " + name + "(const " + ty.T + "* __restrict__ in, const " + ty.T + "* __restrict__ flt, " + ty.T +
       "* __restrict__ out, long N, long ybands, int accumulate) {\n";
  const long VE = ty.f64 ? 2 : 4;                                   // elements of a 16-byte piece
  const bool wide = FO % VE == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0;   // a group's 16 x FO outputs leave as 16-byte pieces
  s += "  __shared__ " + std::string(ty.T) + " band[" + S(b.lds_elems) + "];\n";
  s += "  __shared__ int orig[" + S(G * 16) + "];   // window origin of every output pixel of the band (one decode per pixel, not per use)\n";
  if (wide) s += "  __shared__ __attribute__((aligned(16))) " + std::string(ty.T) + " park[4][" + S(32 * FO) + "];   // per wave: two groups of 16 pixels x FO outputs, as they lie in memory\n";
  s += "  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fk = lane >> 4;\n";
  s += "  const long n0 = (long)(blockIdx.x / ybands) * " + S(b.NB) + ", y0 = (long)(blockIdx.x % ybands) * " + S(b.R) + ";\n";
  s += stage_code(ty, b, HI, WI, CI, PY, PX, "in");
  s += "  for (int q = tid; q < " + S(G * 16) + "; q += 256) {\n";
  s += "    const int qq = q < " + S(b.PT) + " ? q : 0, i = qq / " + S(b.R * Wo) + ", r0 = qq % " + S(b.R * Wo) + ", y = r0 / " + S(Wo) + ", x = r0 % " + S(Wo) + ";\n";
  s += "    orig[q] = ((i * " + S(b.RR) + " + y) * " + S(b.WP) + " + x) * " + S(b.STR) + ";\n  }\n";
  // filter fragments: lane (k = fk, column = fr) of step s holds the bank's value for tap 4 s + fk and output column fr
  s += "  " + std::string(ty.T) + " bf[" + S(KS) + "];\n  int toff[" + S(KS) + "];\n";
  s += "  _Pragma(\"unroll\") for (int s = 0; s < " + S(KS) + "; ++s) {\n";
  s += "    const int kk = 4 * s + fk, dy = kk / " + S(FW * CI) + ", dx = (kk / " + S(CI) + ") % " + S(FW) + ", ci = kk % " + S(CI) + ";\n";
  if (!flip)
    s += "    bf[s] = (fr < " + S(FO) + " && kk < " + S(taps) + ") ? flt[fr * " + S(taps) + " + kk] : " + ty.zero + ";\n";
  else  // flt[f = ci][FH - 1 - dy][FW - 1 - dx][c = fr] of the layer's bank [F = CI][FH][FW][C = FO]
    s += "    bf[s] = (fr < " + S(FO) + " && kk < " + S(taps) + ") ? flt[((ci * " + S(FH) + " + (" + S(FH - 1) + " - dy)) * " + S(FW) + " + (" + S(FW - 1) + " - dx)) * " + S(FO) +
         " + fr] : " + ty.zero + ";\n";
  s += "    toff[s] = kk < " + S(taps) + " ? (dy * " + S(b.WP) + " + dx) * " + S(b.STR) + " + ci : -1;\n  }\n";
  // The band's output pixels are ONE contiguous run of the output (whole rows of one image, or whole images): pixel q of the
  // band is output pixel pix0 + q, and it exists while pix0 + q < lim (the band may hang over the last row / image).
  s += "  const long pix0 = (n0 * " + S(Ho) + " + y0) * " + S(Wo) + ", lim = (n0 + " + S(b.NB) + " < N ? n0 + " + S(b.NB) + " : N) * " + S(Ho * Wo) + ";\n";
  s += "  __syncthreads();\n";
  s += "  " + std::string(ty.T) + "* const run = out + pix0 * " + S(FO) + ";   // the band's outputs: one contiguous run\n";
  s += "  const long left = (lim - pix0) * " + S(FO) + ";                      // elements of the run that exist\n";
  s += "  for (int g = wave * 2; g < " + S(G) + "; g += 8) {\n";
  s += "    const int o0 = orig[g * 16 + fr], o1 = g + 1 < " + S(G) + " ? orig[g * 16 + 16 + fr] : 0;\n";
  s += "    " + std::string(ty.acc) + " acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};\n";
  // fragments of eight steps are read before their sixteen matrix instructions are issued (read -> wait -> multiply step by
  // step costs an LDS round trip per instruction: the image gradient of the 8 -> 16 layer ran 6x its matrix time)
  s += "    _Pragma(\"unroll\") for (int s0 = 0; s0 < " + S(KS) + "; s0 += 8) {\n";
  s += "      " + std::string(ty.T) + " a0[8], a1[8];\n";
  s += "      _Pragma(\"unroll\") for (int u = 0; u < 8; ++u) {\n";
  s += "        if (s0 + u < " + S(KS) + ") { const int to = toff[s0 + u] < 0 ? 0 : toff[s0 + u]; a0[u] = band[o0 + to]; a1[u] = band[o1 + to]; }\n      }\n";
  s += "      _Pragma(\"unroll\") for (int u = 0; u < 8; ++u) {\n";
  s += "        if (s0 + u < " + S(KS) + ") {\n";
  s += "          const bool tap = toff[s0 + u] >= 0;\n";
  s += "          acc0 = " + std::string(ty.mfma) + "(tap ? a0[u] : " + ty.zero + ", bf[s0 + u], acc0, 0, 0, 0);\n";
  s += "          acc1 = " + std::string(ty.mfma) + "(tap ? a1[u] : " + ty.zero + ", bf[s0 + u], acc1, 0, 0, 0);\n        }\n      }\n    }\n";
  if (wide) {
    // D (column fr = output channel, row = pixel of the group) -> the wave's parking rows -> 16-byte pieces of the run
    s += "    if (fr < " + S(FO) + ") {\n";
    s += "      _Pragma(\"unroll\") for (int r = 0; r < 4; ++r) {\n";
    s += "        park[wave][" + ty.drow("r") + " * " + S(FO) + " + fr] = acc0[r];\n";
    s += "        park[wave][(16 + " + ty.drow("r") + ") * " + S(FO) + " + fr] = acc1[r];\n      }\n    }\n";
    s += "    __builtin_amdgcn_fence(__ATOMIC_RELEASE, \"wavefront\");\n    __builtin_amdgcn_wave_barrier();\n    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, \"wavefront\");\n";
    const std::string V = ty.f64 ? "d2v" : "f4";
    s += "    const long e0 = (long)g * " + S(16 * FO) + ";   // first element of the two groups in the run\n";
    s += "    for (int e = lane * " + S(VE) + "; e < " + S(32 * FO) + "; e += " + S(64 * VE) + ") {\n";
    s += "      if (e0 + e >= " + S(b.PT * FO) + " || e0 + e >= left) continue;   // (a piece never straddles the end: FO is a multiple of the piece)\n";
    s += "      " + V + " v = *reinterpret_cast<const " + V + "*>(&park[wave][e]);\n";
    s += "      " + V + "* o = reinterpret_cast<" + V + "*>(run + e0 + e);\n";
    s += "      if (accumulate) v = *o + v;\n      *o = v;\n    }\n";
    s += "    __builtin_amdgcn_wave_barrier();\n";
  } else {
    s += "    if (fr < " + S(FO) + ") {\n";
    s += "      _Pragma(\"unroll\") for (int h = 0; h < 2; ++h)\n";
    s += "        _Pragma(\"unroll\") for (int r = 0; r < 4; ++r) {\n";
    s += "          const int q = (g + h) * 16 + " + ty.drow("r") + ";\n";
    s += "          if (q >= " + S(b.PT) + " || pix0 + q >= lim) continue;\n";
    s += "          " + std::string(ty.T) + "* o = run + (long)q * " + S(FO) + " + fr;\n";
    s += "          const " + std::string(ty.T) + " v = h ? acc1[r] : acc0[r];\n";
    s += "          *o = accumulate ? *o + v : v;\n        }\n    }\n";
  }
  s += "  }\n}\n";
  eg_kernel* k = get_or_build(ctx, name, s);
  if (!k) return EG_ERR_COMPILE;
  long ybands = (Ho + b.R - 1) / b.R;
  const long blocks = ((N + b.NB - 1) / b.NB) * ybands;
  if (blocks > 0x7fffffffL) return EG_OK;
  long Nl = N;
  void* args[] = {(void*)&in, (void*)&flt, (void*)&out, &Nl, &ybands, &accumulate};
  int rc = kernel_launch_raw(k, (unsigned)blocks, 1, 1, 256, args);
  if (rc) return rc;
  *launched = true;
  return EG_OK;
}

int launch_grad_filter(eg_ctx* ctx, const Ty& ty, long N, long H, long W, long C, long F, long FH, long FW, const void* img, const void* gout, void* gflt,
                       int accumulate, bool* launched) {
  *launched = false;
  const long taps = FH * FW * C, Ho = H - FH + 1, Wo = W - FW + 1;
  if (disabled() || C < 1 || F < 1 || F > 16 || C > 16 || taps > 144 || Ho < 1 || Wo < 1 || N * Ho * Wo < 4096) return EG_OK;
  const Band b = plan_band(H, W, C, FH, FW, 0, 0, ty.f64 ? 8 : 4, F);   // (the band's piece of the output gradient is staged too)
  if (!b.ok) return EG_OK;
  const long TB = (taps + 15) / 16, Q = (b.PT + 3) / 4, E = F * taps;
  const long GT = Q * 4 * F;                                            // elements of the staged output-gradient run
  const long GL = (GT + 255) / 256;                                     // ... per thread
  // static LDS: max(band, the fold's 4 x TB accumulator blocks) + the run + the origin table; 64 KB is the limit
  if ((std::max(b.lds_elems, 4 * 16 * TB * 16) + GT) * (ty.f64 ? 8 : 4) + Q * 16 > 60 * 1024 || GL > 64) return EG_OK;
  const std::string name = std::string("eg_conv_band_") + ty.sfx + "_gf_c" + S(C) + "_f" + S(F) + "_" + S(FH) + "x" + S(FW) + "_" + S(H) + "x" + S(W) + "_b" + S(b.NB) + "r" + S(b.R);
  std::string s = prelude();
  s += "extern \"C\" __global__ void __launch_bounds__(256) " + name + "(const " + ty.T + "* __restrict__ img, const " + ty.T + "* __restrict__ gout, " + ty.T +
       "* __restrict__ partial, long N, long ybands, long nbands) {\n";
  s += "  __shared__ " + std::string(ty.T) + " band[" + S(std::max(b.lds_elems, 4 * 16 * TB * 16)) + "];\n";
  s += "  __shared__ int orig[" + S(Q * 4) + "];   // window origin of every output pixel of a band (the same for every band)\n";
  s += "  __shared__ " + std::string(ty.T) + " grun[" + S(GT) + "];   // the band's run of the output gradient, zeros behind its end\n";
  s += "  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fk = lane >> 4;\n";
  s += "  for (int q = tid; q < " + S(Q * 4) + "; q += 256) {\n";
  s += "    const int qq = q < " + S(b.PT) + " ? q : 0, i = qq / " + S(b.R * Wo) + ", r0 = qq % " + S(b.R * Wo) + ", y = r0 / " + S(Wo) + ", x = r0 % " + S(Wo) + ";\n";
  s += "    orig[q] = ((i * " + S(b.RR) + " + y) * " + S(b.WP) + " + x) * " + S(b.STR) + ";\n  }\n";
  // window offsets of this lane's tap columns (tap = 16 t + fr)
  s += "  int toff[" + S(TB) + "];\n";
  s += "  _Pragma(\"unroll\") for (int t = 0; t < " + S(TB) + "; ++t) {\n";
  s += "    const int kk = 16 * t + fr, dy = kk / " + S(FW * C) + ", dx = (kk / " + S(C) + ") % " + S(FW) + ", c = kk % " + S(C) + ";\n";
  s += "    toff[t] = kk < " + S(taps) + " ? (dy * " + S(b.WP) + " + dx) * " + S(b.STR) + " + c : 0;\n  }\n";
  s += "  " + std::string(ty.acc) + " acc[" + S(TB) + "];\n";
  s += "  _Pragma(\"unroll\") for (int t = 0; t < " + S(TB) + "; ++t) acc[t] = " + ty.acc + "{0, 0, 0, 0};\n";
  // a block walks several bands and keeps its sums in registers: fewer partial rows for the second pass, one fold per block
  s += "  for (long bi = blockIdx.x; bi < nbands; bi += gridDim.x) {\n";
  s += "    const long n0 = (bi / ybands) * " + S(b.NB) + ", y0 = (bi % ybands) * " + S(b.R) + ";\n";
  s += "    __syncthreads();   // the previous band's readers are done\n";
  {
    std::string st = stage_code(ty, b, H, W, C, 0, 0, "img");
    // (indent by one level: inside the band loop)
    size_t pos = 0;
    while ((pos = st.find("\n  ", pos)) != std::string::npos) { st.insert(pos + 1, "  "); pos += 3; }
    s += "  " + st;
  }
  // the band's output pixels are one contiguous run of the output gradient (whole rows of one image, or whole images)
  s += "    const long pix0 = (n0 * " + S(Ho) + " + y0) * " + S(Wo) + ", lim = (n0 + " + S(b.NB) + " < N ? n0 + " + S(b.NB) + " : N) * " + S(Ho * Wo) + ";\n";
  s += "    const int live_px = lim - pix0 < " + S(b.PT) + " ? (int)(lim - pix0 > 0 ? lim - pix0 : 0) : " + S(b.PT) + ";   // pixels of the band that exist\n";
  // the run of the output gradient: every thread's loads are issued before the first is stored (one load in flight per wave
  // and iteration made the loop wait for memory: 51 us with one image per block, 102 with seven)
  s += "    {\n      const " + std::string(ty.T) + "* gp = gout + pix0 * " + S(F) + ";\n      const long glive = (long)live_px * " + S(F) + ";\n";
  s += "      " + std::string(ty.T) + " gv[" + S(GL) + "];\n";
  s += "      _Pragma(\"unroll\") for (int u = 0; u < " + S(GL) + "; ++u) { const int e = tid + 256 * u; gv[u] = e < glive ? gp[e] : " + ty.zero + "; }\n";
  s += "      _Pragma(\"unroll\") for (int u = 0; u < " + S(GL) + "; ++u) { const int e = tid + 256 * u; if (e < " + S(GT) + ") grun[e] = gv[u]; }\n    }\n";
  s += "    __syncthreads();\n";
  // four quads per trip: their origins, then their window values and gradient values, then their matrix instructions — an LDS
  // round trip per STEP of the chain origin -> window value -> multiply instead of per quad (one quad per trip left two waits
  // of ~100 cycles in front of every pair of instructions: 50 us on the 28 x 28 x 1 -> 8 layer at batch 4096)
  s += "    for (int qq = wave; qq < " + S(Q) + "; qq += 16) {\n";
  s += "      int o[4];\n      " + std::string(ty.T) + " a[4], bv[4][" + S(TB) + "];\n";
  s += "      _Pragma(\"unroll\") for (int u = 0; u < 4; ++u) o[u] = qq + 4 * u < " + S(Q) + " ? orig[4 * (qq + 4 * u) + fk] : 0;\n";
  s += "      _Pragma(\"unroll\") for (int u = 0; u < 4; ++u) {\n";
  s += "        const int q = 4 * (qq + 4 * u) + fk;\n";
  s += "        a[u] = (qq + 4 * u < " + S(Q) + " && fr < " + S(F) + ") ? grun[q * " + S(F) + " + fr] : " + ty.zero + ";   // (zeros behind the band's last pixel)\n";
  s += "        _Pragma(\"unroll\") for (int t = 0; t < " + S(TB) + "; ++t) bv[u][t] = band[o[u] + toff[t]];\n      }\n";
  s += "      _Pragma(\"unroll\") for (int u = 0; u < 4; ++u) {\n";
  s += "        const bool live = qq + 4 * u < " + S(Q) + " && 4 * (qq + 4 * u) + fk < live_px;\n";
  s += "        _Pragma(\"unroll\") for (int t = 0; t < " + S(TB) + "; ++t) acc[t] = " + std::string(ty.mfma) + "(a[u], live ? bv[u][t] : " + ty.zero + ", acc[t], 0, 0, 0);\n      }\n    }\n  }\n";
  // fold the four waves in wave order: [wave][t][row 16][col 16]
  s += "  __syncthreads();\n";
  s += "  _Pragma(\"unroll\") for (int t = 0; t < " + S(TB) + "; ++t)\n";
  s += "    _Pragma(\"unroll\") for (int r = 0; r < 4; ++r) band[((wave * " + S(TB) + " + t) * 16 + " + ty.drow("r") + ") * 16 + fr] = acc[t][r];\n";
  s += "  __syncthreads();\n";
  s += "  for (int e = tid; e < " + S(E) + "; e += 256) {\n";
  s += "    const int f = e / " + S(taps) + ", kk = e % " + S(taps) + ", t = kk / 16, col = kk % 16;\n";
  s += "    const int at = (t * 16 + f) * 16 + col;\n";
  s += "    partial[(long)blockIdx.x * " + S(E) + " + e] = ((band[at] + band[" + S(TB * 256) + " + at]) + band[" + S(2 * TB * 256) + " + at]) + band[" + S(3 * TB * 256) + " + at];\n  }\n}\n";
  eg_kernel* k = get_or_build(ctx, name, s);
  if (!k) return EG_ERR_COMPILE;
  long ybands = (Ho + b.R - 1) / b.R;
  long nbands = ((N + b.NB - 1) / b.NB) * ybands;
  long blocks = 8L * ctx->compute_units;   // eight blocks (32 waves) per CU, each walking nbands / blocks bands: the loop is latency-bound, 1024 / 2048 / 4096 blocks = 81 / 51 / 52 us on the 28 x 28 x 1 -> 8 layer at batch 4096
  if (blocks > nbands) blocks = nbands;
  const size_t esz = ty.f64 ? 8 : 4;
  const size_t pelems = ((size_t)blocks * E + 3) & ~(size_t)3;
  const size_t selems = (size_t)colsum_scratch_floats(ctx, blocks, E);
  int rc = ensure_workspace(ctx, (pelems + selems) * esz);
  if (rc) return rc;
  void* partial = ctx->workspace;
  long Nl = N;
  void* args[] = {(void*)&img, (void*)&gout, (void*)&partial, &Nl, &ybands, &nbands};
  rc = kernel_launch_raw(k, (unsigned)blocks, 1, 1, 256, args);
  if (rc) return rc;
  if (!ty.f64 && slab_sum_supported(E, static_cast<const float*>(partial), static_cast<float*>(gflt))) {   // one launch, fixed order
    rc = slab_sum(ctx, blocks, E, static_cast<const float*>(partial), static_cast<float*>(gflt), accumulate);
    if (rc) return rc;
    *launched = true;
    return EG_OK;
  }
  if (ty.f64)
    rc = colsum_f64_with_scratch(ctx, blocks, E, static_cast<const double*>(partial), static_cast<double*>(gflt), accumulate,
                                 static_cast<double*>(partial) + pelems);
  else
    rc = colsum_with_scratch(ctx, blocks, E, static_cast<const float*>(partial), static_cast<float*>(gflt), accumulate, static_cast<float*>(partial) + pelems);
  if (rc) return rc;
  *launched = true;
  return EG_OK;
}

}  // namespace

int conv2_band_forward_try(eg_ctx* ctx, bool f64, long N, long H, long W, long C, long F, long FH, long FW, const void* img, const void* flt, void* out,
                           int accumulate, bool* launched) {
  return launch_forward(ctx, f64 ? kF64 : kF32, false, N, H, W, C, F, FH, FW, 0, 0, img, flt, out, accumulate, launched);
}

// gimg [N, H, W, C] from gout [N, H - FH + 1, W - FW + 1, F] and flt [F, FH, FW, C]
int conv2_band_grad_image_try(eg_ctx* ctx, bool f64, long N, long H, long W, long C, long F, long FH, long FW, const void* flt, const void* gout, void* gimg,
                              int accumulate, bool* launched) {
  return launch_forward(ctx, f64 ? kF64 : kF32, true, N, H - FH + 1, W - FW + 1, F, C, FH, FW, FH - 1, FW - 1, gout, flt, gimg, accumulate, launched);
}

int conv2_band_grad_filter_try(eg_ctx* ctx, bool f64, long N, long H, long W, long C, long F, long FH, long FW, const void* img, const void* gout, void* gflt,
                               int accumulate, bool* launched) {
  return launch_grad_filter(ctx, f64 ? kF64 : kF32, N, H, W, C, F, FH, FW, img, gout, gflt, accumulate, launched);
}

}  // namespace eg
