/*
 * refcpu.c — ORACLE (test infrastructure, not product code).
 *
 * A CPU restatement, in plain C, of the loop nests exprgrad's LLVM CPU back-end executes for
 * the compiled-tensor hot path.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product (exprgrad_amd/) never does.
 *
 * Why a restatement: the reference is Nim + LLVM 13 (exprgrad.nimble:6,
 * exprgrad/wrappers/llvm.nim:19-20); neither toolchain exists in the build image, so the
 * reference cannot be compiled here ("unbuildable").  This file is pinned instead against the
 * known-answer vectors of the reference's own tests (tests/test_oracle_known_answers.py,
 * transcribed from tests/test_model.nim, tests/test_talks.nim, tests/test_tensors.nim,
 * tests/test_gpu.nim).
 *
 * What is restated, and from where (paths relative to the exprgrad repository):
 *   - data layout: dense row-major, last dimension contiguous        tensors.nim:20-25, 111-134
 *   - every kernel ACCUMULATES into a zero-initialised result        model.nim:295-300; passes.nim:888-897
 *   - scalar semantics: separate f32 fmul / fadd (no contraction, no fast-math), ordered
 *     compares, select, libm expf/logf/sinf/cosf/powf/sqrtf         llvmgen.nim:212-301; llvm.nim:486-491
 *   - loop order after reorderLoops (reads weigh 10, writes 1)       passes.nim:700-745
 *       matmul  c[y,x] += a[y,it]*b[it,x]           : y, it, x
 *       conv2   out[n,y,x,f] += img[..]*flt[..]     : n, y, f, dy, x, dx, c
 *     => every output element is summed sequentially in increasing reduction index.
 *   - threading: only the outermost independent loop is split, into contiguous chunks
 *                                                                    passes.nim:2415-2437; model.nim:110-132
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (oracle/Makefile).  The flags matter:
 * contraction or reassociation would change the summation the reference performs.
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define EXPORT __attribute__((visibility("default")))

/* ------------------------------------------------------------------ contraction ---------- */

typedef struct {
  int trans_a, trans_b;
  int64_t M, N, K;
  const float* A;
  int64_t lda;
  const float* B;
  int64_t ldb;
  float* C;
  int64_t ldc;
  int64_t lo, hi; /* range of the split loop */
} gemm_job;

/* c[y,x] += a[y,it] * b[it,x]      base.nim:27-28, order y, it, x (x innermost, unit stride). */
static void gemm_nn_rows(const gemm_job* j) {
  for (int64_t y = j->lo; y < j->hi; ++y)
    for (int64_t it = 0; it < j->K; ++it) {
      const float a = j->A[y * j->lda + it];
      const float* b = j->B + it * j->ldb;
      float* c = j->C + y * j->ldc;
      for (int64_t x = 0; x < j->N; ++x) c[x] += a * b[x];
    }
}

/* gradA[y,it] += g[y,x] * b[it,x]  derived from the kernel above for read `a` (passes.nim:519-549);
 * loops inherited from the forward kernel, order y, it, x; x is the reduction.
 * Here: A = g [M,K=x], B = b stored [N=it, K=x]  (trans_b). */
static void gemm_nt_rows(const gemm_job* j) {
  for (int64_t y = j->lo; y < j->hi; ++y)
    for (int64_t it = 0; it < j->N; ++it) {
      float* c = j->C + y * j->ldc + it;
      const float* g = j->A + y * j->lda;
      const float* b = j->B + it * j->ldb;
      for (int64_t x = 0; x < j->K; ++x) *c += g[x] * b[x];
    }
}

/* gradB[it,x] += a[y,it] * g[y,x]  derived for read `b`; order y, it, x; y (outermost) is the
 * reduction, `it` is the independent loop the Threads target would split (Appendix A.4).
 * Here: M = it, N = x, K = y; A stored [K=y, M=it] (trans_a), B = g [K=y, N=x]. */
static void gemm_tn_cols(const gemm_job* j) {
  for (int64_t y = 0; y < j->K; ++y)
    for (int64_t it = j->lo; it < j->hi; ++it) {
      const float a = j->A[y * j->lda + it];
      const float* g = j->B + y * j->ldb;
      float* c = j->C + it * j->ldc;
      for (int64_t x = 0; x < j->N; ++x) c[x] += a * g[x];
    }
}

/* c[y,x] += a[it,y] * b[x,it]  (both transposed; not produced by the layer library, kept for
 * completeness of the operand-layout space): per output, increasing `it`. */
static void gemm_tt_rows(const gemm_job* j) {
  for (int64_t y = j->lo; y < j->hi; ++y)
    for (int64_t x = 0; x < j->N; ++x) {
      float* c = j->C + y * j->ldc + x;
      for (int64_t it = 0; it < j->K; ++it) *c += j->A[it * j->lda + y] * j->B[x * j->ldb + it];
    }
}

static void* gemm_thread(void* p) {
  const gemm_job* j = (const gemm_job*)p;
  if (!j->trans_a && !j->trans_b)
    gemm_nn_rows(j);
  else if (!j->trans_a && j->trans_b)
    gemm_nt_rows(j);
  else if (j->trans_a && !j->trans_b)
    gemm_tn_cols(j);
  else
    gemm_tt_rows(j);
  return NULL;
}

/* C (+)= op(A) op(B).  C must hold the values to accumulate onto (zeros for a fresh result:
 * model.nim:295-300).  threads <= 1 runs inline like builtinRunThreads does when only one
 * chunk qualifies (model.nim:118-121); otherwise the independent outer loop is cut into
 * `threads` contiguous chunks, the first (size mod threads) chunks one longer (model.nim:123-131). */
EXPORT void ref_sgemm(int trans_a, int trans_b, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda,
                      const float* B, int64_t ldb, float* C, int64_t ldc, int threads) {
  gemm_job base = {trans_a, trans_b, M, N, K, A, lda, B, ldb, C, ldc, 0, M};
  if (threads <= 1 || M < 2) {
    gemm_thread(&base);
    return;
  }
  if (threads > M) threads = (int)M;
  pthread_t* tids = (pthread_t*)malloc(sizeof(pthread_t) * threads);
  gemm_job* jobs = (gemm_job*)malloc(sizeof(gemm_job) * threads);
  int64_t offset = 0;
  for (int t = 0; t < threads; ++t) {
    int64_t size = M / threads + (t < M % threads ? 1 : 0);
    jobs[t] = base;
    jobs[t].lo = offset;
    jobs[t].hi = offset + size;
    offset += size;
    pthread_create(&tids[t], NULL, gemm_thread, &jobs[t]);
  }
  for (int t = 0; t < threads; ++t) pthread_join(tids[t], NULL);
  free(tids);
  free(jobs);
}

/* compile[float64] (model.nim:253-260): the same four loop nests over double — C (+)= op(A) op(B), every output element
 * summed sequentially in the reference's loop order (y, it, x; passes.nim:700-745), no FMA contraction. */
EXPORT void ref_dgemm(int trans_a, int trans_b, int64_t M, int64_t N, int64_t K, const double* A, int64_t lda,
                      const double* B, int64_t ldb, double* C, int64_t ldc) {
  if (!trans_a && !trans_b) {          /* c[y,x] += a[y,it] * b[it,x]: y, it, x */
    for (int64_t y = 0; y < M; ++y)
      for (int64_t it = 0; it < K; ++it) {
        const double a = A[y * lda + it];
        for (int64_t x = 0; x < N; ++x) C[y * ldc + x] += a * B[it * ldb + x];
      }
  } else if (!trans_a && trans_b) {    /* gradA[y,it] += g[y,x] * b[it,x]: y, it, x (x the reduction) */
    for (int64_t y = 0; y < M; ++y)
      for (int64_t it = 0; it < N; ++it)
        for (int64_t x = 0; x < K; ++x) C[y * ldc + it] += A[y * lda + x] * B[it * ldb + x];
  } else if (trans_a && !trans_b) {    /* gradB[it,x] += a[y,it] * g[y,x]: y (the reduction) outermost */
    for (int64_t y = 0; y < K; ++y)
      for (int64_t it = 0; it < M; ++it) {
        const double a = A[y * lda + it];
        for (int64_t x = 0; x < N; ++x) C[it * ldc + x] += a * B[y * ldb + x];
      }
  } else {
    for (int64_t y = 0; y < M; ++y)
      for (int64_t x = 0; x < N; ++x)
        for (int64_t it = 0; it < K; ++it) C[y * ldc + x] += A[it * lda + y] * B[x * ldb + it];
  }
}

/* The thread count the reference would use for a loop of `size` iterations whose body costs
 * `work_per_iter` units: minSize = 2^24 / work; threads = clamp(size / minSize, 1, pool)
 * (passes.nim:2415-2437 MIN_WORK_PER_THREAD; model.nim:116-119). */
EXPORT int ref_thread_count(int64_t size, int64_t work_per_iter, int pool) {
  const int64_t min_work = (int64_t)1 << 24;
  if (work_per_iter < 1) work_per_iter = 1;
  int64_t min_size = min_work / work_per_iter;
  int64_t t = pool;
  if (min_size > 0) {
    t = size / min_size;
    if (t > pool) t = pool;
    if (t < 1) t = 1;
  }
  return (int)t;
}

/* ------------------------------------------------------------------ dense pieces --------- */

/* out[y,x] += bias[x]    dnn.nim:22-24, order y, x. */
EXPORT void ref_bias_add(int64_t rows, int64_t cols, const float* bias, float* out) {
  for (int64_t y = 0; y < rows; ++y)
    for (int64_t x = 0; x < cols; ++x) out[y * cols + x] += bias[x];
}

/* gb[x] += g[y,x]        derived from the bias kernel; order y, x (y outermost = reduction). */
EXPORT void ref_colsum(int64_t rows, int64_t cols, const float* in, float* out) {
  for (int64_t y = 0; y < rows; ++y)
    for (int64_t x = 0; x < cols; ++x) out[x] += in[y * cols + x];
}

/* sums[y] += in[y,x]     the shape of softmax.sums (dnn.nim:90-92) without the exp. */
EXPORT void ref_rowsum(int64_t rows, int64_t cols, const float* in, float* out) {
  for (int64_t y = 0; y < rows; ++y)
    for (int64_t x = 0; x < cols; ++x) out[y] += in[y * cols + x];
}

/* s[0] += in[i]          scalar-loss shape (base.nim:57-67). */
EXPORT void ref_sum(int64_t n, const float* in, float* out) {
  for (int64_t i = 0; i < n; ++i) out[0] += in[i];
}

/* param{it} += -grad{it} * rate   base.nim:37-38.  The literal is a float64 in the source and is
 * emitted as an f32 constant for compile[float32] (llvmgen.nim:212-216): alpha = (float)(rate),
 * value = (-g) * alpha.  xor_from_scratch.nim:30-31 writes it as (-0.1) * g: same product. */
EXPORT void ref_gradient_descent(int64_t n, float rate, const float* grad, float* param) {
  for (int64_t i = 0; i < n; ++i) param[i] += (-grad[i]) * rate;
}
/* y{i} += alpha * x{i} */
EXPORT void ref_axpy(int64_t n, float alpha, const float* x, float* y) {
  for (int64_t i = 0; i < n; ++i) y[i] += alpha * x[i];
}

/* ------------------------------------------------------------------ maps ----------------- */
/* Same numbering as enum eg_map_op in include/exprgrad_hip.h. */
enum { MAP_IDENTITY, MAP_RELU, MAP_LEAKY_RELU, MAP_SIGMOID, MAP_TANH, MAP_SCALE, MAP_SIN, MAP_XOR_LEAKY, MAP_EXP };

static float map_fwd(int op, float x, float p) {
  switch (op) {
    case MAP_IDENTITY: return x;
    case MAP_RELU: return (0.0f <= x) ? x : 0.0f; /* dnn.nim:26-27; a >= b is b <= a (dsl.nim:45-46) */
    case MAP_LEAKY_RELU: return ((0.0f <= x) ? 1.0f : p) * x; /* dnn.nim:29-30 */
    case MAP_SIGMOID: {
      const float e = expf(-x);
      return 1.0f / (1.0f + e); /* dnn.nim:32-33 */
    }
    case MAP_TANH: {
      const float a = expf(x), b = expf(-x);
      return (a - b) / (a + b); /* dnn.nim:35-40 */
    }
    case MAP_SCALE: return x * p; /* base.nim:24 */
    case MAP_SIN: return sinf(x); /* dnn.nim:42-43 */
    case MAP_XOR_LEAKY: return (x <= 0.0f) ? p * x : x; /* xor_from_scratch.nim:22 */
    case MAP_EXP: return expf(x);
  }
  return x;
}

/* out{i} += f(in{i}) */
EXPORT void ref_map(int op, int64_t n, const float* in, float* out, float p) {
  for (int64_t i = 0; i < n; ++i) out[i] += map_fwd(op, in[i], p);
}

/* gin{i} += dOut/dIn * gout{i}, written as the instruction sequence `derive` emits
 * (passes.nim:383-517): instructions visited last to first; a register reached twice sums its
 * contributions with `+` in visit order (510-517). */
static float map_bwd(int op, float x, float g, float p) {
  switch (op) {
    case MAP_IDENTITY: return g;
    case MAP_RELU: return (0.0f <= x) ? g : 0.0f; /* select rule 471-476 */
    case MAP_LEAKY_RELU: {
      const float s = (0.0f <= x) ? 1.0f : p;
      return g * s; /* mul rule 399-403: grad(arg1) = g * arg0 */
    }
    case MAP_SIGMOID: {
      const float e = expf(-x);
      const float s = 1.0f + e;
      const float gs = (-1.0f) * (g / (s * s)); /* div rule 404-415 with numerator literal 1 */
      const float gn = gs * e;                  /* exp rule 456-459 */
      return -gn;                               /* negate rule 416-419 */
    }
    case MAP_TANH: {
      const float a = expf(x), b = expf(-x);
      const float d = a - b, t = a + b;
      const float gd = g / t;
      const float gt = (-d) * (g / (t * t));
      float ga = gt, gb = gt; /* add (visited first: it is the later instruction) */
      ga = ga + gd;           /* sub: (g, -g) */
      gb = gb + (-gd);
      const float gn = gb * b;
      return (-gn) + ga * a;
    }
    case MAP_SCALE: return g * p;
    case MAP_SIN: return cosf(x) * g; /* 460-464 */
    case MAP_XOR_LEAKY: {
      const int c = x <= 0.0f;
      const float g1 = c ? g : 0.0f, g2 = c ? 0.0f : g;
      return g2 + g1 * p;
    }
    case MAP_EXP: return g * expf(x);
  }
  return g;
}

EXPORT void ref_map_grad(int op, int64_t n, const float* in, const float* gout, float* gin, float p) {
  for (int64_t i = 0; i < n; ++i) gin[i] += map_bwd(op, in[i], gout[i], p);
}

/* ------------------------------------------------------------------ conv2 ---------------- */
/* out[n,y,x,f] += img[n,y+dy,x+dx,c] * flt[f,dy,dx,c]   dnn.nim:45-49; valid, stride 1.
 * Loop order n, y, f, dy, x, dx, c (passes.nim:700-745; identical to conv2_naive,
 * benchmarks/conv2/conv2.nim:49-55).  Only n is split across threads (Appendix A.4). */
typedef struct {
  int64_t N, H, W, C, F, FH, FW;
  const float* img;
  const float* flt;
  float* out;
  int64_t lo, hi;   /* image range (reference policy) */
  int64_t ylo, yhi; /* row range (non-reference all-core variant) */
} conv_job;

static void* conv_thread(void* p) {
  const conv_job* j = (const conv_job*)p;
  const int64_t Ho = j->H - j->FH + 1, Wo = j->W - j->FW + 1;
  for (int64_t n = j->lo; n < j->hi; ++n)
    for (int64_t y = j->ylo; y < j->yhi; ++y)
      for (int64_t f = 0; f < j->F; ++f)
        for (int64_t dy = 0; dy < j->FH; ++dy)
          for (int64_t x = 0; x < Wo; ++x)
            for (int64_t dx = 0; dx < j->FW; ++dx) {
              const float* ip = j->img + ((n * j->H + y + dy) * j->W + x + dx) * j->C;
              const float* fp = j->flt + ((f * j->FH + dy) * j->FW + dx) * j->C;
              float* op = j->out + ((n * Ho + y) * Wo + x) * j->F + f;
              for (int64_t c = 0; c < j->C; ++c) *op += ip[c] * fp[c];
            }
  return NULL;
}

/* threads_n: threads over images (the reference's policy).  threads_y > 1 additionally splits
 * rows — NOT what the reference does; used only for an "all host cores" baseline figure. */
EXPORT void ref_conv2_nhwc(int64_t N, int64_t H, int64_t W, int64_t C, int64_t F, int64_t FH, int64_t FW,
                           const float* img, const float* flt, float* out, int threads_n, int threads_y) {
  const int64_t Ho = H - FH + 1;
  if (threads_n < 1) threads_n = 1;
  if (threads_y < 1) threads_y = 1;
  if (threads_n > N) threads_n = (int)(N > 0 ? N : 1);
  if (threads_y > Ho) threads_y = (int)(Ho > 0 ? Ho : 1);
  const int total = threads_n * threads_y;
  conv_job* jobs = (conv_job*)malloc(sizeof(conv_job) * total);
  pthread_t* tids = (pthread_t*)malloc(sizeof(pthread_t) * total);
  int64_t noff = 0;
  int k = 0;
  for (int tn = 0; tn < threads_n; ++tn) {
    int64_t nsize = N / threads_n + (tn < N % threads_n ? 1 : 0);
    int64_t yoff = 0;
    for (int ty = 0; ty < threads_y; ++ty) {
      int64_t ysize = Ho / threads_y + (ty < Ho % threads_y ? 1 : 0);
      conv_job j = {N, H, W, C, F, FH, FW, img, flt, out, noff, noff + nsize, yoff, yoff + ysize};
      jobs[k++] = j;
      yoff += ysize;
    }
    noff += nsize;
  }
  if (total == 1) {
    conv_thread(&jobs[0]);
  } else {
    for (int t = 0; t < total; ++t) pthread_create(&tids[t], NULL, conv_thread, &jobs[t]);
    for (int t = 0; t < total; ++t) pthread_join(tids[t], NULL);
  }
  free(jobs);
  free(tids);
}

/* ------------------------------------------------------------------ f64 shadow ----------- */
/* Error budgeting only: the same contraction accumulated in double. */
EXPORT void ref_dgemm_from_f32(int trans_a, int trans_b, int64_t M, int64_t N, int64_t K, const float* A,
                               int64_t lda, const float* B, int64_t ldb, double* C, int64_t ldc) {
  for (int64_t m = 0; m < M; ++m)
    for (int64_t n = 0; n < N; ++n) {
      double s = C[m * ldc + n];
      for (int64_t k = 0; k < K; ++k) {
        const double a = trans_a ? A[k * lda + m] : A[m * lda + k];
        const double b = trans_b ? B[n * ldb + k] : B[k * ldb + n];
        s += a * b;
      }
      C[m * ldc + n] = s;
    }
}

/* The two kernels derive (passes.nim:383-549) produces from conv2 (dnn.nim:45-49), as plain loop
 * nests with the iterators of the forward kernel (n, y, x, f, dy, dx, c), separate float multiply
 * and add, accumulating into the destination like every `++=` kernel:
 *   gflt[f,dy,dx,c]     += gout[n,y,x,f] * img[n,y+dy,x+dx,c]
 *   gimg[n,y+dy,x+dx,c] += gout[n,y,x,f] * flt[f,dy,dx,c]
 * Single threaded (scatter / reduction over every loop): test sizes only. */
EXPORT void ref_conv2_nhwc_grad_filter(int64_t N, int64_t H, int64_t W, int64_t C, int64_t F, int64_t FH, int64_t FW,
                                       const float* img, const float* gout, float* gflt) {
  const int64_t Ho = H - FH + 1, Wo = W - FW + 1;
  for (int64_t n = 0; n < N; ++n)
    for (int64_t y = 0; y < Ho; ++y)
      for (int64_t x = 0; x < Wo; ++x)
        for (int64_t f = 0; f < F; ++f) {
          const float g = gout[((n * Ho + y) * Wo + x) * F + f];
          for (int64_t dy = 0; dy < FH; ++dy)
            for (int64_t dx = 0; dx < FW; ++dx) {
              const float* ip = img + ((n * H + y + dy) * W + x + dx) * C;
              float* gp = gflt + ((f * FH + dy) * FW + dx) * C;
              for (int64_t c = 0; c < C; ++c) gp[c] += g * ip[c];
            }
        }
}

EXPORT void ref_conv2_nhwc_grad_image(int64_t N, int64_t H, int64_t W, int64_t C, int64_t F, int64_t FH, int64_t FW,
                                      const float* flt, const float* gout, float* gimg) {
  const int64_t Ho = H - FH + 1, Wo = W - FW + 1;
  for (int64_t n = 0; n < N; ++n)
    for (int64_t y = 0; y < Ho; ++y)
      for (int64_t x = 0; x < Wo; ++x)
        for (int64_t f = 0; f < F; ++f) {
          const float g = gout[((n * Ho + y) * Wo + x) * F + f];
          for (int64_t dy = 0; dy < FH; ++dy)
            for (int64_t dx = 0; dx < FW; ++dx) {
              const float* fp = flt + ((f * FH + dy) * FW + dx) * C;
              float* gp = gimg + ((n * H + y + dy) * W + x + dx) * C;
              for (int64_t c = 0; c < C; ++c) gp[c] += g * fp[c];
            }
        }
}
