/*
 * refinterp.c — ORACLE (test infrastructure, not product code).
 *
 * Executes ONE lowered kernel of exprgrad's CPU path the way the LLVM back-end's loop nest
 * does: nested loops in the order chosen by reorderLoops (passes.nim:700-745; the caller passes
 * the loops already in that order), and in the innermost body
 *     load every read (llvmgen.nim:277-301, row-major linear index, passes.nim:787-843)
 *     evaluate the expression instruction by instruction in f32 / i64 / i1 (llvmgen.nim:212-276)
 *     out[index] = out[index] + value      (InstrWrite = load, fadd, store: llvmgen.nim:287-297)
 * so every output element accumulates its terms sequentially in loop order, in f32, with no
 * fused multiply-add (built with -ffp-contract=off) — the reference's arithmetic.
 *
 * The graph-level work (autodiff, dead-kernel elimination, loop order, shapes) is restated in
 * oracle/kd.py; this file is only the arithmetic.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#define EXPORT __attribute__((visibility("default")))
#define MAX_LOOPS 12
#define MAX_REGS 4096

typedef union {
  float f;
  int64_t i;
} reg_t;

enum {
  OP_SCALAR_LIT = 0, OP_INDEX_LIT = 1, OP_BOOL_LIT = 2,
  OP_ADD_F = 10, OP_SUB_F = 11, OP_MUL_F = 12, OP_DIV_F = 13, OP_NEG_F = 14,
  OP_ADD_I = 20, OP_SUB_I = 21, OP_MUL_I = 22, OP_IDIV = 23, OP_MOD = 24, OP_WRAP = 25, OP_NEG_I = 26,
  OP_SIN = 30, OP_COS = 31, OP_EXP = 32, OP_POW = 33, OP_SQRT = 34, OP_LOG = 35, OP_LOG10 = 36, OP_LOG2 = 37,
  OP_LN = 38,
  OP_EQ_F = 40, OP_LT_F = 41, OP_LE_F = 42, OP_EQ_I = 43, OP_LT_I = 44, OP_LE_I = 45, OP_AND = 46, OP_OR = 47,
  OP_SELECT = 50,
  OP_TOSCALAR = 60, OP_TOINDEX = 61
};

/* One pass over an instruction list (f32 / i64 / i1 arithmetic of llvmgen.nim:212-276). */
static inline int exec_instrs(reg_t* regs, int ninstr, const int32_t* instr, const double* instr_lit) {
  for (int k = 0; k < ninstr; ++k) {
    const int32_t* in = instr + 5 * k;
    reg_t* d = &regs[in[1]];
    const reg_t a = regs[in[2]], b = regs[in[3]], c = regs[in[4]];
    switch (in[0]) {
      case OP_SCALAR_LIT: d->f = (float)instr_lit[k]; break; /* const_real(float, double) llvmgen.nim:215-216 */
      case OP_INDEX_LIT: d->i = (int64_t)instr_lit[k]; break;
      case OP_BOOL_LIT: d->i = instr_lit[k] != 0.0; break;
      case OP_ADD_F: d->f = a.f + b.f; break;
      case OP_SUB_F: d->f = a.f - b.f; break;
      case OP_MUL_F: d->f = a.f * b.f; break;
      case OP_DIV_F: d->f = a.f / b.f; break;
      case OP_NEG_F: d->f = -a.f; break;
      case OP_ADD_I: d->i = a.i + b.i; break;
      case OP_SUB_I: d->i = a.i - b.i; break;
      case OP_MUL_I: d->i = a.i * b.i; break;
      case OP_IDIV: d->i = a.i / b.i; break;               /* sdiv */
      case OP_MOD: d->i = a.i % b.i; break;                /* srem */
      case OP_WRAP: d->i = ((a.i % b.i) + b.i) % b.i; break; /* llvmgen.nim:227-230 */
      case OP_NEG_I: d->i = -a.i; break;
      case OP_SIN: d->f = sinf(a.f); break;
      case OP_COS: d->f = cosf(a.f); break;
      case OP_EXP: d->f = expf(a.f); break;
      case OP_POW: d->f = powf(a.f, b.f); break;
      case OP_SQRT: d->f = sqrtf(a.f); break;
      case OP_LOG: d->f = logf(a.f) / logf(b.f); break;   /* no CPU lowering in the reference; see kd.py */
      case OP_LOG10: d->f = log10f(a.f); break;
      case OP_LOG2: d->f = log2f(a.f); break;
      case OP_LN: d->f = logf(a.f); break;
      case OP_EQ_F: d->i = a.f == b.f; break;              /* ordered compares: false on NaN */
      case OP_LT_F: d->i = a.f < b.f; break;
      case OP_LE_F: d->i = a.f <= b.f; break;
      case OP_EQ_I: d->i = a.i == b.i; break;
      case OP_LT_I: d->i = a.i < b.i; break;
      case OP_LE_I: d->i = a.i <= b.i; break;
      case OP_AND: d->i = a.i & b.i; break;
      case OP_OR: d->i = a.i | b.i; break;
      case OP_SELECT: *d = a.i ? b : c; break;
      case OP_TOSCALAR: d->f = (float)a.i; break;          /* sitofp */
      case OP_TOINDEX: d->i = (int64_t)a.f; break;         /* fptosi */
      default: return -2;
    }
    }
  return 0;
}

/*
 * loops:      nloops entries outer -> inner: start, stop (step 1), register holding the iterator
 * reads:      nreads tensors; flat index of read r = read_affine[r*(1+nloops)] +
 *             sum_l read_affine[r*(1+nloops)+1+l] * iter_l ; value goes to register read_reg[r]
 * instrs:     ninstr x 5 int32: opcode, res, a0, a1, a2 ; literal i in instr_lit[i]
 * write:      flat index by write_affine (same layout), value = regs[result_reg]
 * overwrite:  0 -> out[idx] += value (InstrWrite) ; 1 -> out[idx] = value (InstrOverwrite)
 */
EXPORT int ref_interp_kernel(int nloops, const int64_t* loop_start, const int64_t* loop_stop, const int32_t* loop_reg,
                             int nregs, int nreads, const float* const* read_ptr, const int32_t* read_reg,
                             const int64_t* read_affine, int ninstr, const int32_t* instr, const double* instr_lit,
                             int result_reg, float* write_ptr, const int64_t* write_affine, int overwrite) {
  if (nloops > MAX_LOOPS || nregs >= MAX_REGS) return -1;
  reg_t regs[MAX_REGS];
  memset(regs, 0, sizeof(regs));
  int64_t it[MAX_LOOPS];
  for (int l = 0; l < nloops; ++l) {
    if (loop_stop[l] <= loop_start[l]) return 0; /* empty iteration space */
    it[l] = loop_start[l];
  }
  const int stride = 1 + nloops;
  for (;;) {
    for (int l = 0; l < nloops; ++l) regs[loop_reg[l]].i = it[l];
    for (int r = 0; r < nreads; ++r) {
      const int64_t* a = read_affine + (int64_t)r * stride;
      int64_t idx = a[0];
      for (int l = 0; l < nloops; ++l) idx += a[1 + l] * it[l];
      regs[read_reg[r]].f = read_ptr[r][idx];
    }
    {
      const int rc = exec_instrs(regs, ninstr, instr, instr_lit);
      if (rc) return rc;
    }
    {
      int64_t idx = write_affine[0];
      for (int l = 0; l < nloops; ++l) idx += write_affine[1 + l] * it[l];
      const float v = regs[result_reg].f;
      if (overwrite)
        write_ptr[idx] = v;
      else
        write_ptr[idx] = write_ptr[idx] + v;
    }
    /* odometer: innermost loop fastest */
    int l = nloops - 1;
    for (; l >= 0; --l) {
      if (++it[l] < loop_stop[l]) break;
      it[l] = loop_start[l];
    }
    if (l < 0) break;
  }
  return 0;
}

/*
 * Kernels with computed (non-affine) indices — LinearIndex.setup of the reference (ir.nim:120-123),
 * e.g. maxpool2's gradient reads `result[image, y div 2, x div 2, chan]` (dnn.nim:59-71).
 * Per innermost iteration: iterators -> registers; the index instructions run (Index typed, may use
 * iterators); every operand's flat index = constant + sum(coefficient * register) over any registers;
 * reads; value instructions; out[idx] += value.
 * ops:        packed per operand (reads, then the write): constant, nterms, nterms x (register, coefficient),
 *             starting at op_offset[operand]
 * op_size:    element count per operand (indices outside [0, size) return -3 instead of touching memory)
 */
EXPORT int ref_interp_kernel2(int nloops, const int64_t* loop_start, const int64_t* loop_stop, const int32_t* loop_reg,
                              int nregs, int nidx, const int32_t* idx_instr, const double* idx_lit, int nreads,
                              const float* const* read_ptr, const int32_t* read_reg, const int64_t* ops,
                              const int32_t* op_offset, int ninstr, const int32_t* instr, const double* instr_lit,
                              int result_reg, float* write_ptr, const int64_t* op_size) {
  if (nloops > MAX_LOOPS || nregs >= MAX_REGS) return -1;
  reg_t regs[MAX_REGS];
  memset(regs, 0, sizeof(regs));
  int64_t it[MAX_LOOPS];
  for (int l = 0; l < nloops; ++l) {
    if (loop_stop[l] <= loop_start[l]) return 0;
    it[l] = loop_start[l];
  }
  for (;;) {
    for (int l = 0; l < nloops; ++l) regs[loop_reg[l]].i = it[l];
    int rc = exec_instrs(regs, nidx, idx_instr, idx_lit);
    if (rc) return rc;
    for (int r = 0; r <= nreads; ++r) {
      const int64_t* o = ops + op_offset[r];
      int64_t idx = o[0];
      for (int64_t t = 0; t < o[1]; ++t) idx += o[3 + 2 * t] * regs[o[2 + 2 * t]].i;
      if (idx < 0 || idx >= op_size[r]) return -3;
      if (r < nreads) {
        regs[read_reg[r]].f = read_ptr[r][idx];
      } else {
        rc = exec_instrs(regs, ninstr, instr, instr_lit);
        if (rc) return rc;
        write_ptr[idx] = write_ptr[idx] + regs[result_reg].f;
      }
    }
    int l = nloops - 1;
    for (; l >= 0; --l) {
      if (++it[l] < loop_stop[l]) break;
      it[l] = loop_start[l];
    }
    if (l < 0) break;
  }
  return 0;
}
