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

/* the reference's arithmetic: float32 registers, libm's float functions */
#define REAL float
#define REG_T reg_f32_t
#define FN(name) name
#define M(fn) fn##f
#define LITERAL(x) ((float)(x))
#include "refinterp_body.h"
#undef REAL
#undef REG_T
#undef FN
#undef M
#undef LITERAL

/* float64 shadow of the same kernels (exported with the suffix _f64) */
#define REAL double
#define REG_T reg_f64_t
#define FN(name) name##_f64
#define M(fn) fn
#define LITERAL(x) ((double)(float)(x)) /* the float32 program's constants */
#include "refinterp_body.h"
#undef REAL
#undef REG_T
#undef FN
#undef M
#undef LITERAL

/* compile[float64] (model.nim:253-260: toScalarType(float64) = Scalar64): the reference's arithmetic for a program
 * instantiated over float64 — double registers, libm's double functions, constants as const_real(double type, v)
 * leaves them (llvmgen.nim:215-216: the literal itself).  Same loop nests, same sequential sums.  Exported with
 * the suffix _c64. */
#define REAL double
#define REG_T reg_c64_t
#define FN(name) name##_c64
#define M(fn) fn
#define LITERAL(x) ((double)(x))
#include "refinterp_body.h"
#undef REAL
#undef REG_T
#undef FN
#undef M
#undef LITERAL
