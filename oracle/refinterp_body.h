/*
 * refinterp_body.h — ORACLE (test infrastructure, not product code).  The kernel interpreter of
 * refinterp.c, written once over REAL: included with REAL = float it is the reference's arithmetic;
 * included with REAL = double it is the float64 SHADOW of the same kernel — the same real-valued
 * function (same kernel list, same constants rounded to float32 as const_real does, llvmgen.nim:215-216)
 * evaluated with 2^-53 rounding, i.e. the exact value both the reference and the GPU approximate.
 * Tests use the shadow to separate the reference's own float32 rounding from the backend's.
 */
typedef union {
  REAL f;
  int64_t i;
} REG_T;


/* One pass over an instruction list (f32 / i64 / i1 arithmetic of llvmgen.nim:212-276). */
static inline int FN(exec_instrs)(REG_T* regs, int ninstr, const int32_t* instr, const double* instr_lit) {
  for (int k = 0; k < ninstr; ++k) {
    const int32_t* in = instr + 5 * k;
    REG_T* d = &regs[in[1]];
    const REG_T a = regs[in[2]], b = regs[in[3]], c = regs[in[4]];
    switch (in[0]) {
      case OP_SCALAR_LIT: d->f = LITERAL(instr_lit[k]); break; /* const_real(scalar type, double) llvmgen.nim:215-216 */
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
      case OP_SIN: d->f = M(sin)(a.f); break;
      case OP_COS: d->f = M(cos)(a.f); break;
      case OP_EXP: d->f = M(exp)(a.f); break;
      case OP_POW: d->f = M(pow)(a.f, b.f); break;
      case OP_SQRT: d->f = M(sqrt)(a.f); break;
      case OP_LOG: d->f = M(log)(a.f) / M(log)(b.f); break;   /* no CPU lowering in the reference; see kd.py */
      case OP_LOG10: d->f = M(log10)(a.f); break;
      case OP_LOG2: d->f = M(log2)(a.f); break;
      case OP_LN: d->f = M(log)(a.f); break;
      case OP_EQ_F: d->i = a.f == b.f; break;              /* ordered compares: false on NaN */
      case OP_LT_F: d->i = a.f < b.f; break;
      case OP_LE_F: d->i = a.f <= b.f; break;
      case OP_EQ_I: d->i = a.i == b.i; break;
      case OP_LT_I: d->i = a.i < b.i; break;
      case OP_LE_I: d->i = a.i <= b.i; break;
      case OP_AND: d->i = a.i & b.i; break;
      case OP_OR: d->i = a.i | b.i; break;
      case OP_SELECT: *d = a.i ? b : c; break;
      case OP_TOSCALAR: d->f = (REAL)a.i; break;          /* sitofp */
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
 * overwrite:  0 -> out[idx] += value (InstrWrite) ; 1 -> out[idx] = value (InstrOverwrite) ;
 *             2 -> out[idx] += |value|: the sum of the magnitudes of the terms a reduction adds up — the scale its
 *             rounding error is measured against (tests/parity.py; never part of a model's arithmetic)
 */
EXPORT int FN(ref_interp_kernel)(int nloops, const int64_t* loop_start, const int64_t* loop_stop, const int32_t* loop_reg,
                             int nregs, int nreads, const REAL* const* read_ptr, const int32_t* read_reg,
                             const int64_t* read_affine, int ninstr, const int32_t* instr, const double* instr_lit,
                             int result_reg, REAL* write_ptr, const int64_t* write_affine, int overwrite) {
  if (nloops > MAX_LOOPS || nregs >= MAX_REGS) return -1;
  REG_T regs[MAX_REGS];
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
      const int rc = FN(exec_instrs)(regs, ninstr, instr, instr_lit);
      if (rc) return rc;
    }
    {
      int64_t idx = write_affine[0];
      for (int l = 0; l < nloops; ++l) idx += write_affine[1 + l] * it[l];
      const REAL v = regs[result_reg].f;
      if (overwrite == 1)
        write_ptr[idx] = v;
      else if (overwrite == 2)
        write_ptr[idx] = write_ptr[idx] + (v < 0 ? -v : v);
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
EXPORT int FN(ref_interp_kernel2)(int nloops, const int64_t* loop_start, const int64_t* loop_stop, const int32_t* loop_reg,
                              int nregs, int nidx, const int32_t* idx_instr, const double* idx_lit, int nreads,
                              const REAL* const* read_ptr, const int32_t* read_reg, const int64_t* ops,
                              const int32_t* op_offset, int ninstr, const int32_t* instr, const double* instr_lit,
                              int result_reg, REAL* write_ptr, const int64_t* op_size) {
  if (nloops > MAX_LOOPS || nregs >= MAX_REGS) return -1;
  REG_T regs[MAX_REGS];
  memset(regs, 0, sizeof(regs));
  int64_t it[MAX_LOOPS];
  for (int l = 0; l < nloops; ++l) {
    if (loop_stop[l] <= loop_start[l]) return 0;
    it[l] = loop_start[l];
  }
  for (;;) {
    for (int l = 0; l < nloops; ++l) regs[loop_reg[l]].i = it[l];
    int rc = FN(exec_instrs)(regs, nidx, idx_instr, idx_lit);
    if (rc) return rc;
    for (int r = 0; r <= nreads; ++r) {
      const int64_t* o = ops + op_offset[r];
      int64_t idx = o[0];
      for (int64_t t = 0; t < o[1]; ++t) idx += o[3 + 2 * t] * regs[o[2 + 2 * t]].i;
      if (idx < 0 || idx >= op_size[r]) return -3;
      if (r < nreads) {
        regs[read_reg[r]].f = read_ptr[r][idx];
      } else {
        rc = FN(exec_instrs)(regs, ninstr, instr, instr_lit);
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
