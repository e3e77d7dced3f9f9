#include "codegen.hpp"

#include <regex>

#include <cmath>
#include <cstdio>
#include <set>

#include "../eg_internal.hpp"

namespace eg {
namespace kd {

namespace {

struct Emitter {
  const Kernel& k;
  std::vector<Ty> ty;
  std::vector<Slot> slots;
  std::string body;

  explicit Emitter(const Kernel& kernel) : k(kernel), ty(infer_types(kernel)) {}

  int slot(Slot::Kind kind, int a = 0, int b = 0) {
    for (size_t i = 0; i < slots.size(); ++i)
      if (slots[i].kind == kind && slots[i].a == a && slots[i].b == b) return (int)i;
    Slot s;
    s.kind = kind;
    s.a = a;
    s.b = b;
    slots.push_back(s);
    return (int)slots.size() - 1;
  }
  const char* S() const { return k.f64 ? "double" : "float"; }  // the program's scalar type (model.nim:253-260)
  std::string p(int i) const { return "p" + std::to_string(i); }
  std::string reg(int r) const { return "r" + std::to_string(r); }

  static std::string f32_literal(double v) {
    const float f = (float)v;  // const_real(float type, double): llvmgen.nim:215-216
    if (std::isinf(f)) return f > 0 ? "__builtin_inff()" : "(-__builtin_inff())";
    if (std::isnan(f)) return "__builtin_nanf(\"\")";
    char buf[64];
    snprintf(buf, sizeof(buf), "%.9gf", (double)f);
    std::string s = buf;
    // "1f" is not a valid literal: make sure there is a '.' or an exponent
    if (s.find('.') == std::string::npos && s.find('e') == std::string::npos && s.find("inf") == std::string::npos)
      s.insert(s.size() - 1, ".0");
    return s;
  }

  // element offset of a tensor op; op_index: reads 0..n-1, write = n
  std::string flat_index(const Op& op, int op_index) {
    std::string s;
    for (size_t d = 0; d < op.dims.size(); ++d) {
      const Lin& l = op.dims[d];
      std::string term = std::to_string(l.constant) + "L";
      for (auto& f : l.factors) term += " + " + std::to_string(f.second) + "L * " + reg(f.first);
      if (!op.raw) term = p(slot(Slot::Stride, op_index, (int)d)) + " * (" + term + ")";
      s += (d ? " + " : "") + term;
    }
    if (s.empty()) s = "0L";
    return s;
  }

  int emit_instr(const Instr& ins, int index, std::string& out) {
    const Ty t = ty[ins.res];
    const char* ctype = t == Ty::Scalar ? S() : (t == Ty::Index ? "long" : "bool");
    std::string special;
    if (ins.kind == IK::Shape || ins.kind == IK::Len || ins.kind == IK::ShapeLen || ins.kind == IK::Epoch)
      special = p(slot(Slot::InstrVal, index));
    out += std::string("      const ") + ctype + " " + reg(ins.res) + " = " + instr_expression(ins, special, "r", k.f64) + ";\n";
    return EG_OK;
  }

  // loads + expression, at indentation of the innermost body
  void emit_body(std::string& out) {
    emit_indices(out);
    for (size_t i = 0; i < k.reads.size(); ++i)
      out += std::string("      const ") + S() + " " + reg(k.reads[i].reg) + " = t" + std::to_string(k.reads[i].tensor) + "[x" + std::to_string(i) + "];\n";
    for (size_t i = 0; i < k.instrs.size(); ++i) emit_instr(k.instrs[i], (int)i, out);
  }

  // computed indices first (`y div 2`): Index instructions over iterators and host values; then the
  // element offset x<i> of every read
  void emit_indices(std::string& out) {
    for (auto& ins : k.index_instrs)
      out += "      const long " + reg(ins.res) + " = " + instr_expression(ins, "0L", "r") + ";\n";
    for (size_t i = 0; i < k.reads.size(); ++i) out += index_decl("x" + std::to_string(i), k.reads[i], (int)i, (int)i);
  }

  // `long <var> = element offset of op`.  Index arithmetic dominates generated kernels over several
  // iterators (64-bit multiply-adds with run-time strides): an operand indexed exactly
  // like an earlier read (`in[n, y, x, c]` and the gradient written at `[n, y, x, c]`) reuses that
  // read's offset when its strides are the same — a comparison of kernel arguments, uniform for the
  // launch — and computes its own only otherwise.
  std::string index_decl(const std::string& var, const Op& op, int op_index, int nreads_before) {
    int same = -1;
    if (!op.raw && op.dims.size() >= 2)
      for (int j = 0; j < nreads_before && same < 0; ++j) {
        const Op& o = k.reads[j];
        if (o.raw || o.dims.size() != op.dims.size()) continue;
        bool eq = true;
        for (size_t d = 0; d < op.dims.size(); ++d) eq = eq && o.dims[d] == op.dims[d];
        if (eq) same = j;
      }
    if (same < 0 && !op.raw && op.dims.size() >= 2) {
      // the same tensor read again at a constant displacement (`in[n, 2y + dy, 2x + dx, c]`, the taps of a
      // stencil): the earlier offset plus stride * displacement
      for (int j = 0; j < nreads_before; ++j) {
        const Op& o = k.reads[j];
        if (o.tensor != op.tensor || o.raw || o.dims.size() != op.dims.size()) continue;
        bool eq = true;
        for (size_t d = 0; d < op.dims.size() && eq; ++d) {
          Lin a = o.dims[d], b = op.dims[d];
          a.constant = b.constant = 0;
          eq = a == b;
        }
        if (!eq) continue;
        std::string e = "x" + std::to_string(j);
        for (size_t d = 0; d < op.dims.size(); ++d) {
          const long delta = op.dims[d].constant - o.dims[d].constant;
          if (delta) e += " + " + p(slot(Slot::Stride, op_index, (int)d)) + " * " + std::to_string(delta) + "L";
        }
        return "      const long " + var + " = " + e + ";\n";
      }
    }
    if (same < 0) return "      const long " + var + " = " + flat_index(op, op_index) + ";\n";
    std::string cond;
    for (size_t d = 0; d < op.dims.size(); ++d)
      cond += (d ? " && " : "") + p(slot(Slot::Stride, op_index, (int)d)) + " == " + p(slot(Slot::Stride, same, (int)d));
    return "      long " + var + ";\n      if (" + cond + ") " + var + " = x" + std::to_string(same) + "; else " + var + " = " +
           flat_index(op, op_index) + ";\n";
  }

  std::string setup_decls() {
    std::string s;
    for (size_t i = 0; i < k.setup.size(); ++i)
      s += "  const long " + reg(k.setup[i].res) + " = " + p(slot(Slot::SetupVal, (int)i)) + ";\n";
    return s;
  }
};

}  // namespace

std::string f32_literal(double v) {
  const float f = (float)v;  // const_real(float type, double): llvmgen.nim:215-216
  if (std::isinf(f)) return f > 0 ? "__builtin_inff()" : "(-__builtin_inff())";
  if (std::isnan(f)) return "__builtin_nanf(\"\")";
  char buf[64];
  snprintf(buf, sizeof(buf), "%.9gf", (double)f);
  std::string s = buf;
  // "1f" is not a valid literal: make sure there is a '.' or an exponent
  if (s.find('.') == std::string::npos && s.find('e') == std::string::npos && s.find("inf") == std::string::npos)
    s.insert(s.size() - 1, ".0");
  return s;
}

// const_real(double type, v) (llvmgen.nim:215-216 with Scalar64): the literal itself, 17 significant digits.
std::string f64_literal(double v) {
  if (std::isinf(v)) return v > 0 ? "__builtin_inf()" : "(-__builtin_inf())";
  if (std::isnan(v)) return "__builtin_nan(\"\")";
  char buf[64];
  snprintf(buf, sizeof(buf), "%.17g", v);
  std::string s = buf;
  if (s.find('.') == std::string::npos && s.find('e') == std::string::npos) s += ".0";
  return s;
}

// One scalar instruction as a C expression over variables `<prefix><register>`; llvmgen.nim:212-276.
// f64: the program's scalar type is float64 — literals keep their double value, the math functions are the double ones.
std::string instr_expression(const Instr& ins, const std::string& special, const std::string& prefix, bool f64) {
  auto a = [&](int i) { return prefix + std::to_string(ins.args[i]); };
  std::string e;
  if (f64) {
    switch (ins.kind) {
      case IK::Scalar: return f64_literal(ins.lit);
      case IK::Sin: return "sin(" + a(0) + ")";
      case IK::Cos: return "cos(" + a(0) + ")";
      case IK::Exp: return "exp(" + a(0) + ")";
      case IK::Pow: return "pow(" + a(0) + ", " + a(1) + ")";
      case IK::Sqrt: return "sqrt(" + a(0) + ")";
      case IK::Log: return "log(" + a(0) + ") / log(" + a(1) + ")";
      case IK::Log10: return "log10(" + a(0) + ")";
      case IK::Log2: return "log2(" + a(0) + ")";
      case IK::Ln: return "log(" + a(0) + ")";
      case IK::ToScalar: return "(double)" + a(0);
      default: break;
    }
  }
  switch (ins.kind) {
      case IK::Scalar: e = f32_literal(ins.lit); break;
      case IK::Index: e = std::to_string((long)ins.lit) + "L"; break;
      case IK::Boolean: e = ins.lit != 0 ? "true" : "false"; break;
      case IK::Add: e = a(0) + " + " + a(1); break;
      case IK::Sub: e = a(0) + " - " + a(1); break;
      case IK::Mul: e = a(0) + " * " + a(1); break;
      case IK::Div: e = a(0) + " / " + a(1); break;
      case IK::IndexDiv: e = a(0) + " / " + a(1); break;
      case IK::Mod: e = a(0) + " % " + a(1); break;
      case IK::Wrap: e = "((" + a(0) + " % " + a(1) + ") + " + a(1) + ") % " + a(1); break;  // llvmgen.nim:227-230
      case IK::Negate: e = "-" + a(0); break;
      case IK::Sin: e = "sinf(" + a(0) + ")"; break;
      case IK::Cos: e = "cosf(" + a(0) + ")"; break;
      case IK::Exp: e = "expf(" + a(0) + ")"; break;
      case IK::Pow: e = "powf(" + a(0) + ", " + a(1) + ")"; break;
      case IK::Sqrt: e = "sqrtf(" + a(0) + ")"; break;
      case IK::Log: e = "logf(" + a(0) + ") / logf(" + a(1) + ")"; break;
      case IK::Log10: e = "log10f(" + a(0) + ")"; break;
      case IK::Log2: e = "log2f(" + a(0) + ")"; break;
      case IK::Ln: e = "logf(" + a(0) + ")"; break;
      case IK::Eq: e = a(0) + " == " + a(1); break;  // ordered compare: false on NaN (llvmgen.nim:253)
      case IK::Lt: e = a(0) + " < " + a(1); break;
      case IK::Le: e = a(0) + " <= " + a(1); break;
      case IK::And: e = a(0) + " && " + a(1); break;
      case IK::Or: e = a(0) + " || " + a(1); break;
      case IK::Select: e = a(0) + " ? " + a(1) + " : " + a(2); break;
      case IK::ToScalar: e = "(float)" + a(0); break;  // sitofp
      case IK::ToIndex: e = "(long)" + a(0); break;    // fptosi
      case IK::Shape: case IK::Len: case IK::ShapeLen: case IK::Epoch:
        e = special;  // host-evaluated builtins (model.nim:83-104): a kernel argument or a constant
        break;
  }
  return e;
}

namespace {

std::vector<int> distinct_tensors(const Kernel& k, bool include_write) {
  std::vector<int> out;
  auto add = [&](int t) {
    for (int x : out)
      if (x == t) return;
    out.push_back(t);
  };
  if (include_write) add(k.write.tensor);
  for (auto& r : k.reads) add(r.tensor);
  return out;
}

std::string signature(const std::string& name, const std::vector<int>& tensors, int write_tensor, bool partial_first,
                      size_t nslots, const std::string& S) {
  std::string s = "extern \"C\" __global__ void __launch_bounds__(256) " + name + "(";
  bool first = true;
  if (partial_first) {
    s += S + "* __restrict__ partial";
    first = false;
  }
  for (int t : tensors) {
    s += first ? "" : ", ";
    first = false;
    // the written tensor may also be read (optimizer kernels read their own parameter): no restrict
    s += (t == write_tensor && !partial_first ? S + "* t" : "const " + S + "* t") + std::to_string(t);
  }
  for (size_t i = 0; i < nslots; ++i) {
    s += first ? "" : ", ";
    first = false;
    s += "long p" + std::to_string(i);
  }
  s += ")";
  return s;
}

}  // namespace

void split_loops(const Kernel& k, std::vector<int>& indep, std::vector<int>& red, bool& scatter) {
  indep.clear();
  red.clear();
  std::set<int> ind_regs;
  // identifyIndependent (passes.nim:1774-1782): iterators that appear bare in a write dimension,
  // taken in write-dimension order so the last one is the fastest varying in memory.
  for (auto& d : k.write.dims) {
    const int r = d.only_register();
    if (!r || ind_regs.count(r)) continue;
    for (size_t l = 0; l < k.loops.size(); ++l)
      if (k.loops[l].reg == r) {
        ind_regs.insert(r);
        indep.push_back((int)l);
      }
  }
  for (size_t l = 0; l < k.loops.size(); ++l)
    if (!ind_regs.count(k.loops[l].reg)) red.push_back((int)l);
  scatter = false;
  for (auto& d : k.write.dims)
    for (auto& f : d.factors) {
      for (int l : red)
        if (k.loops[l].reg == f.first) scatter = true;
      // a computed index (`y div 2`) moves with the iterators it is computed from
      for (auto& ins : k.index_instrs)
        if (ins.res == f.first) scatter = true;
    }
}

bool split_reduction_capable(const Kernel& k) {
  if (!k.index_instrs.empty()) return false;
  std::vector<int> indep, red;
  bool scatter;
  split_loops(k, indep, red, scatter);
  if (scatter || red.empty()) return false;
  // every write dim is a distinct bare iterator, or a constant (only when there is no independent loop)
  std::set<int> seen;
  for (auto& d : k.write.dims) {
    const int r = d.only_register();
    if (r) {
      if (seen.count(r)) return false;
      seen.insert(r);
    } else if (!d.factors.empty() || d.constant != 0) {
      return false;
    }
  }
  if (seen.size() != indep.size()) return false;
  if (indep.empty()) return true;  // single element 0
  // constants mixed with iterators would need the other dims to have extent 1: not handled
  for (auto& d : k.write.dims)
    if (!d.only_register()) return false;
  return true;
}

int generate_mode_a(const Kernel& k, const std::string& name, GenericSource& out) {
  for (auto& ins : k.index_instrs)
    if (ins.kind == IK::Shape || ins.kind == IK::Len || ins.kind == IK::ShapeLen || ins.kind == IK::Epoch) {
      // cannot happen: the parser moves iterator-independent index instructions into the setup (kd.cpp, hoist_host_indices)
      set_error("internal: host-evaluated builtin left among the index instructions of a kernel");
      return EG_ERR_INVALID;
    }
  Emitter em(k);
  out = GenericSource();
  out.name = name;
  split_loops(k, out.indep, out.red, out.scatter);
  const int s_acc = em.slot(Slot::Accumulate);
  const int s_total = em.slot(Slot::Total);
  // Four elements per thread (Slot::Vec4, decided per launch): possible when every operand ends in
  // the bare fastest iterator and nothing else depends on it — then the four elements are adjacent
  // in every operand, the index arithmetic is shared and loads / stores are 16 bytes wide.
  // (float32 only: a float64 kernel moves 8 bytes per element and lane already)
  bool vec = !k.f64 && out.red.empty() && !out.scatter && !out.indep.empty() && k.reads.size() <= 12;
  int rc = 0;
  if (vec) {
    const Loop& fast = k.loops[out.indep.back()];
    rc = fast.reg;
    vec = !fast.has_bounds && k.result != rc;
    auto ends_in_rc = [&](const Op& op) {
      if (op.raw) return op.dims.size() == 1 && op.dims[0].only_register() == rc;  // `out{it} ++= f(in{it})`
      if (op.dims.empty() || op.dims.back().only_register() != rc) return false;
      for (size_t d = 0; d + 1 < op.dims.size(); ++d)
        if (op.dims[d].factor_of(rc) != 0) return false;
      return true;
    };
    vec = vec && ends_in_rc(k.write);
    for (auto& rd : k.reads) vec = vec && ends_in_rc(rd);
    for (auto& ins : k.index_instrs)
      for (int a : ins.args) vec = vec && a != rc;
    for (auto& ins : k.instrs)
      for (int a : ins.args) vec = vec && a != rc;
  }
  const int s_vec = vec ? em.slot(Slot::Vec4) : -1;
  const std::string PV = vec ? em.p(s_vec) : "0L";
  std::string code;
  code += "  long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;\n";
  code += vec ? "  if (gid >= (" + PV + " ? " + em.p(s_total) + " / 4L : " + em.p(s_total) + ")) return;\n"
              : "  if (gid >= " + em.p(s_total) + ") return;\n";
  if (vec) code += "  const long VW = " + PV + " ? 4L : 1L;\n";
  code += em.setup_decls();
  // decode: last independent loop varies fastest.  The divisions by run-time extents dominate a
  // bandwidth-bound kernel when done in 64 bits (pooling gradient over 19 M elements: 100 -> 60 us),
  // so they are done in 32 bits whenever the iteration space fits (wave-uniform branch).
  if (out.indep.size() > 1) {
    for (int l : out.indep) code += "  long " + em.reg(k.loops[l].reg) + ";\n";
    for (int wide = 0; wide < 2; ++wide) {
      code += wide ? "  } else {\n" : "  if (" + em.p(s_total) + " <= 0x7fffffffL) {\n    unsigned g32 = (unsigned)gid;\n";
      const std::string g = wide ? "gid" : "g32";
      for (size_t i = out.indep.size(); i-- > 0;) {
        const int l = out.indep[i];
        const bool fastest = vec && i + 1 == out.indep.size();
        std::string ext = em.p(em.slot(Slot::LoopExtent, l));
        const std::string start = em.p(em.slot(Slot::LoopStart, l));
        if (fastest) ext = "(" + ext + " / VW)";
        const std::string e = wide ? ext : "(unsigned)" + ext;
        const std::string scale = fastest ? " * VW" : "";
        if (i == 0)
          code += "    " + em.reg(k.loops[l].reg) + " = " + start + " + (long)" + g + scale + ";\n";
        else
          code += "    " + em.reg(k.loops[l].reg) + " = " + start + " + (long)(" + g + " % " + e + ")" + scale + "; " + g + " /= " + e + ";\n";
      }
    }
    code += "  }\n";
  } else {
    for (size_t i = out.indep.size(); i-- > 0;) {
      const int l = out.indep[i];
      const std::string start = em.p(em.slot(Slot::LoopStart, l));
      code += "  const long " + em.reg(k.loops[l].reg) + " = " + start + " + gid" + (vec ? " * VW" : "") + ";\n";
    }
  }
  const int write_index = (int)k.reads.size();
  std::string inner;
  em.emit_body(inner);
  if (out.scatter) {
    // write index moves with the serial loops: read-modify-write per iteration (the tensor was zeroed)
    for (int l : out.red) {
      const std::string r = em.reg(k.loops[l].reg);
      code += "  for (long " + r + " = " + em.p(em.slot(Slot::LoopStart, l)) + "; " + r + " < " +
              em.p(em.slot(Slot::LoopStart, l)) + " + " + em.p(em.slot(Slot::LoopExtent, l)) + "; ++" + r + ") {\n";
    }
    code += inner;
    code += "      { " + em.index_decl("w", k.write, write_index, (int)k.reads.size()) + "        t" + std::to_string(k.write.tensor) +
            "[w] = t" + std::to_string(k.write.tensor) + "[w] + " + em.reg(k.result) + "; }\n";
    for (size_t i = 0; i < out.red.size(); ++i) code += "  }\n";
  } else if (out.red.empty()) {
    // one element (or four) per thread: the write offset can share a read's (index_decl)
    const std::string wt = "t" + std::to_string(k.write.tensor);
    std::string idx, loads, instrs;
    em.emit_indices(idx);
    for (size_t i = 0; i < k.reads.size(); ++i)
      loads += std::string("      const ") + em.S() + " " + em.reg(k.reads[i].reg) + " = t" + std::to_string(k.reads[i].tensor) + "[x" + std::to_string(i) + "];\n";
    for (size_t i = 0; i < k.instrs.size(); ++i) em.emit_instr(k.instrs[i], (int)i, instrs);
    code += "    {\n" + idx + em.index_decl("w", k.write, write_index, (int)k.reads.size());
    if (vec) {
      code += "      if (" + PV + ") {\n";
      for (size_t i = 0; i < k.reads.size(); ++i)
        code += "      const eg_f4 v" + std::to_string(i) + " = *reinterpret_cast<const eg_f4*>(t" + std::to_string(k.reads[i].tensor) +
                " + x" + std::to_string(i) + ");\n";
      code += "      eg_f4 res;\n#pragma unroll\n      for (int j = 0; j < 4; ++j) {\n";
      for (size_t i = 0; i < k.reads.size(); ++i)
        code += "      const float " + em.reg(k.reads[i].reg) + " = v" + std::to_string(i) + "[j];\n";
      code += instrs + "      res[j] = " + em.reg(k.result) + ";\n      }\n";
      code += "      eg_f4* wp = reinterpret_cast<eg_f4*>(" + wt + " + w);\n";
      code += "      if (" + em.p(s_acc) + ") { const eg_f4 old = *wp; for (int j = 0; j < 4; ++j) res[j] = old[j] + res[j]; }\n";
      code += "      *wp = res;\n      } else {\n";
    }
    code += loads + instrs;
    code += "      " + wt + "[w] = " + em.p(s_acc) + " ? " + wt + "[w] + " + em.reg(k.result) + " : " + em.reg(k.result) + ";\n";
    if (vec) code += "      }\n";
    code += "    }\n";
  } else {
    code += std::string("  ") + em.S() + (k.f64 ? " acc = 0.0;\n" : " acc = 0.0f;\n");
    for (int l : out.red) {
      const std::string r = em.reg(k.loops[l].reg);
      code += "  for (long " + r + " = " + em.p(em.slot(Slot::LoopStart, l)) + "; " + r + " < " +
              em.p(em.slot(Slot::LoopStart, l)) + " + " + em.p(em.slot(Slot::LoopExtent, l)) + "; ++" + r + ") {\n";
    }
    code += "    {\n" + inner;
    code += out.red.empty() ? "      acc = " + em.reg(k.result) + ";\n" : "      acc = acc + " + em.reg(k.result) + ";\n";
    code += "    }\n";
    for (size_t i = 0; i < out.red.size(); ++i) code += "  }\n";
    const std::string wt = "t" + std::to_string(k.write.tensor);
    code += "  const long w = " + em.flat_index(k.write, write_index) + ";\n";
    code += "  " + wt + "[w] = " + em.p(s_acc) + " ? " + wt + "[w] + acc : acc;\n";
  }
  out.tensor_args = distinct_tensors(k, true);
  // Index arithmetic is most of what these kernels execute.  When every operand has
  // fewer than 2^31 elements (Slot::Narrow, decided per launch) it is exact in 32 bits: the body is
  // emitted a second time with `int` indices and 32-bit copies of the arguments, and a
  // launch-uniform branch picks one.
  const size_t head_end = code.find("return;\n") + 8;
  const std::string head = code.substr(0, head_end), wide = code.substr(head_end);
  const int s_narrow = em.slot(Slot::Narrow);
  std::string narrow = std::regex_replace(wide, std::regex("\\blong\\b"), "int");
  narrow = std::regex_replace(narrow, std::regex("\\b(0x[0-9a-fA-F]+|[0-9]+)L\\b"), "$1");
  narrow = std::regex_replace(narrow, std::regex("\\bp([0-9]+)\\b"), "q$1");
  std::string qdecl;
  for (size_t i = 0; i < em.slots.size(); ++i)
    qdecl += "    const int q" + std::to_string(i) + " = (int)p" + std::to_string(i) + ";\n";
  out.slots = em.slots;
  out.source = std::string("typedef float eg_f4 __attribute__((ext_vector_type(4)));\n") +
               signature(name, out.tensor_args, k.write.tensor, false, out.slots.size(), em.S()) + " {\n" + head + "  if (" +
               em.p(s_narrow) + ") {\n" + qdecl + narrow + "  } else {\n" + wide + "  }\n}\n";
  return EG_OK;
}

int generate_mode_b(const Kernel& k, const std::string& name, int tx, GenericSource& out) {
  Emitter em(k);
  out = GenericSource();
  out.name = name;
  split_loops(k, out.indep, out.red, out.scatter);
  if (!split_reduction_capable(k)) {
    set_error("kernel is not eligible for the split reduction template");
    return EG_ERR_UNSUPPORTED;
  }
  out.tx = tx;
  out.ty = 256 / tx;
  const int s_total = em.slot(Slot::Total);
  const int s_rtotal = em.slot(Slot::RTotal);
  const int s_chunk = em.slot(Slot::Chunk);
  std::string code;
  const std::string S = em.S(), Z = k.f64 ? "0.0" : "0.0f";
  code += "  __shared__ " + S + " red[256];\n";
  code += "  const int tx = threadIdx.x % " + std::to_string(tx) + ", ty = threadIdx.x / " + std::to_string(tx) + ";\n";
  code += "  long ii = (long)blockIdx.y * " + std::to_string(tx) + " + tx;\n";
  code += "  const bool active = ii < " + em.p(s_total) + ";\n";
  code += "  const long flat_out = ii;\n";
  code += em.setup_decls();
  code += "  " + S + " acc = " + Z + ";\n";
  code += "  if (active) {\n";
  for (size_t i = out.indep.size(); i-- > 0;) {
    const int l = out.indep[i];
    const std::string ext = em.p(em.slot(Slot::LoopExtent, l)), start = em.p(em.slot(Slot::LoopStart, l));
    if (i == 0)
      code += "  const long " + em.reg(k.loops[l].reg) + " = " + start + " + ii;\n";
    else
      code += "  const long " + em.reg(k.loops[l].reg) + " = " + start + " + ii % " + ext + "; ii /= " + ext + ";\n";
  }
  code += "  const long r_begin = (long)blockIdx.x * " + em.p(s_chunk) + ";\n";
  code += "  long r_end = r_begin + " + em.p(s_chunk) + "; if (r_end > " + em.p(s_rtotal) + ") r_end = " + em.p(s_rtotal) + ";\n";
  // the element evaluation as a lambda so the loop can be unrolled 4x with independent partial
  // sums: four loads in flight per lane instead of one (bias-gradient sums are HBM bound)
  const std::string TY = std::to_string(out.ty);
  code += "  auto term = [&](long rr) -> " + S + " {\n";
  code += "    long rem = rr;\n";
  for (size_t i = out.red.size(); i-- > 0;) {  // innermost reduction loop varies fastest
    const int l = out.red[i];
    const std::string ext = em.p(em.slot(Slot::LoopExtent, l)), start = em.p(em.slot(Slot::LoopStart, l));
    if (i == 0)
      code += "    const long " + em.reg(k.loops[l].reg) + " = " + start + " + rem;\n";
    else
      code += "    const long " + em.reg(k.loops[l].reg) + " = " + start + " + rem % " + ext + "; rem /= " + ext + ";\n";
  }
  std::string inner;
  em.emit_body(inner);
  code += inner + "    return " + em.reg(k.result) + ";\n  };\n";
  code += "  " + S + " acc1 = " + Z + ", acc2 = " + Z + ", acc3 = " + Z + ";\n";
  code += "  long rr = r_begin + ty;\n";
  code += "  for (; rr + 3 * " + TY + " < r_end; rr += 4 * " + TY + ") {\n";
  code += "    const " + S + " v0 = term(rr), v1 = term(rr + " + TY + "), v2 = term(rr + 2 * " + TY + "), v3 = term(rr + 3 * " + TY + ");\n";
  code += "    acc = acc + v0; acc1 = acc1 + v1; acc2 = acc2 + v2; acc3 = acc3 + v3;\n  }\n";
  code += "  for (; rr < r_end; rr += " + TY + ") acc = acc + term(rr);\n";
  code += "  acc = (acc + acc1) + (acc2 + acc3);\n";
  code += "  }\n";
  code += "  red[threadIdx.x] = acc;\n  __syncthreads();\n";
  code += "  if (ty == 0 && active) {\n    " + S + " s = " + Z + ";\n";
  code += "    for (int t = 0; t < " + std::to_string(out.ty) + "; ++t) s = s + red[t * " + std::to_string(tx) + " + tx];\n";
  code += "    partial[(long)blockIdx.x * " + em.p(s_total) + " + flat_out] = s;\n  }\n";
  out.tensor_args = distinct_tensors(k, false);
  out.slots = em.slots;
  out.source = signature(name, out.tensor_args, k.write.tensor, true, out.slots.size(), S) + " {\n" + code + "}\n";
  return EG_OK;
}

}  // namespace kd
}  // namespace eg
