// Parser, autodiff (`generate`/`derive`), dead-kernel elimination and run-time shape inference
// for kernel-description programs.  Host side of the backend: what exprgrad's passes.nim does
// between `toProgram` and code generation, restricted to what the GPU hot path needs.
#include "kd.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <set>
#include <sstream>

#include "../eg_internal.hpp"

namespace eg {
namespace kd {

static const struct {
  IK k;
  const char* name;
} kNames[] = {
    {IK::Index, "index"},   {IK::Scalar, "scalar"},     {IK::Boolean, "boolean"}, {IK::Add, "add"},
    {IK::Sub, "sub"},       {IK::Mul, "mul"},           {IK::Div, "div"},         {IK::IndexDiv, "indexdiv"},
    {IK::Mod, "mod"},       {IK::Wrap, "wrap"},         {IK::Negate, "negate"},   {IK::Sin, "sin"},
    {IK::Cos, "cos"},       {IK::Exp, "exp"},           {IK::Pow, "pow"},         {IK::Sqrt, "sqrt"},
    {IK::Log, "log"},       {IK::Log10, "log10"},       {IK::Log2, "log2"},       {IK::Ln, "ln"},
    {IK::Eq, "eq"},         {IK::Lt, "lt"},             {IK::Le, "le"},           {IK::And, "and"},
    {IK::Or, "or"},         {IK::Select, "select"},     {IK::ToScalar, "toscalar"}, {IK::ToIndex, "toindex"},
    {IK::Shape, "shape"},   {IK::Len, "len"},           {IK::ShapeLen, "shapelen"}, {IK::Epoch, "epoch"},
};

const char* ik_name(IK k) {
  for (auto& e : kNames)
    if (e.k == k) return e.name;
  return "?";
}
bool ik_from_name(const std::string& s, IK& out) {
  for (auto& e : kNames)
    if (s == e.name) {
      out = e.k;
      return true;
    }
  return false;
}

int Lin::only_register() const {
  if (constant == 0 && factors.size() == 1 && factors[0].second == 1) return factors[0].first;
  return 0;
}
long Lin::factor_of(int reg) const {
  for (auto& f : factors)
    if (f.first == reg) return f.second;
  return 0;
}
bool Lin::operator==(const Lin& o) const {
  if (constant != o.constant || factors.size() != o.factors.size()) return false;
  for (auto& f : factors)
    if (o.factor_of(f.first) != f.second) return false;
  return true;
}

Target* Program::find_target(const std::string& name) {
  for (auto& t : targets)
    if (t.name == name) return &t;
  return nullptr;
}
int Program::alloc_tensor(TK kind, const std::string& name) {
  TensorDef d;
  d.kind = kind;
  d.name = name;
  tensors.push_back(d);
  return (int)tensors.size() - 1;
}

// ------------------------------------------------------------------------------ parsing

namespace {
struct Tok {
  std::vector<std::string> t;
  size_t pos = 0;
  int line = 0;
  bool has() const { return pos < t.size(); }
  const std::string& next() {
    static const std::string empty;
    return pos < t.size() ? t[pos++] : empty;
  }
  bool next_long(long& v) {
    if (!has()) return false;
    char* end = nullptr;
    const std::string& s = t[pos++];
    v = strtol(s.c_str(), &end, 10);
    return end && *end == 0 && !s.empty();
  }
  bool next_double(double& v) {
    if (!has()) return false;
    char* end = nullptr;
    const std::string& s = t[pos++];
    v = strtod(s.c_str(), &end);
    return end && *end == 0 && !s.empty();
  }
};

#define P_FAIL(...)                      \
  do {                                   \
    set_error(__VA_ARGS__);              \
    return EG_ERR_INVALID;               \
  } while (0)

int parse_lin(Tok& tk, Lin& out) {
  if (tk.next() != "L") P_FAIL("kd line %d: expected L", tk.line);
  long c, n;
  if (!tk.next_long(c) || !tk.next_long(n) || n < 0) P_FAIL("kd line %d: bad linear index", tk.line);
  out.constant = c;
  out.factors.clear();
  for (long i = 0; i < n; ++i) {
    long r, f;
    if (!tk.next_long(r) || !tk.next_long(f)) P_FAIL("kd line %d: bad linear index term", tk.line);
    if (f != 0) out.factors.push_back({(int)r, f});
  }
  return EG_OK;
}

int parse_op(Tok& tk, Op& op) {
  long tid, reg, raw, nd;
  if (!tk.next_long(tid) || !tk.next_long(reg) || !tk.next_long(raw) || !tk.next_long(nd) || nd < 0)
    P_FAIL("kd line %d: bad tensor op", tk.line);
  op.tensor = (int)tid;
  op.reg = (int)reg;
  op.raw = raw != 0;
  op.dims.resize(nd);
  for (long i = 0; i < nd; ++i) {
    int rc = parse_lin(tk, op.dims[i]);
    if (rc) return rc;
  }
  return EG_OK;
}

int parse_instr(Tok& tk, Instr& ins) {
  IK k;
  if (!ik_from_name(tk.next(), k)) P_FAIL("kd line %d: unknown instruction", tk.line);
  long res, n;
  if (!tk.next_long(res) || !tk.next_long(n) || n < 0 || n > 3) P_FAIL("kd line %d: bad instruction", tk.line);
  ins.kind = k;
  ins.res = (int)res;
  ins.args.clear();
  for (long i = 0; i < n; ++i) {
    long a;
    if (!tk.next_long(a)) P_FAIL("kd line %d: bad instruction argument", tk.line);
    ins.args.push_back((int)a);
  }
  if (k == IK::Scalar) {
    if (!tk.next_double(ins.lit)) P_FAIL("kd line %d: bad scalar literal", tk.line);
  } else if (k == IK::Index || k == IK::Boolean) {
    long v;
    if (!tk.next_long(v)) P_FAIL("kd line %d: bad literal", tk.line);
    ins.lit = (double)v;
  } else if (k == IK::Shape) {
    long t, d;
    if (!tk.next_long(t) || !tk.next_long(d)) P_FAIL("kd line %d: bad shape()", tk.line);
    ins.tensor = (int)t;
    ins.dim = (int)d;
  } else if (k == IK::Len || k == IK::ShapeLen) {
    long t;
    if (!tk.next_long(t)) P_FAIL("kd line %d: bad len()", tk.line);
    ins.tensor = (int)t;
  }
  return EG_OK;
}
}  // namespace

int parse(const char* text, Program& prog) {
  prog = Program();
  prog.tensors.resize(1);
  std::istringstream in(text);
  std::string line;
  Target* target = nullptr;
  Kernel* kernel = nullptr;
  Kernel cur, grad_cur;    // grad_cur: a kernel of cur's customgrad block
  bool in_custom = false;
  int lineno = 0;
  bool header = false;
  while (std::getline(in, line)) {
    ++lineno;
    Tok tk;
    tk.line = lineno;
    std::istringstream ls(line);
    std::string w;
    while (ls >> w) tk.t.push_back(w);
    if (tk.t.empty() || tk.t[0][0] == '#') continue;
    const std::string kw = tk.next();
    if (kw == "kd") {
      if (tk.next() != "1") P_FAIL("kd line %d: unsupported version/scalar type", lineno);
      const std::string scalar = tk.next();
      if (scalar != "f32" && scalar != "f64") P_FAIL("kd line %d: unsupported version/scalar type", lineno);
      prog.f64 = scalar == "f64";
      header = true;
    } else if (kw == "tensor") {
      long id, rank;
      if (!tk.next_long(id) || id != (long)prog.tensors.size()) P_FAIL("kd line %d: tensor ids must be 1,2,3,...", lineno);
      TensorDef d;
      const std::string kind = tk.next();
      if (kind == "input") d.kind = TK::Input;
      else if (kind == "param") d.kind = TK::Param;
      else if (kind == "result") d.kind = TK::Result;
      else if (kind == "cache") d.kind = TK::Cache;
      else if (kind == "random") d.kind = TK::Random;
      else P_FAIL("kd line %d: unknown tensor kind '%s'", lineno, kind.c_str());
      d.name = tk.next();
      if (d.name == "-") d.name.clear();
      if (!tk.next_long(rank)) P_FAIL("kd line %d: bad rank", lineno);
      if (rank >= 0) {
        d.has_shape = true;
        for (long i = 0; i < rank; ++i) {
          long s;
          if (!tk.next_long(s)) P_FAIL("kd line %d: bad shape", lineno);
          d.shape.push_back(s);
        }
      }
      if (d.kind == TK::Param) {
        if (!d.has_shape) P_FAIL("kd line %d: a parameter needs a static shape", lineno);
        if (!tk.next_double(d.lo) || !tk.next_double(d.hi)) P_FAIL("kd line %d: bad init range", lineno);
      }
      if (d.kind == TK::Random && (!tk.next_double(d.lo) || !tk.next_double(d.hi)))
        P_FAIL("kd line %d: a random tensor needs its range", lineno);
      if (d.kind == TK::Cache && !d.has_shape) P_FAIL("kd line %d: a cache tensor needs a static shape", lineno);
      if (d.kind == TK::Input) prog.inputs[d.name] = (int)id;
      prog.tensors.push_back(d);
    } else if (kw == "shapecopy") {
      long a, b;
      if (!tk.next_long(a) || !tk.next_long(b)) P_FAIL("kd line %d: bad shapecopy", lineno);
      prog.shape_copy[(int)a] = (int)b;
    } else if (kw == "shapedims") {
      long a, n;
      if (!tk.next_long(a) || !tk.next_long(n) || n < 0) P_FAIL("kd line %d: bad shapedims", lineno);
      std::vector<Lin> dims(n);
      for (long i = 0; i < n; ++i) {
        int rc = parse_lin(tk, dims[i]);
        if (rc) return rc;
      }
      prog.shape_dims[(int)a] = dims;
    } else if (kw == "target") {
      Target t;
      t.name = tk.next();
      long out;
      if (t.name.empty() || !tk.next_long(out)) P_FAIL("kd line %d: bad target", lineno);
      t.output = (int)out;
      prog.targets.push_back(t);
      target = &prog.targets.back();
    } else if (kw == "endtarget") {
      target = nullptr;
    } else if (kw == "shapesetup") {
      long a;
      if (!tk.next_long(a)) P_FAIL("kd line %d: bad shapesetup", lineno);
      Instr ins;
      int rc = parse_instr(tk, ins);
      if (rc) return rc;
      prog.shape_setup[(int)a].push_back(ins);
    } else if (kw == "kernel") {
      if (!target) P_FAIL("kd line %d: kernel outside a target", lineno);
      if (kernel) P_FAIL("kd line %d: kernel inside a kernel (missing customgrad?)", lineno);
      Kernel& dst = in_custom ? grad_cur : cur;
      dst = Kernel();
      long n;
      if (!tk.next_long(n)) P_FAIL("kd line %d: bad register count", lineno);
      dst.nregs = (int)n;
      kernel = &dst;
    } else if (kw == "endkernel") {
      if (!kernel || !target) P_FAIL("kd line %d: stray endkernel", lineno);
      if (kernel == &grad_cur) {
        cur.custom_grad.push_back(grad_cur);
        kernel = nullptr;
      } else {
        target->source.push_back(cur);
        kernel = nullptr;
      }
    } else if (kw == "customgrad") {
      if (kernel != &cur || in_custom) P_FAIL("kd line %d: customgrad outside a kernel", lineno);
      cur.has_custom_grad = true;
      in_custom = true;
      kernel = nullptr;
    } else if (kw == "endcustomgrad") {
      if (!in_custom || kernel) P_FAIL("kd line %d: stray endcustomgrad", lineno);
      in_custom = false;
      kernel = &cur;
    } else if (kw == "backwards" || kw == "gradient") {
      if (!target) P_FAIL("kd line %d: generator outside a target", lineno);
      Kernel g;
      long a, b = 0;
      if (!tk.next_long(a)) P_FAIL("kd line %d: bad generator", lineno);
      if (kw == "gradient" && !tk.next_long(b)) P_FAIL("kd line %d: bad generator", lineno);
      g.gen = kw == "backwards" ? Gen::Backwards : Gen::Gradient;
      g.gen_tensor = (int)a;
      g.gen_dest = (int)b;
      target->source.push_back(g);
    } else {
      if (!kernel) P_FAIL("kd line %d: '%s' outside a kernel", lineno, kw.c_str());
      int rc = EG_OK;
      if (kw == "setup") {
        Instr ins;
        rc = parse_instr(tk, ins);
        kernel->setup.push_back(ins);
      } else if (kw == "loop") {
        Loop lp;
        long reg, has;
        if (!tk.next_long(reg)) P_FAIL("kd line %d: bad loop", lineno);
        lp.reg = (int)reg;
        lp.name = tk.next();
        if (!tk.next_long(has)) P_FAIL("kd line %d: bad loop", lineno);
        lp.has_bounds = has != 0;
        if (lp.has_bounds) {
          rc = parse_lin(tk, lp.start);
          if (!rc) rc = parse_lin(tk, lp.stop);
        }
        kernel->loops.push_back(lp);
      } else if (kw == "read") {
        Op op;
        rc = parse_op(tk, op);
        kernel->reads.push_back(op);
      } else if (kw == "idx") {
        Instr ins;
        rc = parse_instr(tk, ins);
        kernel->index_instrs.push_back(ins);
      } else if (kw == "ins") {
        Instr ins;
        rc = parse_instr(tk, ins);
        kernel->instrs.push_back(ins);
      } else if (kw == "result") {
        long r;
        if (!tk.next_long(r)) P_FAIL("kd line %d: bad result", lineno);
        kernel->result = (int)r;
      } else if (kw == "write") {
        rc = parse_op(tk, kernel->write);
      } else {
        P_FAIL("kd line %d: unknown statement '%s'", lineno, kw.c_str());
      }
      if (rc) return rc;
    }
  }
  if (!header) P_FAIL("kd: missing 'kd 1 f32' header");
  // An "idx" instruction whose value does not change inside the loop nest — a literal, shape() / len() / epoch(), or
  // arithmetic on those (`x mod shape(t)[0]`: the literal and the shape) — is a host value: it joins the kernel's setup
  // and reaches the kernel as an argument.  The reference lowers such an instruction like any other index instruction
  // (passes.nim:787-843); the Nim emitter already writes them as "setup" (hipmodel.nim emitKernel), text from another
  // producer is normalised here, so the generated kernels only ever see iterator-dependent index instructions.
  auto hoist_host_indices = [](Kernel& k) {
    std::set<int> varying;
    for (auto& lp : k.loops) varying.insert(lp.reg);
    std::vector<Instr> stay;
    for (auto& ins : k.index_instrs) {
      bool v = false;
      for (int a : ins.args) v = v || varying.count(a) != 0;
      if (v) {
        varying.insert(ins.res);
        stay.push_back(ins);
      } else {
        k.setup.push_back(ins);
      }
    }
    k.index_instrs.swap(stay);
  };
  for (auto& t : prog.targets)
    for (auto& k : t.source) {
      hoist_host_indices(k);
      for (auto& g : k.custom_grad) hoist_host_indices(g);
    }
  // epoch() among the host values (written as "setup" or hoisted just now): plans are then made per epoch
  auto uses_epoch = [](const std::vector<Instr>& v) {
    for (auto& ins : v)
      if (ins.kind == IK::Epoch) return true;
    return false;
  };
  for (auto& t : prog.targets)
    for (auto& k : t.source) {
      prog.epoch_in_setup = prog.epoch_in_setup || uses_epoch(k.setup);
      for (auto& g : k.custom_grad) prog.epoch_in_setup = prog.epoch_in_setup || uses_epoch(g.setup);
    }
  for (auto& ss : prog.shape_setup) prog.epoch_in_setup = prog.epoch_in_setup || uses_epoch(ss.second);
  // validate tensor references
  auto check_tid = [&](int t) { return t >= 1 && t < (int)prog.tensors.size(); };
  for (auto& t : prog.targets) {
    if (t.output != 0 && !check_tid(t.output)) P_FAIL("target '%s': output tensor %d does not exist", t.name.c_str(), t.output);
    for (auto& k : t.source) {
      if (k.gen != Gen::None) {
        if (!check_tid(k.gen_tensor) || (k.gen == Gen::Gradient && !check_tid(k.gen_dest)))
          P_FAIL("target '%s': generator references an unknown tensor", t.name.c_str());
        continue;
      }
      if (!check_tid(k.write.tensor)) P_FAIL("target '%s': kernel writes unknown tensor %d", t.name.c_str(), k.write.tensor);
      for (auto& r : k.reads)
        if (!check_tid(r.tensor)) P_FAIL("target '%s': kernel reads unknown tensor %d", t.name.c_str(), r.tensor);
      if (k.result < 1 || k.result > k.nregs) P_FAIL("target '%s': kernel result register out of range", t.name.c_str());
      for (auto& g : k.custom_grad) {  // gradient placeholders: -t for an existing tensor t
        auto ok = [&](int tid) { return check_tid(tid < 0 ? -tid : tid); };
        if (!ok(g.write.tensor)) P_FAIL("target '%s': custom gradient writes unknown tensor %d", t.name.c_str(), g.write.tensor);
        for (auto& r : g.reads)
          if (!ok(r.tensor)) P_FAIL("target '%s': custom gradient reads unknown tensor %d", t.name.c_str(), r.tensor);
        if (g.result < 1 || g.result > g.nregs) P_FAIL("target '%s': custom gradient result register out of range", t.name.c_str());
      }
    }
  }
  return EG_OK;
}

// ------------------------------------------------------------------------------ types

std::vector<Ty> infer_types(const Kernel& k) {
  std::vector<Ty> ty(k.nregs + 1, Ty::None);
  auto set = [&](int r, Ty t) {
    if (r >= 1 && r <= k.nregs) ty[r] = t;
  };
  for (auto& lp : k.loops) set(lp.reg, Ty::Index);
  for (auto& s : k.setup) set(s.res, Ty::Index);
  for (auto& r : k.reads) set(r.reg, Ty::Scalar);
  std::vector<const Instr*> order;
  for (auto& ins : k.index_instrs) order.push_back(&ins);
  for (auto& ins : k.instrs) order.push_back(&ins);
  for (const Instr* pins : order) {
    const Instr& ins = *pins;
    Ty t = Ty::Scalar;
    switch (ins.kind) {
      case IK::Scalar: t = Ty::Scalar; break;
      case IK::Index: case IK::Shape: case IK::Len: case IK::ShapeLen: case IK::Epoch: case IK::IndexDiv:
      case IK::Mod: case IK::Wrap: case IK::ToIndex: t = Ty::Index; break;
      case IK::Boolean: case IK::Eq: case IK::Lt: case IK::Le: case IK::And: case IK::Or: t = Ty::Boolean; break;
      case IK::Add: case IK::Sub: case IK::Mul: case IK::Negate:
        t = ins.args.empty() ? Ty::Scalar : ty[ins.args[0]];
        break;
      case IK::Select: t = ins.args.size() > 1 ? ty[ins.args[1]] : Ty::Scalar; break;
      default: t = Ty::Scalar; break;
    }
    set(ins.res, t);
  }
  return ty;
}

// ------------------------------------------------------------------------------ autodiff

namespace {

// derive(instrs, regs, gradRegs)  passes.nim:383-517.  Emits into `out`, allocating registers in k.
int derive_instrs(const std::vector<Instr>& instrs, Kernel& k, std::map<int, int>& grad, std::vector<Instr>& out) {
  auto emit = [&](IK kind, std::vector<int> args, double lit = 0) {
    Instr i;
    i.kind = kind;
    i.res = k.alloc();
    i.args = std::move(args);
    i.lit = lit;
    out.push_back(i);
    return i.res;
  };
  for (auto it = instrs.rbegin(); it != instrs.rend(); ++it) {
    const Instr& ins = *it;
    auto g_it = grad.find(ins.res);
    if (g_it == grad.end()) continue;
    const int g = g_it->second;
    const std::vector<int>& a = ins.args;
    std::vector<int> ga;
    switch (ins.kind) {
      case IK::Add: ga = {g, g}; break;                                        // 393-394
      case IK::Sub: ga = {g, emit(IK::Negate, {g})}; break;                    // 395-398
      case IK::Mul: {                                                          // 399-403
        int ga0 = emit(IK::Mul, {g, a[1]});
        int gb0 = emit(IK::Mul, {g, a[0]});
        ga = {ga0, gb0};
        break;
      }
      case IK::Div: {                                                          // 404-415
        int ga0 = emit(IK::Div, {g, a[1]});
        int sq_y = emit(IK::Mul, {a[1], a[1]});
        int div_g = emit(IK::Div, {g, sq_y});
        int neg_x = emit(IK::Negate, {a[0]});
        int gb0 = emit(IK::Mul, {neg_x, div_g});
        ga = {ga0, gb0};
        break;
      }
      case IK::Negate: ga = {emit(IK::Negate, {g})}; break;                    // 416-419
      case IK::Ln: case IK::Log10: case IK::Log2: {                            // 420-436
        const double base = ins.kind == IK::Ln ? 1.0 : (ins.kind == IK::Log10 ? std::log(10.0) : std::log(2.0));
        int den = a[0];
        if (base != 1.0) {
          int factor = emit(IK::Scalar, {}, base);
          den = emit(IK::Mul, {a[0], factor});
        }
        ga = {emit(IK::Div, {g, den})};
        break;
      }
      case IK::Log: {                                                          // 437-455
        int log_y = emit(IK::Ln, {a[1]});
        int mul = emit(IK::Mul, {a[0], log_y});
        int gx = emit(IK::Div, {g, mul});
        int log_x = emit(IK::Ln, {a[0]});
        int neg_log_x = emit(IK::Negate, {log_x});
        int log_y_sq = emit(IK::Mul, {log_y, log_y});
        int den = emit(IK::Mul, {a[1], log_y_sq});
        int num = emit(IK::Mul, {g, neg_log_x});
        int gy = emit(IK::Div, {num, den});
        ga = {gx, gy};
        break;
      }
      case IK::Exp: ga = {emit(IK::Mul, {g, ins.res})}; break;                 // 456-459
      case IK::Sin: {                                                          // 460-464
        int c = emit(IK::Cos, {a[0]});
        ga = {emit(IK::Mul, {c, g})};
        break;
      }
      case IK::Cos: {                                                          // 465-470
        int s = emit(IK::Sin, {a[0]});
        int ns = emit(IK::Negate, {s});
        ga = {emit(IK::Mul, {ns, g})};
        break;
      }
      case IK::Select: {                                                       // 471-476
        int zero = emit(IK::Scalar, {}, 0.0);
        int ga0 = emit(IK::Select, {a[0], g, zero});
        int gb0 = emit(IK::Select, {a[0], zero, g});
        ga = {0, ga0, gb0};
        break;
      }
      case IK::Sqrt: {                                                         // 477-484
        int two = emit(IK::Scalar, {}, 2.0);
        int den = emit(IK::Mul, {two, ins.res});
        ga = {emit(IK::Div, {g, den})};
        break;
      }
      case IK::Pow: {                                                          // 485-503
        int one = emit(IK::Scalar, {}, 1.0);
        int new_exp = emit(IK::Sub, {a[1], one});
        int pw = emit(IK::Pow, {a[0], new_exp});
        int pw_factor = emit(IK::Mul, {a[1], pw});
        int g_base = emit(IK::Mul, {g, pw_factor});
        int lg = emit(IK::Ln, {a[0]});
        int product = emit(IK::Mul, {ins.res, lg});
        int g_exp = emit(IK::Mul, {g, product});
        ga = {g_base, g_exp};
        break;
      }
      case IK::ToScalar: case IK::ToIndex: ga = {0}; break;                    // 504
      default: break;
    }
    if (ga.size() != a.size()) {                                               // 507-508
      set_error("Unable to derive %s", ik_name(ins.kind));
      return EG_ERR_UNSUPPORTED;
    }
    for (size_t i = 0; i < a.size(); ++i) {                                    // 510-517
      if (ga[i] == 0) continue;
      auto f = grad.find(a[i]);
      if (f != grad.end())
        f->second = emit(IK::Add, {f->second, ga[i]});
      else
        grad[a[i]] = ga[i];
    }
  }
  return EG_OK;
}

// Kernel-level deadCodeElim (passes.nim:306-317).
void dead_code_elim(Kernel& k) {
  std::vector<char> used(k.nregs + 1, 0);
  auto use = [&](int r) {
    if (r >= 1 && r <= k.nregs) used[r] = 1;
  };
  use(k.write.reg);
  for (auto& d : k.write.dims)
    for (auto& f : d.factors) use(f.first);
  std::vector<Instr> kept;
  for (auto it = k.instrs.rbegin(); it != k.instrs.rend(); ++it)
    if (used[it->res]) {
      kept.push_back(*it);
      for (int a : it->args) use(a);
    }
  std::reverse(kept.begin(), kept.end());
  k.instrs.swap(kept);
  std::vector<Op> reads;
  for (auto& r : k.reads)
    if (used[r.reg]) {
      reads.push_back(r);
      for (auto& d : r.dims)
        for (auto& f : d.factors) use(f.first);
    }
  k.reads.swap(reads);
  std::vector<Instr> kept_idx;
  for (auto it = k.index_instrs.rbegin(); it != k.index_instrs.rend(); ++it)
    if (used[it->res]) {
      kept_idx.push_back(*it);
      for (int a : it->args) use(a);
    }
  std::reverse(kept_idx.begin(), kept_idx.end());
  k.index_instrs.swap(kept_idx);
  std::vector<Loop> loops;
  for (auto& lp : k.loops)
    if (used[lp.reg]) {
      loops.push_back(lp);
      if (lp.has_bounds) {
        for (auto& f : lp.start.factors) use(f.first);
        for (auto& f : lp.stop.factors) use(f.first);
      }
    }
  k.loops.swap(loops);
  std::vector<Instr> setup;
  for (auto& s : k.setup)
    if (used[s.res]) setup.push_back(s);
  k.setup.swap(setup);
}

// derive(kernel, gradTensors)  passes.nim:519-549: one gradient kernel per read, in read order.
int derive_kernel(const Kernel& kernel, const std::map<int, int>& grad_tensors, std::vector<Kernel>& out) {
  Kernel base = kernel;
  std::map<int, int> grad;
  const int write_grad = base.alloc();
  Op gop;
  gop.tensor = grad_tensors.at(kernel.write.tensor);
  gop.reg = write_grad;
  gop.raw = kernel.write.raw;
  gop.dims = kernel.write.dims;
  base.reads.push_back(gop);
  grad[kernel.write.reg] = write_grad;
  std::vector<Instr> extra;
  int rc = derive_instrs(kernel.instrs, base, grad, extra);
  if (rc) return rc;
  base.instrs.insert(base.instrs.end(), extra.begin(), extra.end());
  for (auto& read : kernel.reads) {
    auto g = grad.find(read.reg);
    if (g == grad.end()) continue;
    Kernel gk = base;
    gk.result = g->second;
    gk.write.tensor = grad_tensors.at(read.tensor);
    gk.write.raw = read.raw;
    gk.write.dims = read.dims;
    gk.write.reg = g->second;
    dead_code_elim(gk);
    out.push_back(gk);
  }
  return EG_OK;
}

// generate (passes.nim:558-640), GenBackwards / GenGradient only.
int generate(Program& prog, Target& t) {
  std::vector<Kernel> ks = t.source;
  size_t i = 0;
  while (i < ks.size()) {
    if (ks[i].gen == Gen::Backwards) {
      std::map<int, int> grad_tensors;
      std::vector<Kernel> grads;
      const int loss = ks[i].gen_tensor;
      const int grad_loss = prog.alloc_tensor(TK::Result, "grad_loss");
      prog.shape_copy[grad_loss] = loss;
      {  // gradLoss{i} = 1 for i in 0 ..< len(loss)   passes.nim:575-606
        Kernel seed;
        const int r_val = seed.alloc(), r_it = seed.alloc(), r_len = seed.alloc();
        Instr len;
        len.kind = IK::Len;
        len.res = r_len;
        len.tensor = loss;
        seed.setup.push_back(len);
        Loop lp;
        lp.reg = r_it;
        lp.name = "i";
        lp.has_bounds = true;
        lp.stop.factors.push_back({r_len, 1});
        seed.loops.push_back(lp);
        Instr one;
        one.kind = IK::Scalar;
        one.res = r_val;
        one.lit = 1.0;
        seed.instrs.push_back(one);
        seed.result = r_val;
        seed.write.tensor = grad_loss;
        seed.write.reg = r_val;
        seed.write.raw = true;
        Lin idx;
        idx.factors.push_back({r_it, 1});
        seed.write.dims.push_back(idx);
        seed.is_seed = true;
        grads.push_back(seed);
      }
      grad_tensors[loss] = grad_loss;
      for (size_t j = i + 1; j < ks.size(); ++j)  // 608-612
        if (ks[j].gen == Gen::Gradient) {
          grad_tensors[ks[j].gen_tensor] = ks[j].gen_dest;
          prog.shape_copy[ks[j].gen_dest] = ks[j].gen_tensor;
        }
      for (size_t jj = i; jj-- > 0;) {  // 614-636
        const Kernel& k2 = ks[jj];
        if (k2.gen != Gen::None) continue;
        for (auto& read : k2.reads)
          if (!grad_tensors.count(read.tensor)) {
            const int gt = prog.alloc_tensor(TK::Result, "grad");
            prog.shape_copy[gt] = read.tensor;
            grad_tensors[read.tensor] = gt;
          }
        if (!grad_tensors.count(k2.write.tensor)) continue;  // does not influence the loss
        if (k2.has_custom_grad) {  // 626-634: the user's gradient kernels, last first, placeholders bound
          for (size_t g = k2.custom_grad.size(); g-- > 0;) {
            Kernel gk = k2.custom_grad[g];
            gk.has_custom_grad = false;
            gk.custom_grad.clear();
            auto bind = [&](Op& op) {
              if (op.tensor >= 0) return true;
              auto it = grad_tensors.find(-op.tensor);
              if (it == grad_tensors.end()) return false;
              op.tensor = it->second;
              return true;
            };
            bool ok = bind(gk.write);
            for (auto& rd : gk.reads) ok = bind(rd) && ok;
            if (!ok) {
              set_error("custom gradient refers to the gradient of a tensor the kernel does not touch");
              return EG_ERR_INVALID;
            }
            grads.push_back(gk);
          }
          continue;
        }
        int rc = derive_kernel(k2, grad_tensors, grads);
        if (rc) return rc;
      }
      ks.erase(ks.begin() + i);
      ks.insert(ks.begin() + i, grads.begin(), grads.end());
      i += grads.size();
    } else if (ks[i].gen == Gen::Gradient) {
      ks.erase(ks.begin() + i);  // 641-642
    } else {
      ++i;
    }
  }
  t.all.swap(ks);
  return EG_OK;
}

// deadKernelElim (passes.nim:331-350)
void dead_kernel_elim(const Program& prog, Target& t) {
  std::vector<char> used(prog.tensors.size(), 0);
  for (size_t i = 1; i < prog.tensors.size(); ++i)
    if (prog.tensors[i].kind != TK::Result) used[i] = 1;
  if (t.output) used[t.output] = 1;
  std::vector<int> live;
  for (int i = (int)t.all.size() - 1; i >= 0; --i) {
    const Kernel& k = t.all[i];
    if (used[k.write.tensor]) {
      for (auto& r : k.reads) used[r.tensor] = 1;
      live.push_back(i);
    }
  }
  std::reverse(live.begin(), live.end());
  t.live.swap(live);
  t.first_update = -1;
  for (size_t p = 0; p < t.live.size(); ++p)
    // persistent state: parameters and optimizer caches (adam.m / adam.v, base.nim:45)
    if (prog.tensors[t.all[t.live[p]].write.tensor].kind == TK::Param ||
        prog.tensors[t.all[t.live[p]].write.tensor].kind == TK::Cache) {
      t.first_update = (int)p;
      break;
    }
}

}  // namespace

int compile_program(Program& prog) {
  for (auto& t : prog.targets) {
    int rc = generate(prog, t);
    if (rc) return rc;
  }
  // grad tensors allocated by one target enlarge prog.tensors: eliminate after all are generated
  for (auto& t : prog.targets) dead_kernel_elim(prog, t);
  if (prog.f64) {
    std::function<void(Kernel&)> mark = [&](Kernel& k) {
      k.f64 = true;
      for (auto& c : k.custom_grad) mark(c);
    };
    for (auto& t : prog.targets) {
      for (auto& k : t.source) mark(k);
      for (auto& k : t.all) mark(k);
    }
  }
  return EG_OK;
}

// ------------------------------------------------------------------------------ shapes

static long prod(const std::vector<long>& s) {
  long p = 1;
  for (long v : s) p *= v;
  return p;
}

// Host evaluation of shape()/len()/index arithmetic (kernel setup, shape-constraint setup).
// later_tensor / later: instructions that need the shape of `later_tensor` while it is unknown, and
// everything computed from them, are skipped and their registers collected in `later` (a kernel
// whose loop bounds name the shape of the tensor it writes, see infer_kernel).
static int eval_host_instrs(const std::vector<Instr>& instrs, const Shapes& shapes, long epoch, std::map<int, long>& vals,
                            int later_tensor = 0, std::set<int>* later = nullptr) {
  for (auto& s : instrs) {
    long v = 0;
    auto arg = [&](size_t i) -> long {
      auto it = i < s.args.size() ? vals.find(s.args[i]) : vals.end();
      return it == vals.end() ? 0 : it->second;
    };
    if (later) {
      bool skip = (s.kind == IK::Shape || s.kind == IK::Len || s.kind == IK::ShapeLen) && s.tensor == later_tensor &&
                  !shapes.count(s.tensor);
      for (int a : s.args) skip = skip || later->count(a);
      if (skip) {
        later->insert(s.res);
        continue;
      }
    }
    switch (s.kind) {
      case IK::Shape: {
        auto it = shapes.find(s.tensor);
        if (it == shapes.end()) {
          set_error("shape of tensor %d is needed before it is known", s.tensor);
          return EG_ERR_SHAPE;
        }
        int d = s.dim < 0 ? s.dim + (int)it->second.size() : s.dim;
        if (d < 0 || d >= (int)it->second.size()) {
          set_error("shape()[%d] out of range for a rank-%zu tensor", s.dim, it->second.size());
          return EG_ERR_SHAPE;
        }
        v = it->second[d];
        break;
      }
      case IK::Len: case IK::ShapeLen: {
        auto it = shapes.find(s.tensor);
        if (it == shapes.end()) {
          set_error("shape of tensor %d is needed before it is known", s.tensor);
          return EG_ERR_SHAPE;
        }
        v = s.kind == IK::Len ? prod(it->second) : (long)it->second.size();
        break;
      }
      case IK::Index: v = (long)s.lit; break;
      case IK::Epoch: v = epoch; break;
      case IK::Add: v = arg(0) + arg(1); break;
      case IK::Sub: v = arg(0) - arg(1); break;
      case IK::Mul: v = arg(0) * arg(1); break;
      case IK::Negate: v = -arg(0); break;
      case IK::IndexDiv: v = arg(1) ? arg(0) / arg(1) : 0; break;  // sdiv: truncation toward zero
      case IK::Mod: v = arg(1) ? arg(0) % arg(1) : 0; break;
      default:
        set_error("unsupported host instruction %s", ik_name(s.kind));
        return EG_ERR_UNSUPPORTED;
    }
    vals[s.res] = v;
  }
  return EG_OK;
}

// withShape / reshape dims of `tid`, if they can be evaluated now.  Registers of the dims come from
// the constraint's own host instructions (shapesetup) or, failing that, from the writing kernel's.
static bool user_shape(const Program& prog, int tid, const Shapes& shapes, long epoch, const std::map<int, long>& kernel_vals,
                       std::vector<long>& shape_out) {
  auto sd = prog.shape_dims.find(tid);
  if (sd == prog.shape_dims.end()) return false;
  std::map<int, long> vals = kernel_vals;
  auto ss = prog.shape_setup.find(tid);
  if (ss != prog.shape_setup.end() && eval_host_instrs(ss->second, shapes, epoch, vals) != EG_OK) {
    eg::clear_error();
    return false;
  }
  shape_out.clear();
  for (auto& l : sd->second) {
    long v = l.constant;
    for (auto& f : l.factors) {
      auto it = vals.find(f.first);
      if (it == vals.end()) return false;
      v += f.second * it->second;
    }
    shape_out.push_back(v);
  }
  return true;
}

int infer_kernel(const Program& prog, const Kernel& k, Shapes& shapes, long epoch, KernelInfo& out) {
  out = KernelInfo();
  std::map<int, long>& vals = out.vals;
  // Explicit loop bounds may name the shape of the very tensor the kernel writes
  // (`res[x] ++= ... | (x in 0..<res.shape[0])`, tests/test_model.nim:99-107): the reference's
  // constraint solver gets that shape from the reads; the bounds are applied once it is known.
  std::set<int> later;
  {
    int rc = eval_host_instrs(k.setup, shapes, epoch, vals, k.write.tensor, &later);
    if (rc) return rc;
  }
  std::set<int> idx_regs;  // computed indices never bound a loop
  for (auto& ins : k.index_instrs) idx_regs.insert(ins.res);
  auto lin_const = [&](const Lin& l) {
    long v = l.constant;
    for (auto& f : l.factors) v += f.second * vals.at(f.first);
    return v;
  };
  std::map<int, std::pair<long, long>> bounds;
  std::set<int> loop_regs;
  auto known = [&](const Lin& l) {
    for (auto& f : l.factors)
      if (later.count(f.first)) return false;
    return true;
  };
  std::set<int> postponed;  // loops whose explicit bounds wait for the written tensor's shape
  for (auto& lp : k.loops) {
    loop_regs.insert(lp.reg);
    if (!lp.has_bounds) continue;
    if (known(lp.start) && known(lp.stop))
      bounds[lp.reg] = {lin_const(lp.start), lin_const(lp.stop)};
    else
      postponed.insert(lp.reg);
  }
  // user constraints (withShape / copyShape, parser.nim:683-697) fix the written tensor's shape
  // before its loops are bounded: PriorityUser outranks the inferred constraints
  if (!shapes.count(k.write.tensor)) {
    auto sc = prog.shape_copy.find(k.write.tensor);
    std::vector<long> shp;
    if (user_shape(prog, k.write.tensor, shapes, epoch, vals, shp)) {
      shapes[k.write.tensor] = shp;
    } else if (sc != prog.shape_copy.end() && shapes.count(sc->second)) {
      shapes[k.write.tensor] = shapes[sc->second];
    }
  }
  std::vector<const Op*> ops;
  for (auto& r : k.reads) ops.push_back(&r);
  ops.push_back(&k.write);
  // inferLoopBounds (passes.nim:986-1010): first op that indexes a dimension with the bare iterator
  for (const Op* op : ops) {
    auto it = shapes.find(op->tensor);
    if (it == shapes.end()) continue;
    const std::vector<long>& shp = it->second;
    if (!op->raw && op->dims.size() != shp.size()) {
      set_error("tensor %d has rank %zu but is indexed with %zu dimensions", op->tensor, shp.size(), op->dims.size());
      return EG_ERR_SHAPE;
    }
    for (size_t d = 0; d < op->dims.size(); ++d) {
      const int r = op->dims[d].only_register();
      // (loops with explicit bounds never take part: inferLoopBounds skips them, passes.nim:1030-1038)
      if (r && loop_regs.count(r) && !bounds.count(r) && !postponed.count(r)) bounds[r] = {0, op->raw ? prod(shp) : shp[d]};
    }
  }
  // iterators that never appear bare (y in img[n, y+dy, ...]): max(index) = extent - 1
  // (the ShapeLinear solve of passes.nim:1420-1436 specialised to one unknown per dimension)
  bool progress = true;
  while (progress && bounds.size() < loop_regs.size()) {
    progress = false;
    for (const Op* op : ops) {
      auto it = shapes.find(op->tensor);
      if (it == shapes.end() || op->raw) continue;
      for (size_t d = 0; d < op->dims.size(); ++d) {
        const Lin& lin = op->dims[d];
        bool computed = false;
        for (auto& f : lin.factors)
          if (idx_regs.count(f.first)) computed = true;
        if (computed) continue;
        int unknown = 0, n_unknown = 0;
        for (auto& f : lin.factors)
          if (!bounds.count(f.first) && !vals.count(f.first)) {
            unknown = f.first;
            ++n_unknown;
          }
        if (n_unknown != 1 || lin.factor_of(unknown) <= 0 || !loop_regs.count(unknown)) continue;
        // several reads of one tensor that differ only in the constant (image[x], image[x + 1],
        // image[x + 2]): the largest offset decides (simplifyMaxIndex, passes.nim:1040-1057)
        long rest = lin.constant;
        if (op != &k.write)
          for (auto& other : k.reads)
            if (other.tensor == op->tensor && !other.raw && other.dims.size() == op->dims.size()) {
              Lin a = other.dims[d], b = lin;
              a.constant = b.constant = 0;
              if (a == b && other.dims[d].constant > rest) rest = other.dims[d].constant;
            }
        for (auto& f : lin.factors) {
          if (f.first == unknown) continue;
          if (vals.count(f.first))
            rest += f.second * vals[f.first];
          else
            rest += f.second * (f.second > 0 ? bounds[f.first].second - 1 : bounds[f.first].first);
        }
        const long fu = lin.factor_of(unknown);
        long num = it->second[d] - 1 - rest;
        long ext = num >= 0 ? num / fu + 1 : 0;
        bounds[unknown] = {0, ext};
        progress = true;
      }
    }
  }
  for (auto& lp : k.loops)
    if (!bounds.count(lp.reg)) {
      set_error("unable to infer the bounds of loop '%s'", lp.name.c_str());
      return EG_ERR_SHAPE;
    }
  // shape of the written tensor
  const int wt = k.write.tensor;
  if (!shapes.count(wt)) {
    auto sd = prog.shape_dims.find(wt);
    auto sc = prog.shape_copy.find(wt);
    if (sd != prog.shape_dims.end()) {
      std::vector<long> shp;
      if (!user_shape(prog, wt, shapes, epoch, vals, shp)) {
        set_error("withShape of tensor %d uses a value that is not known yet", wt);
        return EG_ERR_SHAPE;
      }
      shapes[wt] = shp;
    } else if (sc != prog.shape_copy.end() && shapes.count(sc->second)) {
      shapes[wt] = shapes[sc->second];
    } else if (k.write.raw) {
      if (k.reads.size() == 1) {  // ShapeCopy, passes.nim:1061-1067
        shapes[wt] = shapes.at(k.reads[0].tensor);
      } else {
        set_error("shape of tensor %d is under-constrained (raw write with %zu reads; use copyShape/withShape)", wt,
                  k.reads.size());
        return EG_ERR_SHAPE;
      }
    } else {
      std::vector<long> shp;
      for (auto& lin : k.write.dims) {
        for (auto& f : lin.factors)
          if (idx_regs.count(f.first)) {
            set_error("shape of tensor %d is under-constrained (computed write index; use withShape)", wt);
            return EG_ERR_SHAPE;
          }
        long hi = lin.constant;
        for (auto& f : lin.factors) {
          if (vals.count(f.first))
            hi += f.second * vals[f.first];
          else
            hi += f.second * (f.second > 0 ? bounds[f.first].second - 1 : bounds[f.first].first);
        }
        shp.push_back(hi + 1);
      }
      shapes[wt] = shp;
    }
  } else if (!k.write.raw && k.write.dims.size() != shapes[wt].size()) {
    set_error("tensor %d has rank %zu but is written with %zu dimensions", wt, shapes[wt].size(), k.write.dims.size());
    return EG_ERR_SHAPE;
  }
  if (!later.empty()) {  // the written tensor has its shape now: the postponed bounds
    int rc = eval_host_instrs(k.setup, shapes, epoch, vals);
    if (rc) return rc;
    for (auto& lp : k.loops)
      if (lp.has_bounds) bounds[lp.reg] = {lin_const(lp.start), lin_const(lp.stop)};
  }
  out.bounds.clear();
  for (auto& lp : k.loops) out.bounds.push_back(bounds[lp.reg]);
  out.ok = true;
  return EG_OK;
}

// ------------------------------------------------------------------------------ printing

static std::string lin_text(const Lin& l) {
  std::string s;
  char buf[64];
  bool first = true;
  for (auto& f : l.factors) {
    if (f.second == 1)
      snprintf(buf, sizeof(buf), "%sr%d", first ? "" : " + ", f.first);
    else
      snprintf(buf, sizeof(buf), "%s%ld*r%d", first ? "" : " + ", f.second, f.first);
    s += buf;
    first = false;
  }
  if (l.constant != 0 || first) {
    snprintf(buf, sizeof(buf), "%s%ld", first ? "" : " + ", l.constant);
    s += buf;
  }
  return s;
}

static std::string op_text(const Op& op) {
  std::string s = "t" + std::to_string(op.tensor) + (op.raw ? "{" : "[");
  for (size_t i = 0; i < op.dims.size(); ++i) s += (i ? ", " : "") + lin_text(op.dims[i]);
  s += op.raw ? "}" : "]";
  return s;
}

std::string to_text(const Kernel& k) {
  std::string s = op_text(k.write) + " += r" + std::to_string(k.result) + " | loops(";
  for (size_t i = 0; i < k.loops.size(); ++i) s += (i ? "," : "") + k.loops[i].name + ":r" + std::to_string(k.loops[i].reg);
  s += ") reads(";
  for (size_t i = 0; i < k.reads.size(); ++i) s += (i ? ", " : "") + ("r" + std::to_string(k.reads[i].reg) + "=" + op_text(k.reads[i]));
  s += ")";
  auto list = [&](const char* title, const std::vector<Instr>& instrs) {
    s += std::string(" ") + title + "(";
    for (size_t i = 0; i < instrs.size(); ++i) {
      const Instr& ins = instrs[i];
      s += (i ? "; " : "") + ("r" + std::to_string(ins.res) + "=" + ik_name(ins.kind));
      for (int a : ins.args) s += " r" + std::to_string(a);
      if (ins.kind == IK::Scalar || ins.kind == IK::Index) {
        char buf[48];
        snprintf(buf, sizeof(buf), " %g", ins.lit);
        s += buf;
      }
    }
    s += ")";
  };
  if (!k.index_instrs.empty()) list("index", k.index_instrs);
  list("instrs", k.instrs);
  return s;
}

}  // namespace kd
}  // namespace eg
