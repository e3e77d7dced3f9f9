#include "epilogue.hpp"

#include <algorithm>
#include <map>
#include <set>

#include "../eg_internal.hpp"
#include "codegen.hpp"

namespace eg {
namespace kd {
namespace {

long prodv(const std::vector<long>& v) {
  long p = 1;
  for (long x : v) p *= x;
  return p;
}

// flat element index of an operand as constant + sum(factor * register)
bool flat_index(const Op& op, const std::vector<long>& shape, std::map<int, long>& factors, long& constant) {
  factors.clear();
  constant = 0;
  if (op.raw) {
    if (op.dims.size() != 1) return false;
    constant = op.dims[0].constant;
    for (auto& f : op.dims[0].factors) factors[f.first] += f.second;
  } else {
    if (op.dims.size() != shape.size()) return false;
    long stride = 1;
    for (size_t d = shape.size(); d-- > 0;) {
      constant += stride * op.dims[d].constant;
      for (auto& f : op.dims[d].factors) factors[f.first] += stride * f.second;
      stride *= shape[d];
    }
  }
  for (auto it = factors.begin(); it != factors.end();)
    it = it->second == 0 ? factors.erase(it) : std::next(it);
  return true;
}

}  // namespace

bool epilogue_capable(const Kernel& k, const KernelInfo& info, const Shapes& shapes, int c_tensor, long M, long N) {
  if (!info.ok || k.is_seed || k.gen != Gen::None || !k.index_instrs.empty()) return false;
  std::vector<int> indep, red;
  bool scatter;
  split_loops(k, indep, red, scatter);
  if (scatter || !red.empty() || k.loops.empty()) return false;
  if (k.write.tensor == c_tensor) return false;
  bool reads_c = false;
  std::set<int> tensors;
  for (auto& rd : k.reads) {
    if (rd.tensor == c_tensor) reads_c = true;
    if (rd.tensor == k.write.tensor) return false;
    tensors.insert(rd.tensor);
  }
  tensors.insert(k.write.tensor);
  tensors.erase(c_tensor);
  if (!reads_c || (int)tensors.size() > 6) return false;

  std::map<int, long> ref, cur;
  long ref_c = 0, cur_c = 0;
  bool have_ref = false;
  auto check = [&](const Op& op) {
    auto it = shapes.find(op.tensor);
    if (it == shapes.end() || prodv(it->second) != M * N) return false;
    if (!flat_index(op, it->second, cur, cur_c)) return false;
    if (!have_ref) {
      ref = cur;
      ref_c = cur_c;
      have_ref = true;
      return true;
    }
    return cur == ref && cur_c == ref_c;
  };
  for (auto& rd : k.reads)
    if (!check(rd)) return false;
  if (!check(k.write)) return false;
  auto cs = shapes.find(c_tensor);
  if (cs == shapes.end() || cs->second.size() != 2 || cs->second[0] != M || cs->second[1] != N) return false;

  // the flat index must be a mixed-radix number over all loops: 0 .. M*N-1, each exactly once
  if (ref_c != 0 || ref.size() != k.loops.size()) return false;
  std::vector<std::pair<long, long>> digits;  // (factor, extent)
  for (size_t l = 0; l < k.loops.size(); ++l) {
    auto f = ref.find(k.loops[l].reg);
    if (f == ref.end() || f->second <= 0) return false;
    if (info.bounds[l].first != 0 || info.bounds[l].second <= 0) return false;
    digits.push_back({f->second, info.bounds[l].second});
  }
  std::sort(digits.begin(), digits.end());
  long expect = 1;
  for (auto& d : digits) {
    if (d.first != expect) return false;
    expect *= d.second;
  }
  return expect == M * N;
}

bool only_predicate_uses(const Kernel& k, int tensor, PredicateSpec& out) {
  std::set<int> regs;
  for (auto& rd : k.reads)
    if (rd.tensor == tensor) regs.insert(rd.reg);
  if (regs.empty() || regs.count(k.result) || k.write.tensor == tensor) return false;
  std::map<int, const Instr*> def;
  for (auto& ins : k.instrs) def[ins.res] = &ins;
  bool any = false;
  for (auto& ins : k.instrs) {
    int uses = 0;
    for (int a : ins.args) uses += regs.count(a) ? 1 : 0;
    if (!uses) continue;
    if (uses != 1 || ins.args.size() != 2 || (ins.kind != IK::Le && ins.kind != IK::Lt && ins.kind != IK::Eq)) return false;
    const bool literal_first = regs.count(ins.args[1]) != 0;
    auto d = def.find(ins.args[literal_first ? 0 : 1]);
    if (d == def.end() || d->second->kind != IK::Scalar) return false;
    PredicateSpec spec;
    spec.kind = ins.kind;
    spec.literal = (double)(float)d->second->lit;
    spec.literal_first = literal_first;
    if (any && !(spec == out)) return false;
    out = spec;
    any = true;
  }
  for (auto& ins : k.index_instrs)
    for (int a : ins.args)
      if (regs.count(a)) return false;
  return any;
}

namespace {
std::string predicate_expression(const PredicateSpec& p, const std::string& x) {
  const std::string lit = f32_literal(p.literal);
  const char* op = p.kind == IK::Le ? " <= " : (p.kind == IK::Lt ? " < " : " == ");
  return p.literal_first ? lit + op + x : x + op + lit;
}
}  // namespace

const char* const kNoRowProduct = "  static constexpr int RD_N = 0, RD_W = 0, RD_OUT = 0, RD_BIAS = -1, RD_LDW = 0, RD_LDO = 0;\n";

int generate_epilogue(const Kernel& k, const KernelInfo& info, const Shapes& shapes, int c_tensor, bool store_c,
                      bool accumulate, EpilogueSpec& out, const std::map<int, PredicateSpec>* pred_reads,
                      const PredicateSpec* pred_write) {
  out = EpilogueSpec();
  out.struct_name = "EgEpi";
  std::set<int> tensors;
  for (auto& rd : k.reads)
    if (rd.tensor != c_tensor) tensors.insert(rd.tensor);
  tensors.insert(k.write.tensor);
  out.operands.assign(tensors.begin(), tensors.end());
  if (pred_write) out.operands.push_back(c_tensor);  // the bits of the contraction result: a.epi[PRED]
  auto is_bits = [&](int t) { return pred_reads && pred_reads->count(t) != 0; };

  // operands other than the contraction result, in load order: x[i]
  std::vector<int> loads;
  for (int t : out.operands)
    if (t != k.write.tensor && !(pred_write && t == c_tensor)) loads.push_back(t);
  if (accumulate) loads.push_back(k.write.tensor);
  const int nx = loads.empty() ? 1 : (int)loads.size();
  auto slot_of = [&](int t) {
    for (size_t i = 0; i < loads.size(); ++i)
      if (loads[i] == t) return (int)i;
    return -1;
  };
  auto operand_index = [&](int t) {
    for (size_t i = 0; i < out.operands.size(); ++i)
      if (out.operands[i] == t) return (int)i;
    return -1;
  };
  const std::string nxs = std::to_string(nx);
  std::string c = "struct EgEpi {\n  static constexpr bool ACTIVE = true;\n  static constexpr int NX = " + nxs + ";\n"
                  "  static constexpr bool STORE_C = " + std::string(store_c ? "true" : "false") + ";\n"
                  "  static constexpr int OUT = " + std::to_string(operand_index(k.write.tensor)) + ";\n"
                  "  static constexpr int PRED = " + std::to_string(pred_write ? operand_index(c_tensor) : -1) + ";\n" +
                  std::string(kNoRowProduct) +
                  "  __device__ __forceinline__ static bool predicate(float v) { return " +
                  (pred_write ? predicate_expression(*pred_write, "v") : std::string("false")) + "; }\n"
                  "  __device__ __forceinline__ static void prefetch(const eg::gemm::GemmArgs& a, long idx, float (&x)[" + nxs + "]) {\n";
  for (size_t i = 0; i < loads.size(); ++i) {
    const std::string ep = "a.epi[" + std::to_string(operand_index(loads[i])) + "]";
    if (is_bits(loads[i]))  // one bit per element: 1.0f / 0.0f, turned back into the comparison's result in compute()
      c += "    x[" + std::to_string(i) + "] = (float)((((const unsigned*)" + ep + ")[idx >> 5] >> (idx & 31)) & 1u);\n";
    else
      c += "    x[" + std::to_string(i) + "] = ((const float*)" + ep + ")[idx];\n";
  }
  c += "  }\n  __device__ __forceinline__ static void prefetch4(const eg::gemm::GemmArgs& a, long idx, eg::gemm::f32x4 (&x)[" + nxs + "]) {\n";
  for (size_t i = 0; i < loads.size(); ++i) {
    const std::string ep = "a.epi[" + std::to_string(operand_index(loads[i])) + "]";
    if (is_bits(loads[i])) {  // idx is a multiple of 4: the four bits sit in one word
      c += "    { const unsigned w = ((const unsigned*)" + ep + ")[idx >> 5] >> (idx & 31);\n";
      for (int e = 0; e < 4; ++e)
        c += "      x[" + std::to_string(i) + "][" + std::to_string(e) + "] = (float)((w >> " + std::to_string(e) + ") & 1u);\n";
      c += "    }\n";
    } else {
      c += "    x[" + std::to_string(i) + "] = *reinterpret_cast<const eg::gemm::f32x4*>((const float*)" + ep + " + idx);\n";
    }
  }
  c += "  }\n  __device__ __forceinline__ static float compute(const eg::gemm::GemmArgs& a, long idx, float v, const float (&x)[" + nxs + "]) {\n";
  // loop registers from the flat index (dead code unless the expression uses an iterator value)
  std::map<int, long> fac;
  long cst = 0;
  flat_index(k.write, shapes.at(k.write.tensor), fac, cst);
  for (size_t l = 0; l < k.loops.size(); ++l) {
    const int r = k.loops[l].reg;
    c += "    const long r" + std::to_string(r) + " = (idx / " + std::to_string(fac.at(r)) + "L) % " +
         std::to_string(info.bounds[l].second) + "L;\n";
  }
  for (auto& s : k.setup) c += "    const long r" + std::to_string(s.res) + " = " + std::to_string(info.vals.at(s.res)) + "L;\n";
  for (auto& rd : k.reads)
    c += "    const float r" + std::to_string(rd.reg) + " = " +
         (rd.tensor == c_tensor ? std::string("v") : "x[" + std::to_string(slot_of(rd.tensor)) + "]") + ";\n";
  const std::vector<Ty> ty = infer_types(k);
  std::set<int> bit_regs;  // registers that hold a predicate bit (1.0f / 0.0f) instead of the tensor's value
  for (auto& rd : k.reads)
    if (is_bits(rd.tensor)) bit_regs.insert(rd.reg);
  for (auto& ins : k.instrs) {
    const Ty t = ty[ins.res];
    const char* ctype = t == Ty::Scalar ? "float" : (t == Ty::Index ? "long" : "bool");
    bool on_bit = false;
    for (int a_ : ins.args)
      if (bit_regs.count(a_)) {  // the comparison only_predicate_uses() found: its answer is the stored bit
        c += "    const bool r" + std::to_string(ins.res) + " = r" + std::to_string(a_) + " != 0.0f;\n";
        on_bit = true;
        break;
      }
    if (on_bit) continue;
    std::string special;
    if (ins.kind == IK::Epoch) {
      special = "a.epi_ep";
    } else if (ins.kind == IK::Shape || ins.kind == IK::Len || ins.kind == IK::ShapeLen) {
      const std::vector<long>& shp = shapes.at(ins.tensor);
      long val = 0;
      if (ins.kind == IK::Len) val = prodv(shp);
      else if (ins.kind == IK::ShapeLen) val = (long)shp.size();
      else {
        const int d = ins.dim < 0 ? ins.dim + (int)shp.size() : ins.dim;
        val = (d >= 0 && d < (int)shp.size()) ? shp[d] : 0;
      }
      special = std::to_string(val) + "L";
    }
    c += std::string("    const ") + ctype + " r" + std::to_string(ins.res) + " = " + instr_expression(ins, special, "r") + ";\n";
  }
  const std::string val = "(0.0f + r" + std::to_string(k.result) + ")";
  c += "    return " + (accumulate ? "x[" + std::to_string(slot_of(k.write.tensor)) + "] + " + val : val) + ";\n  }\n};\n";
  out.struct_code = c;
  return EG_OK;
}

}  // namespace kd
}  // namespace eg
