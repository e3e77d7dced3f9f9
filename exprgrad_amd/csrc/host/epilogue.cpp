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

int generate_epilogue(const Kernel& k, const KernelInfo& info, const Shapes& shapes, int c_tensor, bool store_c,
                      bool accumulate, EpilogueSpec& out) {
  out = EpilogueSpec();
  out.struct_name = "EgEpi";
  std::set<int> tensors;
  for (auto& rd : k.reads)
    if (rd.tensor != c_tensor) tensors.insert(rd.tensor);
  tensors.insert(k.write.tensor);
  out.operands.assign(tensors.begin(), tensors.end());

  // operands other than the contraction result, in load order: x[i]
  std::vector<int> loads;
  for (int t : out.operands)
    if (t != k.write.tensor) loads.push_back(t);
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
                  "  __device__ __forceinline__ static void prefetch(const eg::gemm::GemmArgs& a, long idx, float (&x)[" + nxs + "]) {\n";
  for (size_t i = 0; i < loads.size(); ++i)
    c += "    x[" + std::to_string(i) + "] = ((const float*)a.epi[" + std::to_string(operand_index(loads[i])) + "])[idx];\n";
  c += "  }\n  __device__ __forceinline__ static void prefetch4(const eg::gemm::GemmArgs& a, long idx, eg::gemm::f32x4 (&x)[" + nxs + "]) {\n";
  for (size_t i = 0; i < loads.size(); ++i)
    c += "    x[" + std::to_string(i) + "] = *reinterpret_cast<const eg::gemm::f32x4*>((const float*)a.epi[" +
         std::to_string(operand_index(loads[i])) + "] + idx);\n";
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
  for (auto& ins : k.instrs) {
    const Ty t = ty[ins.res];
    const char* ctype = t == Ty::Scalar ? "float" : (t == Ty::Index ? "long" : "bool");
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
