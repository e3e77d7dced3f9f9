// Pattern matching of lowered kernels against the hand-written library: contraction, bias add,
// convolution and its two gradients.
#include "model_types.hpp"


namespace eg {
namespace model {

bool bare2(const Op& op, int& r0, int& r1) {
  if (op.raw || op.dims.size() != 2) return false;
  r0 = op.dims[0].only_register();
  r1 = op.dims[1].only_register();
  return r0 && r1 && r0 != r1;
}

int loop_index(const Kernel& k, int reg) {
  for (size_t l = 0; l < k.loops.size(); ++l)
    if (k.loops[l].reg == reg) return (int)l;
  return -1;
}

// c[i,j] += a(i,k) * b(k,j)      base.nim:27-28 and its two derived forms (passes.nim:519-549)
bool match_gemm(const Kernel& k, GemmMatch& m) {
  if (!k.index_instrs.empty()) return false;
  if (k.instrs.size() != 1 || k.instrs[0].kind != IK::Mul || k.result != k.instrs[0].res) return false;
  if (k.reads.size() != 2 || k.loops.size() != 3 || !k.setup.empty()) return false;
  for (auto& lp : k.loops)
    if (lp.has_bounds) return false;
  const std::vector<int>& args = k.instrs[0].args;
  if (!((args[0] == k.reads[0].reg && args[1] == k.reads[1].reg) || (args[0] == k.reads[1].reg && args[1] == k.reads[0].reg)))
    return false;
  int wi, wj;
  if (!bare2(k.write, wi, wj)) return false;
  int kk = 0;
  for (auto& lp : k.loops)
    if (lp.reg != wi && lp.reg != wj) kk = lp.reg;
  if (!kk || loop_index(k, wi) < 0 || loop_index(k, wj) < 0) return false;
  int r[2][2];
  if (!bare2(k.reads[0], r[0][0], r[0][1]) || !bare2(k.reads[1], r[1][0], r[1][1])) return false;
  auto is = [](const int* p, int x, int y) { return (p[0] == x && p[1] == y) || (p[0] == y && p[1] == x); };
  for (int a = 0; a < 2; ++a) {
    const int b = 1 - a;
    if (is(r[a], wi, kk) && is(r[b], kk, wj)) {
      m.a_read = a;
      m.b_read = b;
      m.trans_a = r[a][0] == kk;
      m.trans_b = r[b][0] == wj;
      m.li = loop_index(k, wi);
      m.lj = loop_index(k, wj);
      m.lk = loop_index(k, kk);
      return true;
    }
  }
  return false;
}

// out[y,x] += bias[x] on the tensor the preceding contraction wrote   dnn.nim:22-24
bool match_bias(const Kernel& k, int tensor) {
  if (!k.index_instrs.empty()) return false;
  if (!k.instrs.empty() || k.reads.size() != 1 || k.result != k.reads[0].reg || k.write.tensor != tensor) return false;
  if (k.loops.size() != 2 || !k.setup.empty()) return false;
  for (auto& lp : k.loops)
    if (lp.has_bounds) return false;
  int wi, wj;
  if (!bare2(k.write, wi, wj)) return false;
  const Op& b = k.reads[0];
  return !b.raw && b.dims.size() == 1 && b.dims[0].only_register() == wj;
}

// out[n,y,x,f] += img[n,y+dy,x+dx,c] * flt[f,dy,dx,c]   dnn.nim:45-49 (4-D) / conv2.nim:128-132 (3-D)
// and the two kernels derive (passes.nim:383-549) makes of it, which are the same loop nest with
// another of the three tensors written:
//   gimg[n,y+dy,x+dx,c] += gout[n,y,x,f] * flt[f,dy,dx,c]      gflt[f,dy,dx,c] += gout[n,y,x,f] * img[n,y+dy,x+dx,c]
bool match_conv(const Kernel& k, ConvMatch& m) {
  if (!k.index_instrs.empty()) return false;
  if (k.instrs.size() != 1 || k.instrs[0].kind != IK::Mul || k.result != k.instrs[0].res || k.reads.size() != 2) return false;
  if (!k.setup.empty() || k.write.raw || k.reads[0].raw || k.reads[1].raw) return false;
  const std::vector<int>& args = k.instrs[0].args;
  if (!((args[0] == k.reads[0].reg && args[1] == k.reads[1].reg) || (args[0] == k.reads[1].reg && args[1] == k.reads[0].reg)))
    return false;
  for (auto& lp : k.loops)
    if (lp.has_bounds) return false;
  const Op* ops[3] = {&k.write, &k.reads[0], &k.reads[1]};  // operand index + 1
  static const int perms[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};
  for (auto& pm : perms) {
    const Op& out = *ops[pm[0]];
    const Op& img = *ops[pm[1]];
    const Op& flt = *ops[pm[2]];
    const size_t nd = out.dims.size();
    if ((nd != 4 && nd != 3) || img.dims.size() != nd || flt.dims.size() != 4) continue;
    const bool batched = nd == 4;
    if (k.loops.size() != (batched ? 7u : 6u)) continue;
    std::vector<int> w;
    for (auto& d : out.dims) w.push_back(d.only_register());
    const int off = batched ? 1 : 0;
    const int n = batched ? w[0] : 0, y = w[off], x = w[off + 1], f = w[off + 2];
    if (!y || !x || !f || (batched && !n)) continue;
    const int ff = flt.dims[0].only_register(), dy = flt.dims[1].only_register(), dx = flt.dims[2].only_register(),
              c = flt.dims[3].only_register();
    if (ff != f || !dy || !dx || !c) continue;
    std::set<int> all = {y, x, f, dy, dx, c};
    if (batched) all.insert(n);
    if (all.size() != (batched ? 7u : 6u)) continue;
    auto pair_sum = [](const Lin& l, int p, int q) {
      return l.constant == 0 && l.factors.size() == 2 && l.factor_of(p) == 1 && l.factor_of(q) == 1;
    };
    if (batched && img.dims[0].only_register() != n) continue;
    if (!pair_sum(img.dims[off], y, dy) || !pair_sum(img.dims[off + 1], x, dx) || img.dims[off + 2].only_register() != c)
      continue;
    m.out_op = pm[0] - 1;
    m.img_op = pm[1] - 1;
    m.flt_op = pm[2] - 1;
    m.batched = batched;
    m.role = pm[0] == 0 ? ConvMatch::Forward : (pm[1] == 0 ? ConvMatch::GradImage : ConvMatch::GradFilter);
    return true;
  }
  return false;
}

}  // namespace model
}  // namespace eg
