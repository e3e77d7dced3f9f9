// Plan construction: shapes, overwrite / accumulate decisions, launch arguments, arena layout.
#include "model_types.hpp"


namespace eg {
namespace model {

// The kernels generated for one plan (row / small / map groups, split reductions): one program.
int build_plan_kernels(eg_model* m, Plan& plan) {
  if (plan.pending.empty()) return EG_OK;
  std::string source;
  std::vector<std::string> names;
  for (auto& pk : plan.pending) {
    if (*pk.slot) continue;  // the same split-reduction kernel requested twice
    bool dup = false;
    for (auto& n : names) dup = dup || n == pk.name;
    if (dup) continue;
    source += pk.source + "\n";
    names.push_back(pk.name);
  }
  std::vector<eg_kernel*> built;
  int rc = names.empty() ? EG_OK : eg::kernels_compile_batch(m->ctx, "eg_plan_kernels", source.c_str(), names, built);
  if (rc) {  // name the culprit
    eg::clear_error();
    for (auto& pk : plan.pending) {
      if (*pk.slot) continue;
      rc = eg_kernel_compile(m->ctx, pk.name.c_str(), pk.source.c_str(), pk.slot);
      if (rc) {
        std::string msg = eg_last_error();
        set_error("%s\n--- generated source ---\n%s", msg.c_str(), pk.source.c_str());
        return rc;
      }
      m->kernels.push_back(*pk.slot);
    }
  } else {
    for (size_t i = 0; i < built.size(); ++i) {
      m->kernels.push_back(built[i]);
      for (auto& pk : plan.pending)
        if (pk.name == names[i]) *pk.slot = built[i];
    }
  }
  plan.pending.clear();
  return EG_OK;
}

float* tensor_ptr(eg_model* m, TargetState& ts, Plan& plan, int tid) {
  for (auto al = plan.alias.find(tid); al != plan.alias.end(); al = plan.alias.find(tid)) tid = al->second;
  const TensorDef& d = m->prog.tensors[tid];
  if (d.kind == TK::Param || d.kind == TK::Cache) return m->params[tid].ptr;
  if (d.kind == TK::Input) {
    auto it = m->inputs.find(tid);
    return (it == m->inputs.end() || !it->second.bound) ? nullptr : const_cast<float*>(it->second.device);
  }
  auto b = ts.bucket_offset.find(tid);
  if (b != ts.bucket_offset.end()) return ts.bucket + b->second;
  auto a = plan.arena_offset.find(tid);
  if (a != plan.arena_offset.end()) return plan.arena + a->second;
  return nullptr;
}

// write covers the whole tensor with plain stores?
bool full_cover(const Kernel& k, const KernelInfo& info, const std::vector<long>& shape) {
  std::set<int> seen;
  if (k.write.raw) {
    if (k.write.dims.size() != 1) return false;
    const int r = k.write.dims[0].only_register();
    const int l = r ? loop_index(k, r) : -1;
    return l >= 0 && info.bounds[l].first == 0 && info.bounds[l].second == prod(shape);
  }
  if (k.write.dims.size() != shape.size()) return false;
  for (size_t d = 0; d < shape.size(); ++d) {
    const Lin& lin = k.write.dims[d];
    const int r = lin.only_register();
    if (r) {
      const int l = loop_index(k, r);
      if (l < 0 || seen.count(r) || info.bounds[l].first != 0 || info.bounds[l].second != shape[d]) return false;
      seen.insert(r);
    } else if (!(lin.factors.empty() && lin.constant == 0 && shape[d] == 1)) {
      return false;
    }
  }
  return true;
}

// Remember where a generated kernel takes its four-elements-per-thread flag and what fill_params decided.
void note_vec4(Launch& L, long total) {
  L.total_items = total;
  L.vec_slot = -1;
  for (size_t i = 0; i < L.generic->src.slots.size(); ++i)
    if (L.generic->src.slots[i].kind == Slot::Vec4) L.vec_slot = (int)i;
  L.vec_ok = L.vec_slot >= 0 && L.params[L.vec_slot] != 0;
}

int fill_params(eg_model* m, const Kernel& k, const KernelInfo& info, const Shapes& shapes, const GenericSource& src,
                bool accumulate, long total, long rtotal, long chunk, std::vector<long>& out) {
  out.clear();
  for (const Slot& s : src.slots) {
    long v = 0;
    switch (s.kind) {
      case Slot::Accumulate: v = accumulate ? 1 : 0; break;
      case Slot::Total: v = total; break;
      case Slot::RTotal: v = rtotal; break;
      case Slot::Chunk: v = chunk; break;
      case Slot::LoopStart: v = info.bounds[s.a].first; break;
      case Slot::LoopExtent: v = info.bounds[s.a].second - info.bounds[s.a].first; break;
      case Slot::Stride: {
        const Op& op = s.a < (int)k.reads.size() ? k.reads[s.a] : k.write;
        const std::vector<long>& shp = shapes.at(op.tensor);
        long stride = 1;
        for (int d = (int)shp.size() - 1; d > s.b; --d) stride *= shp[d];
        v = stride;
        break;
      }
      case Slot::SetupVal: v = info.vals.at(k.setup[s.a].res); break;
      case Slot::Vec4: {
        constexpr bool off = false;
        const int l = src.indep.empty() ? -1 : src.indep.back();
        v = !off && l >= 0 && info.bounds[l].first == 0 && info.bounds[l].second % 4 == 0 && total % 4 == 0 && total > 0;
        auto rows_of_four = [&](int tensor) {
          auto sh = shapes.find(tensor);
          return sh != shapes.end() && !sh->second.empty() && sh->second.back() % 4 == 0;
        };
        for (auto& rd : k.reads) v = v && rows_of_four(rd.tensor);
        v = v && rows_of_four(k.write.tensor);
        break;
      }
      case Slot::Narrow: {
        static const bool off = eg::sw::raw("EG_NO_NARROW_INDEX") != nullptr;
        const long lim = 1L << 31;
        v = !off && total < lim && rtotal < lim;
        auto small = [&](int tensor) {
          auto sh = shapes.find(tensor);
          return sh != shapes.end() && prod(sh->second) < lim;
        };
        for (auto& rd : k.reads) v = v && small(rd.tensor);
        v = v && small(k.write.tensor);
        for (auto& b : info.bounds) v = v && b.first > -lim && b.second < lim;
        // The 32-bit copy of the body narrows every `long`, Index-typed VALUES included
        // (`toScalar(i * 100000)`, a large Index literal): only addressing is known to fit, so a
        // kernel that computes with Index values keeps the 64-bit body.
        // Iterators, shape() / len() / epoch values are checked above and below, so converting one of
        // them with toScalar stays exact; arithmetic on them and Index literals are not bounded.
        if (v) {
          const std::vector<Ty> types = infer_types(k);
          for (auto& ins : k.instrs) {
            const bool index_typed = ins.res > 0 && ins.res < (int)types.size() && types[ins.res] == Ty::Index;
            const bool derived = ins.kind != IK::Shape && ins.kind != IK::Len && ins.kind != IK::ShapeLen && ins.kind != IK::Epoch;
            if (index_typed && derived) v = 0;
          }
        }
        break;
      }
      case Slot::InstrVal: {
        const Instr& ins = k.instrs[s.a];
        if (ins.kind == IK::Epoch) {
          v = m->epoch;
        } else {
          const std::vector<long>& shp = shapes.at(ins.tensor);
          if (ins.kind == IK::Len) v = prod(shp);
          else if (ins.kind == IK::ShapeLen) v = (long)shp.size();
          else {
            int d = ins.dim < 0 ? ins.dim + (int)shp.size() : ins.dim;
            if (d < 0 || d >= (int)shp.size()) {
              set_error("shape()[%d] out of range for a rank-%zu tensor", ins.dim, shp.size());
              return EG_ERR_SHAPE;
            }
            v = shp[d];
          }
        }
        break;
      }
      default: break;
    }
    out.push_back(v);
  }
  // 32-bit copies of the arguments must be exact too
  for (size_t i = 0; i < src.slots.size(); ++i)
    if (src.slots[i].kind == Slot::Narrow)
      for (long v : out)
        if (v >= (1L << 31) || v < -(1L << 31)) out[i] = 0;
  return EG_OK;
}

// Device resources of a plan (graphs, row-group partials, the arena).  The caller has waited for the stream.
void release_plan(Plan& plan) {
  for (auto& g : plan.graphs) {
    if (g.exec) hipGraphExecDestroy(g.exec);
    g.exec = nullptr;
  }
  for (auto& rg : plan.row_groups) {
    if (rg->partial) hipFree(rg->partial);
    rg->partial = nullptr;
    if (rg->counter) hipFree(rg->counter);
    rg->counter = nullptr;
  }
  if (plan.sample_group && plan.sample_group->slab) {
    hipFree(plan.sample_group->slab);
    plan.sample_group->slab = nullptr;
  }
  if (plan.arena) hipFree(plan.arena);
  plan.arena = nullptr;
}

std::string shape_key(eg_model* m) {
  std::ostringstream os;
  for (auto& in : m->inputs) {
    if (!in.second.bound) continue;
    os << in.first << ":";
    for (long s : in.second.shape) os << s << ",";
    os << ";";
  }
  os << "e" << (m->prog.epoch_in_setup ? m->epoch : 0);  // host values computed from epoch() are fixed per plan (kd.hpp)
  if (m->keep_values) os << "v";
  return os.str();
}

// Is live kernel p a whole-tensor raw copy `dst{it} ++= src{it}` whose destination can simply share
// the source's storage?  (reshape and its gradient.)  Requires: dst is written by this kernel only,
// src is complete by then (no later writer), same element count, dst not in the gradient bucket.
bool copy_can_alias(eg_model* m, TargetState& ts, const Kernel& k, const KernelInfo& info, const Shapes& shapes, int p) {
  static const bool off = [] {
    const char* e = eg::sw::raw("EG_NO_ALIAS");
    return e && e[0] && e[0] != '0';
  }();
  if (off || !info.ok) return false;
  if (k.reads.size() != 1 || !k.instrs.empty() || !k.index_instrs.empty() || k.loops.size() != 1) return false;
  const Op& rd = k.reads[0];
  if (k.result != rd.reg || !k.write.raw || !rd.raw || k.write.dims.size() != 1 || rd.dims.size() != 1) return false;
  const int it = k.loops[0].reg;
  if (k.write.dims[0].only_register() != it || rd.dims[0].only_register() != it) return false;
  const int dst = k.write.tensor, src = rd.tensor;
  if (dst == src || ts.bucket_offset.count(dst) || m->prog.tensors[dst].kind != TK::Result) return false;
  const long n = prod(shapes.at(dst));
  if (info.bounds[0].first != 0 || info.bounds[0].second != n || prod(shapes.at(src)) != n) return false;
  const Target& t = *ts.target;
  for (size_t q = 0; q < t.live.size(); ++q) {
    const Kernel& o = t.all[t.live[q]];
    if ((int)q != p && o.write.tensor == dst) return false;   // another contribution to dst
    if ((int)q > p && o.write.tensor == src) return false;    // src still changes after the copy
  }
  return true;
}

int make_plan(eg_model* m, TargetState& ts, Plan& plan) {
  Target& t = *ts.target;
  Shapes& shapes = plan.shapes;
  plan.esz = m->esz;
  // A float64 program (compile[float64], model.nim:253-260) runs as the plain launch list: library contractions in
  // float64 (eg_dgemm), everything else as generated kernels over `double`; the float32 fusion passes (row / sample /
  // map groups, contraction epilogues, predicate bits, side lanes) generate float32 code and stay out of it.
  const bool fuse = !m->f64;
  for (auto& in : m->inputs) {
    if (!in.second.bound) continue;
    const TensorDef& d = m->prog.tensors[in.first];
    if (d.has_shape && !d.shape.empty()) {  // staticShapeMismatch (tests/test_errors.nim:56-59)
      bool ok = d.shape.size() == in.second.shape.size();
      for (size_t i = 0; ok && i < d.shape.size(); ++i)
        if (d.shape[i] >= 0 && d.shape[i] != in.second.shape[i]) ok = false;
      if (!ok) {
        set_error("input \"%s\" does not match its static shape", d.name.c_str());
        return EG_ERR_SHAPE;
      }
    }
    shapes[in.first] = in.second.shape;
  }
  for (auto& p : m->params) shapes[p.first] = p.second.shape;

  // shape inference over EVERY kernel (eliminated ones included: the reference collects shape
  // constraints before dead kernels are dropped, model.nim:46-77)
  std::vector<KernelInfo> infos(t.all.size());
  std::set<int> live_set(t.live.begin(), t.live.end());
  for (size_t i = 0; i < t.all.size(); ++i) {
    const Kernel& k = t.all[i];
    bool ready = true;
    int missing = 0;
    for (auto& r : k.reads)  // TensorRandom: shaped like the tensor `rand` was given (parser.nim:378-383)
      if (m->prog.tensors[r.tensor].kind == TK::Random && !shapes.count(r.tensor)) {
        auto sc = m->prog.shape_copy.find(r.tensor);
        if (sc != m->prog.shape_copy.end() && shapes.count(sc->second)) shapes[r.tensor] = shapes[sc->second];
      }
    for (auto& r : k.reads)
      if (!shapes.count(r.tensor)) {
        ready = false;
        missing = r.tensor;
      }
    if (ready)
      for (auto& s : k.setup)  // (bounds that name the written tensor's own shape are resolved by infer_kernel)
        if (s.tensor && s.tensor != k.write.tensor && !shapes.count(s.tensor)) {
          ready = false;
          missing = s.tensor;
        }
    if (!ready) {
      if (live_set.count((int)i)) {
        const TensorDef& d = m->prog.tensors[missing];
        if (d.kind == TK::Input) {
          set_error("input \"%s\" of target \"%s\" was not provided", d.name.c_str(), t.name.c_str());
          return EG_ERR_RUNTIME;
        }
        set_error("the shape of tensor %d is under-constrained", missing);
        return EG_ERR_SHAPE;
      }
      continue;
    }
    int rc = infer_kernel(m->prog, k, shapes, m->epoch, infos[i]);
    if (rc) {
      if (live_set.count((int)i)) return rc;
      eg::clear_error();
    }
  }

  // which result tensors does the live list write; who writes first
  std::map<int, int> first_writer;  // tensor -> position in live
  std::vector<int> result_tensors;
  for (size_t p = 0; p < t.live.size(); ++p) {
    if (ts.lowered[p].inlined) continue;  // its tensor is never materialised
    const Kernel& k = t.all[t.live[p]];
    const int wt = k.write.tensor;
    if (m->prog.tensors[wt].kind == TK::Result && !first_writer.count(wt)) {
      first_writer[wt] = (int)p;
      result_tensors.push_back(wt);
    }
  }
  if (t.output && m->prog.tensors[t.output].kind == TK::Result && !first_writer.count(t.output)) {
    if (!shapes.count(t.output)) {
      set_error("the shape of the output of target \"%s\" is under-constrained", t.name.c_str());
      return EG_ERR_SHAPE;
    }
    result_tensors.push_back(t.output);  // never written: stays zero
  }

  // ---- row fusion (rowfuse.hpp): runs of per-sample kernels become one generated kernel each
  std::vector<int> group_of(t.live.size(), -1);
  plan.row_groups.clear();
  // decide overwrite vs accumulate per launch; collect tensors that must be zeroed
  std::set<int> needs_zero;
  plan.sample_group.reset();
  if (fuse) {
    int rc = form_sample_group(m, ts, plan, infos, first_writer, group_of, needs_zero);
    if (rc) return rc;
  }
  int rc_groups = fuse ? form_row_groups(m, ts, plan, infos, first_writer, group_of) : EG_OK;
  if (rc_groups) return rc_groups;

  plan.launches.clear();
  plan.n_backward = -1;
  std::set<int> folded;  // consumers that run inside the kernel before them
  for (size_t p = 0; p < t.live.size(); ++p) {
    if ((int)p == t.first_update) plan.n_backward = (int)plan.launches.size();
    if (folded.count((int)p)) continue;
    if (group_of[p] == SAMPLE_GROUP_CODE) {
      if (!plan.sample_group->positions.empty() && plan.sample_group->positions[0] == (int)p) {
        Launch L;
        L.lowered = (int)p;
        L.kind = StepKind::SampleFused;
        L.accumulate = false;
        plan.launches.push_back(L);
      }
      continue;
    }
    if (group_of[p] <= -2) {
      const int wt = t.all[t.live[p]].write.tensor;  // small groups always accumulate
      if (m->prog.tensors[wt].kind == TK::Result && first_writer[wt] == (int)p) needs_zero.insert(wt);
      if (p == 0 || group_of[p - 1] != group_of[p]) {
        Launch L;
        L.lowered = (int)p;
        L.kind = StepKind::SmallFused;
        L.row_group = -2 - group_of[p];
        plan.launches.push_back(L);
      }
      continue;
    }
    if (group_of[p] >= 0) {
      if (p == 0 || group_of[p - 1] != group_of[p]) {  // first kernel of the group: one launch for all
        Launch L;
        L.lowered = (int)p;
        L.kind = StepKind::RowFused;
        L.row_group = group_of[p];
        L.accumulate = false;
        plan.launches.push_back(L);
      }
      continue;
    }
    Lowered& lo = ts.lowered[p];
    if (lo.absorbed) continue;
    const Kernel& k = t.all[lo.all_index];
    const KernelInfo& info = infos[lo.all_index];
    const int wt = k.write.tensor;
    const bool is_result = m->prog.tensors[wt].kind == TK::Result;
    const bool first = is_result && first_writer[wt] == (int)p;
    const std::vector<long>& wshape = shapes.at(wt);
    if (lo.kind == StepKind::GenericA && first && copy_can_alias(m, ts, k, info, shapes, (int)p)) {
      plan.alias[wt] = k.reads[0].tensor;
      continue;
    }
    if (lo.consumer >= 0 && group_of[lo.consumer] == -1 && first && info.ok && full_cover(k, info, wshape)) {
      // consumer inlining: P covers T, and T, U and the consumer's other operands have one shape
      const int q = lo.consumer;
      const Kernel& C = t.all[t.live[q]];
      const KernelInfo& cinfo = infos[t.live[q]];
      const Kernel& F = *lo.with_consumer;
      const int U = C.write.tensor;
      const long count = prod(wshape);
      bool same = cinfo.ok && cinfo.bounds[0].first == 0 && cinfo.bounds[0].second == count;
      auto matches = [&](int tensor) {
        auto sh = shapes.find(tensor);
        if (sh == shapes.end()) return false;
        return k.write.raw ? prod(sh->second) == count : sh->second == wshape;
      };
      same = same && matches(U);
      for (auto& rd : C.reads) same = same && matches(rd.tensor);
      if (same) {
        const bool u_result = m->prog.tensors[U].kind == TK::Result;
        const bool u_first = u_result && first_writer[U] == q;
        Launch L;
        L.lowered = (int)p;
        L.kind = StepKind::GenericA;
        L.generic = &lo.with_consumer_code;
        L.consumer = q;
        L.blocks_x = (count + 255) / 256;
        int rc = fill_params(m, F, info, shapes, lo.with_consumer_code.src, !u_first, count, 1, 0, L.params);
        if (rc) return rc;
        note_vec4(L, count);
        for (size_t si = 0; si < L.generic->src.slots.size(); ++si) {
          const Slot& sl = L.generic->src.slots[si];
          if (sl.kind == Slot::InstrVal && F.instrs[sl.a].kind == IK::Epoch) L.epoch_slots.push_back((int)si);
        }
        L.c_tensor = U;
        L.accumulate = !u_first;
        plan.launches.push_back(L);
        folded.insert(q);
        continue;
      }
    }
    Launch L;
    L.lowered = (int)p;
    L.kind = lo.kind;
    if (lo.kind == StepKind::Seed) {
      L.count = prod(wshape);
      L.c_tensor = wt;
      L.accumulate = false;
      plan.launches.push_back(L);
      continue;
    }
    bool overwrite = first && full_cover(k, info, wshape);
    if (lo.kind == StepKind::Gemm) {
      const GemmMatch& g = lo.gemm;
      const Op& A = k.reads[g.a_read];
      const Op& B = k.reads[g.b_read];
      L.M = info.bounds[g.li].second;
      L.N = info.bounds[g.lj].second;
      L.K = info.bounds[g.lk].second;
      const std::vector<long>& as = shapes.at(A.tensor);
      const std::vector<long>& bs = shapes.at(B.tensor);
      // loop bounds come from the first tensor that names the iterator; the other operands must agree
      const long a_m = g.trans_a ? as[1] : as[0], a_k = g.trans_a ? as[0] : as[1];
      const long b_k = g.trans_b ? bs[1] : bs[0], b_n = g.trans_b ? bs[0] : bs[1];
      if (a_m < L.M || a_k < L.K || b_k < L.K || b_n < L.N || wshape[0] < L.M || wshape[1] < L.N) {
        set_error("contraction operands have inconsistent shapes ([%ld,%ld] x [%ld,%ld])", a_m, a_k, b_k, b_n);
        return EG_ERR_SHAPE;
      }
      L.lda = as[1];
      L.ldb = bs[1];
      L.ldc = wshape[1];
      L.a_tensor = A.tensor;
      L.b_tensor = B.tensor;
      L.c_tensor = wt;
      L.trans_a = g.trans_a;
      L.trans_b = g.trans_b;
      L.bias_tensor = lo.bias_tensor;
    } else if (lo.kind == StepKind::Conv || lo.kind == StepKind::ConvGradImage || lo.kind == StepKind::ConvGradFilter) {
      auto operand = [&](int which) -> const Op& { return which < 0 ? k.write : k.reads[which]; };
      const Op& img = operand(lo.conv.img_op);
      const Op& flt = operand(lo.conv.flt_op);
      const Op& out = operand(lo.conv.out_op);
      const std::vector<long>& is = shapes.at(img.tensor);
      const std::vector<long>& fs = shapes.at(flt.tensor);
      const std::vector<long>& os = shapes.at(out.tensor);
      const int off = lo.conv.batched ? 1 : 0;
      L.cN = lo.conv.batched ? is[0] : 1;
      L.cH = is[off];
      L.cW = is[off + 1];
      L.cC = is[off + 2];
      L.cF = fs[0];
      L.cFH = fs[1];
      L.cFW = fs[2];
      if (fs[3] != L.cC) {
        set_error("conv2: image has %ld channels, filters have %ld", L.cC, fs[3]);
        return EG_ERR_SHAPE;
      }
      if (lo.kind != StepKind::Conv) {
        // the gradient kernels cover their destination completely only if the three shapes are
        // the ones of a valid convolution
        const bool ok = os.size() == is.size() && (!lo.conv.batched || os[0] == L.cN) && os[off] == L.cH - L.cFH + 1 &&
                        os[off + 1] == L.cW - L.cFW + 1 && os[off + 2] == L.cF;
        if (!ok) {
          set_error("conv2 gradient: output gradient shape does not match image and filter shapes");
          return EG_ERR_SHAPE;
        }
        if (first) overwrite = true;
      }
      // a_tensor: image (forward, filter gradient) or filters (image gradient); b_tensor: filters
      // (forward) or the output gradient
      if (lo.kind == StepKind::Conv) {
        L.a_tensor = img.tensor;
        L.b_tensor = flt.tensor;
      } else if (lo.kind == StepKind::ConvGradFilter) {
        L.a_tensor = img.tensor;
        L.b_tensor = out.tensor;
      } else {
        L.a_tensor = flt.tensor;
        L.b_tensor = out.tensor;
      }
      L.c_tensor = wt;
    } else {
      // generic: choose the template
      std::vector<int> indep, red;
      bool scatter;
      split_loops(k, indep, red, scatter);
      long total = 1, rtotal = 1;
      for (int l : indep) total *= info.bounds[l].second - info.bounds[l].first;
      for (int l : red) rtotal *= info.bounds[l].second - info.bounds[l].first;
      if (total < 0) total = 0;
      if (rtotal < 0) rtotal = 0;
      const bool use_b = lo.b_capable && !scatter && rtotal >= 2048 && total <= 8192 && total * 64 <= rtotal &&
                         full_cover(k, info, wshape);
      if (scatter) overwrite = false;
      if (use_b) {
        int tx = 1;
        while (tx < total && tx < 64) tx <<= 1;
        Generic& g = lo.mode_b[tx];
        if (!g.handle) {
          char name[64];
          snprintf(name, sizeof(name), "eg_k%d_b%d", m->kernel_serial++, tx);
          int rc = generate_mode_b(k, name, tx, g.src);
          if (rc) return rc;
          plan.pending.push_back({g.src.name, g.src.source, &g.handle});
        }
        const int ty = 256 / tx;
        const long col_tiles = (total + tx - 1) / tx;
        long nchunks = (4L * m->ctx->compute_units + col_tiles - 1) / col_tiles;
        const long max_chunks = (rtotal + ty * 8 - 1) / (ty * 8);
        if (nchunks > max_chunks) nchunks = max_chunks;
        if (nchunks < 1) nchunks = 1;
        const long chunk = (rtotal + nchunks - 1) / nchunks;
        nchunks = (rtotal + chunk - 1) / chunk;
        L.kind = StepKind::GenericB;
        L.generic = &g;
        L.blocks_x = nchunks;
        L.blocks_y = col_tiles;
        L.partial_rows = nchunks;
        L.partial_cols = total;
        int rc = fill_params(m, k, info, shapes, g.src, !overwrite, total, rtotal, chunk, L.params);
        if (rc) return rc;
      } else {
        L.kind = StepKind::GenericA;
        L.generic = &lo.mode_a;
        L.blocks_x = (total + 255) / 256;
        int rc = fill_params(m, k, info, shapes, lo.mode_a.src, !overwrite, total, rtotal, 0, L.params);
        if (rc) return rc;
        note_vec4(L, total);
      }
      ConvMatch cm;
      if (m->f64 && match_conv(k, cm)) {
        // operands of a matched convolution kernel: -1 = the written tensor, else a read (ConvMatch)
        auto tensor_of = [&](int which) { return which < 0 ? k.write.tensor : k.reads[which].tensor; };
        const int t_img = tensor_of(cm.img_op), t_flt = tensor_of(cm.flt_op), t_out = tensor_of(cm.out_op);
        const std::vector<long>& is = shapes.at(t_img);
        const std::vector<long>& fs = shapes.at(t_flt);
        const std::vector<long>& os = shapes.at(t_out);
        const int o = cm.batched ? 1 : 0;
        const bool valid = fs.size() == 4 && is.size() == (size_t)(3 + o) && os.size() == is.size() && fs[3] == is[o + 2] &&
                           os[o] == is[o] - fs[1] + 1 && os[o + 1] == is[o + 1] - fs[2] + 1 && os[o + 2] == fs[0] &&
                           (!cm.batched || os[0] == is[0]) && (cm.role != ConvMatch::Forward || full_cover(k, info, wshape));
        if (valid) {
          L.conv_direct64 = cm.role == ConvMatch::Forward ? 1 : (cm.role == ConvMatch::GradImage ? 2 : 3);
          L.cN = cm.batched ? is[0] : 1;
          L.cH = is[o];
          L.cW = is[o + 1];
          L.cC = is[o + 2];
          L.cF = fs[0];
          L.cFH = fs[1];
          L.cFW = fs[2];
          // a / b as the float32 library launches take them: image + filters (forward), image + output gradient (filter
          // gradient), filters + output gradient (image gradient)
          L.a_tensor = cm.role == ConvMatch::GradImage ? t_flt : t_img;
          L.b_tensor = cm.role == ConvMatch::Forward ? t_flt : t_out;
        }
      }
      for (size_t si = 0; si < L.generic->src.slots.size(); ++si) {
        const Slot& sl = L.generic->src.slots[si];
        if (sl.kind == Slot::InstrVal && k.instrs[sl.a].kind == IK::Epoch) L.epoch_slots.push_back((int)si);
      }
      L.c_tensor = wt;
    }
    L.accumulate = !overwrite;
    if (is_result && first && !overwrite) needs_zero.insert(wt);
    plan.launches.push_back(L);
  }
  if (plan.n_backward < 0) plan.n_backward = (int)plan.launches.size();
  if (t.output && m->prog.tensors[t.output].kind == TK::Result && !first_writer.count(t.output)) needs_zero.insert(t.output);
  if (fuse) {
    int rc = fuse_epilogues(m, ts, plan, infos);
    if (rc) return rc;
  }
  if (fuse) {
    int rc = fold_bias_gradients(m, ts, plan, infos);
    if (rc) return rc;
  }
  if (fuse) {
    int rc = fold_row_products(m, ts, plan);
    if (rc) return rc;
  }

  if (fuse) plan_overlap(m, ts, plan);
  if (fuse) plan_pipeline(m, ts, plan);
  if (fuse) {
    int rc = fuse_row_tails(m, ts, plan, infos);
    if (rc) return rc;
    rc = fuse_slab_fold(m, ts, plan, infos);
    if (rc) return rc;
  }
  {
    int rc = build_plan_kernels(m, plan);
    if (rc) return rc;
  }
  // arena layout: tensors that need zeroing first (one memset), then the rest
  plan.arena_offset.clear();
  plan.bucket_zero.clear();
  long off = 0;
  for (int pass = 0; pass < 2; ++pass) {
    for (int tid : result_tensors) {
      if (plan.alias.count(tid)) continue;  // lives in its source's storage
      if (ts.bucket_offset.count(tid)) {
        if (pass == 0 && needs_zero.count(tid)) plan.bucket_zero.push_back(tid);
        continue;
      }
      // (predicate bits may be OR-ed in; a row product is added to its destination)
      const bool z = needs_zero.count(tid) != 0 || (plan.predicated.count(tid) != 0 && plan.pred_unzeroed.count(tid) == 0) ||
                     plan.zero_extra.count(tid) != 0;
      if ((pass == 0) != z) continue;
      // large tensors start on a 256-byte boundary: a streaming kernel's 1 KiB wave stores then cover whole 64-byte
      // sectors (65 536 x 512 floats written from a base 16 bytes past a boundary: 34 us instead of 27, tools/narrow_k_harness.hip OUT_OFFSET=4)
      const long floats = storage_floats(plan, tid);
      if (floats >= 16384) off = (off + 63) & ~63L;
      plan.arena_offset[tid] = off;
      off += align4(floats);
    }
    if (pass == 0) plan.zero_floats = off;
  }
  plan.random_tensors.clear();
  for (int p : t.live)
    for (auto& rd : t.all[p].reads)
      if (m->prog.tensors[rd.tensor].kind == TK::Random && !plan.arena_offset.count(rd.tensor)) {
        plan.arena_offset[rd.tensor] = off;
        off += align4(prod(shapes.at(rd.tensor)) * plan.esz);
        plan.random_tensors.push_back(rd.tensor);
      }
  if (!plan.random_tensors.empty()) {
    int rc = ensure_rng(m);
    if (rc) return rc;
  }
  plan.arena_floats = off;
  if (off > 0) {
    EG_HIP_CHECK(hipSetDevice(m->ctx->device));
    EG_HIP_CHECK(hipMalloc((void**)&plan.arena, (size_t)off * sizeof(float)));
  }
  return check_plan(m, ts, plan);
}

int get_plan(eg_model* m, const char* target, TargetState** ts_out, Plan** plan_out) {
  EG_REQUIRE(m && target, EG_ERR_INVALID, "NULL model or target");
  auto it = m->targets.find(target);
  // model.nim:395-396
  EG_REQUIRE(it != m->targets.end(), EG_ERR_RUNTIME, "%s is not a target of the model", target);
  TargetState& ts = it->second;
  const bool per_epoch = m->prog.epoch_in_setup;
  if (ts.last && ts.last_stamp == m->inputs_gen && (!per_epoch || ts.last_epoch == m->epoch)) {  // same bindings as the last lookup: same shapes, same plan
    *ts_out = &ts;
    *plan_out = ts.last;
    return EG_OK;
  }
  const std::string key = shape_key(m);
  auto p = ts.plans.find(key);
  if (p == ts.plans.end()) {
    if (per_epoch) {
      // Plans of other epochs will not be asked for again (the epoch only grows): release them instead of keeping one
      // arena per epoch.  Their launches may still be queued, hence the wait.
      bool stale = false;
      for (auto& old : ts.plans) stale = stale || old.second->epoch != m->epoch;
      if (stale) {
        EG_HIP_CHECK(hipStreamSynchronize(m->ctx->stream));
        for (auto it = ts.plans.begin(); it != ts.plans.end();) {
          if (it->second->epoch != m->epoch) {
            release_plan(*it->second);
            it = ts.plans.erase(it);
          } else {
            ++it;
          }
        }
        ts.last = nullptr;
      }
    }
    std::unique_ptr<Plan> plan(new Plan());
    plan->key = key;
    plan->epoch = m->epoch;
    int rc = make_plan(m, ts, *plan);
    if (rc) {
      if (plan->arena) hipFree(plan->arena);
      return rc;
    }
    p = ts.plans.emplace(key, std::move(plan)).first;
  }
  ts.last = p->second.get();
  ts.last_stamp = m->inputs_gen;
  ts.last_epoch = m->epoch;
  *ts_out = &ts;
  *plan_out = p->second.get();
  return EG_OK;
}

}  // namespace model
}  // namespace eg
