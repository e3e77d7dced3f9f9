// Static (shape independent) lowering of a target: producer / consumer inlining, library matches,
// template-A source for everything else, and the one hiprtc program they are built in.
#include "model_types.hpp"


namespace eg {
namespace model {

int build_generic(eg_model* m, Generic& g) {
  int rc = eg_kernel_compile(m->ctx, g.src.name.c_str(), g.src.source.c_str(), &g.handle);
  if (rc) {
    std::string msg = eg_last_error();
    set_error("%s\n--- generated source ---\n%s", msg.c_str(), g.src.source.c_str());
    return rc;
  }
  m->kernels.push_back(g.handle);
  return EG_OK;
}

// Producer inlining.  A unary elementwise kernel `T{it} ++= f(S{it})` whose result is read only by
// generated kernels (not by a contraction / convolution the library runs) is recomputed inside
// those consumers: every read `T[index]` becomes `f(S[index])`, the producer is never launched and T
// never touches memory (conv2 -> leakyRelu -> maxpool2: the activation disappears into the pooling
// kernel and into maxpool2's hand-written gradient).  Same operations in the same order per element:
// results are bit-identical.  The reference's CPU target gets a similar effect from fuseLoops
// (passes.nim:1929-2004); its GPU target launches every kernel.
void inline_producers(eg_model* m, TargetState& ts) {
  static const bool off = [] {
    const char* e = eg::sw::raw("EG_NO_INLINE");
    return e && e[0] && e[0] != '0';
  }();
  if (off) return;
  Target& t = *ts.target;
  const Program& prog = m->prog;
  auto is_library = [&](const Kernel& k) {
    GemmMatch g;
    ConvMatch c;
    return k.is_seed || match_gemm(k, g) || (!prog.f64 && match_conv(k, c));
  };
  for (size_t p = 0; p < t.live.size(); ++p) {
    if (ts.lowered[p].absorbed || (int)p == t.first_update) continue;
    const Kernel P = t.all[t.live[p]];  // copy: the consumers below are edited in place
    // ---- a pure unary map over whole tensors?
    if (P.loops.size() != 1 || !P.index_instrs.empty() || !P.setup.empty() || P.is_seed || P.instrs.size() > 32) continue;
    if (P.reads.empty() || !P.write.raw || P.write.dims.size() != 1) continue;
    const int it = P.loops[0].reg;
    if (P.loops[0].has_bounds || P.write.dims[0].only_register() != it) continue;
    const int T = P.write.tensor, S = P.reads[0].tensor;
    bool pure = prog.tensors[T].kind == TK::Result && T != S && T != t.output && !ts.bucket_offset.count(T);
    for (auto& rd : P.reads)
      if (rd.tensor != S || !rd.raw || rd.dims.size() != 1 || rd.dims[0].only_register() != it) pure = false;
    for (auto& ins : P.instrs) {
      if (ins.kind == IK::Shape || ins.kind == IK::Len || ins.kind == IK::ShapeLen || ins.kind == IK::Epoch) pure = false;
      for (int a : ins.args)
        if (a == it) pure = false;  // the value depends on the position
    }
    if (P.result == it) pure = false;
    // T must be shaped like S for an index into T to address the same element of S
    auto sc = prog.shape_copy.find(T);
    if (prog.shape_dims.count(T) || (sc != prog.shape_copy.end() ? sc->second != S : P.reads.size() != 1)) pure = false;
    if (!pure) continue;
    // ---- every other kernel: nobody else writes T, S is final, all readers of T are generated kernels
    std::vector<size_t> consumers;
    bool ok = true;
    for (size_t q = 0; q < t.live.size() && ok; ++q) {
      if (q == p) continue;
      const Kernel& K = t.all[t.live[q]];
      if (K.write.tensor == T) ok = false;
      if (q > p && K.write.tensor == S) ok = false;
      bool reads_t = false;
      for (auto& rd : K.reads) reads_t = reads_t || rd.tensor == T;
      if (!reads_t) continue;
      if (q < p || ts.lowered[q].absorbed || is_library(K) || K.instrs.size() + P.instrs.size() * K.reads.size() > 96) ok = false;
      if ((t.first_update >= 0) && ((int)p < t.first_update) != ((int)q < t.first_update)) ok = false;  // stay on one side
      consumers.push_back(q);
    }
    if (!ok || consumers.empty()) continue;
    // ---- rewrite the consumers
    for (size_t q : consumers) {
      Kernel& K = t.all[t.live[q]];
      std::vector<Op> reads;
      std::vector<Instr> prefix;
      for (auto& rd : K.reads) {
        if (rd.tensor != T) {
          reads.push_back(rd);
          continue;
        }
        // one load of S at the same index, then P's instructions with fresh registers
        std::map<int, int> rename;
        Op load = rd;
        load.tensor = S;
        load.reg = K.alloc();
        for (auto& pr : P.reads) rename[pr.reg] = load.reg;
        reads.push_back(load);
        for (auto& ins : P.instrs) {
          Instr c = ins;
          c.res = K.alloc();
          rename[ins.res] = c.res;
          for (int& a : c.args) a = rename.count(a) ? rename[a] : a;
          prefix.push_back(c);
        }
        // the old data register of the T read now names the recomputed value (0 + f, as P stored it)
        Instr zero, sum;
        zero.kind = IK::Scalar;
        zero.lit = 0.0;
        zero.res = K.alloc();
        sum.kind = IK::Add;
        sum.args = {zero.res, rename.count(P.result) ? rename[P.result] : P.result};
        sum.res = rd.reg;
        prefix.push_back(zero);
        prefix.push_back(sum);
      }
      K.reads.swap(reads);
      K.instrs.insert(K.instrs.begin(), prefix.begin(), prefix.end());
    }
    ts.lowered[p].absorbed = true;
    ts.lowered[p].inlined = true;
    ts.lowered[p].all_index = t.live[p];
  }
}

// Consumer inlining, the mirror image of inline_producers.  A generated kernel P without a reduction
// (every iteration writes its own element: maxpool2's hand-written gradient, upsample2, a binary
// map) followed directly by an elementwise kernel `U{it} ++= g(T{it}, V{it}...)` that is the only
// reader of P's result T: P stores g(value, V...) into U right away and T never exists.  The
// combined kernel is generated here; a plan uses it only when P covers T completely and T, U and
// the V have one shape (make_plan), otherwise both kernels run as they are.
void inline_consumers(eg_model* m, TargetState& ts) {
  static const bool off = [] {
    const char* e = eg::sw::raw("EG_NO_INLINE");
    return e && e[0] && e[0] != '0';
  }();
  if (off) return;
  Target& t = *ts.target;
  const Program& prog = m->prog;
  const int n = (int)t.live.size();
  for (int p = 0; p < n; ++p) {
    Lowered& lo = ts.lowered[p];
    if (lo.absorbed || lo.kind != StepKind::GenericA) continue;
    int q = p + 1;
    while (q < n && ts.lowered[q].absorbed) ++q;
    if (q >= n || ts.lowered[q].kind != StepKind::GenericA || q == t.first_update) continue;
    if (t.first_update >= 0 && (p < t.first_update) != (q < t.first_update)) continue;
    const Kernel& P = t.all[t.live[p]];
    const Kernel& C = t.all[t.live[q]];
    // ---- P: one element per iteration
    std::vector<int> indep, red;
    bool scatter = false;
    split_loops(P, indep, red, scatter);
    if (!red.empty() || scatter || P.is_seed || P.gen != Gen::None) continue;
    const int T = P.write.tensor, U = C.write.tensor;
    if (prog.tensors[T].kind != TK::Result || T == t.output || ts.bucket_offset.count(T) || T == U) continue;
    // ---- C: a map over whole tensors that reads T
    if (C.loops.size() != 1 || !C.index_instrs.empty() || !C.setup.empty() || C.is_seed || C.gen != Gen::None) continue;
    if (C.loops[0].has_bounds || C.instrs.size() + P.instrs.size() > 96) continue;
    const int it = C.loops[0].reg;
    bool ok = C.write.raw && C.write.dims.size() == 1 && C.write.dims[0].only_register() == it && C.result != it;
    bool reads_t = false;
    for (auto& rd : C.reads) {
      if (!rd.raw || rd.dims.size() != 1 || rd.dims[0].only_register() != it || rd.tensor == U) ok = false;
      reads_t = reads_t || rd.tensor == T;
    }
    for (auto& ins : C.instrs) {
      if (ins.kind == IK::Shape || ins.kind == IK::Len || ins.kind == IK::ShapeLen || ins.kind == IK::Epoch) ok = false;
      for (int a : ins.args)
        if (a == it) ok = false;
    }
    for (auto& rd : P.reads)
      if (rd.tensor == U) ok = false;
    if (!ok || !reads_t) continue;
    // ---- nobody else writes or reads T
    for (int s2 = 0; s2 < n && ok; ++s2) {
      if (s2 == p || s2 == q) continue;
      const Kernel& K = t.all[t.live[s2]];
      if (K.write.tensor == T) ok = false;
      for (auto& rd : K.reads)
        if (rd.tensor == T) ok = false;
    }
    if (!ok) continue;
    // ---- P + C
    std::unique_ptr<Kernel> F(new Kernel(P));
    std::map<int, int> rename;
    Instr zero, value;  // what C would have loaded: 0 + P's value, as P stored it
    zero.kind = IK::Scalar;
    zero.lit = 0.0;
    zero.res = F->alloc();
    value.kind = IK::Add;
    value.args = {zero.res, P.result};
    value.res = F->alloc();
    F->instrs.push_back(zero);
    F->instrs.push_back(value);
    for (auto& rd : C.reads) {
      if (rd.tensor == T) {
        rename[rd.reg] = value.res;
        continue;
      }
      Op op = P.write;  // same element as the one P writes
      op.tensor = rd.tensor;
      op.reg = F->alloc();
      rename[rd.reg] = op.reg;
      F->reads.push_back(op);
    }
    for (auto& ins : C.instrs) {
      Instr c = ins;
      c.res = F->alloc();
      rename[ins.res] = c.res;
      for (int& a : c.args) a = rename.count(a) ? rename[a] : a;
      F->instrs.push_back(c);
    }
    F->result = rename.count(C.result) ? rename[C.result] : C.result;
    F->write.tensor = U;
    char name[64];
    snprintf(name, sizeof(name), "eg_k%d_ac", m->kernel_serial++);
    if (generate_mode_a(*F, name, lo.with_consumer_code.src) != EG_OK) {
      eg::clear_error();
      continue;
    }
    lo.consumer = q;
    lo.with_consumer = std::move(F);
    m->pending.push_back(&lo.with_consumer_code);
  }
}

int lower_target(eg_model* m, TargetState& ts) {
  Target& t = *ts.target;
  ts.lowered.clear();
  ts.lowered.resize(t.live.size());
  inline_producers(m, ts);
  for (size_t p = 0; p < t.live.size(); ++p) {
    Lowered& lo = ts.lowered[p];
    lo.all_index = t.live[p];
    const Kernel& k = t.all[lo.all_index];
    if (lo.absorbed) continue;
    if (k.is_seed) {
      lo.kind = StepKind::Seed;
      continue;
    }
    if (match_gemm(k, lo.gemm)) {
      lo.kind = StepKind::Gemm;
      // dense = contraction followed by the bias kernel on the same tensor (dnn.nim:21-24):
      // fold it into the epilogue.  Not across the backward/update boundary.
      if (p + 1 < t.live.size() && (int)(p + 1) != t.first_update) {
        const Kernel& nk = t.all[t.live[p + 1]];
        if (match_bias(nk, k.write.tensor) && nk.write.dims[1].only_register() &&
            k.write.dims.size() == 2) {
          lo.bias_tensor = nk.reads[0].tensor;
          ts.lowered[p + 1].absorbed = true;
          ts.lowered[p + 1].all_index = t.live[p + 1];
        }
      }
      continue;
    }
    // (float64: no library convolution yet — conv2 and its gradients run as generated kernels over `double`)
    if (!m->prog.f64 && match_conv(k, lo.conv)) {
      lo.kind = lo.conv.role == ConvMatch::Forward     ? StepKind::Conv
                : lo.conv.role == ConvMatch::GradImage ? StepKind::ConvGradImage
                                                       : StepKind::ConvGradFilter;
      continue;
    }
    lo.kind = StepKind::GenericA;
    lo.b_capable = split_reduction_capable(k);
    char name[64];
    snprintf(name, sizeof(name), "eg_k%d_a", m->kernel_serial++);
    int rc = generate_mode_a(k, name, lo.mode_a.src);
    if (rc) return rc;
    m->pending.push_back(&lo.mode_a);  // built together with the model's other generated kernels
  }
  inline_consumers(m, ts);
  return EG_OK;
}

// All template-A kernels of the model in one hiprtc program (seconds -> fractions of a second for a
// 40-kernel network); if the program fails to build, build one by one to name the culprit.
int build_pending(eg_model* m) {
  if (m->pending.empty()) return EG_OK;
  std::string source;
  std::vector<std::string> names;
  for (Generic* g : m->pending) {
    source += g->src.source + "\n";
    names.push_back(g->src.name);
  }
  std::vector<eg_kernel*> built;
  int rc = eg::kernels_compile_batch(m->ctx, "eg_model_kernels", source.c_str(), names, built);
  if (rc) {
    eg::clear_error();
    for (Generic* g : m->pending) {
      rc = build_generic(m, *g);
      if (rc) return rc;
    }
  } else {
    for (size_t i = 0; i < built.size(); ++i) {
      m->pending[i]->handle = built[i];
      m->kernels.push_back(built[i]);
    }
  }
  m->pending.clear();
  return EG_OK;
}

void describe(eg_model* m) {
  std::ostringstream os;
  for (auto& kv : m->targets) {
    TargetState& ts = kv.second;
    os << "target " << kv.first << " (" << ts.target->live.size() << " kernels)\n";
    for (size_t p = 0; p < ts.lowered.size(); ++p) {
      const Lowered& lo = ts.lowered[p];
      const Kernel& k = ts.target->all[lo.all_index];
      const char* kind = lo.absorbed ? "fused-into-previous"
                         : lo.kind == StepKind::Gemm ? (lo.bias_tensor ? "gemm+bias" : "gemm")
                         : lo.kind == StepKind::Conv ? "conv2"
                         : lo.kind == StepKind::ConvGradImage ? "conv2-grad-image"
                         : lo.kind == StepKind::ConvGradFilter ? "conv2-grad-filter"
                         : lo.kind == StepKind::Seed ? "seed-fill"
                         : (lo.b_capable ? "generic(map|split-reduce)" : "generic(map)");
      os << "  [" << p << "] " << kind;
      if (lo.kind == StepKind::Gemm) os << (lo.gemm.trans_a ? " T" : " N") << (lo.gemm.trans_b ? "T" : "N");
      os << " : " << to_text(k) << "\n";
    }
  }
  m->plan_text = os.str();
}

}  // namespace model
}  // namespace eg
