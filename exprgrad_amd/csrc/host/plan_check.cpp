// Invariants of a finished plan.  make_plan ends with check_plan: a plan that violates one of them
// is refused (EG_ERR_RUNTIME naming the invariant) instead of producing numbers.
//
// The plan is the place where the backend departs furthest from the reference's "one launch per
// kernel, every result zeroed, every kernel accumulates" (model.nim:295-300, 392-411): kernels are
// fused, folded into epilogues, inlined, run on a second stream, and results that their first
// writer covers completely are neither zeroed nor read.  Each of those decisions is only valid
// under conditions on who reads and writes what, checked here once more, independently of the code
// that made the decision:
//   1. coverage     every live kernel of the target is represented exactly once
//   2. storage      every tensor a launch touches has device storage; arena slots are aligned,
//                   inside the arena, pairwise disjoint; to-be-zeroed tensors lie in the zero prefix
//   3. order        no launch reads (or accumulates into) a result tensor that is neither zeroed
//                   nor written by an earlier launch
//   (3b) predicate tensors are read and written as bits by every launch that touches them;
//   4. side lane    the launches of an overlap group touch nothing the contraction they run next to
//                   writes, and write nothing it reads; groups are disjoint, ordered, and stay on one
//                   side of the backward | update boundary
#include <algorithm>

#include "model_types.hpp"

namespace eg {
namespace model {

namespace {

struct IO {
  std::set<int> reads;         // tensors whose previous contents the launch needs
  std::set<int> writes;        // tensors the launch writes
  std::set<int> needs_prior;   // subset of writes that are accumulated into / partially written
};

int root(const Plan& plan, int t) {
  for (int guard = 0; guard < 64; ++guard) {
    auto a = plan.alias.find(t);
    if (a == plan.alias.end()) break;
    t = a->second;
  }
  return t;
}

void kernel_io(const Plan& plan, const Kernel& k, bool accumulate, IO& io) {
  for (auto& rd : k.reads) io.reads.insert(root(plan, rd.tensor));
  const int w = root(plan, k.write.tensor);
  io.writes.insert(w);
  if (accumulate) io.needs_prior.insert(w);
}

// What a launch reads and writes, derived from the kernels it stands for.
void launch_io(eg_model* m, TargetState& ts, const Plan& plan, const Launch& L, IO& io) {
  const Target& t = *ts.target;
  auto rd = [&](int x) {
    if (x) io.reads.insert(root(plan, x));
  };
  auto wr = [&](int x, bool acc) {
    if (!x) return;
    io.writes.insert(root(plan, x));
    if (acc) io.needs_prior.insert(root(plan, x));
  };
  switch (L.kind) {
    case StepKind::Seed: wr(L.c_tensor, false); break;
    case StepKind::Gemm:
    case StepKind::Conv:
    case StepKind::ConvGradImage:
    case StepKind::ConvGradFilter:
      rd(L.a_tensor);
      rd(L.b_tensor);
      rd(L.bias_tensor);
      wr(L.c_tensor, L.accumulate);
      wr(L.ones_tensor, false);
      break;
    case StepKind::GemmFused: {
      const PlanEpilogue& pe = *plan.epilogues[L.epilogue];
      rd(L.a_tensor);
      rd(L.b_tensor);
      rd(L.bias_tensor);
      const Kernel& ck = t.all[ts.lowered[pe.consumer.lowered].all_index];
      if (pe.store_c || pe.pred_write) wr(L.c_tensor, L.accumulate);
      for (auto& r : ck.reads)
        if (r.tensor != L.c_tensor) rd(r.tensor);
      wr(ck.write.tensor, pe.consumer.accumulate);
      if (pe.row_product) {  // adds to its destination: that must be zeroed
        rd(pe.product.b_tensor);
        rd(pe.product.bias_tensor);
        wr(pe.product.c_tensor, true);
      }
      break;
    }
    case StepKind::GenericA:
    case StepKind::GenericB: {
      const Kernel& k = L.consumer >= 0 ? *ts.lowered[L.lowered].with_consumer : t.all[ts.lowered[L.lowered].all_index];
      for (auto& r : k.reads) rd(r.tensor);
      wr(L.consumer >= 0 ? L.c_tensor : k.write.tensor, L.accumulate);
      break;
    }
    case StepKind::RowFused: {
      const PlanRowGroup& pg = *plan.row_groups[L.row_group];
      for (auto& kv : pg.g.tensors) {
        const RowGroupTensor& gt = kv.second;
        switch (gt.role) {
          case RowGroupTensor::RowExternal:
          case RowGroupTensor::SmallExternal: rd(kv.first); break;
          case RowGroupTensor::RowLocal:
            if (gt.load_first) rd(kv.first);
            if (gt.store) wr(kv.first, gt.load_first);
            break;
          case RowGroupTensor::Reduction: wr(kv.first, gt.accumulate); break;
          case RowGroupTensor::SmallLocal: break;  // lives in registers only
        }
      }
      break;
    }
    case StepKind::SampleFused: {
      const PlanSampleGroup& sg = *plan.sample_group;
      std::set<int> inside;
      for (size_t i = 0; i < sg.g.kernel_index.size(); ++i) {
        const Kernel& k = t.all[sg.g.kernel_index[i]];
        for (auto& r : k.reads)
          if (!inside.count(root(plan, r.tensor))) rd(r.tensor);
        const int w = root(plan, k.write.tensor);
        io.writes.insert(w);
        // a summed tensor is written whole by the fold behind the kernel; a plain store needs nothing before it
        // (a tensor that lives in the block's LDS starts from zero there)
        if (!sg.g.infos[i].reduced && !sg.g.overwrite[i] && !inside.count(w) && !sg.g.lds.count(k.write.tensor)) io.needs_prior.insert(w);
        inside.insert(w);
      }
      break;
    }
    case StepKind::SmallFused: {
      const PlanSmallGroup& sg = *plan.small_groups[L.row_group];
      // every kernel of a small / map group accumulates; a tensor written earlier in the group
      // counts as present for the later kernels
      std::set<int> inside;
      for (int ki : sg.g.kernel_index) {
        const Kernel& k = t.all[ki];
        for (auto& r : k.reads)
          if (!inside.count(root(plan, r.tensor))) rd(r.tensor);
        const int w = root(plan, k.write.tensor);
        io.writes.insert(w);
        if (!inside.count(w)) io.needs_prior.insert(w);
        inside.insert(w);
      }
      break;
    }
  }
  (void)m;
}

#define EG_PLAN_REQUIRE(cond, ...)                                     \
  do {                                                                 \
    if (!(cond)) {                                                     \
      char _msg[512];                                                  \
      snprintf(_msg, sizeof(_msg), __VA_ARGS__);                       \
      set_error("plan invariant violated (target %s): %s", ts.target->name.c_str(), _msg); \
      return EG_ERR_RUNTIME;                                           \
    }                                                                  \
  } while (0)

}  // namespace

int check_plan(eg_model* m, TargetState& ts, Plan& plan) {
  static const bool off = [] {
    const char* e = eg::sw::raw("EG_NO_PLAN_CHECK");
    return e && e[0] && e[0] != '0';
  }();
  if (off) return EG_OK;
  const Target& t = *ts.target;
  const Program& prog = m->prog;
  const int n = (int)plan.launches.size();

  // ---- 1. coverage: live kernel -> how it is represented
  {
    std::vector<int> seen(t.live.size(), 0);
    std::map<int, int> pos_of_all;  // index into t.all -> position in live
    for (size_t p = 0; p < t.live.size(); ++p) pos_of_all[t.live[p]] = (int)p;
    auto mark = [&](int p) {
      if (p >= 0 && p < (int)seen.size()) seen[p]++;
    };
    for (const Launch& L : plan.launches) {
      switch (L.kind) {
        case StepKind::RowFused:
          for (int ki : plan.row_groups[L.row_group]->g.kernel_index) mark(pos_of_all.count(ki) ? pos_of_all[ki] : -1);
          break;
        case StepKind::SmallFused:
          for (int ki : plan.small_groups[L.row_group]->g.kernel_index) mark(pos_of_all.count(ki) ? pos_of_all[ki] : -1);
          break;
        case StepKind::SampleFused:
          for (int pos : plan.sample_group->positions) mark(pos);
          break;
        case StepKind::GemmFused:
          mark(L.lowered);
          mark(plan.epilogues[L.epilogue]->consumer.lowered);
          if (plan.epilogues[L.epilogue]->row_product) mark(plan.epilogues[L.epilogue]->product.lowered);
          break;
        default:
          mark(L.lowered);
          if (L.consumer >= 0) mark(L.consumer);
          if (L.ones_lowered >= 0) mark(L.ones_lowered);
      }
    }
    for (size_t p = 0; p < t.live.size(); ++p) {
      const Lowered& lo = ts.lowered[p];
      const Kernel& k = t.all[t.live[p]];
      const bool aliased = plan.alias.count(k.write.tensor) != 0 && lo.kind == StepKind::GenericA && !lo.absorbed;
      // an absorbed kernel (folded bias, inlined producer) rides with another one, unless its whole
      // run went into a row group, where it is a member in its own right
      if (lo.absorbed) {
        EG_PLAN_REQUIRE(seen[p] <= 1, "absorbed kernel %zu is launched %d times", p, seen[p]);
        continue;
      }
      if (aliased) {
        EG_PLAN_REQUIRE(seen[p] == 0, "kernel %zu is both a storage-sharing copy and a launch", p);
        continue;
      }
      EG_PLAN_REQUIRE(seen[p] == 1, "live kernel %zu (lowered as %d, writes tensor %d, consumer %d) is represented %d times in the launch list",
                      p, (int)lo.kind, k.write.tensor, lo.consumer, seen[p]);
    }
  }
  EG_PLAN_REQUIRE(plan.n_backward >= 0 && plan.n_backward <= n, "backward | update boundary %d outside [0, %d]", plan.n_backward, n);

  // ---- 2. storage
  {
    std::vector<std::pair<long, long>> slots;  // [begin, end) in floats
    for (auto& kv : plan.arena_offset) {
      auto sh = plan.shapes.find(kv.first);
      EG_PLAN_REQUIRE(sh != plan.shapes.end(), "arena tensor %d has no shape", kv.first);
      const long count = storage_floats(plan, kv.first);
      EG_PLAN_REQUIRE(kv.second % 4 == 0, "arena slot of tensor %d is not 16-byte aligned", kv.first);
      EG_PLAN_REQUIRE(kv.second >= 0 && kv.second + count <= plan.arena_floats, "tensor %d leaves the arena", kv.first);
      slots.push_back({kv.second, kv.second + count});
    }
    std::sort(slots.begin(), slots.end());
    for (size_t i = 1; i < slots.size(); ++i)
      EG_PLAN_REQUIRE(slots[i].first >= slots[i - 1].second, "arena slots [%ld, %ld) and [%ld, %ld) overlap", slots[i - 1].first,
                      slots[i - 1].second, slots[i].first, slots[i].second);
    EG_PLAN_REQUIRE(plan.zero_floats >= 0 && plan.zero_floats <= plan.arena_floats, "zero prefix larger than the arena");
    for (auto& kv : plan.alias) {
      const int r = root(plan, kv.first);
      EG_PLAN_REQUIRE(!plan.alias.count(r), "storage-sharing chain of tensor %d does not end", kv.first);
      auto a = plan.shapes.find(kv.first), b = plan.shapes.find(r);
      EG_PLAN_REQUIRE(a != plan.shapes.end() && b != plan.shapes.end() && prod(a->second) == prod(b->second),
                      "tensor %d shares the storage of tensor %d of another size", kv.first, r);
    }
  }

  // ---- 3. order: what exists before the first launch
  std::set<int> present;
  for (size_t tid = 1; tid < prog.tensors.size(); ++tid) {
    const TK kind = prog.tensors[tid].kind;
    if (kind != TK::Result) present.insert((int)tid);  // inputs, parameters, caches, random tensors (refilled per run)
  }
  for (auto& kv : plan.arena_offset) {
    auto sh = plan.shapes.find(kv.first);
    const long count = sh == plan.shapes.end() ? 0 : storage_floats(plan, kv.first);
    if (kv.second + count <= plan.zero_floats) present.insert(kv.first);
  }
  for (int tid : plan.bucket_zero) present.insert(tid);
  std::vector<IO> ios((size_t)n);
  for (int i = 0; i < n; ++i) {
    const Launch& L = plan.launches[i];
    IO& io = ios[(size_t)i];
    launch_io(m, ts, plan, L, io);
    for (int r : io.reads) {
      auto sh = plan.shapes.find(r);
      const long count = sh == plan.shapes.end() ? 0 : prod(sh->second);
      EG_PLAN_REQUIRE(present.count(r) || count == 0, "launch %d reads tensor %d, which is neither zeroed nor written before", i, r);
    }
    for (int w : io.needs_prior) {
      auto sh = plan.shapes.find(w);
      const long count = sh == plan.shapes.end() ? 0 : prod(sh->second);
      EG_PLAN_REQUIRE(present.count(w) || count == 0,
                      "launch %d accumulates into tensor %d, which is neither zeroed nor written before", i, w);
    }
    for (int x : io.reads)
      if (prog.tensors[x].kind == TK::Result || prog.tensors[x].kind == TK::Random) {
        auto sh = plan.shapes.find(x);
        if (sh != plan.shapes.end() && prod(sh->second) > 0)
          EG_PLAN_REQUIRE(tensor_ptr(m, ts, plan, x) != nullptr, "tensor %d (read by launch %d) has no storage", x, i);
      }
    for (int x : io.writes) {
      auto sh = plan.shapes.find(x);
      if (sh != plan.shapes.end() && prod(sh->second) > 0)
        EG_PLAN_REQUIRE(tensor_ptr(m, ts, plan, x) != nullptr, "tensor %d (written by launch %d) has no storage", x, i);
      present.insert(x);
    }
    // predicate tensors: one bit per element — only launches that know may touch them
    for (int x : io.reads)
      if (plan.predicated.count(x))
        EG_PLAN_REQUIRE(L.kind == StepKind::GemmFused && plan.epilogues[L.epilogue]->pred_reads.count(x),
                        "launch %d reads tensor %d, which exists as predicate bits only, as values", i, x);
    for (int x : io.writes)
      if (plan.predicated.count(x))
        EG_PLAN_REQUIRE(L.kind == StepKind::GemmFused && plan.epilogues[L.epilogue]->pred_write && L.c_tensor == x,
                        "launch %d writes values into tensor %d, which exists as predicate bits only", i, x);
  }

  // ---- 4. side lane
  int last_big = -1;
  for (const Plan::Overlap& ov : plan.overlaps) {
    EG_PLAN_REQUIRE(ov.first >= 0 && ov.first < ov.big && ov.big < n, "overlap group [%d, %d) outside the launch list", ov.first, ov.big);
    EG_PLAN_REQUIRE(ov.first > last_big, "overlap groups [.., %d] and [%d, %d) intersect", last_big, ov.first, ov.big);
    EG_PLAN_REQUIRE(!(ov.first < plan.n_backward && ov.big >= plan.n_backward),
                    "overlap group [%d, %d] crosses the backward | update boundary %d", ov.first, ov.big, plan.n_backward);
    const IO& big = ios[(size_t)ov.big];
    for (int s = ov.first; s < ov.big; ++s) {
      const IO& side = ios[(size_t)s];
      for (int x : side.reads) EG_PLAN_REQUIRE(!big.writes.count(x), "side launch %d reads tensor %d, written by launch %d next to it", s, x, ov.big);
      for (int x : side.writes) {
        EG_PLAN_REQUIRE(!big.writes.count(x), "side launch %d and launch %d both write tensor %d", s, ov.big, x);
        EG_PLAN_REQUIRE(!big.reads.count(x), "side launch %d writes tensor %d, read by launch %d next to it", s, x, ov.big);
      }
    }
    last_big = ov.big;
  }
  return EG_OK;
}

}  // namespace model
}  // namespace eg
