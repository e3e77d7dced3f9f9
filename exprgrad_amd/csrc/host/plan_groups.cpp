// Fusion groups of a plan: row groups (rowfuse.hpp), small-kernel groups, map groups.
#include "model_types.hpp"


namespace eg {
namespace model {

bool row_fusion_enabled() {
  static const bool on = [] {
    const char* e = getenv("EG_NO_ROWFUSE");
    return !(e && e[0] && e[0] != '0');
  }();
  return on;
}

bool row_tails_enabled() {  // (read when a plan is made: a test builds one model each way)
  const char* e = getenv("EG_NO_ROW_TAIL");
  return !(e && e[0] && e[0] != '0');
}

// Is the tail of this RowFused launch part of the range being issued?  (run.cpp: MODE 2 and the tail launch skipped.)
bool row_tail_active(const Plan& plan, const Launch& L) {
  return L.kind == StepKind::RowFused && L.tail_launch >= 0 && L.tail_launch < plan.active_end && L.tail_launch >= plan.active_begin;
}

// A small / map group (the optimizer updates of a small net) that DIRECTLY follows a multi-block row group whose last
// block folds the partial rows: that block goes on with the group's kernels — everything they read is complete by then
// (every block has arrived), they are small (is_small_kernel: tensors of at most 4096 elements), and one 256-thread
// block runs them exactly as a small group's kernel would.  The XOR step (rows + finalize + update: three dependent
// launches of ~5 us each) becomes ONE launch.  Not across an overlap group or under the batch pipeline.
int fuse_row_tails(eg_model* m, TargetState& ts, Plan& plan, const std::vector<KernelInfo>& infos) {
  if (!row_tails_enabled() || plan.pipe.active) return EG_OK;
  Target& t = *ts.target;
  auto in_overlap = [&](int i) {
    for (auto& ov : plan.overlaps)
      if (i >= ov.first && i <= ov.big) return true;
    return false;
  };
  for (int i = 0; i + 1 < (int)plan.launches.size(); ++i) {
    Launch& R = plan.launches[i];
    Launch& S = plan.launches[i + 1];
    if (R.kind != StepKind::RowFused || S.kind != StepKind::SmallFused) continue;
    PlanRowGroup& pg = *plan.row_groups[R.row_group];
    if (!pg.g.in_kernel_finalize || pg.tail_group >= 0 || in_overlap(i) || in_overlap(i + 1)) continue;
    PlanSmallGroup& sg = *plan.small_groups[S.row_group];
    long work = 0;
    bool ok = true;
    for (int ki : sg.g.kernel_index) {
      const Kernel& k = t.all[ki];
      ok = ok && is_small_kernel(m->prog, k, infos[ki], plan.shapes);
      long w = 1;
      for (size_t l = 0; l < k.loops.size() && ok; ++l) w *= std::max(0L, infos[ki].bounds[l].second - infos[ki].bounds[l].first);
      work += w;
    }
    if (!ok || work > 16384) continue;
    pg.g.tail_kernels = sg.g.kernel_index;
    int rc = generate_row_group(m->prog, t.all, infos, plan.shapes, pg.g);
    if (rc) return rc;
    bool replaced = false;
    for (auto& pk : plan.pending)
      if (pk.slot == &pg.handle) {
        pk.source = pg.g.source;
        replaced = true;
      }
    if (!replaced) {  // (cannot happen: the group's kernel is built with the plan)
      pg.g.tail_kernels.clear();
      return generate_row_group(m->prog, t.all, infos, plan.shapes, pg.g);
    }
    pg.tail_group = S.row_group;
    R.tail_launch = i + 1;
    S.tail_of = i;
    // A launch-bound step: fewer, longer blocks (the kernel walks its samples with a grid stride) — 64 arrivals at the
    // ticket counter instead of 256 (a fan-in of 255 costs ~3.3 us, MI355X_MICROARCH.md price list) and 64 partial rows
    // for the last block; 786 KB of input do not need 256 CUs.  Per-thread totals then span several samples: the sums
    // change their order (not their terms) against the one-sample-per-thread launch.
    long cap = 64;
    if (const char* e = getenv("EG_ROW_TAIL_BLOCKS")) cap = std::max(1L, atol(e));
    if (pg.nblocks > cap) pg.nblocks = (int)cap;
  }
  return EG_OK;
}

// Partition the live kernel list into row groups (rowfuse.hpp) and build their kernels.
// group_of[p] = index into plan.row_groups, or -1 for kernels that keep their own launch.
int form_row_groups(eg_model* m, TargetState& ts, Plan& plan, const std::vector<KernelInfo>& infos,
                    const std::map<int, int>& first_writer, std::vector<int>& group_of) {
  Target& t = *ts.target;
  const Shapes& shapes = plan.shapes;
  if (!row_fusion_enabled()) return EG_OK;
  long B = 0;
  for (auto& in : m->inputs)
    if (in.second.bound && !in.second.shape.empty()) {
      B = in.second.shape[0];
      break;
    }
  if (B <= 0) return EG_OK;
  const int n = (int)t.live.size();
  std::vector<RowKernelInfo> rki(n);
  for (int p = 0; p < n; ++p) rki[p] = analyse_row_kernel(m->prog, t.all[t.live[p]], infos[t.live[p]], shapes, B);
  // a bias folded into a library contraction is not available on its own
  for (int p = 1; p < n; ++p)
    if (ts.lowered[p].absorbed && !rki[p - 1].ok) rki[p].ok = false;

  constexpr long LOCAL_BUDGET = 192;  // floats of per-thread state
  constexpr long RED_MAX = 512;
  int p = 0;
  while (p < n) {
    if (!rki[p].ok) {
      ++p;
      continue;
    }
    // grow a group from p
    std::unique_ptr<PlanRowGroup> pg(new PlanRowGroup());
    RowGroup& g = pg->g;
    g.B = B;
    int q = p;
    int row_kernels = 0;
    while (q < n && rki[q].ok && !(q != p && q == t.first_update) && !(p < t.first_update && q >= t.first_update && t.first_update >= 0)) {
      const Kernel& k = t.all[t.live[q]];
      const RowKernelInfo& ri = rki[q];
      // tentative roles with this kernel added
      std::map<int, RowGroupTensor> roles = g.tensors;
      bool ok = true;
      const int yreg = ri.row_loop >= 0 ? k.loops[ri.row_loop].reg : 0;
      auto op_is_row = [&](const Op& op) {
        if (ri.row_loop < 0) return false;
        for (auto& d : op.dims)
          if (d.factor_of(yreg)) return true;
        return false;
      };
      auto touch = [&](const Op& op, bool write) {
        const std::vector<long>& shp = shapes.at(op.tensor);
        auto it = roles.find(op.tensor);
        RowGroupTensor gt;
        if (it != roles.end()) gt = it->second;
        gt.tensor = op.tensor;
        if (op_is_row(op)) {
          const long inner = prod(shp) / B;
          if (it != roles.end() && gt.role != RowGroupTensor::RowLocal && gt.role != RowGroupTensor::RowExternal) ok = false;
          gt.inner = inner;
          if (write) gt.role = RowGroupTensor::RowLocal;
          else if (it == roles.end()) gt.role = RowGroupTensor::RowExternal;
        } else {
          const long count = prod(shp);
          if (it != roles.end() && (gt.role == RowGroupTensor::RowLocal || gt.role == RowGroupTensor::RowExternal)) ok = false;
          gt.inner = count;
          if (write) {
            const RowGroupTensor::Role want = ri.small_only ? RowGroupTensor::SmallLocal : RowGroupTensor::Reduction;
            if (it != roles.end() && gt.role != want && gt.role != RowGroupTensor::SmallExternal) ok = false;
            if (it != roles.end() && gt.role == RowGroupTensor::SmallExternal) ok = false;  // read earlier in the group
            gt.role = want;
          } else {
            if (it != roles.end() && gt.role == RowGroupTensor::Reduction) ok = false;  // needs the grid-wide total
            if (it == roles.end()) gt.role = RowGroupTensor::SmallExternal;
          }
        }
        roles[op.tensor] = gt;
      };
      for (auto& rd : k.reads) touch(rd, false);
      touch(k.write, true);
      if (ok && ri.small_only) {
        // thread-local recomputation only: the tensor must not exist outside the group
        const int wt = k.write.tensor;
        auto fw = first_writer.find(wt);
        if (fw == first_writer.end() || fw->second != q || wt == t.output) ok = false;
      }
      long locals = 0, reds = 0;
      int segs = 0;
      for (auto& kv : roles) {
        if (kv.second.role == RowGroupTensor::RowLocal || kv.second.role == RowGroupTensor::SmallLocal) locals += kv.second.inner;
        if (kv.second.role == RowGroupTensor::Reduction) {
          locals += kv.second.inner;
          reds += kv.second.inner;
          ++segs;
        }
      }
      if (locals > LOCAL_BUDGET || reds > RED_MAX || segs > 16) ok = false;
      // a contraction keeps its folded bias in the same group
      if (ok && q + 1 < n && ts.lowered[q + 1].absorbed && !rki[q + 1].ok) ok = false;
      if (!ok) break;
      g.tensors = roles;
      g.kernel_index.push_back(t.live[q]);
      g.infos.push_back(ri);
      if (!ri.small_only) ++row_kernels;
      ++q;
    }
    // an absorbed bias must not be split from its contraction by the end of the group
    while (q > p && q < n && ts.lowered[q].absorbed) {
      --q;
      g.kernel_index.pop_back();
      g.infos.pop_back();
    }
    row_kernels = 0;
    for (auto& ri : g.infos)
      if (!ri.small_only) ++row_kernels;
    if (q - p < 2 || row_kernels < 2) {
      p = std::max(q, p + 1);
      continue;
    }
    // roles may contain tensors of kernels popped above or of the kernel that failed: rebuild exactly
    g.tensors.clear();
    {
      std::vector<int> keep = g.kernel_index;
      std::vector<RowKernelInfo> keep_infos = g.infos;
      g.kernel_index.clear();
      g.infos.clear();
      for (size_t i = 0; i < keep.size(); ++i) {
        const Kernel& k = t.all[keep[i]];
        const RowKernelInfo& ri = keep_infos[i];
        const int yreg = ri.row_loop >= 0 ? k.loops[ri.row_loop].reg : 0;
        auto touch = [&](const Op& op, bool write) {
          bool row = false;
          if (ri.row_loop >= 0)
            for (auto& d : op.dims)
              if (d.factor_of(yreg)) row = true;
          RowGroupTensor& gt = g.tensors[op.tensor];
          gt.tensor = op.tensor;
          const long count = prod(shapes.at(op.tensor));
          if (row) {
            gt.inner = count / B;
            if (write) gt.role = RowGroupTensor::RowLocal;
            else if (gt.role != RowGroupTensor::RowLocal) gt.role = RowGroupTensor::RowExternal;
          } else {
            gt.inner = count;
            if (write) gt.role = ri.small_only ? RowGroupTensor::SmallLocal : RowGroupTensor::Reduction;
            else if (gt.role != RowGroupTensor::SmallLocal && gt.role != RowGroupTensor::Reduction)
              gt.role = RowGroupTensor::SmallExternal;
          }
        };
        for (auto& rd : k.reads) touch(rd, false);
        touch(k.write, true);
        g.kernel_index.push_back(keep[i]);
        g.infos.push_back(ri);
      }
    }
    // liveness: what must come from / go to memory
    auto written_outside_before = [&](int tensor) {
      for (int s = 0; s < p; ++s)
        if (t.all[t.live[s]].write.tensor == tensor) return true;
      return false;
    };
    auto used_after = [&](int tensor) {
      if (tensor == t.output) return true;
      for (int s = q; s < n; ++s) {
        const Kernel& k = t.all[t.live[s]];
        if (k.write.tensor == tensor) return true;
        for (auto& rd : k.reads)
          if (rd.tensor == tensor) return true;
      }
      return false;
    };
    bool leaks = false;  // a thread-local small tensor somebody outside the group wants
    for (auto& kv : g.tensors)
      if (kv.second.role == RowGroupTensor::SmallLocal && (used_after(kv.first) || written_outside_before(kv.first)))
        leaks = true;
    if (leaks) {
      p = std::max(q, p + 1);
      continue;
    }
    long red_off = 0;
    for (auto& kv : g.tensors) {
      RowGroupTensor& gt = kv.second;
      if (gt.role == RowGroupTensor::RowLocal) {
        gt.load_first = written_outside_before(kv.first);
        gt.store = used_after(kv.first);
      } else if (gt.role == RowGroupTensor::Reduction) {
        const TK kind = m->prog.tensors[kv.first].kind;
        gt.accumulate = kind != TK::Result || written_outside_before(kv.first);
        gt.red_offset = red_off;
        red_off += gt.inner;
        pg->red_tensors.push_back(kv.first);
      }
    }
    g.red_total = red_off;
    char name[64];
    snprintf(name, sizeof(name), "eg_rows%d", m->kernel_serial++);
    g.name = name;
    pg->nblocks = (int)((B + 255) / 256);
    // several blocks: the last one to arrive folds the partial rows itself (rowfuse.hpp) — every thread of it keeps one
    // accumulator per total in registers (at most 64 totals), rows in strides of 256
    g.in_kernel_finalize = row_tails_enabled() && g.red_total > 0 && g.red_total <= 64 && pg->nblocks > 1 && pg->nblocks <= 4096;
    int rc = generate_row_group(m->prog, t.all, infos, shapes, g);
    if (rc) return rc;
    plan.pending.push_back({g.name, g.source, &pg->handle});
    if (g.red_total > 0) {
      EG_HIP_CHECK(hipSetDevice(m->ctx->device));
      EG_HIP_CHECK(hipMalloc((void**)&pg->partial, (size_t)pg->nblocks * g.red_total * sizeof(float)));
      if (g.in_kernel_finalize) {
        EG_HIP_CHECK(hipMalloc((void**)&pg->counter, 64));
        EG_HIP_CHECK(hipMemsetAsync(pg->counter, 0, 64, m->ctx->stream));  // (the context's stream: ordered against the launches)
      }
    }
    const int gi = (int)plan.row_groups.size();
    for (int s = p; s < q; ++s) group_of[s] = gi;
    plan.row_groups.push_back(std::move(pg));
    p = q;
  }
  // ---- small-kernel groups among what is left (encoded as -2 - index in group_of)
  p = 0;
  while (p < n) {
    auto eligible = [&](int s) {
      if (group_of[s] != -1 || ts.lowered[s].absorbed || ts.lowered[s].bias_tensor) return false;
      return is_small_kernel(m->prog, t.all[t.live[s]], infos[t.live[s]], shapes);
    };
    if (!eligible(p)) {
      ++p;
      continue;
    }
    int q = p;
    while (q < n && eligible(q) && !(q != p && q == t.first_update)) ++q;
    if (q - p >= 2) {
      // a run made of elementwise maps only (adam: m, v and parameter updates of every parameter) is
      // better off as a map group below: many blocks instead of one
      bool all_maps = true;
      for (int s = p; s < q && all_maps; ++s) {
        long count = 0;
        all_maps = ts.lowered[s].kind == StepKind::GenericA &&
                   is_map_kernel(m->prog, t.all[t.live[s]], infos[t.live[s]], shapes, count);
      }
      if (all_maps) {
        p = q;
        continue;
      }
      std::unique_ptr<PlanSmallGroup> sg(new PlanSmallGroup());
      for (int s = p; s < q; ++s) sg->g.kernel_index.push_back(t.live[s]);
      char name[64];
      snprintf(name, sizeof(name), "eg_small%d", m->kernel_serial++);
      sg->g.name = name;
      int rc = generate_small_group(m->prog, t.all, infos, shapes, sg->g);
      if (rc) return rc;
      plan.pending.push_back({sg->g.name, sg->g.source, &sg->handle});
      const int gi = (int)plan.small_groups.size();
      for (int s = p; s < q; ++s) group_of[s] = -2 - gi;
      plan.small_groups.push_back(std::move(sg));
    }
    p = std::max(q, p + 1);
  }
  // ---- map groups: runs of raw elementwise kernels of any size among what is still separate (the
  //      per-parameter optimizer kernels); same encoding and launch path as the small groups
  p = 0;
  while (p < n) {
    auto eligible = [&](int s) {
      if (group_of[s] != -1 || ts.lowered[s].absorbed || ts.lowered[s].kind != StepKind::GenericA) return false;
      long count = 0;
      return is_map_kernel(m->prog, t.all[t.live[s]], infos[t.live[s]], shapes, count);
    };
    if (!eligible(p)) {
      ++p;
      continue;
    }
    int q = p;
    while (q < n && eligible(q) && !(q != p && q == t.first_update)) ++q;
    if (q - p >= 2) {
      std::unique_ptr<PlanSmallGroup> sg(new PlanSmallGroup());
      for (int s = p; s < q; ++s) sg->g.kernel_index.push_back(t.live[s]);
      char name[64];
      snprintf(name, sizeof(name), "eg_maps%d", m->kernel_serial++);
      sg->g.name = name;
      int rc = generate_map_group(m->prog, t.all, infos, shapes, sg->g);
      if (rc) return rc;
      plan.pending.push_back({sg->g.name, sg->g.source, &sg->handle});
      const int gi = (int)plan.small_groups.size();
      for (int s = p; s < q; ++s) group_of[s] = -2 - gi;
      plan.small_groups.push_back(std::move(sg));
    }
    p = std::max(q, p + 1);
  }
  return EG_OK;
}

}  // namespace model
}  // namespace eg
