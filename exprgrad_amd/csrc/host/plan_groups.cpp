// Fusion groups of a plan: row groups (rowfuse.hpp), small-kernel groups, map groups.
#include "model_types.hpp"


namespace eg {
namespace model {

bool row_fusion_enabled() {
  static const bool on = [] {
    const char* e = eg::sw::raw("EG_NO_ROWFUSE");
    return !(e && e[0] && e[0] != '0');
  }();
  return on;
}

bool row_tails_enabled() {  // (read when a plan is made: a test builds one model each way)
  const char* e = eg::sw::raw("EG_NO_ROW_TAIL");
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
    {  // the grid this launch will use (below), known to the generator: rowfuse.hpp grid_blocks
      long row_floats = 0;
      for (auto& tt : pg.g.tensors)
        if (tt.second.role == RowGroupTensor::RowExternal || (tt.second.role == RowGroupTensor::RowLocal && (tt.second.store || tt.second.load_first)))
          row_floats += tt.second.inner;
      long cap = std::max(64L, (pg.g.B * std::max(1L, row_floats) * 4 + 12287) / 12288);
      cap = eg::sw::integer("EG_ROW_TAIL_BLOCKS", cap);   // tuning aid
      pg.g.grid_blocks = std::min<long>(pg.nblocks, std::max(1L, cap));
    }
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
    // The figure belongs to THAT size: 64 blocks for 786 KB of rows are 12 KB per block.  A larger batch keeps that
    // share per block instead of the count (a 1 M-row group on 64 of 256 CUs with 64-sample serial loops per
    // thread was a multi-x slowdown against ceil(B / 256) blocks); the fan-in it adds is microseconds.
    long row_floats = 0;
    for (auto& tt : pg.g.tensors)
      if (tt.second.role == RowGroupTensor::RowExternal || (tt.second.role == RowGroupTensor::RowLocal && (tt.second.store || tt.second.load_first)))
        row_floats += tt.second.inner;
    const long bytes_touched = pg.g.B * std::max(1L, row_floats) * 4;
    long cap = std::max(64L, (bytes_touched + 12287) / 12288);
    cap = std::max(1L, eg::sw::integer("EG_ROW_TAIL_BLOCKS", cap));   // tuning aid
    if (pg.nblocks > cap) pg.nblocks = (int)cap;
  }
  return EG_OK;
}


bool slab_fold_active(const Plan& plan, const Launch& L) {
  return L.kind == StepKind::SampleFused && L.fold_launch >= 0 && L.fold_launch >= plan.active_begin && L.fold_launch < plan.active_end;
}

// The optimizer's map group directly behind a sample group adds up the slab rows itself (rowfuse.hpp, SmallGroup::fold_*):
// the slab_sum launch between the two disappears when a range holds both (Model.apply / fit; the data-parallel step
// ends its backward range between them and keeps the launch, because the exchange needs the totals in the bucket).
// Every summed tensor must be read by the group as a raw map of its own size; EG_NO_SLAB_FOLD=1 off.
int fuse_slab_fold(eg_model* m, TargetState& ts, Plan& plan, const std::vector<KernelInfo>& infos) {
  {
    const char* e = eg::sw::raw("EG_NO_SLAB_FOLD");
    if (e && e[0] && e[0] != '0') return EG_OK;
  }
  if (!plan.sample_group || plan.sample_group->g.slab_floats <= 0 || plan.pipe.active) return EG_OK;
  // (a thread adds its element's B rows one after the other: beyond a few dozen samples the slab_sum launch, which
  //  spreads the rows over lanes, is the faster fold — batch 256: 69 us per step with the fold, 58 without)
  if (plan.sample_group->g.B > 64) return EG_OK;
  Target& t = *ts.target;
  for (int i = 0; i + 1 < (int)plan.launches.size(); ++i) {
    Launch& S = plan.launches[i];
    Launch& M = plan.launches[i + 1];
    if (S.kind != StepKind::SampleFused || M.kind != StepKind::SmallFused) continue;
    PlanSampleGroup& sg = *plan.sample_group;
    PlanSmallGroup& mg = *plan.small_groups[M.row_group];
    if (mg.g.blocks <= 1 || M.tail_of >= 0) return EG_OK;  // (a map group: one block per 256 elements)
    std::set<int> read_as_map;
    for (int ki : mg.g.kernel_index) {
      const Kernel& k = t.all[ki];
      long count = 0;
      if (!is_map_kernel(m->prog, k, infos[ki], plan.shapes, count)) return EG_OK;
      for (auto& rd : k.reads)
        if (prod(plan.shapes.at(rd.tensor)) == count) read_as_map.insert(rd.tensor);
    }
    for (int tid : sg.sum_tensors)
      if (!read_as_map.count(tid)) return EG_OK;
    for (int ki : mg.g.kernel_index)  // nobody in the group writes a summed tensor (the fold stores into it first)
      if (std::find(sg.sum_tensors.begin(), sg.sum_tensors.end(), t.all[ki].write.tensor) != sg.sum_tensors.end()) return EG_OK;
    mg.g.fold_offset = sg.g.slab_offset;
    mg.g.fold_rows = sg.g.B;
    mg.g.fold_row_floats = sg.g.slab_floats;
    int rc = generate_map_group(m->prog, t.all, infos, plan.shapes, mg.g);
    if (rc) return rc;
    bool replaced = false;
    for (auto& pk : plan.pending)
      if (pk.slot == &mg.handle) {
        pk.source = mg.g.source;
        replaced = true;
      }
    if (!replaced) {
      mg.g.fold_offset.clear();
      return generate_map_group(m->prog, t.all, infos, plan.shapes, mg.g);
    }
    S.fold_launch = i + 1;
    M.fold_of = i;
    return EG_OK;
  }
  return EG_OK;
}

// A run of per-sample kernels of the backward range as ONE kernel with one block per sample (rowfuse.hpp, "sample
// groups"): for small batches, where every launch of the step sits at the floor of a dependent launch.  At most one group
// per plan: the longest run of consecutive live kernels each of which
//   * walks the samples with one loop that indexes dimension 0 of every [B, ...] tensor it touches (or is a raw map over
//     B * inner elements, or the gradLoss seed), and
//   * reads nothing that an earlier member sums over the batch (a block sees its own sample only), and
//   * if it sums over the batch itself, writes a member of the gradient bucket without scatter (its per-sample
//     contributions go to the slab; one slab_sum launch behind the kernel folds them, in a fixed order).
// Switches: EG_NO_SAMPLE_FUSE=1, EG_SAMPLE_FUSE_MAX_BATCH (default 1280), EG_SAMPLE_THREADS (512), EG_SAMPLE_NO_LDS.
int form_sample_group(eg_model* m, TargetState& ts, Plan& plan, const std::vector<KernelInfo>& infos, const std::map<int, int>& first_writer,
                      std::vector<int>& group_of, std::set<int>& needs_zero) {
  {
    const char* e = eg::sw::raw("EG_NO_SAMPLE_FUSE");
    if ((e && e[0] && e[0] != '0') || !row_fusion_enabled()) return EG_OK;
  }
  Target& t = *ts.target;
  const Shapes& shapes = plan.shapes;
  long B = 0;
  for (auto& in : m->inputs)
    if (in.second.bound && !in.second.shape.empty()) {
      B = in.second.shape[0];
      break;
    }
  // (fashion_mnist step, one box, sample group vs launch chain: 43.7 vs 76.5 us at batch 32, 57 vs 119 at 256, 94 vs 136 at 512, 170 vs 183 at
  //  1024, 299 vs 245 at 2048 — until the small-channel convolutions of round 5 made the chain's five convolution launches twice as fast, the
  //  chain took 381 there and the limit was 2048)
  long max_batch = 1280;
  if (const char* e = eg::sw::raw("EG_SAMPLE_FUSE_MAX_BATCH")) max_batch = atol(e);
  if (B < 2 || B > max_batch) return EG_OK;
  const int n = (int)t.live.size();
  const int limit = t.first_update >= 0 ? t.first_update : n;
  std::vector<SampleKernelInfo> ski((size_t)n);
  for (int p = 0; p < limit; ++p) {
    const Lowered& lo = ts.lowered[p];
    if (lo.absorbed && lo.inlined) continue;
    const Kernel& k = t.all[t.live[p]];
    ski[p] = analyse_sample_kernel(m->prog, k, infos[t.live[p]], shapes, B);
    std::vector<int> indep, red;
    bool scatter = false;
    split_loops(k, indep, red, scatter);
    ConvMatch cm;
    if (ski[p].ok && match_conv(k, cm) && cm.role == ConvMatch::GradImage && cm.batched) {
      ski[p].gather = true;
      ski[p].g_img = k.write.tensor;
      ski[p].g_out = k.reads[cm.out_op].tensor;
      ski[p].g_flt = k.reads[cm.flt_op].tensor;
      if (shapes.at(ski[p].g_img).size() != 4 || shapes.at(ski[p].g_out).size() != 4 || shapes.at(ski[p].g_flt).size() != 4) ski[p].ok = false;
    } else if (ski[p].ok && scatter && ski[p].reduced) {
      ski[p].ok = false;  // (the slab row would have to start from zero)
    }
    if (ski[p].ok && !eg::sw::present("EG_SAMPLE_NO_MFMA") && match_conv(k, cm) && cm.batched) {
      // a convolution member on the matrix cores (rowfuse.hpp conv_role): small operand fragments must fit registers
      auto tensor_of = [&](int op) { return op < 0 ? k.write.tensor : k.reads[op].tensor; };
      const int ti = tensor_of(cm.img_op), to = tensor_of(cm.out_op), tf = tensor_of(cm.flt_op);
      const std::vector<long>&is = shapes.at(ti), &os = shapes.at(to), &fs = shapes.at(tf);
      if (is.size() == 4 && os.size() == 4 && fs.size() == 4) {
        const long C = is[3], F = os[3], taps = fs[1] * fs[2];
        long frag = 0;   // registers of the small operand's fragments per lane
        if (cm.role == ConvMatch::Forward) frag = ((F + 15) / 16) * ((taps * C + 3) / 4);
        else if (cm.role == ConvMatch::GradImage) frag = ((C + 15) / 16) * ((taps * F + 3) / 4);
        else frag = ((F + 15) / 16) * ((taps * C + 15) / 16) * 4;   // accumulator blocks
        if (frag <= 64 && (cm.role != ConvMatch::GradFilter || ski[p].reduced)) {
          ski[p].conv_role = cm.role == ConvMatch::Forward ? 1 : cm.role == ConvMatch::GradFilter ? 2 : 3;
          ski[p].conv_img = ti;
          ski[p].conv_out = to;
          ski[p].conv_flt = tf;
        }
      }
    }
    if (ski[p].ok && ski[p].reduced && !ts.bucket_offset.count(k.write.tensor)) ski[p].ok = false;
    if (ski[p].ok && plan.alias.count(k.write.tensor)) ski[p].ok = false;
    // What belongs on the matrix cores stays there: a contraction whose two other extents are both >= 32 (a hidden layer of
    // 64 x 64 and up: at batch 2048 the MFMA tiles finish it in ~10 us, 2048 scalar blocks in ~150), a convolution with
    // >= 16 channels AND >= 16 filters (the halo / implicit-GEMM kernels).  The run ends at such a kernel.
    if (ski[p].ok && ski[p].batch_loop >= 0) {
      GemmMatch gm;
      if (match_gemm(k, gm)) {
        long lo_ext = -1;
        for (size_t l = 0; l < k.loops.size(); ++l) {
          if ((int)l == ski[p].batch_loop) continue;
          const long e = infos[t.live[p]].bounds[l].second - infos[t.live[p]].bounds[l].first;
          lo_ext = lo_ext < 0 ? e : std::min(lo_ext, e);
        }
        if (lo_ext >= 32) ski[p].ok = false;
      } else if (match_conv(k, cm) && cm.batched) {
        const int ft = cm.flt_op < 0 ? k.write.tensor : k.reads[cm.flt_op].tensor;
        const std::vector<long>& fs = shapes.at(ft);
        if (fs.size() == 4 && fs[0] >= 16 && fs[3] >= 16) ski[p].ok = false;
      }
    }
  }
  static const bool debug = eg::sw::raw("EG_DEBUG_SAMPLE") != nullptr;
  if (debug)
    for (int p = 0; p < limit; ++p)
      fprintf(stderr, "[eg] sample: live %d ok %d loop %d raw %d reduced %d seed %d gather %d work %ld absorbed %d inlined %d | %s\n", p, (int)ski[p].ok,
              ski[p].batch_loop, (int)ski[p].raw, (int)ski[p].reduced, (int)ski[p].seed, (int)ski[p].gather, ski[p].work,
              (int)ts.lowered[p].absorbed, (int)ts.lowered[p].inlined, to_text(t.all[t.live[p]]).substr(0, 110).c_str());
  // longest run
  int best_p = 0, best_q = 0, best_members = 0;
  for (int p = 0; p < limit;) {
    if (!(ski[p].ok || (ts.lowered[p].absorbed && ts.lowered[p].inlined))) {
      ++p;
      continue;
    }
    std::set<int> local, summed;
    int q = p, members = 0, last_member = -1;
    long work = 0;
    for (; q < limit; ++q) {
      const Lowered& lo = ts.lowered[q];
      if (lo.absorbed && lo.inlined) continue;  // recomputed inside its readers: nothing to run
      if (!ski[q].ok) break;
      // a bias folded into the contraction in front of it is part of THAT launch: a member only next to its contraction
      if (lo.absorbed && last_member != q - 1) break;
      // ... and a contraction does not leave its folded bias behind
      if (q + 1 < n && ts.lowered[q + 1].absorbed && !ts.lowered[q + 1].inlined && (q + 1 >= limit || !ski[q + 1].ok)) break;
      const Kernel& k = t.all[t.live[q]];
      const int yreg = ski[q].batch_loop >= 0 ? k.loops[ski[q].batch_loop].reg : 0;
      auto by_sample = [&](const Op& op) {
        if (!yreg) return false;
        for (auto& d : op.dims)
          if (d.factor_of(yreg)) return true;
        return false;
      };
      bool ok = true;
      for (auto& rd : k.reads) {
        if (summed.count(rd.tensor)) ok = false;
        if (local.count(rd.tensor) && !by_sample(rd)) ok = false;
      }
      if (ski[q].reduced) {
        if (local.count(k.write.tensor)) ok = false;
      } else if (ski[q].seed) {
        if (summed.count(k.write.tensor) || local.count(k.write.tensor)) ok = false;
      } else {
        if (summed.count(k.write.tensor)) ok = false;
      }
      if (!ok) break;
      if (ski[q].reduced) summed.insert(k.write.tensor);
      else if (!ski[q].seed) local.insert(k.write.tensor);
      work += ski[q].work;
      ++members;
      last_member = q;
    }
    // the run must not end between a contraction and the bias folded into it
    if (q < n && ts.lowered[q].absorbed && !ts.lowered[q].inlined && last_member == q - 1) {
      p = std::max(q, p + 1);
      continue;
    }
    if (work >= 4096 && work <= (1L << 20) && members > best_members) {  // (per sample; a 784 x 512 layer belongs on the matrix cores at any batch)
      best_p = p;
      best_q = q;
      best_members = members;
    }
    p = std::max(q, p + 1);
  }
  if (debug) fprintf(stderr, "[eg] sample: best run [%d, %d) with %d members\n", best_p, best_q, best_members);
  if (best_members < 4) return EG_OK;
  std::unique_ptr<PlanSampleGroup> sg(new PlanSampleGroup());
  SampleGroup& g = sg->g;
  g.B = B;
  std::set<int> sums;
  std::vector<int> zero_these;
  for (int q = best_p; q < best_q; ++q) {
    const Lowered& lo = ts.lowered[q];
    if (lo.absorbed && lo.inlined) continue;
    const Kernel& k = t.all[t.live[q]];
    const KernelInfo& info = infos[t.live[q]];
    const int wt = k.write.tensor;
    std::vector<int> indep, red;
    bool scatter = false;
    split_loops(k, indep, red, scatter);
    char over = 0;
    if (ski[q].reduced) {
      sums.insert(wt);
    } else {
      const bool is_result = m->prog.tensors[wt].kind == TK::Result;
      auto fw = first_writer.find(wt);
      const bool first = is_result && fw != first_writer.end() && fw->second == q;
      const bool whole = ski[q].gather || (!scatter && full_cover(k, info, shapes.at(wt)));
      over = first && whole && !ts.bucket_offset.count(wt) ? 1 : 0;
      if (ski[q].seed && !over) return EG_OK;  // every block would add its own 1
      if (first && !over) zero_these.push_back(wt);
    }
    g.kernel_index.push_back(t.live[q]);
    g.infos.push_back(ski[q]);
    g.overwrite.push_back(over);
    sg->positions.push_back(q);
  }
  // the summed tensors: written by members only, one contiguous range of the bucket
  if (!sums.empty()) {
    for (int p = 0; p < n; ++p) {
      const bool member = std::find(sg->positions.begin(), sg->positions.end(), p) != sg->positions.end();
      if (!member && !(ts.lowered[p].absorbed && ts.lowered[p].inlined) && sums.count(t.all[t.live[p]].write.tensor)) return EG_OK;
    }
    long base = -1, end = 0;
    for (int tid : sums) {
      const long off = ts.bucket_offset.at(tid);
      if (base < 0 || off < base) base = off;
      end = std::max(end, off + align4(prod(shapes.at(tid))));
    }
    for (auto& b : ts.bucket_offset)
      if (b.second >= base && b.second < end && !sums.count(b.first)) return EG_OK;
    sg->bucket_base = base;
    g.slab_floats = end - base;
    for (int tid : sums) {
      g.slab_offset[tid] = ts.bucket_offset.at(tid) - base;
      sg->sum_tensors.push_back(tid);
    }
  }
  // Tensors that live inside the group only stay in the block's LDS (rowfuse.hpp): written by members, every access
  // by sample, nobody outside the run touches them, not the target's output, and the model does not keep values.
  {
    long thr = 512;  // (measured on the fashion_mnist step at batch 32: 256 / 512 / 1024 threads -> 51.9 / 43.9 / 49.8 us per batch)
    g.threads = (int)std::min(1024L, std::max(64L, thr / 64 * 64));
    std::map<int, bool> first_plain;  // candidate -> its first member write is a plain store
    std::set<int> candidates;
    for (size_t i = 0; i < g.kernel_index.size(); ++i) {
      const Kernel& k = t.all[g.kernel_index[i]];
      if (g.infos[i].reduced || g.infos[i].seed) continue;
      const int wt = k.write.tensor;
      if (m->prog.tensors[wt].kind != TK::Result || ts.bucket_offset.count(wt) || wt == t.output) continue;
      if (!candidates.count(wt)) first_plain[wt] = g.overwrite[i] != 0;
      candidates.insert(wt);
    }
    for (int p = 0; p < n; ++p) {
      const bool member = std::find(sg->positions.begin(), sg->positions.end(), p) != sg->positions.end();
      if (member || (ts.lowered[p].absorbed && ts.lowered[p].inlined)) continue;
      const Kernel& k = t.all[t.live[p]];
      candidates.erase(k.write.tensor);
      for (auto& rd : k.reads) candidates.erase(rd.tensor);
    }
    constexpr bool no_lds = false;
    long budget = 34L * 1024;  // floats (136 KB of the 160 KB a block may own; up to 16 KB more for the split reductions)
    if (!m->keep_values && !no_lds)
      for (int tid : candidates) {
        const long inner = prod(shapes.at(tid)) / B;
        if (inner <= 0 || inner > budget) continue;
        budget -= inner;
        g.lds[tid] = inner;
        if (!first_plain[tid]) g.lds_zero.insert(tid);
      }
    // parameters the members only read, into what is left of the block's LDS (at most 48 KB of them): rowfuse.hpp `staged`
    if (!eg::sw::present("EG_SAMPLE_NO_STAGE")) {
      std::set<int> read_only, written_by;
      for (size_t i = 0; i < g.kernel_index.size(); ++i) {
        const Kernel& k = t.all[g.kernel_index[i]];
        written_by.insert(k.write.tensor);
        for (auto& rd : k.reads) read_only.insert(rd.tensor);
      }
      long room = std::min(budget, 12L * 1024);
      std::vector<std::pair<long, int>> by_size;
      for (int tid : read_only)
        if (!written_by.count(tid) && m->prog.tensors[tid].kind == TK::Param) by_size.push_back({prod(shapes.at(tid)), tid});
      std::sort(by_size.begin(), by_size.end());
      for (auto& pr : by_size)
        if (pr.first > 0 && pr.first <= room) {
          g.staged[pr.second] = pr.first;
          room -= pr.first;
        }
    }
    // (what starts from zero in LDS needs no zeroed global storage)
    zero_these.erase(std::remove_if(zero_these.begin(), zero_these.end(), [&](int tid) { return g.lds.count(tid) != 0; }), zero_these.end());
  }
  char name[64];
  snprintf(name, sizeof(name), "eg_samples%d", m->kernel_serial++);
  g.name = name;
  int rc = generate_sample_group(m->prog, t.all, infos, shapes, g);
  if (rc) return rc;
  if (g.slab_floats > 0) {
    EG_HIP_CHECK(hipSetDevice(m->ctx->device));
    EG_HIP_CHECK(hipMalloc((void**)&sg->slab, (size_t)B * g.slab_floats * sizeof(float)));
    // (padding floats between the tensors of a row are never written by a block: zero once, so that the fold writes zeros there)
    EG_HIP_CHECK(hipMemsetAsync(sg->slab, 0, (size_t)B * g.slab_floats * sizeof(float), m->ctx->stream));
  }
  plan.pending.push_back({g.name, g.source, &sg->handle});
  for (int wt : zero_these) needs_zero.insert(wt);
  for (int q = best_p; q < best_q; ++q) group_of[q] = SAMPLE_GROUP_CODE;
  plan.sample_group = std::move(sg);
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
  for (int p = 0; p < n; ++p)
    if (group_of[p] == -1) rki[p] = analyse_row_kernel(m->prog, t.all[t.live[p]], infos[t.live[p]], shapes, B);
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
      EG_HIP_CHECK(hipMalloc((void**)&pg->partial, (size_t)pg->nblocks * g.red_stride() * sizeof(float)));
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
