// Side lane: bandwidth-bound launches that run next to a long contraction (second stream).
#include "model_types.hpp"


namespace eg {
namespace model {

// ---- side lane --------------------------------------------------------------------------------
int ensure_side_lane(eg_ctx* ctx) {
  if (ctx->side_stream) return EG_OK;
  EG_HIP_CHECK(hipSetDevice(ctx->device));
  EG_HIP_CHECK(hipStreamCreateWithFlags(&ctx->side_stream, hipStreamNonBlocking));
  EG_HIP_CHECK(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
  EG_HIP_CHECK(hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
  return EG_OK;
}

// Tensors a launch reads / writes (storage-sharing tensors folded onto their source); false for
// launch kinds that do not take part in overlap groups.
bool launch_tensors(const Plan& plan, const Launch& L, std::set<int>& reads, std::set<int>& writes) {
  auto root = [&](int t) {
    for (int guard = 0; guard < 64; ++guard) {
      auto a = plan.alias.find(t);
      if (a == plan.alias.end()) break;
      t = a->second;
    }
    return t;
  };
  auto rd = [&](int t) {
    if (t) reads.insert(root(t));
  };
  auto wr = [&](int t) {
    if (t) {
      writes.insert(root(t));
      reads.insert(root(t));  // accumulate / partial overwrite: conservative
    }
  };
  switch (L.kind) {
    case StepKind::Gemm:
      rd(L.a_tensor);
      rd(L.b_tensor);
      rd(L.bias_tensor);
      wr(L.c_tensor);
      wr(L.ones_tensor);
      return true;
    case StepKind::GemmFused:
      rd(L.a_tensor);
      rd(L.b_tensor);
      rd(L.bias_tensor);
      wr(L.c_tensor);
      for (int t : plan.epilogues[L.epilogue]->spec.operands) wr(t);
      return true;
    case StepKind::GenericA:
    case StepKind::GenericB:
      // mode A lists the written tensor first; mode B writes its partial sums and lists reads only
      for (size_t i = 0; i < L.generic->src.tensor_args.size(); ++i) {
        if (i == 0 && L.kind == StepKind::GenericA) wr(L.generic->src.tensor_args[i]);
        else rd(L.generic->src.tensor_args[i]);
      }
      wr(L.c_tensor);
      return true;
    default: return false;
  }
}

// A long contraction keeps the matrix cores busy and leaves the memory system idle; the
// bandwidth-bound launches just before it that it does not depend on (dense backward: the bias
// gradient's column sum and the small weight gradient before the large weight gradient) run next to
// it on the side lane instead of in front of it (tools/overlap_probe.py: 582 -> 511 us).
void plan_overlap(eg_model* m, TargetState& ts, Plan& plan) {
  (void)ts;
  static const bool off = [] {
    const char* e = eg::sw::raw("EG_NO_OVERLAP");
    return e && e[0] && e[0] != '0';
  }();
  plan.overlaps.clear();
  if (off) return;
  const int n = (int)plan.launches.size();
  for (int j = 1; j < n; ++j) {
    const Launch& B = plan.launches[j];
    if (B.kind != StepKind::Gemm && B.kind != StepKind::GemmFused) continue;
    const double flops = 2.0 * (double)B.M * (double)B.N * (double)B.K;
    if (flops < 8e9) continue;
    std::set<int> br, bw;
    if (!launch_tensors(plan, B, br, bw)) continue;
    int first = j;
    while (first > 0 && j - first < 4) {
      const int i = first - 1;
      if (i + 1 == plan.n_backward) break;  // never across the backward | update boundary
      if (!plan.overlaps.empty() && i <= plan.overlaps.back().big) break;
      const Launch& S = plan.launches[i];
      if (S.kind != StepKind::Gemm && S.kind != StepKind::GenericA && S.kind != StepKind::GenericB) break;
      if (S.kind == StepKind::Gemm && 2.0 * (double)S.M * (double)S.N * (double)S.K * 4 > flops) break;
      std::set<int> sr, sw;
      if (!launch_tensors(plan, S, sr, sw)) break;
      bool clash = false;
      for (int t : sr) clash = clash || bw.count(t);   // S reads (or writes) what the contraction writes
      for (int t : sw) clash = clash || br.count(t);   // S writes what the contraction reads (or writes)
      if (clash) break;
      first = i;
    }
    static const bool debug = eg::sw::raw("EG_DEBUG_OVERLAP") != nullptr;
    if (debug) fprintf(stderr, "[eg] overlap: contraction %d (%.1f GFLOP) takes launches [%d, %d)\n", j, flops / 1e9, first, j);
    if (first < j && ensure_side_lane(m->ctx) == EG_OK) {
      plan.overlaps.push_back({first, j});
      // Round 6: a row group shortly in front of the group whose totals (a bias gradient's column sums) nobody reads before
      // the contraction is over hands its fold to the side lane: in the dense step the row kernel's in-kernel fold was
      // 5 of its 12.8 us, on the critical path between the forward product and the backward chain.
      const bool keep_tail = eg::sw::raw("EG_NO_DEFERRED_FOLD") != nullptr;   // (read per plan: a test builds one model each way)
      for (int r = first - 1; !keep_tail && r >= 0 && r >= first - 3; --r) {
        const Launch& R = plan.launches[r];
        if (r + 1 == plan.n_backward) break;
        if (R.kind != StepKind::RowFused) continue;
        const PlanRowGroup& pg = *plan.row_groups[R.row_group];
        if (pg.g.single_block || pg.g.red_total <= 0 || pg.tail_group >= 0 || !pg.g.in_kernel_finalize) break;
        std::set<int> totals;
        for (int tid : pg.red_tensors) {
          int t = tid;
          for (int guard = 0; guard < 64; ++guard) {
            auto al = plan.alias.find(t);
            if (al == plan.alias.end()) break;
            t = al->second;
          }
          totals.insert(t);
        }
        bool free_of_readers = true;
        for (int k = r + 1; k <= j && free_of_readers; ++k) {
          std::set<int> kr, kw;
          if (!launch_tensors(plan, plan.launches[k], kr, kw)) free_of_readers = false;
          for (int t : totals) free_of_readers = free_of_readers && !kr.count(t) && !kw.count(t);
        }
        if (free_of_readers) plan.overlaps.back().deferred_row = r;
        break;
      }
    }
  }
}

}  // namespace model
}  // namespace eg
