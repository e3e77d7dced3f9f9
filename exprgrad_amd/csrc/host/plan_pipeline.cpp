// Batch pipeline: a training step as two half batches whose stages interleave on two lanes.
//
// A dense training step alternates long contractions (matrix-core bound: the first layer's forward
// product and weight gradient, 430 - 490 us each at cfg 5) with launches that only stream memory (the
// 10-wide second layer, softmax / loss gradient, the activation gradient: ~150 us together), strictly
// one after the other: every launch needs its predecessor's whole output.  Cut along the batch, the
// dependence is per sample: half A's streaming launches need half A's forward product only, so they
// can run while the matrix cores do half B's forward product, and half B's while they do half A's
// weight gradient:
//
//     main lane   fwd A | fwd B            | gW1 A              | gW1 B (+=)  | update
//     side lane         | D2 A .. gW2 A    | D2 B .. gW2 B (+=) |
//
// The reference has nothing of the kind (every kernel is its own launch, in order: model.nim:392-411);
// this is scheduling only — the same kernels on the same numbers, except that reductions over the batch
// (weight / bias gradients, losses) are formed as (first half) + (second half).  Deterministic.
//
// A plan qualifies when every launch of its backward range can be cut:
//   rows       a contraction whose M is the batch (A rows, C rows, bias per column, epilogue operands
//              [batch, N]): forward products, input-gradient products;
//   reduction  a TN contraction whose K is the batch (weight gradients, with or without the ones row):
//              the second half accumulates;
//   row group  a row-fused launch: row tensors by rows, batch reductions accumulate in the finalize;
// and when nothing inside the range reads a reduction result (they are complete only after both halves).
#include "model_types.hpp"

namespace eg {
namespace model {

void plan_pipeline(eg_model* m, TargetState& ts, Plan& plan) {
  plan.pipe = Plan::Pipeline();
  {
    // OFF unless EG_PIPELINE=1.  Measured on the cfg-5 step (profiles/r02_pipeline_trace.txt): the lanes
    // do overlap, but a 256x256 contraction leaves one small wave per SIMD to whatever runs next to it, and
    // bandwidth-bound kernels live on having many waves in flight — next to it they run 3 to 7 times
    // slower (a 7 us row group: 51 us, an 18 us skinny product: 120 us, even at raised wave priority),
    // so each half's streaming chain (296 us) outlasts the contraction it should hide under (248 us) and
    // slows it by 35 us: 1.144 ms per step against 1.122 ms without the pipeline.
    const char* e = eg::sw::raw("EG_PIPELINE");
    if (!(e && e[0] && e[0] != '0')) return;
  }
  if (!plan.predicated.empty()) return;  // (half-batch slices of predicate-bit tensors are not addressed by run_launch_sliced)
  double min_flops = 2e10;  // only steps with long contractions have something to hide work under
  if (const char* e = eg::sw::raw("EG_PIPELINE_MIN_FLOPS")) min_flops = atof(e);
  const Target& t = *ts.target;
  const int nb = plan.n_backward;
  if (nb < 3) return;
  long B = 0;
  for (auto& in : m->inputs)
    if (in.second.bound && !in.second.shape.empty()) {
      B = in.second.shape[0];
      break;
    }
  if (B < 512 || B % 512 != 0) return;  // halves that are whole 256-row tiles
  auto batch_major = [&](int tensor) {
    auto sh = plan.shapes.find(tensor);
    return sh != plan.shapes.end() && !sh->second.empty() && sh->second[0] == B;
  };
  std::set<int> reductions;  // complete only after both halves
  double heavy_flops = 0;
  int n_heavy = 0, n_light = 0;
  for (int i = 0; i < nb; ++i) {
    Launch& L = plan.launches[i];
    L.slice_mode = 0;
    L.heavy = false;
    auto reads_reduction = [&](std::initializer_list<int> tensors) {
      for (int x : tensors)
        if (x && reductions.count(x)) return true;
      return false;
    };
    if (L.kind == StepKind::Gemm || L.kind == StepKind::GemmFused) {
      if (reads_reduction({L.a_tensor, L.b_tensor, L.bias_tensor})) return;
      const double flops = 2.0 * (double)L.M * (double)L.N * (double)L.K;
      if (L.M == B && !L.trans_a && batch_major(L.a_tensor) && batch_major(L.c_tensor) && L.ldc == L.N) {
        L.slice_mode = 1;
        if (L.kind == StepKind::GemmFused) {
          const PlanEpilogue& pe = *plan.epilogues[L.epilogue];
          for (int op : pe.spec.operands)
            if (!batch_major(op) || reductions.count(op)) return;
        }
      } else if (L.kind == StepKind::Gemm && L.K == B && L.trans_a && !L.trans_b && batch_major(L.a_tensor) && batch_major(L.b_tensor)) {
        L.slice_mode = 2;
        reductions.insert(L.c_tensor);
        if (L.ones_tensor) reductions.insert(L.ones_tensor);
      } else {
        return;
      }
      L.heavy = flops >= min_flops / 4;
      if (L.heavy) {
        heavy_flops += flops;
        ++n_heavy;
      } else {
        ++n_light;
      }
    } else if (L.kind == StepKind::RowFused) {
      const PlanRowGroup& pg = *plan.row_groups[L.row_group];
      if (pg.g.B != B) return;
      for (auto& kv : pg.g.tensors) {
        const RowGroupTensor& gt = kv.second;
        const bool row = gt.role == RowGroupTensor::RowLocal || gt.role == RowGroupTensor::RowExternal;
        if (row && !batch_major(kv.first)) return;
        if ((gt.role == RowGroupTensor::RowExternal || gt.role == RowGroupTensor::SmallExternal) && reductions.count(kv.first)) return;
        if (gt.role == RowGroupTensor::Reduction) reductions.insert(kv.first);
      }
      L.slice_mode = 1;
      ++n_light;
    } else {
      return;  // seeds, generated kernels, convolutions: not cut (yet)
    }
  }
  (void)t;
  if (n_heavy < 2 || n_light < 1 || heavy_flops < min_flops) return;
  // the first and the last launch of the range should be long contractions: otherwise a half starts or
  // ends with streaming work nothing can hide
  plan.pipe.active = true;
  plan.pipe.batch = B;
  plan.pipe.half = B / 2;
  // the pipeline subsumes the side-lane groups of the backward range
  std::vector<Plan::Overlap> keep;
  for (auto& ov : plan.overlaps)
    if (ov.first >= nb) keep.push_back(ov);
  plan.overlaps.swap(keep);
}

}  // namespace model
}  // namespace eg
