// Contraction + elementwise consumer -> one launch (epilogue.hpp).
#include "model_types.hpp"


namespace eg {
namespace model {

// Contraction + elementwise consumer -> one launch (epilogue.hpp).  Only large outputs: the
// fused kernel is built at run time from the matrix kernel's source (seconds), which pays when
// the saved round trip through HBM is megabytes; small chains are launch bound and handled by
// row fusion / graphs.  EG_EPILOGUE_MIN_ELEMS overrides the threshold, EG_NO_EPILOGUE=1 disables.
int fuse_epilogues(eg_model* m, TargetState& ts, Plan& plan, const std::vector<KernelInfo>& infos) {
  static const bool off = [] {
    const char* e = getenv("EG_NO_EPILOGUE");
    return e && e[0] && e[0] != '0';
  }();
  if (off) return EG_OK;
  long min_elems = 1L << 20;
  if (const char* e = getenv("EG_EPILOGUE_MIN_ELEMS")) min_elems = atol(e);
  Target& t = *ts.target;
  plan.epilogues.clear();
  for (size_t i = 0; i + 1 < plan.launches.size(); ++i) {
    Launch& G = plan.launches[i];
    if (G.kind != StepKind::Gemm || G.accumulate) continue;
    if (G.ldc != G.N || G.M * G.N < min_elems || G.M * G.N <= 0) continue;
    // the consumer: the first later launch that reads the contraction result.  It need not be
    // adjacent (derive emits the other gradient contraction of a layer in between), as long as
    // moving it up to the contraction is legal.
    size_t j = i + 1;
    bool found = false;
    for (; j < plan.launches.size() && j <= i + 4; ++j) {
      const Launch& X = plan.launches[j];
      if (X.kind == StepKind::RowFused || X.kind == StepKind::SmallFused || X.kind == StepKind::GemmFused) break;
      const Kernel& kx = t.all[ts.lowered[X.lowered].all_index];
      bool reads_c = false;
      for (auto& rd : kx.reads)
        if (rd.tensor == G.c_tensor) reads_c = true;
      if (reads_c) {
        found = X.kind == StepKind::GenericA;
        break;
      }
    }
    if (!found) continue;
    if (plan.n_backward > (int)i && plan.n_backward <= (int)j) continue;  // straddles the backward / update boundary
    Launch& E = plan.launches[j];
    const Lowered& le = ts.lowered[E.lowered];
    const Kernel& ke = t.all[le.all_index];
    const KernelInfo& ie = infos[le.all_index];
    if (!epilogue_capable(ke, ie, plan.shapes, G.c_tensor, G.M, G.N)) continue;
    bool legal = true;
    for (int p = plan.launches[i + 1].lowered; p < E.lowered && j > i + 1; ++p) {
      const Kernel& kx = t.all[t.live[p]];
      if (kx.write.tensor == ke.write.tensor) legal = false;
      for (auto& rd : ke.reads)
        if (rd.tensor == kx.write.tensor) legal = false;
      for (auto& rd : kx.reads)
        if (rd.tensor == ke.write.tensor) legal = false;
    }
    if (!legal) continue;
    eg::gemm::FusedLaunch probe;
    float* aligned = reinterpret_cast<float*>(uintptr_t(256));
    if (eg::gemm::plan_fused(m->ctx, G.trans_a, G.trans_b, G.M, G.N, G.K, aligned, G.lda, aligned, G.ldb, aligned, G.ldc,
                             nullptr, probe)) {
      eg::clear_error();
      continue;
    }
    if (probe.splits > 1) continue;
    // is the contraction result itself needed by anything but the consumer?
    bool store_c = G.c_tensor == t.output || ts.bucket_offset.count(G.c_tensor) != 0;
    for (size_t p = (size_t)G.lowered + 1; p < t.live.size() && !store_c; ++p) {
      if ((int)p == E.lowered) continue;
      const Kernel& k = t.all[t.live[p]];
      if (k.write.tensor == G.c_tensor && !ts.lowered[p].absorbed) store_c = true;
      for (auto& rd : k.reads)
        if (rd.tensor == G.c_tensor) store_c = true;
    }
    auto pe = std::make_unique<PlanEpilogue>();
    int rc = generate_epilogue(ke, ie, plan.shapes, G.c_tensor, store_c, E.accumulate, pe->spec);
    if (rc) return rc;
    pe->consumer = E;
    pe->store_c = store_c;
    G.kind = StepKind::GemmFused;
    G.epilogue = (int)plan.epilogues.size();
    plan.epilogues.push_back(std::move(pe));
    plan.launches.erase(plan.launches.begin() + j);
    if (plan.n_backward > (int)j) plan.n_backward--;
  }
  return EG_OK;
}

}  // namespace model
}  // namespace eg
