// Contraction + elementwise consumer -> one launch (epilogue.hpp).
#include <algorithm>

#include "model_types.hpp"


namespace eg {
namespace model {

// Predicate tensors.  `h = x * W + b` feeds relu in the forward pass (fused above: relu(h) is written from the
// accumulators) and is read once more, by relu's derived gradient `gIn{it} ++= select(0 <= h{it}, g{it}, 0)`
// (dnn.nim:26-27 through passes.nim:475-483) — which asks ONE yes / no question of every element and never uses the value.
// When every remaining reader of a fused contraction's result is itself a fused consumer that only asks the same question,
// the result is not stored at all: the producer writes the answer, one bit per element, and the readers fetch the bit.
// cfg 5 at 65 536 samples: 134 MB less written by the forward product, 130 MB less read by the activation-gradient
// product (both are bound by exactly that traffic).  Same comparison on the same float32 value, evaluated where the value
// is produced instead of where it is consumed: results are bit-identical (tests/test_gpu_epilogue.py).
// EG_NO_PREDICATE=1 switches it off.
static int predicate_tensors(eg_model* m, TargetState& ts, Plan& plan, const std::vector<KernelInfo>& infos) {
  plan.predicated.clear();
  plan.pred_unzeroed.clear();
  if (m->keep_values) return EG_OK;  // eg_model_keep_values
  {
    const char* e = eg::sw::raw("EG_NO_PREDICATE");
    if (e && e[0] && e[0] != '0') return EG_OK;
    // the batch pipeline (an experiment, off unless EG_PIPELINE=1) addresses half batches of [batch, N] tensors by rows;
    // bit-packed tensors are not addressed that way: the experiment keeps the values
    const char* p = eg::sw::raw("EG_PIPELINE");
    if (p && p[0] && p[0] != '0') return EG_OK;
  }
  Target& t = *ts.target;
  auto fused_launch_of_consumer = [&](int live_pos) -> Launch* {
    for (auto& L : plan.launches)
      if (L.kind == StepKind::GemmFused && plan.epilogues[L.epilogue]->consumer.lowered == live_pos) return &L;
    return nullptr;
  };
  auto regenerate = [&](Launch& G, const PredicateSpec* write_spec) {
    PlanEpilogue& pe = *plan.epilogues[G.epilogue];
    const Lowered& le = ts.lowered[pe.consumer.lowered];
    return generate_epilogue(t.all[le.all_index], infos[le.all_index], plan.shapes, G.c_tensor, pe.store_c, pe.consumer.accumulate,
                             pe.spec, &pe.pred_reads, write_spec);
  };
  for (auto& G : plan.launches) {
    if (G.kind != StepKind::GemmFused) continue;
    PlanEpilogue& pe = *plan.epilogues[G.epilogue];
    const int T = G.c_tensor;
    if (!pe.store_c || T == t.output || ts.bucket_offset.count(T) || m->prog.tensors[T].kind != TK::Result) continue;
    if (G.N % 32 != 0 || G.ldc != G.N) continue;  // rows start on word boundaries
    bool shares = plan.alias.count(T) != 0;
    for (auto& kv : plan.alias) shares = shares || kv.second == T;
    if (shares || (int)pe.spec.operands.size() + 1 > eg::gemm::MAX_EPILOGUE_OPERANDS) continue;
    // every other live kernel that touches T: no writer, and every reader a fused consumer asking the same question
    bool ok = true;
    PredicateSpec spec;
    std::vector<Launch*> readers;
    for (size_t p = 0; p < t.live.size() && ok; ++p) {
      if ((int)p == G.lowered || (int)p == pe.consumer.lowered || ts.lowered[p].absorbed || ts.lowered[p].inlined) continue;
      const Kernel& k = t.all[t.live[p]];
      if (k.write.tensor == T) ok = false;
      bool reads = false;
      for (auto& rd : k.reads) reads = reads || rd.tensor == T;
      if (!reads || !ok) continue;
      Launch* R = fused_launch_of_consumer((int)p);
      PredicateSpec s;
      if (!R || R->M * R->N != G.M * G.N || !only_predicate_uses(k, T, s) || (!readers.empty() && !(s == spec))) {
        ok = false;
        break;
      }
      spec = s;
      readers.push_back(R);
    }
    if (!ok || readers.empty()) continue;
    pe.store_c = false;
    pe.pred_write = true;
    int rc = regenerate(G, &spec);
    if (rc) return rc;
    for (Launch* R : readers) {
      PlanEpilogue& pr = *plan.epilogues[R->epilogue];
      pr.pred_reads[T] = spec;
      PredicateSpec own;  // a reader may itself produce predicate bits (decided earlier in this loop)
      const bool writes_bits = pr.pred_write;
      if (writes_bits) own = plan.predicated.at(R->c_tensor);
      rc = regenerate(*R, writes_bits ? &own : nullptr);
      if (rc) return rc;
    }
    plan.predicated[T] = spec;
    {
      // Bits are OR-ed into zeroed words by ragged tiles and by launches whose operands turn out unaligned; a launch
      // whose tiles are all whole and leave through LDS stores whole words (gemm_f32_mfma.hpp: `packed`), and its
      // tensor can stay out of the zero prefix (4 MB per step at cfg 5).  run.cpp zeroes it by hand if the launch
      // cannot take that path after all.
      eg::gemm::FusedLaunch probe;
      float* aligned = reinterpret_cast<float*>(uintptr_t(256));
      if (eg::gemm::plan_fused(m->ctx, G.trans_a, G.trans_b, G.M, G.N, G.K, aligned, G.lda, aligned, G.ldb, aligned, G.ldc, nullptr,
                               probe) == EG_OK) {
        if (probe.splits <= 1 && eg::gemm::fused_wide_store(probe) && G.N % 32 == 0 && G.ldc == G.N) {
          pe.pred_whole_words = true;
          plan.pred_unzeroed.insert(T);
        }
      } else {
        eg::clear_error();
      }
    }
  }
  return EG_OK;
}

// Contraction + elementwise consumer -> one launch (epilogue.hpp).  Only large outputs: the
// fused kernel is built at run time from the matrix kernel's source (seconds), which pays when
// the saved round trip through HBM is megabytes; small chains are launch bound and handled by
// row fusion / graphs.  EG_EPILOGUE_MIN_ELEMS overrides the threshold, EG_NO_EPILOGUE=1 disables.
int fuse_epilogues(eg_model* m, TargetState& ts, Plan& plan, const std::vector<KernelInfo>& infos) {
  static const bool off = [] {
    const char* e = eg::sw::raw("EG_NO_EPILOGUE");
    return e && e[0] && e[0] != '0';
  }();
  if (off) return EG_OK;
  long min_elems = 1L << 20;
  if (const char* e = eg::sw::raw("EG_EPILOGUE_MIN_ELEMS")) min_elems = atol(e);
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
      if (X.kind == StepKind::RowFused || X.kind == StepKind::SmallFused || X.kind == StepKind::GemmFused || X.kind == StepKind::SampleFused) break;
      const Kernel& kx = t.all[ts.lowered[X.lowered].all_index];
      bool reads_c = false;
      for (auto& rd : kx.reads)
        if (rd.tensor == G.c_tensor) reads_c = true;
      if (reads_c) {
        // (a launch that already carries an inlined consumer of its own stays as it is: the epilogue is
        // generated from ONE kernel, the second one would be lost — plan invariant 1 caught exactly that
        // under EG_NO_ROWFUSE, tests/test_gpu_fuzz.py chain 21)
        found = X.kind == StepKind::GenericA && X.consumer < 0;
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
  return predicate_tensors(m, ts, plan, infos);
}

// Row products.  `dense(…, 512) -> relu -> dense(512, 10)` (examples: the classifier head of cfg 5): the second layer's
// contraction `z[y, x] ++= a[y, it] * W2[it, x]` + bias (dnn.nim:19-24) has 10 columns — its whole cost is reading a, the
// 134 MB the first layer's fused epilogue has just written.  When the first layer runs on whole 256 x 256 tiles, the
// rows of a pass through LDS on their way out (wide-store pass, gemm_f32_mfma.hpp), and the second layer's product is
// taken there: 16 x 16 x 4 MFMAs over the staged rows against the tile's 256 rows of W2, each N-tile adding its partial
// product to z with one float atomic per element.  z is zeroed with the other accumulated results; with at most two
// N-tiles an element is 0 + p + q in either order — bit-identical from run to run (more N-tiles: not folded).
// The narrow contraction disappears from the launch list (36 us at cfg 5).  EG_NO_ROW_PRODUCT=1 switches it off.
int fold_row_products(eg_model* m, TargetState& ts, Plan& plan) {
  plan.zero_extra.clear();
  for (const char* name : {"EG_NO_ROW_PRODUCT", "EG_PIPELINE"}) {  // read per plan: a test compares folded and unfolded plans
    const char* e = eg::sw::raw(name);
    if (e && e[0] && e[0] != '0') return EG_OK;
  }
  const Target& t = *ts.target;
  auto shares_storage = [&](int x) {
    bool s = plan.alias.count(x) != 0;
    for (auto& kv : plan.alias) s = s || kv.second == x;
    return s;
  };
  for (size_t gi = 0; gi + 1 < plan.launches.size(); ++gi) {
    Launch& G = plan.launches[gi];
    if (G.kind != StepKind::GemmFused || G.accumulate) continue;
    PlanEpilogue& pe = *plan.epilogues[G.epilogue];
    if (pe.row_product || pe.consumer.accumulate) continue;
    if (G.M % 256 != 0 || G.N % 256 != 0 || G.N > 512 || G.ldc != G.N) continue;  // whole tiles, at most two N-tiles
    if ((int)pe.spec.operands.size() + 3 > eg::gemm::MAX_EPILOGUE_OPERANDS) continue;
    eg::gemm::FusedLaunch probe;
    float* aligned = reinterpret_cast<float*>(uintptr_t(256));
    if (eg::gemm::plan_fused(m->ctx, G.trans_a, G.trans_b, G.M, G.N, G.K, aligned, G.lda, aligned, G.ldb, aligned, G.ldc, nullptr,
                             probe)) {
      eg::clear_error();
      continue;
    }
    eg::gemm::fused_withdraw_narrow(probe);   // (a row product rides on the matrix tile whatever K is: run.cpp withdraws it too)
    if (probe.bm != 256 || probe.bn != 256 || probe.splits > 1 || !eg::gemm::fused_wide_store(probe)) continue;
    const int R = t.all[ts.lowered[pe.consumer.lowered].all_index].write.tensor;  // the rows the product is taken of
    // the narrow contraction: the next launches up to it may not touch what moves
    for (size_t pj = gi + 1; pj < plan.launches.size() && pj <= gi + 3; ++pj) {
      const Launch& P = plan.launches[pj];
      if (P.kind != StepKind::Gemm && P.kind != StepKind::GenericA && P.kind != StepKind::GenericB) break;
      if (P.consumer >= 0 || P.ones_tensor) break;
      const bool candidate = P.kind == StepKind::Gemm && !P.trans_a && !P.trans_b && P.a_tensor == R && P.lda == G.N &&
                             P.M == G.M && P.K == G.N && P.N >= 1 && P.N <= 16 && !P.accumulate && P.ldc == P.N;
      if (!candidate) {
        const Kernel& kx = t.all[ts.lowered[P.lowered].all_index];
        if (kx.write.tensor == R) break;
        continue;
      }
      const int Z = P.c_tensor;
      // (the target's output lives in the arena like any other result: zeroed with it)
      if (ts.bucket_offset.count(Z) || m->prog.tensors[Z].kind != TK::Result || shares_storage(Z) || plan.predicated.count(Z))
        break;
      if ((int)gi < plan.n_backward && (int)pj >= plan.n_backward) break;
      bool legal = true;
      for (size_t x = gi + 1; x < pj && legal; ++x) {
        const Kernel& kx = t.all[ts.lowered[plan.launches[x].lowered].all_index];
        if (kx.write.tensor == Z || kx.write.tensor == P.b_tensor || (P.bias_tensor && kx.write.tensor == P.bias_tensor)) legal = false;
        for (auto& rd : kx.reads)
          if (rd.tensor == Z) legal = false;
      }
      if (!legal) break;
      pe.row_product = true;
      pe.product = P;
      pe.plain_struct_code = pe.spec.struct_code;
      const int iw = (int)pe.spec.operands.size();
      pe.spec.operands.push_back(P.b_tensor);
      pe.spec.operands.push_back(Z);
      if (P.bias_tensor) pe.spec.operands.push_back(P.bias_tensor);
      char line[256];
      snprintf(line, sizeof(line), "  static constexpr int RD_N = %ld, RD_W = %d, RD_OUT = %d, RD_BIAS = %d, RD_LDW = %ld, RD_LDO = %ld;\n",
               P.N, iw, iw + 1, P.bias_tensor ? iw + 2 : -1, P.ldb, P.ldc);
      const size_t at = pe.spec.struct_code.find(kNoRowProduct);
      EG_REQUIRE(at != std::string::npos, EG_ERR_RUNTIME, "generated epilogue without the row-product line");
      pe.spec.struct_code.replace(at, strlen(kNoRowProduct), line);
      plan.zero_extra.insert(Z);
      plan.launches.erase(plan.launches.begin() + (long)pj);
      if (plan.n_backward > (int)pj) plan.n_backward--;
      break;
    }
  }
  return EG_OK;
}

// dense = `out[y,x] ++= in[y,it] * W[it,x]` + `out[y,x] ++= b[x]` (dnn.nim:19-24); derive turns the two into
// `gW[it,x] ++= in[y,it] * g[y,x]` and `gb[x] ++= g[y,x]` (passes.nim:519-549): two reductions over the same
// batch that both stream g.  With A = [in | 1] they are ONE contraction whose last row is gb — the
// kernel supplies the ones (GemmArgs::ones_row), and because gW and gb are neighbours in the gradient
// bucket the contraction's M + 1 rows of output land exactly in [gW; gb].  The separate column sum
// (a second pass over g: 134 MB at cfg 5) disappears.
int fold_bias_gradients(eg_model* m, TargetState& ts, Plan& plan, const std::vector<KernelInfo>& infos) {
  {  // read per plan (not once per process): a test switches it to compare folded and unfolded plans
    const char* e = eg::sw::raw("EG_NO_ONES_ROW");
    if (e && e[0] && e[0] != '0') return EG_OK;
  }
  const Target& t = *ts.target;
  for (size_t gi = 0; gi < plan.launches.size(); ++gi) {
    Launch& G = plan.launches[gi];
    if (G.kind != StepKind::Gemm || G.accumulate || !G.trans_a || G.trans_b || G.bias_tensor || G.ldc != G.N) continue;
    auto gw = ts.bucket_offset.find(G.c_tensor);
    if (gw == ts.bucket_offset.end()) continue;
    // candidate: a column sum of the same B operand whose destination follows gW in the bucket
    for (size_t ci = 0; ci < plan.launches.size(); ++ci) {
      if (ci == gi) continue;
      Launch& C = plan.launches[ci];
      if (C.kind != StepKind::GenericA && C.kind != StepKind::GenericB) continue;
      if (C.accumulate || C.consumer >= 0) continue;
      const Kernel& k = t.all[ts.lowered[C.lowered].all_index];
      const KernelInfo& info = infos[ts.lowered[C.lowered].all_index];
      // gb[x] ++= g[y,x]: no arithmetic, one read indexed (y, x), written at (x)
      if (!k.instrs.empty() || k.reads.size() != 1 || k.result != k.reads[0].reg || !k.index_instrs.empty() || !k.setup.empty()) continue;
      if (k.loops.size() != 2 || k.write.raw || k.write.dims.size() != 1 || k.reads[0].raw || k.reads[0].dims.size() != 2) continue;
      const int x = k.write.dims[0].only_register();
      const int ry = k.reads[0].dims[0].only_register(), rx = k.reads[0].dims[1].only_register();
      if (!x || rx != x || !ry || ry == x) continue;
      if (k.reads[0].tensor != G.b_tensor) continue;
      bool bounds_ok = info.ok;
      for (size_t l = 0; l < k.loops.size() && bounds_ok; ++l) {
        const long want = k.loops[l].reg == x ? G.N : G.K;
        bounds_ok = !k.loops[l].has_bounds && info.bounds[l].first == 0 && info.bounds[l].second == want;
      }
      if (!bounds_ok) continue;
      auto gb = ts.bucket_offset.find(k.write.tensor);
      if (gb == ts.bucket_offset.end() || gb->second != gw->second + G.M * G.N) continue;
      auto sh = plan.shapes.find(k.write.tensor);
      if (sh == plan.shapes.end() || prod(sh->second) != G.N) continue;
      // both on one side of the backward | update boundary, and nobody between them touches what moves
      const size_t lo = std::min(gi, ci), hi = std::max(gi, ci);
      if ((int)lo < plan.n_backward && (int)hi >= plan.n_backward) continue;
      bool legal = true;
      for (size_t j = lo + 1; j < hi && legal; ++j) {
        const Launch& X = plan.launches[j];
        if (X.kind == StepKind::RowFused || X.kind == StepKind::SmallFused || X.kind == StepKind::GemmFused || X.kind == StepKind::SampleFused) {
          legal = false;  // (their tensor sets are not worth analysing here: adjacent launches are the case that matters)
          break;
        }
        const Kernel& kx = t.all[ts.lowered[X.lowered].all_index];
        if (kx.write.tensor == k.write.tensor || kx.write.tensor == G.b_tensor || kx.write.tensor == G.c_tensor) legal = false;
        for (auto& rd : kx.reads)
          if (rd.tensor == k.write.tensor || rd.tensor == G.c_tensor) legal = false;
      }
      if (!legal) continue;
      G.ones_tensor = k.write.tensor;
      G.ones_lowered = C.lowered;
      plan.launches.erase(plan.launches.begin() + (long)ci);
      if (plan.n_backward > (int)ci) plan.n_backward--;
      if (ci < gi) --gi;
      break;
    }
  }
  (void)m;
  return EG_OK;
}

}  // namespace model
}  // namespace eg
