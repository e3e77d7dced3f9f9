// Execution of a plan: one launch, a range of launches (with the side lane), HIP-graph capture and replay.
#include <algorithm>

#include <cstring>

#include "model_types.hpp"
#include <random>

namespace eg {
namespace model {

// Device-resident generator state of the model's TensorRandom tensors: {seed, fills drawn so far}.
int ensure_rng(eg_model* m, uint64_t seed, bool reseed) {
  EG_HIP_CHECK(hipSetDevice(m->ctx->device));
  const bool fresh = m->rng_state == nullptr;
  if (fresh) EG_HIP_CHECK(hipMalloc((void**)&m->rng_state, 2 * sizeof(uint64_t)));
  if (fresh || reseed) {
    const uint64_t init[2] = {seed, 0};
    EG_HIP_CHECK(hipStreamSynchronize(m->ctx->stream));
    EG_HIP_CHECK(hipMemcpy(m->rng_state, init, sizeof(init), hipMemcpyHostToDevice));
  }
  return EG_OK;
}

int run_launch(eg_model* m, TargetState& ts, Plan& plan, Launch& L) {
  eg_ctx* ctx = m->ctx;
  switch (L.kind) {
    case StepKind::Seed:
      if (m->f64) return eg_fill_f64(ctx, L.count, (double)m->grad_scale, reinterpret_cast<double*>(tensor_ptr(m, ts, plan, L.c_tensor)));
      return eg_fill_f32(ctx, L.count, m->grad_scale, tensor_ptr(m, ts, plan, L.c_tensor));
    case StepKind::Gemm:
      if (m->f64) {
        auto dp = [&](int t) { return reinterpret_cast<double*>(tensor_ptr(m, ts, plan, t)); };
        return eg_dgemm(ctx, L.trans_a, L.trans_b, L.M, L.N, L.K, dp(L.a_tensor), L.lda, dp(L.b_tensor), L.ldb, dp(L.c_tensor), L.ldc,
                        L.accumulate, L.bias_tensor ? dp(L.bias_tensor) : nullptr);
      }
      if (L.ones_tensor) {
        // weight gradient + bias gradient in one contraction; when the operands of this run do not
        // qualify (alignment of a caller-owned input), the two reductions run separately
        int rc = eg::gemm::sgemm_ones_row(ctx, L.trans_a, L.trans_b, L.M, L.N, L.K, tensor_ptr(m, ts, plan, L.a_tensor), L.lda,
                                          tensor_ptr(m, ts, plan, L.b_tensor), L.ldb, tensor_ptr(m, ts, plan, L.c_tensor), L.ldc,
                                          L.accumulate);
        if (rc != EG_ERR_UNSUPPORTED) return rc;
        eg::clear_error();
        rc = eg_colsum(ctx, L.K, L.N, tensor_ptr(m, ts, plan, L.b_tensor), tensor_ptr(m, ts, plan, L.ones_tensor), 0);
        if (rc) return rc;
      }
      return eg_sgemm(ctx, L.trans_a, L.trans_b, L.M, L.N, L.K, tensor_ptr(m, ts, plan, L.a_tensor), L.lda,
                      tensor_ptr(m, ts, plan, L.b_tensor), L.ldb, tensor_ptr(m, ts, plan, L.c_tensor), L.ldc,
                      L.accumulate, L.bias_tensor ? tensor_ptr(m, ts, plan, L.bias_tensor) : nullptr);
    case StepKind::GemmFused: {
      PlanEpilogue& pe = *plan.epilogues[L.epilogue];
      const float* bias = L.bias_tensor ? tensor_ptr(m, ts, plan, L.bias_tensor) : nullptr;
      eg::gemm::FusedLaunch f;
      int rc = eg::gemm::plan_fused(ctx, L.trans_a, L.trans_b, L.M, L.N, L.K, tensor_ptr(m, ts, plan, L.a_tensor), L.lda,
                                    tensor_ptr(m, ts, plan, L.b_tensor), L.ldb, tensor_ptr(m, ts, plan, L.c_tensor), L.ldc,
                                    bias, f);
      if (rc) return rc;
      // The tile model may ask for k-slices at run time where it did not when the plan was made (it depends on
      // ctx->compute_units, which the data-parallel tail range lowers around its launches).  The unfused route below is
      // only valid when C holds plain values and the consumer reads plain values: a predicate tensor has M*N/32 words of
      // storage (eg_sgemm would overrun it), predicate operands hold bits, and a row product's destination would stay
      // zero.  Those launches run fused on one slice — the args of plan_fused always describe the whole K.
      if (pe.row_product) eg::gemm::fused_withdraw_narrow(f);   // (a row product rides on the 256 x 256 matrix tile)
      const bool must_fuse = pe.pred_write || !pe.pred_reads.empty() || pe.row_product || !pe.store_c;
      if (f.splits > 1 && must_fuse) f.splits = 1;
      if (f.splits > 1) {
        rc = eg_sgemm(ctx, L.trans_a, L.trans_b, L.M, L.N, L.K, tensor_ptr(m, ts, plan, L.a_tensor), L.lda,
                      tensor_ptr(m, ts, plan, L.b_tensor), L.ldb, tensor_ptr(m, ts, plan, L.c_tensor), L.ldc, 0, bias);
        if (rc) return rc;
        return run_launch(m, ts, plan, pe.consumer);
      }
      void* operands[eg::gemm::MAX_EPILOGUE_OPERANDS] = {};
      for (size_t o = 0; o < pe.spec.operands.size(); ++o) operands[o] = tensor_ptr(m, ts, plan, pe.spec.operands[o]);
      eg::gemm::set_epilogue_operands(f, operands, (int)pe.spec.operands.size(), m->grad_scale, m->epoch);
      // a row product rides on whole 256 x 256 tiles that leave through LDS; anything else (cannot happen for the shapes
      // the plan was made for, unless an operand turns out unaligned) runs the plain functor and the product by itself
      if (pe.pred_write && plan.pred_unzeroed.count(L.c_tensor) && !eg::gemm::fused_wide_store(f)) {
        // planned to store whole words of predicate bits, but this launch will OR single bits in: zero the words first
        rc = eg::zero_ranges(ctx, {{tensor_ptr(m, ts, plan, L.c_tensor), storage_floats(plan, L.c_tensor)}});
        if (rc) return rc;
      }
      const bool with_product = pe.row_product && f.bm == 256 && f.bn == 256 && eg::gemm::fused_wide_store(f);
      const std::string variant = eg::gemm::fused_variant(f);
      eg_kernel*& handle = (pe.row_product && !with_product) ? pe.built_plain[variant] : pe.built[variant];
      const std::string& struct_code = (pe.row_product && !with_product) ? pe.plain_struct_code : pe.spec.struct_code;
      if (!handle) {
        const std::string name = "eg_gemm_epi" + std::to_string(m->kernel_serial++);
        const std::string src = eg::gemm::fused_source(f, struct_code, pe.spec.struct_name, name);
        if (const char* dump = eg::sw::raw("EG_DUMP_FUSED")) {  // debugging aid: the generated translation unit
          if (FILE* fp = fopen((std::string(dump) + "/" + name + "_" + variant + ".hip").c_str(), "w")) {
            fputs(src.c_str(), fp);
            fclose(fp);
          }
        }
        rc = eg_kernel_compile(ctx, name.c_str(), src.c_str(), &handle);
        if (rc) {
          std::string msg = eg_last_error();
          set_error("%s\n--- generated epilogue (%s) ---\n%s", msg.c_str(), variant.c_str(), struct_code.c_str());
          handle = nullptr;
          return rc;
        }
        m->kernels.push_back(handle);
      }
      static const bool gemm_trace = eg::sw::raw("EG_GEMM_TRACE") != nullptr;
      long long* trace = nullptr;
      if (gemm_trace && !f.narrow) {
        trace = eg::gemm::trace_begin(ctx, f.grid, (unsigned)f.nt / 64);
        eg::gemm::fused_set_trace(f, trace);
      }
      void* args[] = {f.args};
      rc = f.narrow ? eg::kernel_launch_raw(handle, f.narrow_grid, 1, 1, 256, args)
                    : eg::kernel_launch_raw(handle, f.grid, 1, 1, (unsigned)f.nt, args);
      if (trace) {
        eg::gemm::trace_end(ctx, trace, f.grid, (unsigned)f.nt / 64, "fused contraction");
        eg::gemm::fused_set_trace(f, nullptr);
      }
      if (rc || !pe.row_product || with_product) return rc;
      return run_launch(m, ts, plan, pe.product);
    }
    case StepKind::ConvGradFilter:
      return eg_conv2_nhwc_grad_filter(ctx, L.cN, L.cH, L.cW, L.cC, L.cF, L.cFH, L.cFW, tensor_ptr(m, ts, plan, L.a_tensor),
                                       tensor_ptr(m, ts, plan, L.b_tensor), tensor_ptr(m, ts, plan, L.c_tensor),
                                       L.accumulate);
    case StepKind::ConvGradImage:
      return eg_conv2_nhwc_grad_image(ctx, L.cN, L.cH, L.cW, L.cC, L.cF, L.cFH, L.cFW, tensor_ptr(m, ts, plan, L.a_tensor),
                                      tensor_ptr(m, ts, plan, L.b_tensor), tensor_ptr(m, ts, plan, L.c_tensor),
                                      L.accumulate);
    case StepKind::Conv:
      return eg_conv2_nhwc(ctx, L.cN, L.cH, L.cW, L.cC, L.cF, L.cFH, L.cFW, tensor_ptr(m, ts, plan, L.a_tensor),
                           tensor_ptr(m, ts, plan, L.b_tensor), tensor_ptr(m, ts, plan, L.c_tensor), L.accumulate);
    case StepKind::SmallFused: {
      PlanSmallGroup& sg = *plan.small_groups[L.row_group];
      std::vector<float*> ptrs;
      for (int tid : sg.g.ptr_args) ptrs.push_back(tensor_ptr(m, ts, plan, tid));
      std::vector<void*> args;
      for (auto& p : ptrs) args.push_back(&p);
      float GS = m->grad_scale;
      long EP = m->epoch;
      args.push_back(&GS);
      args.push_back(&EP);
      const float* slab = nullptr;
      long FOLD = 0;
      if (!sg.g.fold_offset.empty()) {  // the group can add up a sample group's slab rows itself (fuse_slab_fold)
        slab = plan.sample_group ? plan.sample_group->slab : nullptr;
        // only when the sample kernel ran in THIS range: behind a backward | exchange | update split the bucket holds the
        // all-reduced totals and the slab only this rank's rows
        FOLD = (L.fold_of >= plan.active_begin && L.fold_of < plan.active_end && slab && slab_fold_active(plan, plan.launches[L.fold_of])) ? 1 : 0;
        args.push_back(&slab);
        args.push_back(&FOLD);
      }
      return eg::kernel_launch_raw(sg.handle, (unsigned)sg.g.blocks, 1, 1, 256, args.data());
    }
    case StepKind::SampleFused: {
      PlanSampleGroup& sg = *plan.sample_group;
      std::vector<float*> ptrs;
      ptrs.push_back(sg.slab);
      for (int tid : sg.g.ptr_args) {
        float* p = tensor_ptr(m, ts, plan, tid);
        if (!p && prod(plan.shapes.at(tid)) > 0) {
          set_error("tensor %d has no device storage", tid);
          return EG_ERR_INVALID;
        }
        ptrs.push_back(p);
      }
      std::vector<void*> args;
      for (auto& p : ptrs) args.push_back(&p);
      float GS = m->grad_scale;
      long EP = m->epoch;
      args.push_back(&GS);
      args.push_back(&EP);
      int rc = eg::kernel_launch_raw(sg.handle, (unsigned)sg.g.B, 1, 1, (unsigned)sg.g.threads, args.data());
      if (rc || sg.g.slab_floats <= 0) return rc;
      if (slab_fold_active(plan, L)) return EG_OK;  // the optimizer's map group behind this launch adds the rows up itself
      // every sample's contribution to the parameter gradients -> the gradient bucket, in a fixed order
      float* dst = ts.bucket + sg.bucket_base;
      if (eg::slab_sum_supported(sg.g.slab_floats, sg.slab, dst)) return eg::slab_sum(ctx, sg.g.B, sg.g.slab_floats, sg.slab, dst, 0);
      // a bucket bound to caller memory that is not 16-byte aligned (a sliced view), a row that is no multiple of four
      // floats, or EG_NO_SLAB_SUM: the scalar two-pass column sum, as kernels/conv2_band.cpp falls back
      rc = eg::ensure_workspace(ctx, (size_t)eg::colsum_scratch_floats(ctx, sg.g.B, sg.g.slab_floats) * sizeof(float));
      if (rc) return rc;
      return eg::colsum_with_scratch(ctx, sg.g.B, sg.g.slab_floats, sg.slab, dst, 0, static_cast<float*>(ctx->workspace));
    }
    case StepKind::RowFused: {
      PlanRowGroup& pg = *plan.row_groups[L.row_group];
      std::vector<float*> ptrs;
      ptrs.push_back(pg.partial);
      for (int tid : pg.g.ptr_args) ptrs.push_back(tensor_ptr(m, ts, plan, tid));
      std::vector<void*> args;
      for (auto& p : ptrs) args.push_back(&p);
      long B = pg.g.B, EP = m->epoch;
      float GS = m->grad_scale;
      args.push_back(&B);
      args.push_back(&GS);
      args.push_back(&EP);
      std::vector<float*> dsts;
      if (pg.g.single_block || pg.g.in_kernel_finalize) {  // totals go to their destinations from inside the kernel (rowfuse.hpp)
        dsts.reserve(pg.red_tensors.size() + pg.g.tail_ptr_args.size());
        for (int tid : pg.red_tensors) dsts.push_back(tensor_ptr(m, ts, plan, tid));
        for (auto& p : dsts) args.push_back(&p);
      }
      // several blocks: the last one to arrive folds the partial rows (MODE 1) and, when the range being issued
      // holds the small group behind this launch as well, runs it (MODE 2; run_range_eager skips that launch)
      long MODE = 0;
      unsigned* counter = pg.counter;
      const bool deferred = plan.defer_finalize_of == (int)(&L - plan.launches.data());   // the next fork folds the rows
      if (pg.g.in_kernel_finalize) {
        MODE = deferred ? 0 : row_tail_active(plan, L) ? 2 : 1;
        args.push_back(&counter);
        args.push_back(&MODE);
        const size_t first_tail = dsts.size();
        for (int tid : pg.g.tail_ptr_args) dsts.push_back(tensor_ptr(m, ts, plan, tid));
        for (size_t a = first_tail; a < dsts.size(); ++a) args.push_back(&dsts[a]);
      }
      int rc = eg::kernel_launch_raw(pg.handle, (unsigned)pg.nblocks, 1, 1, 256, args.data());
      if (rc) return rc;
      if (deferred) return EG_OK;
      if (pg.g.red_total > 0 && !pg.g.single_block && MODE == 0) {
        eg::RowFinalizeArgs fa = {};
        fa.nseg = (int)pg.red_tensors.size();
        for (int s = 0; s < fa.nseg; ++s) {
          const RowGroupTensor& gt = pg.g.tensors.at(pg.red_tensors[s]);
          fa.dst[s] = tensor_ptr(m, ts, plan, pg.red_tensors[s]);
          fa.offset[s] = (int)gt.red_offset;
          fa.accumulate[s] = gt.accumulate ? 1 : 0;
        }
        fa.offset[fa.nseg] = (int)pg.g.red_total;
        return eg::row_finalize(ctx, pg.partial, pg.nblocks, (int)pg.g.red_total, (int)pg.g.red_stride(), fa);
      }
      return EG_OK;
    }
    case StepKind::GenericA:
    case StepKind::GenericB: {
      if (L.blocks_x <= 0 || L.blocks_y <= 0) return EG_OK;
      if (L.conv_direct64) {
        bool launched = false;
        auto dp = [&](int t) { return reinterpret_cast<double*>(tensor_ptr(m, ts, plan, t)); };
        int rc = EG_OK;
        if (L.conv_direct64 == 1) {
          rc = eg::conv2_band_forward_try(ctx, true, L.cN, L.cH, L.cW, L.cC, L.cF, L.cFH, L.cFW, dp(L.a_tensor), dp(L.b_tensor), dp(L.c_tensor), L.accumulate, &launched);
          if (!rc && !launched)   // (images too wide for a band in LDS: column strips)
            rc = eg::conv2_direct_f64_try(ctx, L.cN, L.cH, L.cW, L.cC, L.cF, L.cFH, L.cFW, dp(L.a_tensor), dp(L.b_tensor), dp(L.c_tensor), L.accumulate, &launched);
        } else if (L.conv_direct64 == 2) {
          rc = eg::conv2_band_grad_image_try(ctx, true, L.cN, L.cH, L.cW, L.cC, L.cF, L.cFH, L.cFW, dp(L.a_tensor), dp(L.b_tensor), dp(L.c_tensor), L.accumulate, &launched);
        } else {
          rc = eg::conv2_band_grad_filter_try(ctx, true, L.cN, L.cH, L.cW, L.cC, L.cF, L.cFH, L.cFW, dp(L.a_tensor), dp(L.b_tensor), dp(L.c_tensor), L.accumulate, &launched);
        }
        if (rc || launched) return rc;
      }
      const GenericSource& src = L.generic->src;
      std::vector<void*> args;
      std::vector<float*> ptrs;
      ptrs.reserve(src.tensor_args.size() + 1);
      float* partial = nullptr;
      float* scratch = nullptr;
      if (L.kind == StepKind::GenericB) {
        const long pfloats = ((L.partial_rows * L.partial_cols + 3) & ~3L) * m->esz;
        const long sfloats = eg::colsum_scratch_floats(ctx, L.partial_rows, L.partial_cols) * m->esz;
        int rc = eg::ensure_workspace(ctx, (size_t)(pfloats + sfloats) * sizeof(float));
        if (rc) return rc;
        partial = static_cast<float*>(ctx->workspace);
        scratch = partial + pfloats;
        ptrs.push_back(partial);
      }
      for (int tid : src.tensor_args) {
        float* p = tensor_ptr(m, ts, plan, tid);
        if (!p && prod(plan.shapes.at(tid)) > 0) {
          set_error("tensor %d has no device storage", tid);
          return EG_ERR_INVALID;
        }
        ptrs.push_back(p);
      }
      for (auto& p : ptrs) args.push_back(&p);
      for (auto& v : L.params) args.push_back(&v);
      for (int slot : L.epoch_slots) L.params[slot] = m->epoch;
      long blocks_x = L.blocks_x;
      if (L.kind == StepKind::GenericA && L.vec_slot >= 0) {
        // four elements per thread need 16-byte aligned operands (arena tensors are; caller-owned ones may not be)
        bool aligned = L.vec_ok;
        for (float* p : ptrs) aligned = aligned && (reinterpret_cast<uintptr_t>(p) & 15) == 0;
        L.params[L.vec_slot] = aligned ? 1 : 0;
        blocks_x = ((aligned ? L.total_items / 4 : L.total_items) + 255) / 256;
      }
      int rc = eg::kernel_launch_raw(L.generic->handle, (unsigned)blocks_x, (unsigned)L.blocks_y, 1, 256, args.data());
      if (rc) return rc;
      if (L.kind == StepKind::GenericB && m->f64)
        return eg::colsum_f64_with_scratch(ctx, L.partial_rows, L.partial_cols, reinterpret_cast<const double*>(partial),
                                           reinterpret_cast<double*>(tensor_ptr(m, ts, plan, L.c_tensor)), L.accumulate,
                                           reinterpret_cast<double*>(scratch));
      if (L.kind == StepKind::GenericB)
        return eg::colsum_with_scratch(ctx, L.partial_rows, L.partial_cols, partial,
                                       tensor_ptr(m, ts, plan, L.c_tensor), L.accumulate, scratch);
      return EG_OK;
    }
  }
  return EG_OK;
}

// the fold of a row group's partial rows as a launch of its own (what run_launch does behind the row kernel when the kernel
// does not fold them itself)
static int run_row_fold(eg_model* m, TargetState& ts, Plan& plan, const Launch& L) {
  PlanRowGroup& pg = *plan.row_groups[L.row_group];
  eg::RowFinalizeArgs fa = {};
  fa.nseg = (int)pg.red_tensors.size();
  for (int s = 0; s < fa.nseg; ++s) {
    const RowGroupTensor& gt = pg.g.tensors.at(pg.red_tensors[s]);
    fa.dst[s] = tensor_ptr(m, ts, plan, pg.red_tensors[s]);
    fa.offset[s] = (int)gt.red_offset;
    fa.accumulate[s] = gt.accumulate ? 1 : 0;
  }
  fa.offset[fa.nseg] = (int)pg.g.red_total;
  return eg::row_finalize(m->ctx, pg.partial, pg.nblocks, (int)pg.g.red_total, (int)pg.g.red_stride(), fa);
}

int run_range_eager(eg_model* m, TargetState& ts, Plan& plan, int begin, int end, bool zero, const SideHook* hook) {
  plan.defer_finalize_of = -1;
  if (zero) {
    // allocShapes / zeroResultTensor (model.nim:318, 383): results start from zero on every call
    {
      std::vector<std::pair<float*, long>> ranges;
      if (plan.zero_floats > 0) ranges.emplace_back(plan.arena, plan.zero_floats);
      for (int tid : plan.bucket_zero) {
        float* p = ts.bucket + ts.bucket_offset[tid];
        const long n = prod(plan.shapes.at(tid)) * m->esz;
        // neighbours in the bucket (a layer's weights and bias) are one range
        if (!ranges.empty() && ranges.back().first + ranges.back().second == p)
          ranges.back().second += n;
        else
          ranges.emplace_back(p, n);
      }
      int rc = eg::zero_ranges(m->ctx, ranges);
      if (rc) return rc;
    }
    if (eg::poison_enabled()) {
      // EG_POISON: whatever is not zeroed must be overwritten completely by its first writer
      if (plan.arena_floats > plan.zero_floats)
        EG_HIP_CHECK(hipMemsetAsync(plan.arena + plan.zero_floats, 0xFF,
                                    (size_t)(plan.arena_floats - plan.zero_floats) * sizeof(float), m->ctx->stream));
      std::set<int> zeroed(plan.bucket_zero.begin(), plan.bucket_zero.end());
      for (auto& b : ts.bucket_offset)
        if (!zeroed.count(b.first) && plan.shapes.count(b.first))
          EG_HIP_CHECK(hipMemsetAsync(ts.bucket + b.second, 0xFF, (size_t)prod(plan.shapes.at(b.first)) * m->esz * sizeof(float),
                                      m->ctx->stream));
      for (auto& rg : plan.row_groups)
        if (rg->partial)
          EG_HIP_CHECK(hipMemsetAsync(rg->partial, 0xFF, (size_t)rg->nblocks * rg->g.red_stride() * sizeof(float), m->ctx->stream));
    }
    // fresh random tensors for this call (model.nim:310-314 does it on the host); the draw counter
    // is bumped on the device so a captured sequence advances on every replay
    for (size_t r = 0; r < plan.random_tensors.size(); ++r) {
      const int tid = plan.random_tensors[r];
      const TensorDef& d = m->prog.tensors[tid];
      int rc = m->f64 ? eg_fill_uniform_f64(m->ctx, prod(plan.shapes.at(tid)), d.lo, d.hi, m->rng_state, (uint64_t)tid,
                                            reinterpret_cast<double*>(plan.arena + plan.arena_offset[tid]))
                      : eg_fill_uniform(m->ctx, prod(plan.shapes.at(tid)), (float)d.lo, (float)d.hi, m->rng_state, (uint64_t)tid,
                               plan.arena + plan.arena_offset[tid]);
      if (rc) return rc;
    }
    if (!plan.random_tensors.empty()) {
      int rc = eg_rng_advance(m->ctx, m->rng_state);
      if (rc) return rc;
    }
  }
  if (plan.pipe.active && begin == 0 && end >= plan.n_backward) {
    int rc = run_pipelined(m, ts, plan, hook);
    if (rc) return rc;
    begin = plan.n_backward;
  }
  plan.active_begin = begin;
  plan.active_end = end;
  size_t next_overlap = 0;
  plan.defer_finalize_of = -1;
  for (int i = begin; i < end; ++i) {
    // a row group whose fold the fork of an overlap group inside this range runs on the side lane (plan_overlap)
    for (const Plan::Overlap& ov : plan.overlaps)
      if (ov.deferred_row == i && ov.first >= begin && ov.big < end) plan.defer_finalize_of = i;
    // a small group that the last block of the row group in front of it has run already (fuse_row_tails)
    if (plan.launches[i].tail_of >= begin && row_tail_active(plan, plan.launches[plan.launches[i].tail_of])) continue;
    while (next_overlap < plan.overlaps.size() && plan.overlaps[next_overlap].first < i) ++next_overlap;
    if (next_overlap < plan.overlaps.size() && plan.overlaps[next_overlap].first == i &&
        plan.overlaps[next_overlap].big < end) {
      // fork: the side lane takes launches [i, big) after everything issued so far, the main stream
      // goes on with the contraction; join before whatever follows
      const int big = plan.overlaps[next_overlap].big;
      eg_ctx* ctx = m->ctx;
      EG_HIP_CHECK(hipEventRecord(ctx->ev_fork, ctx->stream));
      EG_HIP_CHECK(hipStreamWaitEvent(ctx->side_stream, ctx->ev_fork, 0));
      // The contraction is issued FIRST (round 6): in a captured graph the branch whose node is created first stays on the
      // queue of what precedes it and the other branch moves to a second queue, 10 - 12 us behind (r06_step_timeline.txt: the
      // long weight gradient started 12 us after the kernel that makes its operand, and the update 10 us after the
      // gradient's last kernel).  Measured on one box, six alternating runs: 0.9708 against 0.9737 ms per step — the gaps
      // shrink by 11 us, the contraction shares its CUs from the start and takes 14 us longer.  A data-parallel step keeps
      // the side lane first: its early gradient exchange rides that lane and must not queue behind 256 resident blocks.
      // EG_OVERLAP_SIDE_FIRST=1: the order of rounds 2 - 5 everywhere.
      const bool side_first = eg::sw::raw("EG_OVERLAP_SIDE_FIRST") != nullptr || (hook && hook->big == big);
      if (!side_first) {
        int rc = run_launch(m, ts, plan, plan.launches[big]);
        if (rc) return rc;
      }
      {
        LaneSwap lane(ctx);
        if (plan.overlaps[next_overlap].deferred_row >= 0 && plan.defer_finalize_of == plan.overlaps[next_overlap].deferred_row) {
          int rc = run_row_fold(m, ts, plan, plan.launches[plan.defer_finalize_of]);
          plan.defer_finalize_of = -1;
          if (rc) return rc;
        }
        for (int s2 = i; s2 < big; ++s2) {
          int rc = run_launch(m, ts, plan, plan.launches[s2]);
          if (rc) return rc;
        }
        if (hook && hook->big == big) {  // data-parallel step: the early gradients' exchange rides the side lane
          int rc = hook->fn(hook->user);
          if (rc) return rc;
        }
        EG_HIP_CHECK(hipEventRecord(ctx->ev_join, ctx->stream));  // (the side stream, while swapped)
      }
      if (side_first) {
        int rc = run_launch(m, ts, plan, plan.launches[big]);
        if (rc) return rc;
      }
      EG_HIP_CHECK(hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
      i = big;
      continue;
    }
    // Two independent tiny contractions next to each other (a dense layer's two gradients at a small batch): one launch
    // (EG_NO_SMALL_PAIR=1: two).  Independence is checked on the storage: neither writes what the other touches.
    const bool pair_off = eg::sw::raw("EG_NO_SMALL_PAIR") != nullptr;   // (read per launch sequence: a test builds one model each way)
    if (!pair_off && !m->f64 && i + 1 < end && i + 1 != plan.n_backward &&
        !(next_overlap < plan.overlaps.size() && plan.overlaps[next_overlap].first == i + 1)) {
      const Launch &A = plan.launches[i], &B = plan.launches[i + 1];
      if (A.kind == StepKind::Gemm && B.kind == StepKind::Gemm && !A.ones_tensor && !B.ones_tensor &&
          eg::gemm_small_suits(A.M, A.N, A.K) && eg::gemm_small_suits(B.M, B.N, B.K)) {
        auto ptr = [&](int t) -> float* { return t ? tensor_ptr(m, ts, plan, t) : nullptr; };
        float *ca = ptr(A.c_tensor), *cb = ptr(B.c_tensor);
        const float* touched_by_b[] = {ptr(B.a_tensor), ptr(B.b_tensor), ptr(B.bias_tensor), cb};
        const float* read_by_a[] = {ptr(A.a_tensor), ptr(A.b_tensor), ptr(A.bias_tensor)};
        bool independent = ca && cb;
        for (const float* p : touched_by_b) independent = independent && p != ca;
        for (const float* p : read_by_a) independent = independent && p != cb;
        if (independent) {
          int rc = eg::gemm_small_pair(
              m->ctx,
              eg::small_gemm(A.trans_a, A.trans_b, A.M, A.N, A.K, ptr(A.a_tensor), A.lda, ptr(A.b_tensor), A.ldb, ca, A.ldc, A.accumulate,
                             ptr(A.bias_tensor)),
              eg::small_gemm(B.trans_a, B.trans_b, B.M, B.N, B.K, ptr(B.a_tensor), B.lda, ptr(B.b_tensor), B.ldb, cb, B.ldc, B.accumulate,
                             ptr(B.bias_tensor)));
          if (rc) return rc;
          ++i;
          continue;
        }
      }
    }
    int rc = run_launch(m, ts, plan, plan.launches[i]);
    if (rc) return rc;
  }
  return EG_OK;
}

bool graphs_enabled() {
  static const bool on = [] {
    const char* e = eg::sw::raw("EG_NO_GRAPH");
    return !(e && e[0] && e[0] != '0');
  }();
  return on;
}

std::string capture_key(eg_model* m, TargetState& ts) {
  std::ostringstream key;
  for (auto& in : m->inputs)
    if (in.second.bound) key << in.first << "=" << (const void*)in.second.device << ";";
  key << "w" << m->ctx->workspace << "x" << m->ctx->aux << "s" << m->ctx->side_workspace << "y" << m->ctx->side_aux << "b"
      << (void*)ts.bucket << "g" << m->grad_scale << "e" << m->epoch;
  return key.str();
}

// slot: 0 = whole call, 1 = backward part, 2 = update part
int run_range(eg_model* m, TargetState& ts, Plan& plan, int begin, int end, bool zero, int slot) {
  int rc = eg::set_device(m->ctx);
  if (rc) return rc;
  Plan::Captured& cap = plan.graphs[slot];
  // (a range that comes down to ONE kernel — a row group whose last block runs the small group behind it — gains nothing
  //  from a graph: replaying a one-node graph measured 19 us per XOR step against 13.7 us for the plain launch)
  int launches = 0;
  for (int i = begin; i < end; ++i) {
    const int of = plan.launches[i].tail_of;
    if (!(of >= begin && plan.launches[of].tail_launch == i)) ++launches;
  }
  if (!graphs_enabled() || launches < 2) return run_range_eager(m, ts, plan, begin, end, zero);
  // every pointer / scalar that ends up in a kernel argument: as plain words first (a replay formats no string)
  const void* now[5] = {m->ctx->workspace, m->ctx->aux, m->ctx->side_workspace, m->ctx->side_aux, ts.bucket};
  if (cap.exec && cap.stamp == m->inputs_gen && cap.grad_scale == m->grad_scale && cap.epoch == m->epoch &&
      memcmp(cap.ptrs, now, sizeof(now)) == 0) {
    EG_HIP_CHECK(hipGraphLaunch(cap.exec, m->ctx->stream));
    return EG_OK;
  }
  const std::string k = capture_key(m, ts);
  if (cap.exec && cap.key == k) {  // (the bindings were renewed with the same values)
    cap.stamp = m->inputs_gen;
    memcpy(cap.ptrs, now, sizeof(now));
    cap.grad_scale = m->grad_scale;
    cap.epoch = m->epoch;
    EG_HIP_CHECK(hipGraphLaunch(cap.exec, m->ctx->stream));
    return EG_OK;
  }
  if (cap.exec) {
    // the old sequence may still be running (launches are asynchronous): let it finish before its
    // executable graph goes away
    EG_HIP_CHECK(hipStreamSynchronize(m->ctx->stream));
    hipGraphExecDestroy(cap.exec);
    cap.exec = nullptr;
  }
  // first execution with these arguments runs eagerly (lazy kernel builds, workspace growth);
  // the second one is captured
  if (cap.key != k || cap.runs < 1) {
    if (cap.key != k) cap.runs = 0;
    cap.key = k;
    cap.runs++;
    return run_range_eager(m, ts, plan, begin, end, zero);
  }
  hipGraph_t graph = nullptr;
  hipError_t e = hipStreamBeginCapture(m->ctx->stream, hipStreamCaptureModeThreadLocal);
  static const bool debug = eg::sw::raw("EG_DEBUG_GRAPH") != nullptr;
  if (e != hipSuccess) {  // capture unavailable on this stream: stay eager
    if (debug) fprintf(stderr, "[eg] begin capture failed: %s\n", hipGetErrorString(e));
    (void)hipGetLastError();
    return run_range_eager(m, ts, plan, begin, end, zero);
  }
  rc = run_range_eager(m, ts, plan, begin, end, zero);
  e = hipStreamEndCapture(m->ctx->stream, &graph);
  if (rc) {
    if (graph) hipGraphDestroy(graph);
    return rc;
  }
  if (e != hipSuccess || !graph) {
    if (debug) fprintf(stderr, "[eg] end capture failed: %s\n", hipGetErrorString(e));
    (void)hipGetLastError();
    return run_range_eager(m, ts, plan, begin, end, zero);
  }
  if (debug) {
    size_t n = 0;
    hipGraphGetNodes(graph, nullptr, &n);
    fprintf(stderr, "[eg] captured %zu nodes for slot %d of target %s\n", n, slot, ts.target->name.c_str());
  }
  e = hipGraphInstantiate(&cap.exec, graph, nullptr, nullptr, 0);
  hipGraphDestroy(graph);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    cap.exec = nullptr;
    return run_range_eager(m, ts, plan, begin, end, zero);
  }
  cap.stamp = m->inputs_gen;
  memcpy(cap.ptrs, now, sizeof(now));
  cap.grad_scale = m->grad_scale;
  cap.epoch = m->epoch;
  EG_HIP_CHECK(hipGraphLaunch(cap.exec, m->ctx->stream));
  return EG_OK;
}

// ---- batch pipeline (plan_pipeline.cpp) ------------------------------------------------------------
int run_launch_sliced(eg_model* m, TargetState& ts, Plan& plan, Launch& L, const Slice& sl) {
  eg_ctx* ctx = m->ctx;
  switch (L.kind) {
    case StepKind::Gemm: {
      const float* A = tensor_ptr(m, ts, plan, L.a_tensor);
      const float* B = tensor_ptr(m, ts, plan, L.b_tensor);
      float* C = tensor_ptr(m, ts, plan, L.c_tensor);
      const float* bias = L.bias_tensor ? tensor_ptr(m, ts, plan, L.bias_tensor) : nullptr;
      if (L.slice_mode == 1)  // rows of the batch: A(m, k) = A[m * lda + k]
        return eg_sgemm(ctx, L.trans_a, L.trans_b, sl.rows, L.N, L.K, A + sl.row0 * L.lda, L.lda, B, L.ldb, C + sl.row0 * L.ldc,
                        L.ldc, L.accumulate, bias);
      // the batch is K: A(k, m) = A[k * lda + m], B(k, n) = B[k * ldb + n]; the second half adds to the first
      const int accumulate = L.accumulate || sl.second;
      A += sl.row0 * L.lda;
      B += sl.row0 * L.ldb;
      if (L.ones_tensor) {
        int rc = eg::gemm::sgemm_ones_row(ctx, L.trans_a, L.trans_b, L.M, L.N, sl.rows, A, L.lda, B, L.ldb, C, L.ldc, accumulate);
        if (rc != EG_ERR_UNSUPPORTED) return rc;
        eg::clear_error();
        rc = eg_colsum(ctx, sl.rows, L.N, B, tensor_ptr(m, ts, plan, L.ones_tensor), sl.second ? 1 : 0);
        if (rc) return rc;
      }
      return eg_sgemm(ctx, L.trans_a, L.trans_b, L.M, L.N, sl.rows, A, L.lda, B, L.ldb, C, L.ldc, accumulate, bias);
    }
    case StepKind::GemmFused: {
      PlanEpilogue& pe = *plan.epilogues[L.epilogue];
      const float* bias = L.bias_tensor ? tensor_ptr(m, ts, plan, L.bias_tensor) : nullptr;
      const float* A = tensor_ptr(m, ts, plan, L.a_tensor) + sl.row0 * L.lda;
      float* C = tensor_ptr(m, ts, plan, L.c_tensor) + sl.row0 * L.ldc;
      eg::gemm::FusedLaunch f;
      int rc = eg::gemm::plan_fused(ctx, L.trans_a, L.trans_b, sl.rows, L.N, L.K, A, L.lda, tensor_ptr(m, ts, plan, L.b_tensor),
                                    L.ldb, C, L.ldc, bias, f);
      if (rc) return rc;
      eg::gemm::fused_withdraw_narrow(f);   // (the batch pipeline, an experiment, keeps the matrix tile)
      if (f.splits > 1) {
        set_error("batch pipeline: a half of a fused contraction needs split-K");
        return EG_ERR_RUNTIME;
      }
      const std::string variant = eg::gemm::fused_variant(f);
      eg_kernel*& handle = pe.built[variant];
      if (!handle) {
        const std::string name = "eg_gemm_epi" + std::to_string(m->kernel_serial++);
        const std::string src = eg::gemm::fused_source(f, pe.spec.struct_code, pe.spec.struct_name, name);
        rc = eg_kernel_compile(ctx, name.c_str(), src.c_str(), &handle);
        if (rc) {
          handle = nullptr;
          return rc;
        }
        m->kernels.push_back(handle);
      }
      void* operands[eg::gemm::MAX_EPILOGUE_OPERANDS] = {};
      for (size_t o = 0; o < pe.spec.operands.size(); ++o)
        operands[o] = tensor_ptr(m, ts, plan, pe.spec.operands[o]) + sl.row0 * L.N;  // [batch, N] tensors
      eg::gemm::set_epilogue_operands(f, operands, (int)pe.spec.operands.size(), m->grad_scale, m->epoch);
      void* args[] = {f.args};
      return eg::kernel_launch_raw(handle, f.grid, 1, 1, (unsigned)f.nt, args);
    }
    case StepKind::RowFused: {
      PlanRowGroup& pg = *plan.row_groups[L.row_group];
      std::vector<float*> ptrs;
      ptrs.push_back(pg.partial);
      for (int tid : pg.g.ptr_args) {
        float* p = tensor_ptr(m, ts, plan, tid);
        const RowGroupTensor& gt = pg.g.tensors.at(tid);
        if (gt.role == RowGroupTensor::RowLocal || gt.role == RowGroupTensor::RowExternal) p += sl.row0 * gt.inner;
        ptrs.push_back(p);
      }
      std::vector<void*> args;
      for (auto& p : ptrs) args.push_back(&p);
      long B = sl.rows, EP = m->epoch;
      float GS = m->grad_scale;
      args.push_back(&B);
      args.push_back(&GS);
      args.push_back(&EP);
      const int nblocks = (int)((sl.rows + 255) / 256);
      if (pg.g.single_block) {
        set_error("batch pipeline: a single-block row group cannot be cut");
        return EG_ERR_RUNTIME;
      }
      // (a half batch is folded by row_finalize, which can add the second half to the first: MODE 0)
      std::vector<float*> extra;
      long MODE = 0;
      unsigned* counter = pg.counter;
      if (pg.g.in_kernel_finalize) {
        extra.reserve(pg.red_tensors.size() + pg.g.tail_ptr_args.size());
        for (int tid : pg.red_tensors) extra.push_back(tensor_ptr(m, ts, plan, tid));
        for (int tid : pg.g.tail_ptr_args) extra.push_back(tensor_ptr(m, ts, plan, tid));
        for (size_t a = 0; a < pg.red_tensors.size(); ++a) args.push_back(&extra[a]);
        args.push_back(&counter);
        args.push_back(&MODE);
        for (size_t a = pg.red_tensors.size(); a < extra.size(); ++a) args.push_back(&extra[a]);
      }
      int rc = eg::kernel_launch_raw(pg.handle, (unsigned)nblocks, 1, 1, 256, args.data());
      if (rc) return rc;
      if (pg.g.red_total > 0) {
        eg::RowFinalizeArgs fa = {};
        fa.nseg = (int)pg.red_tensors.size();
        for (int s = 0; s < fa.nseg; ++s) {
          const RowGroupTensor& gt = pg.g.tensors.at(pg.red_tensors[s]);
          fa.dst[s] = tensor_ptr(m, ts, plan, pg.red_tensors[s]);
          fa.offset[s] = (int)gt.red_offset;
          fa.accumulate[s] = (gt.accumulate || sl.second) ? 1 : 0;
        }
        fa.offset[fa.nseg] = (int)pg.g.red_total;
        return eg::row_finalize(ctx, pg.partial, nblocks, (int)pg.g.red_total, (int)pg.g.red_stride(), fa);
      }
      return EG_OK;
    }
    default:
      set_error("batch pipeline: launch kind %d cannot be cut", (int)L.kind);
      return EG_ERR_RUNTIME;
  }
}

// Stage j = the j-th long contraction of the range and the streaming launches that follow it (stage
// -1: what precedes the first one).  Main lane: contraction j of half A, of half B, contraction j + 1
// of half A, ...  Side lane: the streaming launches of stage j, half A then half B.  Events carry the
// per-half dependencies between the lanes; both lanes are in order, so nothing else needs saying.
int run_pipelined(eg_model* m, TargetState& ts, Plan& plan, const SideHook* hook) {
  eg_ctx* ctx = m->ctx;
  int rc = ensure_side_lane(ctx);
  if (rc) return rc;
  const int nb = plan.n_backward;
  std::vector<int> heavy;  // positions of the long contractions
  for (int i = 0; i < nb; ++i)
    if (plan.launches[i].heavy) heavy.push_back(i);
  const int stages = (int)heavy.size();
  const size_t need = (size_t)(stages + 1) * 4;
  while (ctx->pipe_events.size() < need) {
    hipEvent_t e;
    EG_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    ctx->pipe_events.push_back(e);
  }
  auto ev_main = [&](int stage, int h) { return ctx->pipe_events[(size_t)((stage + 1) * 4 + h)]; };      // contraction (stage, h) done
  auto ev_side = [&](int stage, int h) { return ctx->pipe_events[(size_t)((stage + 1) * 4 + 2 + h)]; };  // streaming part (stage, h) done
  const Slice halves[2] = {{0, plan.pipe.half, false}, {plan.pipe.half, plan.pipe.batch - plan.pipe.half, true}};
  // everything queued so far on the main lane (zero fills, random tensors) precedes the side lane's work
  EG_HIP_CHECK(hipEventRecord(ctx->ev_fork, ctx->stream));
  EG_HIP_CHECK(hipStreamWaitEvent(ctx->side_stream, ctx->ev_fork, 0));
  for (int stage = -1; stage < stages; ++stage) {
    const int first_light = stage < 0 ? 0 : heavy[(size_t)stage] + 1;
    const int last_light = stage + 1 < stages ? heavy[(size_t)stage + 1] : nb;  // exclusive
    if (stage >= 0) {
      for (int h = 0; h < 2; ++h) {
        if (stage > 0 || heavy[0] > 0)  // the streaming launches before it, same half
          EG_HIP_CHECK(hipStreamWaitEvent(ctx->stream, ev_side(stage - 1, h), 0));
        rc = run_launch_sliced(m, ts, plan, plan.launches[heavy[(size_t)stage]], halves[h]);
        if (rc) return rc;
        EG_HIP_CHECK(hipEventRecord(ev_main(stage, h), ctx->stream));
      }
    }
    const bool any_light = first_light < last_light;
    {
      LaneSwap lane(ctx);  // ctx->stream is the side lane's from here
      for (int h = 0; h < 2; ++h) {
        if (stage >= 0) EG_HIP_CHECK(hipStreamWaitEvent(ctx->stream, ev_main(stage, h), 0));
        for (int i = first_light; i < last_light; ++i) {
          rc = run_launch_sliced(m, ts, plan, plan.launches[i], halves[h]);
          if (rc) return rc;
        }
        if (stage == stages - 1 && h == 1 && hook && hook->big == -2) {  // data-parallel step: the early gradients go out here
          rc = hook->fn(hook->user);
          if (rc) return rc;
        }
        EG_HIP_CHECK(hipEventRecord(ev_side(stage, h), ctx->stream));
      }
    }
    (void)any_light;
  }
  // join: the main lane continues after the side lane's last piece
  EG_HIP_CHECK(hipStreamWaitEvent(ctx->stream, ev_side(stages - 1, 0), 0));
  EG_HIP_CHECK(hipStreamWaitEvent(ctx->stream, ev_side(stages - 1, 1), 0));
  return EG_OK;
}

// ---- data-parallel step with the exchange overlapped (dp_rccl.cpp) -------------------
// The gradient bucket is exchanged in two parts when the backward range ends in an overlap group whose
// long contraction produces the LAST gradient (dense nets: the first layer's weight gradient, by far
// the largest): every other gradient is complete before that contraction finishes, so its all-reduce
// is issued on the side lane and runs under the contraction; only the contraction's own gradient is
// reduced after it.  Both parts are contiguous runs of the bucket (at most a few segments).
int plan_exchange(eg_model* m, TargetState& ts, Plan& plan, ExchangePlan& ex) {
  (void)m;
  ex = ExchangePlan();
  // last launch of the backward range that writes each bucket tensor
  std::map<int, int> last_writer;
  auto note = [&](int tensor, int launch) {
    if (ts.bucket_offset.count(tensor)) last_writer[tensor] = launch;
  };
  const Target& t = *ts.target;
  for (int i = 0; i < plan.n_backward; ++i) {
    const Launch& L = plan.launches[i];
    switch (L.kind) {
      case StepKind::RowFused:
        for (int tid : plan.row_groups[L.row_group]->red_tensors) note(tid, i);
        for (auto& kv : plan.row_groups[L.row_group]->g.tensors)
          if (kv.second.role == RowGroupTensor::RowLocal && kv.second.store) note(kv.first, i);
        break;
      case StepKind::SmallFused:
        for (int ki : plan.small_groups[L.row_group]->g.kernel_index) note(t.all[ki].write.tensor, i);
        break;
      case StepKind::SampleFused:
        for (int ki : plan.sample_group->g.kernel_index) note(t.all[ki].write.tensor, i);
        break;
      case StepKind::GemmFused:
        note(L.c_tensor, i);
        note(t.all[ts.lowered[plan.epilogues[L.epilogue]->consumer.lowered].all_index].write.tensor, i);
        break;
      default:
        note(L.c_tensor, i);
        note(L.ones_tensor, i);
    }
  }
  if (plan.pipe.active) {
    // batch pipeline: gradients written by the streaming launches (side lane) are complete when the side
    // lane ends, before the last contraction of the main lane: they go out there (hook->big == -2)
    bool late_any = false, early_any = false;
    std::vector<std::pair<long, int>> order;
    for (auto& b : ts.bucket_offset) order.push_back({b.second, b.first});
    std::sort(order.begin(), order.end());
    for (size_t i = 0; i < order.size(); ++i) {
      const int tid = order[i].second;
      const long begin = order[i].first, end = i + 1 < order.size() ? order[i + 1].first : ts.bucket_floats;
      auto lw = last_writer.find(tid);
      const bool late = lw != last_writer.end() && plan.launches[(size_t)lw->second].heavy;
      (late ? late_any : early_any) = true;
      std::vector<std::pair<long, long>>& segs = late ? ex.late : ex.early;
      if (!segs.empty() && segs.back().first + segs.back().second == begin) segs.back().second += end - begin;
      else segs.push_back({begin, end - begin});
    }
    ex.big = -2;
    if (!late_any || !early_any) {
      ex.big = -1;
      ex.late.clear();
      ex.early.clear();
      if (ts.bucket_floats > 0) ex.late.push_back({0, ts.bucket_floats});
    }
    return EG_OK;
  }
  int big = -1;
  for (auto& ov : plan.overlaps)
    if (ov.big < plan.n_backward) big = ov.big;
  if (big >= 0) {
    bool late_any = false, early_any = false;
    for (auto& b : ts.bucket_offset) {
      auto lw = last_writer.find(b.first);
      const bool late = lw != last_writer.end() && lw->second >= big;
      late_any = late_any || late;
      early_any = early_any || !late;
    }
    if (!late_any || !early_any) big = -1;  // nothing to split: one all-reduce of the whole bucket
  }
  // bucket tensors in offset order -> maximal contiguous runs of one class
  std::vector<std::pair<long, int>> order;  // (offset, tensor)
  for (auto& b : ts.bucket_offset) order.push_back({b.second, b.first});
  std::sort(order.begin(), order.end());
  for (size_t i = 0; i < order.size(); ++i) {
    const int tid = order[i].second;
    const long begin = order[i].first;
    const long end = i + 1 < order.size() ? order[i + 1].first : ts.bucket_floats;  // padding travels with its tensor
    auto lw = last_writer.find(tid);
    const bool late = big >= 0 && lw != last_writer.end() && lw->second >= big;
    std::vector<std::pair<long, long>>& segs = late ? ex.late : ex.early;
    if (!segs.empty() && segs.back().first + segs.back().second == begin) segs.back().second += end - begin;
    else segs.push_back({begin, end - begin});
  }
  ex.big = big;
  if (big < 0) {  // everything in one piece, after the backward range
    ex.late.clear();
    ex.early.clear();
    if (ts.bucket_floats > 0) ex.late.push_back({0, ts.bucket_floats});
  }
  return EG_OK;
}

}  // namespace model
}  // namespace eg
