// Group 3 of the C ABI: compile a kernel-description program for the GPU and run its targets.
//
// What this replaces in the reference (for a CompileGpu target):
//   newModel             model.nim:215-251   passes + parameter allocation + kernel build
//   allocShapes (gpu)    model.nim:302-318   result tensors (re)allocated and zero-filled per call
//   flushStateTensors    model.nim:326-345   parameters live on the device (and, unlike the
//                                            reference, device-side updates are what readers see)
//   writeInput/readOutput model.nim:357-376
//   call / apply         model.nim:392-411   infer shapes, run the target's kernel list in order
//   runGpuKernel & co.   model.nim:148-172   one launch per lowered kernel
//
// Every live kernel of a target is matched against the hand-written library
// (contraction -> eg_sgemm with the following bias kernel folded into its epilogue,
// convolution -> eg_conv2_nhwc, gradLoss seed -> eg_fill_f32); everything else gets generated
// HIP source (codegen.hpp) built once with hiprtc.  A plan (shapes, launch arguments, result
// arena) is cached per input-shape signature — the reference re-solves shapes on every call
// (passes.nim:1386-1436).
#include <random>
#include "model_types.hpp"
#include "dp_schedule.hpp"

using namespace eg::kd;
using namespace eg::model;
using eg::set_error;


namespace eg {

eg_ctx* model_context(eg_model* model) { return model ? model->ctx : nullptr; }

int model_backward_with_exchange(eg_model* m, const char* target, const GradExchange& gx, int* pieces) try {
  EG_REQUIRE(m && target && gx.allreduce, EG_ERR_INVALID, "model_backward_with_exchange: NULL argument");
  TargetState* ts;
  Plan* plan;
  int rc = get_plan(m, target, &ts, &plan);
  if (rc) return rc;
  ExchangePlan ex;
  rc = plan_exchange(m, *ts, *plan, ex);
  if (rc) return rc;
  int issued = 0;
  auto reduce = [&](const std::vector<std::pair<long, long>>& segs) {
    for (auto& seg : segs) {
      if (seg.second <= 0) continue;
      int r = gx.allreduce(gx.user, ts->bucket + seg.first, seg.second);
      if (r) return r;
      ++issued;
    }
    return (int)EG_OK;
  };
  // WHICH all-reduce calls this step issues is the (group, target)'s exchange schedule, not this plan's own cut
  // (host/dp_schedule.hpp): negotiated at step counts every rank reaches together, served by whatever plan a rank
  // currently holds — overlapped when the plan proposes the agreed cut, behind its whole backward range otherwise.
  eg::dp::Proposal mine;
  mine.bucket_floats = ts->bucket_floats;
  mine.split = ex.big >= 0 && !plan->pipe.active;
  mine.early = ex.early;
  mine.late = ex.late;
  static const long reagree_every = [] {
    const char* e = eg::sw::raw("EG_DP_REAGREE_STEPS");
    return e ? atol(e) : 256L;
  }();
  // keyed by what every rank derives identically from the communicator, not by this rank's group address (a one-rank
  // caller without an identity — gx.group 0 — has no peer to disagree with: its address serves)
  eg::dp::Schedule& sched = ts->dp_schedules.of(gx.group ? gx.group : (uint64_t)reinterpret_cast<uintptr_t>(gx.user));
  eg::dp::How how = eg::dp::How::Whole;
  std::string why;
  rc = eg::dp::step_decision(sched, mine, gx.split, gx.agree, gx.user, reagree_every, &how, &why);
  if (rc == -1) {
    set_error("%s", why.c_str());
    return EG_ERR_INVALID;
  }
  if (rc) return rc;
  const bool split = how == eg::dp::How::Overlapped;
  if (how == eg::dp::How::Sequential) {
    // the agreed cut, without the overlap this plan cannot give: same calls, same order as on the other ranks
    rc = run_range(m, *ts, *plan, 0, plan->n_backward, true, 1);
    if (rc) return rc;
    rc = reduce(sched.early);
    if (!rc) rc = reduce(sched.late);
    if (pieces) *pieces = issued;
    return rc;
  }
  if (!split) {
    // nothing to overlap with: the captured backward range, then one all-reduce of the whole bucket
    rc = run_range(m, *ts, *plan, 0, plan->n_backward, true, 1);
    if (rc) return rc;
    std::vector<std::pair<long, long>> whole;
    if (ts->bucket_floats > 0) whole.push_back({0, ts->bucket_floats});
    rc = reduce(whole);
  } else {
    // Three captured ranges around the two collectives (a collective is not captured: RCCL decides that):
    //   main lane:  [0, first)  ......................  [big, n_backward)  ..  all-reduce(late)
    //   side lane:              [first, big)  all-reduce(early)  ......  join ^
    // [first, big) are the bandwidth-bound launches of the overlap group (the small weight gradient, column sums):
    // they and the early exchange run under the long contraction `big`, which leaves gx.reserve_cus compute units
    // free so that the collective's kernel has somewhere to run.
    int first = -1;
    for (auto& ov : plan->overlaps)
      if (ov.big == ex.big) first = ov.first;
    EG_REQUIRE(first >= 0 && first <= ex.big, EG_ERR_RUNTIME, "data-parallel step: the overlap group of launch %d is gone", ex.big);
    eg_ctx* ctx = m->ctx;
    rc = eg::set_device(ctx);
    if (rc) return rc;
    rc = run_range(m, *ts, *plan, 0, first, true, 3);
    if (rc) return rc;
    EG_HIP_CHECK(hipEventRecord(ctx->ev_fork, ctx->stream));
    EG_HIP_CHECK(hipStreamWaitEvent(ctx->side_stream, ctx->ev_fork, 0));
    int side_rc = EG_OK;
    {
      LaneSwap lane(ctx);
      side_rc = run_range(m, *ts, *plan, first, ex.big, false, 4);
      if (!side_rc) side_rc = reduce(ex.early);
      // (recorded whatever happened: the main lane below waits for it, and a failed step must not leave the side
      // stream unjoined)
      hipEventRecord(ctx->ev_join, ctx->stream);
    }
    const int cus = ctx->compute_units;
    if (gx.reserve_cus > 0 && gx.reserve_cus < cus) ctx->compute_units = cus - gx.reserve_cus;
    rc = side_rc ? side_rc : run_range(m, *ts, *plan, ex.big, plan->n_backward, false, 5);
    ctx->compute_units = cus;
    EG_HIP_CHECK(hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
    if (rc) return rc;
    rc = reduce(ex.late);
  }
  if (pieces) *pieces = issued;
  return rc;
}
EG_CATCH_ALL

}  // namespace eg

extern "C" {

int eg_model_compile(eg_ctx* ctx, const char* program_text, eg_model** out) try {
  EG_REQUIRE(ctx && program_text && out, EG_ERR_INVALID, "eg_model_compile: NULL argument");
  std::unique_ptr<eg_model> m(new eg_model());
  m->ctx = ctx;
  m->source_text = program_text;
  int rc = parse(program_text, m->prog);
  if (rc) return rc;
  rc = compile_program(m->prog);
  if (rc) return rc;
  m->f64 = m->prog.f64;
  m->esz = m->f64 ? 2 : 1;
  rc = eg::set_device(ctx);
  if (rc) return rc;
  // parameters: uniform in initRange (model.nim:241-247); deterministic here, tests overwrite them
  std::mt19937 rng(10);
  for (size_t tid = 1; tid < m->prog.tensors.size(); ++tid) {
    const TensorDef& d = m->prog.tensors[tid];
    // caches (model.nim:248-249: zero tensors) live next to the parameters: same lifetime, same access
    if (d.kind != TK::Param && d.kind != TK::Cache) continue;
    DevTensor dt;
    dt.shape = d.shape;
    dt.count = prod(d.shape);
    if (dt.count > 0 && m->f64) {  // newRandTensor[float64] (model.nim:241-247)
      EG_HIP_CHECK(hipMalloc((void**)&dt.ptr, (size_t)dt.count * sizeof(double)));
      std::vector<double> host(dt.count);
      std::uniform_real_distribution<double> dist(d.lo, d.hi);
      for (auto& v : host) v = d.kind == TK::Cache ? 0.0 : (d.hi > d.lo ? dist(rng) : d.lo);
      EG_HIP_CHECK(hipMemcpy(dt.ptr, host.data(), host.size() * sizeof(double), hipMemcpyHostToDevice));
    } else if (dt.count > 0) {
      EG_HIP_CHECK(hipMalloc((void**)&dt.ptr, (size_t)dt.count * sizeof(float)));
      std::vector<float> host(dt.count);
      std::uniform_real_distribution<float> dist((float)d.lo, (float)d.hi);
      for (auto& v : host) v = d.kind == TK::Cache ? 0.0f : (d.hi > d.lo ? dist(rng) : (float)d.lo);
      EG_HIP_CHECK(hipMemcpy(dt.ptr, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    m->params[(int)tid] = dt;
  }
  for (auto& t : m->prog.targets) {
    TargetState& ts = m->targets[t.name];
    ts.target = &t;
    // gradient bucket: GenGradient destinations of parameters, in kernel-list order
    long off = 0;
    for (auto& k : t.source)
      if (k.gen == Gen::Gradient && m->prog.tensors[k.gen_tensor].kind == TK::Param &&
          !ts.bucket_offset.count(k.gen_dest)) {
        ts.grad_tensors.push_back(k.gen_dest);
        ts.bucket_offset[k.gen_dest] = off;
        off += align4(prod(m->prog.tensors[k.gen_tensor].shape) * m->esz);
      }
    ts.bucket_floats = off;
    if (off > 0) {
      EG_HIP_CHECK(hipMalloc((void**)&ts.bucket, (size_t)off * sizeof(float)));
      EG_HIP_CHECK(hipMemsetAsync(ts.bucket, 0, (size_t)off * sizeof(float), ctx->stream));  // (ordered against the launches that follow)
      ts.bucket_owned = true;
    }
    rc = lower_target(m.get(), ts);
    if (rc) return rc;
  }
  rc = build_pending(m.get());
  if (rc) return rc;
  describe(m.get());
  *out = m.release();
  return EG_OK;
}
EG_CATCH_ALL

int eg_model_free(eg_model* m) try {
  if (!m) return EG_OK;
  hipSetDevice(m->ctx->device);
  hipStreamSynchronize(m->ctx->stream);
  if (m->fit_graph.exec) hipGraphExecDestroy(m->fit_graph.exec);
  m->fit_graph.exec = nullptr;
  if (m->fit_graph.exec2) hipGraphExecDestroy(m->fit_graph.exec2);
  m->fit_graph.exec2 = nullptr;
  for (auto& ev : m->fit_graph.done)
    if (ev) hipEventDestroy(ev);
  if (m->fit_graph.graph) hipGraphDestroy(m->fit_graph.graph);
  m->fit_graph.graph = nullptr;
  for (auto& kv : m->targets) {
    for (auto& p : kv.second.plans) release_plan(*p.second);
    if (kv.second.bucket_owned && kv.second.bucket) hipFree(kv.second.bucket);
  }
  for (auto& p : m->params)
    if (p.second.ptr) hipFree(p.second.ptr);
  for (auto& in : m->inputs)
    if (in.second.owned) hipFree(in.second.owned);
  if (m->rng_state) hipFree(m->rng_state);
  for (float* p : m->fit_data)
    if (p) hipFree(p);
  if (m->copy_event) hipEventDestroy(m->copy_event);
  if (m->main_event) hipEventDestroy(m->main_event);
  if (m->copy_stream) hipStreamDestroy(m->copy_stream);
  for (eg_kernel* k : m->kernels) eg_kernel_free(k);
  delete m;
  return EG_OK;
}
EG_CATCH_ALL

const char* eg_model_plan_text(eg_model* m) { return m ? m->plan_text.c_str() : ""; }

const char* eg_model_launch_text(eg_model* m, const char* target) {
  if (!m || !target) return "";
  auto it = m->targets.find(target);
  if (it == m->targets.end() || !it->second.last) return "";
  TargetState& ts = it->second;
  Plan& plan = *ts.last;
  std::ostringstream os;
  if (plan.pipe.active)
    os << "-- batch pipeline: launches up to the update run as two halves of " << plan.pipe.half
       << " rows; long contractions on the main lane, the rest on the side lane under them --\n";
  for (size_t i = 0; i < plan.launches.size(); ++i) {
    const Launch& L = plan.launches[i];
    if ((int)i == plan.n_backward) os << "-- update --\n";
    os << "[" << i << "] ";
    switch (L.kind) {
      case StepKind::Seed: os << "seed-fill t" << L.c_tensor; break;
      case StepKind::Gemm:
      case StepKind::GemmFused:
        os << (L.kind == StepKind::GemmFused ? "gemm+epilogue " : "gemm ") << (L.trans_a ? "T" : "N") << (L.trans_b ? "T" : "N")
           << " " << L.M << "x" << L.N << "x" << L.K << " -> t" << L.c_tensor << (L.bias_tensor ? " +bias" : "")
           << (L.accumulate ? " accumulate" : "");
        if (L.ones_tensor) os << " +ones-row -> t" << L.ones_tensor << " (bias gradient, kernel " << L.ones_lowered << ")";
        if (L.kind == StepKind::GemmFused) {
          const PlanEpilogue& pe = *plan.epilogues[L.epilogue];
          os << " | consumer kernel " << pe.consumer.lowered << " operands";
          for (int t : pe.spec.operands) os << " t" << t << (pe.pred_reads.count(t) || (pe.pred_write && t == L.c_tensor) ? "(bits)" : "");
          if (pe.pred_write) os << " | t" << L.c_tensor << " stored as predicate bits";
          if (pe.row_product)
            os << " | row product NN " << pe.product.M << "x" << pe.product.N << "x" << pe.product.K << " -> t" << pe.product.c_tensor
               << (pe.product.bias_tensor ? " +bias" : "") << " (kernel " << pe.product.lowered << ")";
        }
        break;
      case StepKind::Conv: os << "conv2 -> t" << L.c_tensor << (L.accumulate ? " accumulate" : ""); break;
      case StepKind::ConvGradImage: os << "conv2-grad-image -> t" << L.c_tensor << (L.accumulate ? " accumulate" : ""); break;
      case StepKind::ConvGradFilter: os << "conv2-grad-filter -> t" << L.c_tensor << (L.accumulate ? " accumulate" : ""); break;
      case StepKind::GenericA:
        os << "generated(map) kernel " << L.lowered << " -> t" << L.c_tensor;
        if (L.conv_direct64) os << " | " << (L.conv_direct64 == 1 ? "convolution" : L.conv_direct64 == 2 ? "convolution's image gradient" : "convolution's filter gradient")
                                 << ": the float64 matrix-core kernel (eg_conv_band_f64_* / eg_conv_mfma64_*) when the shape suits it";
        if (L.consumer >= 0) os << " (with its consumer, kernel " << L.consumer << ")";
        break;
      case StepKind::GenericB: os << "generated(split-reduce) kernel " << L.lowered << " -> t" << L.c_tensor; break;
      case StepKind::RowFused: {
        const PlanRowGroup& pg = *plan.row_groups[L.row_group];
        os << "row-fused " << pg.g.kernel_index.size() << " kernels";
        if (pg.g.in_kernel_finalize) os << " | partial rows folded by the last block to arrive";
        if (L.tail_launch >= 0) os << " | which goes on with launch " << L.tail_launch << " when a range holds both";
        break;
      }
      case StepKind::SampleFused: {
        const PlanSampleGroup& sg = *plan.sample_group;
        os << "sample-fused " << sg.g.kernel_index.size() << " kernels, one block per sample (" << sg.g.B << " blocks)";
        if (sg.g.slab_floats > 0) os << " | " << sg.sum_tensors.size() << " batch sums " << (L.fold_launch >= 0 ? "folded by launch " + std::to_string(L.fold_launch) + " (one slab pass when a range ends between them)" : std::string("folded by one slab pass"));
        break;
      }
      case StepKind::SmallFused: {
        const PlanSmallGroup& sg = *plan.small_groups[L.row_group];
        os << (sg.g.blocks > 1 ? "map-fused " : "small-fused ") << sg.g.kernel_index.size() << " kernels";
        if (L.tail_of >= 0) os << " (run by the last block of launch " << L.tail_of << " when a range holds both)";
        break;
      }
    }
    for (auto& ov : plan.overlaps) {
      if ((int)i >= ov.first && (int)i < ov.big) os << "   || side lane, next to launch " << ov.big;
      if ((int)i == ov.deferred_row) os << "   || its fold runs beside launch " << ov.big << " (when a range holds both)";
    }
    if (plan.pipe.active && (int)i < plan.n_backward)
      os << (L.heavy ? "   || main lane" : "   || side lane") << (L.slice_mode == 2 ? ", halves accumulate" : ", by rows");
    os << "\n";
  }
  m->launch_text = os.str();
  return m->launch_text.c_str();
}

int eg_model_kernel_count(eg_model* m, const char* target) try {
  if (!m || !target) return -1;
  auto it = m->targets.find(target);
  return it == m->targets.end() ? -1 : (int)it->second.target->live.size();
}
EG_CATCH_ALL

int eg_model_tensor_count(eg_model* m) { return m ? (int)m->prog.tensors.size() - 1 : 0; }

int eg_model_param_info(eg_model* m, int tensor_id, int* kind, int* rank, int64_t* shape8, char* name,
                        size_t name_cap) try {
  EG_REQUIRE(m && tensor_id >= 1 && tensor_id < (int)m->prog.tensors.size(), EG_ERR_INVALID, "bad tensor id %d", tensor_id);
  const TensorDef& d = m->prog.tensors[tensor_id];
  if (kind) *kind = (int)d.kind;
  if (rank) *rank = d.has_shape ? (int)d.shape.size() : -1;
  if (shape8)
    for (size_t i = 0; i < d.shape.size() && i < 8; ++i) shape8[i] = d.shape[i];
  if (name && name_cap) {
    snprintf(name, name_cap, "%s", d.name.c_str());
  }
  return EG_OK;
}
EG_CATCH_ALL

// The float32-typed entry points refuse a float64 model and the other way round: a Tensor[float32] handed to a
// Model[float64] does not compile in the reference either (model.nim:357-376 are generic over the model's T).
#define EG_REQUIRE_SCALAR(m, want64, fn)                                                                         \
  EG_REQUIRE((m)->f64 == (want64), EG_ERR_INVALID, "%s: the model computes in %s (the T of compile[T]); call the %s", fn, \
             (m)->f64 ? "float64" : "float32", (m)->f64 ? "_f64 entry points" : "entry points without _f64")

static int param_write(eg_model* m, int tensor_id, const void* host, int64_t count, bool f64, const char* fn) {
  EG_REQUIRE(m && host, EG_ERR_INVALID, "%s: NULL argument", fn);
  EG_REQUIRE_SCALAR(m, f64, fn);
  auto it = m->params.find(tensor_id);
  EG_REQUIRE(it != m->params.end(), EG_ERR_INVALID, "tensor %d is not a parameter", tensor_id);
  EG_REQUIRE(count == it->second.count, EG_ERR_SIZE, "parameter %d has %ld elements, got %ld", tensor_id,
             it->second.count, (long)count);
  if (count == 0) return EG_OK;
  return eg::copy_h2d(m->ctx, it->second.ptr, host, (size_t)count * sizeof(float) * m->esz);
}

static int param_read(eg_model* m, int tensor_id, void* host, int64_t count, bool f64, const char* fn) {
  EG_REQUIRE(m && host, EG_ERR_INVALID, "%s: NULL argument", fn);
  EG_REQUIRE_SCALAR(m, f64, fn);
  auto it = m->params.find(tensor_id);
  EG_REQUIRE(it != m->params.end(), EG_ERR_INVALID, "tensor %d is not a parameter", tensor_id);
  EG_REQUIRE(count == it->second.count, EG_ERR_SIZE, "parameter %d has %ld elements, got %ld", tensor_id,
             it->second.count, (long)count);
  if (count == 0) return EG_OK;
  return eg::copy_d2h(m->ctx, host, it->second.ptr, (size_t)count * sizeof(float) * m->esz);
}

int eg_model_scalar_bytes(eg_model* m) { return m ? 4 * m->esz : 0; }

int eg_model_param_write_f64(eg_model* m, int tensor_id, const double* host, int64_t count) try {
  return param_write(m, tensor_id, host, count, true, "eg_model_param_write_f64");
}
EG_CATCH_ALL

int eg_model_param_read_f64(eg_model* m, int tensor_id, double* host, int64_t count) try {
  return param_read(m, tensor_id, host, count, true, "eg_model_param_read_f64");
}
EG_CATCH_ALL

int eg_model_param_write(eg_model* m, int tensor_id, const float* host, int64_t count) try {
  EG_REQUIRE(m && host, EG_ERR_INVALID, "eg_model_param_write: NULL argument");
  EG_REQUIRE_SCALAR(m, false, "eg_model_param_write");
  auto it = m->params.find(tensor_id);
  EG_REQUIRE(it != m->params.end(), EG_ERR_INVALID, "tensor %d is not a parameter", tensor_id);
  EG_REQUIRE(count == it->second.count, EG_ERR_SIZE, "parameter %d has %ld elements, got %ld", tensor_id,
             it->second.count, (long)count);
  if (count == 0) return EG_OK;
  return eg::copy_h2d(m->ctx, it->second.ptr, host, (size_t)count * sizeof(float));
}
EG_CATCH_ALL

int eg_model_param_read(eg_model* m, int tensor_id, float* host, int64_t count) try {
  EG_REQUIRE(m && host, EG_ERR_INVALID, "eg_model_param_read: NULL argument");
  EG_REQUIRE_SCALAR(m, false, "eg_model_param_read");
  auto it = m->params.find(tensor_id);
  EG_REQUIRE(it != m->params.end(), EG_ERR_INVALID, "tensor %d is not a parameter", tensor_id);
  EG_REQUIRE(count == it->second.count, EG_ERR_SIZE, "parameter %d has %ld elements, got %ld", tensor_id,
             it->second.count, (long)count);
  if (count == 0) return EG_OK;
  return eg::copy_d2h(m->ctx, host, it->second.ptr, (size_t)count * sizeof(float));
}
EG_CATCH_ALL

int eg_model_param_ptr(eg_model* m, int tensor_id, float** device_ptr, int64_t* count) try {
  EG_REQUIRE(m, EG_ERR_INVALID, "NULL model");
  auto it = m->params.find(tensor_id);
  EG_REQUIRE(it != m->params.end(), EG_ERR_INVALID, "tensor %d is not a parameter", tensor_id);
  if (device_ptr) *device_ptr = it->second.ptr;
  if (count) *count = it->second.count;
  return EG_OK;
}
EG_CATCH_ALL

int eg_model_grad_bucket(eg_model* m, const char* target, float** device_ptr, int64_t* count) try {
  EG_REQUIRE(m && target, EG_ERR_INVALID, "NULL argument");
  auto it = m->targets.find(target);
  EG_REQUIRE(it != m->targets.end(), EG_ERR_RUNTIME, "%s is not a target of the model", target);
  if (device_ptr) *device_ptr = it->second.bucket;
  if (count) *count = it->second.bucket_floats / m->esz;  // elements (doubles of a float64 model)
  return EG_OK;
}
EG_CATCH_ALL

int eg_model_bind_grad_bucket(eg_model* m, const char* target, float* device_ptr, int64_t count) try {
  EG_REQUIRE(m && target && device_ptr, EG_ERR_INVALID, "NULL argument");
  auto it = m->targets.find(target);
  EG_REQUIRE(it != m->targets.end(), EG_ERR_RUNTIME, "%s is not a target of the model", target);
  TargetState& ts = it->second;
  EG_REQUIRE(count * m->esz >= ts.bucket_floats, EG_ERR_SIZE, "gradient bucket needs %ld elements, got %ld", ts.bucket_floats / m->esz,
             (long)count);
  EG_HIP_CHECK(hipSetDevice(m->ctx->device));
  EG_HIP_CHECK(hipStreamSynchronize(m->ctx->stream));
  if (ts.bucket_owned && ts.bucket) hipFree(ts.bucket);
  ts.bucket = device_ptr;
  ts.bucket_owned = false;
  return EG_OK;
}
EG_CATCH_ALL

static int bind_input(eg_model* m, const char* name, const float* device, const float* host, int rank,
                      const int64_t* shape, bool f64 = false) {
  EG_REQUIRE(m && name, EG_ERR_INVALID, "NULL argument");
  EG_REQUIRE_SCALAR(m, f64, f64 ? "eg_model_set_input_*_f64" : "eg_model_set_input_*");
  auto it = m->prog.inputs.find(name);
  // model.nim:358-359
  EG_REQUIRE(it != m->prog.inputs.end(), EG_ERR_RUNTIME, "%s is not an input to the model", name);
  EG_REQUIRE(rank >= 0 && rank <= 8 && (rank == 0 || shape), EG_ERR_INVALID, "bad input rank");
  BoundInput& b = m->inputs[it->second];
  m->inputs_gen++;
  b.bound = true;
  b.shape.assign(shape, shape + rank);
  const long count = prod(b.shape);
  if (host) {
    EG_HIP_CHECK(hipSetDevice(m->ctx->device));
    if (b.owned_count < count) {
      EG_HIP_CHECK(hipStreamSynchronize(m->ctx->stream));
      if (b.owned) EG_HIP_CHECK(hipFree(b.owned));
      b.owned = nullptr;
      b.owned_count = 0;
      EG_HIP_CHECK(hipMalloc((void**)&b.owned, (size_t)(count > 0 ? count : 1) * sizeof(float) * m->esz));
      b.owned_count = count;
    }
    if (count > 0) {
      // blocking H2D on every call, as the reference does (model.nim:364-368 -> cl.nim:111-116)
      int rc = eg::copy_h2d(m->ctx, b.owned, host, (size_t)count * sizeof(float) * m->esz);
      if (rc) return rc;
    }
    b.device = b.owned;
  } else {
    EG_REQUIRE(device || count == 0, EG_ERR_INVALID, "NULL device pointer for input %s", name);
    b.device = device;
  }
  return EG_OK;
}

int eg_model_set_input_host(eg_model* m, const char* name, const float* host, int rank, const int64_t* shape) try {
  EG_REQUIRE(host || rank == 0, EG_ERR_INVALID, "NULL host pointer");
  static const float dummy = 0;
  return bind_input(m, name, nullptr, host ? host : &dummy, rank, shape);
}
EG_CATCH_ALL

int eg_model_set_input_device(eg_model* m, const char* name, const float* device_ptr, int rank, const int64_t* shape) try {
  return bind_input(m, name, device_ptr, nullptr, rank, shape);
}
EG_CATCH_ALL

int eg_model_set_input_host_f64(eg_model* m, const char* name, const double* host, int rank, const int64_t* shape) try {
  EG_REQUIRE(host || rank == 0, EG_ERR_INVALID, "NULL host pointer");
  static const double dummy = 0;
  return bind_input(m, name, nullptr, reinterpret_cast<const float*>(host ? host : &dummy), rank, shape, true);
}
EG_CATCH_ALL

int eg_model_set_input_device_f64(eg_model* m, const char* name, const double* device_ptr, int rank, const int64_t* shape) try {
  return bind_input(m, name, reinterpret_cast<const float*>(device_ptr), nullptr, rank, shape, true);
}
EG_CATCH_ALL

int eg_model_clear_inputs(eg_model* m) try {
  EG_REQUIRE(m, EG_ERR_INVALID, "NULL model");
  // Host-staged inputs keep their staging buffer (reused by the next host bind); only the
  // bindings are forgotten.  No synchronisation: nothing is freed.
  m->inputs_gen++;
  for (auto it = m->inputs.begin(); it != m->inputs.end();) {
    if (it->second.owned) {
      it->second.device = nullptr;
      it->second.shape.clear();
      it->second.bound = false;
      ++it;
    } else {
      it = m->inputs.erase(it);
    }
  }
  return EG_OK;
}
EG_CATCH_ALL

int eg_model_run(eg_model* m, const char* target) try {
  TargetState* ts;
  Plan* plan;
  int rc = get_plan(m, target, &ts, &plan);
  if (rc) return rc;
  return run_range(m, *ts, *plan, 0, (int)plan->launches.size(), true, 0);
}
EG_CATCH_ALL

int eg_model_run_backward(eg_model* m, const char* target) try {
  TargetState* ts;
  Plan* plan;
  int rc = get_plan(m, target, &ts, &plan);
  if (rc) return rc;
  return run_range(m, *ts, *plan, 0, plan->n_backward, true, 1);
}
EG_CATCH_ALL

int eg_model_run_update(eg_model* m, const char* target) try {
  TargetState* ts;
  Plan* plan;
  int rc = get_plan(m, target, &ts, &plan);
  if (rc) return rc;
  return run_range(m, *ts, *plan, plan->n_backward, (int)plan->launches.size(), false, 2);
}
EG_CATCH_ALL

int eg_model_set_grad_scale(eg_model* m, float scale) try {
  EG_REQUIRE(m, EG_ERR_INVALID, "NULL model");
  m->grad_scale = scale;
  return EG_OK;
}
EG_CATCH_ALL

static int find_tensor(eg_model* m, const char* target, int* tid, TargetState** ts) {
  auto it = m->targets.find(target);
  EG_REQUIRE(it != m->targets.end(), EG_ERR_RUNTIME, "%s is not a target of the model", target);
  *ts = &it->second;
  *tid = it->second.target->output;
  EG_REQUIRE(*tid != 0, EG_ERR_INVALID, "target %s has no output tensor", target);
  EG_REQUIRE(it->second.last, EG_ERR_INVALID, "target %s has not been run", target);
  return EG_OK;
}

static int tensor_shape(eg_model* m, TargetState& ts, int tid, int* rank, int64_t* shape8) {
  EG_REQUIRE(ts.last, EG_ERR_INVALID, "the target has not been run");
  auto s = ts.last->shapes.find(tid);
  EG_REQUIRE(s != ts.last->shapes.end(), EG_ERR_INVALID, "tensor %d has no shape in the last run", tid);
  if (rank) *rank = (int)s->second.size();
  if (shape8)
    for (size_t i = 0; i < s->second.size() && i < 8; ++i) shape8[i] = s->second[i];
  return EG_OK;
}

static int read_tensor(eg_model* m, TargetState& ts, int tid, void* host, int64_t count, bool f64 = false) {
  EG_REQUIRE(ts.last && host, EG_ERR_INVALID, "nothing to read");
  EG_REQUIRE_SCALAR(m, f64, f64 ? "eg_model_read_*_f64" : "eg_model_read_*");
  auto s = ts.last->shapes.find(tid);
  EG_REQUIRE(s != ts.last->shapes.end(), EG_ERR_INVALID, "tensor %d has no shape in the last run", tid);
  const long n = prod(s->second);
  EG_REQUIRE(count == n, EG_ERR_SIZE, "Buffer size is not equal to target size (%ld vs %ld)", n, (long)count);
  if (n == 0) return EG_OK;
  EG_REQUIRE(!ts.last->predicated.count(tid), EG_ERR_INVALID,
             "tensor %d exists only as predicate bits in the last run's plan (eg_model_keep_values(model, 1) makes the plans keep values)", tid);
  EG_REQUIRE(!(ts.last->sample_group && ts.last->sample_group->g.lds.count(tid)), EG_ERR_INVALID,
             "tensor %d lived in the LDS of a sample group's blocks in the last run's plan: its values were never stored "
             "(eg_model_keep_values(model, 1) makes the plans keep values)", tid);
  float* p = tensor_ptr(m, ts, *ts.last, tid);
  EG_REQUIRE(p, EG_ERR_INVALID, "tensor %d was not materialised by the last run", tid);
  return eg::copy_d2h(m->ctx, host, p, (size_t)n * sizeof(float) * m->esz);
}

int eg_model_output_shape(eg_model* m, const char* target, int* rank, int64_t* shape8) try {
  EG_REQUIRE(m && target, EG_ERR_INVALID, "NULL argument");
  int tid;
  TargetState* ts;
  int rc = find_tensor(m, target, &tid, &ts);
  if (rc) return rc;
  return tensor_shape(m, *ts, tid, rank, shape8);
}
EG_CATCH_ALL

int eg_model_read_output(eg_model* m, const char* target, float* host, int64_t count) try {
  EG_REQUIRE(m && target, EG_ERR_INVALID, "NULL argument");
  int tid;
  TargetState* ts;
  int rc = find_tensor(m, target, &tid, &ts);
  if (rc) return rc;
  return read_tensor(m, *ts, tid, host, count);
}
EG_CATCH_ALL

int eg_model_read_output_f64(eg_model* m, const char* target, double* host, int64_t count) try {
  EG_REQUIRE(m && target, EG_ERR_INVALID, "NULL argument");
  int tid;
  TargetState* ts;
  int rc = find_tensor(m, target, &tid, &ts);
  if (rc) return rc;
  return read_tensor(m, *ts, tid, host, count, true);
}
EG_CATCH_ALL

static TargetState* last_target(eg_model* m, const char* target) {
  auto it = m->targets.find(target ? target : "");
  return it == m->targets.end() ? nullptr : &it->second;
}

int eg_model_tensor_shape(eg_model* m, const char* target, int tensor_id, int* rank, int64_t* shape8) try {
  EG_REQUIRE(m, EG_ERR_INVALID, "NULL model");
  TargetState* ts = last_target(m, target);
  EG_REQUIRE(ts, EG_ERR_RUNTIME, "%s is not a target of the model", target ? target : "(null)");
  return tensor_shape(m, *ts, tensor_id, rank, shape8);
}
EG_CATCH_ALL

int eg_model_read_tensor(eg_model* m, const char* target, int tensor_id, float* host, int64_t count) try {
  EG_REQUIRE(m, EG_ERR_INVALID, "NULL model");
  TargetState* ts = last_target(m, target);
  EG_REQUIRE(ts, EG_ERR_RUNTIME, "%s is not a target of the model", target ? target : "(null)");
  return read_tensor(m, *ts, tensor_id, host, count);
}
EG_CATCH_ALL

int eg_model_read_tensor_f64(eg_model* m, const char* target, int tensor_id, double* host, int64_t count) try {
  EG_REQUIRE(m, EG_ERR_INVALID, "NULL model");
  TargetState* ts = last_target(m, target);
  EG_REQUIRE(ts, EG_ERR_RUNTIME, "%s is not a target of the model", target ? target : "(null)");
  return read_tensor(m, *ts, tensor_id, host, count, true);
}
EG_CATCH_ALL

int eg_model_tensor_ptr(eg_model* m, const char* target, int tensor_id, float** device_ptr, int64_t* count) try {
  EG_REQUIRE(m, EG_ERR_INVALID, "NULL model");
  TargetState* ts = last_target(m, target);
  EG_REQUIRE(ts && ts->last, EG_ERR_RUNTIME, "target has not been run");
  auto s = ts->last->shapes.find(tensor_id);
  EG_REQUIRE(s != ts->last->shapes.end(), EG_ERR_INVALID, "tensor %d has no shape in the last run", tensor_id);
  // the same refusals as eg_model_read_tensor: an address that was allocated but never written would read as garbage
  EG_REQUIRE(!ts->last->predicated.count(tensor_id), EG_ERR_INVALID,
             "tensor %d exists only as predicate bits in the last run's plan (eg_model_keep_values(model, 1) makes the plans keep values)",
             tensor_id);
  EG_REQUIRE(!(ts->last->sample_group && ts->last->sample_group->g.lds.count(tensor_id)), EG_ERR_INVALID,
             "tensor %d lived in the LDS of a sample group's blocks in the last run's plan: its values were never stored "
             "(eg_model_keep_values(model, 1) makes the plans keep values)", tensor_id);
  if (device_ptr) *device_ptr = tensor_ptr(m, *ts, *ts->last, tensor_id);
  if (count) *count = prod(s->second);
  return EG_OK;
}
EG_CATCH_ALL

int eg_model_set_epoch(eg_model* m, int64_t epoch) try {
  EG_REQUIRE(m, EG_ERR_INVALID, "NULL model");
  m->epoch = epoch;
  return EG_OK;
}
EG_CATCH_ALL

int64_t eg_model_epoch(eg_model* m) { return m ? m->epoch : 0; }

int eg_model_keep_values(eg_model* m, int on) try {
  EG_REQUIRE(m, EG_ERR_INVALID, "NULL model");
  if (m->keep_values != (on != 0)) {
    m->keep_values = on != 0;
    m->inputs_gen++;  // the plan key changed: the next run looks its plan up again
  }
  return EG_OK;
}
EG_CATCH_ALL

int eg_model_set_seed(eg_model* m, uint64_t seed) try {
  EG_REQUIRE(m, EG_ERR_INVALID, "NULL model");
  return ensure_rng(m, seed, true);
}
EG_CATCH_ALL

}  // extern "C"
