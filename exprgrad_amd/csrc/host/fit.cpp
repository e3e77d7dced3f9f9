// eg_model_fit: Model.fit (model.nim:413-454) as one C-ABI call.
#include <random>
#include <sstream>
#include "model_types.hpp"

using namespace eg::kd;
using namespace eg::model;
using eg::set_error;


namespace {

// Batches per graph launch (EG_FIT_GROUP; 1 = every batch its own launch).  Between two launches of a captured sequence the
// device idles for ~8 us whatever the sequence is, which a batch-32 step of 77 us notices and a batch-4096
// step of 550 us does not: groups only for small batches.
long fit_group_size(long batch_size) {
  if (!graphs_enabled()) return 1;
  if (const char* e = eg::sw::raw("EG_FIT_GROUP")) {
    const long g = atol(e);
    return g >= 1 && g <= 64 ? g : 1;
  }
  return batch_size <= 256 ? 16 : 1;   // (measured at batch 32: 4 -> 78.4 us, 8 -> 78.1, 16 -> 76.8, 32 -> 79.0, 64 -> 82.2; every batch by itself 82.1)
}

// Is the sample group's kernel the only launch of the plan that reads an input tensor?  (Then a batch's rows can be read
// where they lie.)  Conservative: any launch kind whose operands are not enumerated here counts as a reader.
bool inputs_read_by_sample_kernel_only(eg_model* m, TargetState& ts, Plan& plan) {
  if (!plan.sample_group || eg::sw::raw("EG_FIT_NO_DIRECT")) return false;
  const Target& t = *ts.target;
  auto is_input = [&](int tid) { return tid > 0 && m->prog.tensors[tid].kind == TK::Input; };
  bool sample_reads = false;
  for (int tid : plan.sample_group->g.ptr_args) sample_reads = sample_reads || is_input(tid);
  if (!sample_reads) return false;
  for (const Launch& L : plan.launches) {
    switch (L.kind) {
      case StepKind::SampleFused: break;
      case StepKind::Seed: break;
      case StepKind::SmallFused:
        for (int ki : plan.small_groups[L.row_group]->g.kernel_index)
          for (auto& rd : t.all[ki].reads)
            if (is_input(rd.tensor)) return false;
        break;
      case StepKind::Gemm:
      case StepKind::Conv:
      case StepKind::ConvGradImage:
      case StepKind::ConvGradFilter:
        if (is_input(L.a_tensor) || is_input(L.b_tensor) || is_input(L.bias_tensor)) return false;
        break;
      default: return false;
    }
  }
  return true;
}

// One graph launch for the batches [b, b + group): the graph holds `group` times (segment copy, launch sequence); its copy
// nodes are re-pointed at the rows of these batches first.  *done = false: not available here (capture refused, node
// parameters not updatable) — the caller goes on batch by batch.
// (Round 5: when the sample group's kernel is the only reader of the inputs, the batches are captured without their copies
// and the kernel nodes' input arguments are re-pointed instead — FitGraph::direct.)
template <class RowsOf>
int launch_group(eg_model* m, TargetState& ts, Plan& plan, long group, long b, RowsOf rows_of, const std::vector<BoundInput*>& col_inputs,
                 const std::vector<int>& col_tids, bool* done) {
  *done = false;
  eg_ctx* ctx = m->ctx;
  eg_model::FitGraph& fg = m->fit_graph;
  const bool want_direct = inputs_read_by_sample_kernel_only(m, ts, plan);
  // argument index of every column's tensor in the sample kernel: (slab, t<ptr_args>..., GS, EP)
  std::vector<int> arg_of(col_tids.size(), -1);
  if (want_direct)
    for (size_t i = 0; i < col_tids.size(); ++i)
      for (size_t a = 0; a < plan.sample_group->g.ptr_args.size(); ++a)
        if (plan.sample_group->g.ptr_args[a] == col_tids[i]) arg_of[i] = 1 + (int)a;
  std::ostringstream k;
  k << capture_key(m, ts) << "|plan" << (const void*)&plan << "|g" << group << "|n" << plan.launches.size();
  const std::string key = k.str();
  bool fresh = false;
  if (!fg.exec || fg.key != key) {
    fresh = true;
    if (fg.exec) {
      EG_HIP_CHECK(hipStreamSynchronize(ctx->stream));   // (an older group may still be running)
      hipGraphExecDestroy(fg.exec);
      fg.exec = nullptr;
    }
    if (fg.exec2) hipGraphExecDestroy(fg.exec2);
    fg.exec2 = nullptr;
    fg.launched[0] = fg.launched[1] = false;
    fg.turn = 0;
    if (fg.graph) hipGraphDestroy(fg.graph);
    fg.graph = nullptr;
    fg.copies.clear();
    hipGraph_t graph = nullptr;
    if (hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
      (void)hipGetLastError();
      return EG_OK;
    }
    int rc = EG_OK;
    for (long j = 0; j < group && !rc; ++j) {
      if (want_direct) {  // the kernels read the batch's rows in place
        const eg::CopySegments cs = rows_of(b + j);
        for (size_t i = 0; i < col_inputs.size(); ++i) col_inputs[i]->device = cs.src[i];
      } else {
        rc = eg::copy_segments(ctx, rows_of(b + j));
      }
      if (!rc) rc = run_range_eager(m, ts, plan, 0, (int)plan.launches.size(), true);
    }
    if (want_direct)
      for (BoundInput* in : col_inputs) in->device = in->owned;  // (the staging buffers stay what single batches use)
    const hipError_t e = hipStreamEndCapture(ctx->stream, &graph);
    if (rc) {
      if (graph) hipGraphDestroy(graph);
      return rc;
    }
    if (e != hipSuccess || !graph) {
      (void)hipGetLastError();
      return EG_OK;
    }
    // the copy nodes, in batch order: the node whose first source is the first source of batch b + j
    size_t count = 0;
    std::vector<hipGraphNode_t> nodes;
    bool ok = hipGraphGetNodes(graph, nullptr, &count) == hipSuccess;
    if (ok) {
      nodes.resize(count);
      ok = hipGraphGetNodes(graph, nodes.data(), &count) == hipSuccess;
    }
    std::vector<hipGraphNode_t> copies((size_t)group, nullptr);
    void* const reader_fn = want_direct ? eg::kernel_function(plan.sample_group->handle) : nullptr;
    int first_col = -1;
    for (size_t i = 0; i < arg_of.size() && first_col < 0; ++i)
      if (arg_of[i] >= 0) first_col = (int)i;
    if (want_direct && (first_col < 0 || !reader_fn)) ok = false;
    for (size_t i = 0; ok && i < count; ++i) {
      hipGraphNodeType type;
      if (hipGraphNodeGetType(nodes[i], &type) != hipSuccess || type != hipGraphNodeTypeKernel) continue;
      hipKernelNodeParams p = {};
      if (hipGraphKernelNodeGetParams(nodes[i], &p) != hipSuccess) continue;
      if (want_direct) {  // the sample kernel's node whose first input argument is batch b + j's rows
        if (p.func != reader_fn || !p.kernelParams || !p.kernelParams[arg_of[(size_t)first_col]]) continue;
        const float* have = *static_cast<const float* const*>(p.kernelParams[arg_of[(size_t)first_col]]);
        for (long j = 0; j < group; ++j)
          if (have == rows_of(b + j).src[first_col]) copies[(size_t)j] = nodes[i];
        continue;
      }
      if (p.func != eg::copy_segments_function() || !p.kernelParams || !p.kernelParams[0]) continue;
      const eg::CopySegments* cs = static_cast<const eg::CopySegments*>(p.kernelParams[0]);
      for (long j = 0; j < group; ++j)
        if (cs->src[0] == rows_of(b + j).src[0]) copies[(size_t)j] = nodes[i];
    }
    for (hipGraphNode_t n : copies) ok = ok && n != nullptr;
    fg.direct = want_direct;
    static const bool debug = eg::sw::raw("EG_DEBUG_GRAPH") != nullptr;
    if (debug) fprintf(stderr, "[eg] fit group of %ld batches: %zu nodes captured, copy nodes %s\n", group, count, ok ? "found" : "NOT found");
    if (ok) ok = hipGraphInstantiate(&fg.exec, graph, nullptr, nullptr, 0) == hipSuccess;
    if (!ok) {
      (void)hipGetLastError();
      hipGraphDestroy(graph);
      fg.exec = nullptr;
      return EG_OK;
    }
    if (hipGraphInstantiate(&fg.exec2, graph, nullptr, nullptr, 0) != hipSuccess) {  // (one executable then: its event is waited for before every update)
      (void)hipGetLastError();
      fg.exec2 = nullptr;
    }
    for (int i = 0; i < 2; ++i)
      if (!fg.done[i]) EG_HIP_CHECK(hipEventCreateWithFlags(&fg.done[i], hipEventDisableTiming));
    fg.graph = graph;
    fg.copies.swap(copies);
    fg.key = key;
  }
  const int t = fg.exec2 ? fg.turn : 0;
  hipGraphExec_t exec = t ? fg.exec2 : fg.exec;
  if (!fresh || t == 1) {
    if (fg.launched[t]) EG_HIP_CHECK(hipEventSynchronize(fg.done[t]));  // this executable's previous launch has run
    for (long j = 0; j < group; ++j) {
      const eg::CopySegments cs = rows_of(b + j);
      hipKernelNodeParams p;
      void* arg[1];
      bool set = false;
      if (fg.direct) {
        // the node's own argument list with the input pointers replaced (the other entries keep pointing at what the
        // capture stored: slab, parameters, arena tensors, scale, epoch)
        hipKernelNodeParams q = {};
        const float* rows[8];
        std::vector<void*> argv;
        if (hipGraphKernelNodeGetParams(fg.copies[(size_t)j], &q) == hipSuccess && q.kernelParams) {
          const size_t nargs = plan.sample_group->g.ptr_args.size() + 3;
          argv.assign(q.kernelParams, q.kernelParams + nargs);
          for (size_t i = 0; i < arg_of.size() && i < 8; ++i)
            if (arg_of[i] >= 0) {
              rows[i] = cs.src[i];
              argv[(size_t)arg_of[i]] = &rows[i];
            }
          q.kernelParams = argv.data();
          set = hipGraphExecKernelNodeSetParams(exec, fg.copies[(size_t)j], &q) == hipSuccess;
        }
      } else {
        set = eg::copy_segments_node_params(ctx, cs, &p, arg) && hipGraphExecKernelNodeSetParams(exec, fg.copies[(size_t)j], &p) == hipSuccess;
      }
      if (!set) {
        if (eg::sw::raw("EG_DEBUG_GRAPH")) fprintf(stderr, "[eg] fit group: hipGraphExecKernelNodeSetParams refused: %s\n", hipGetErrorString(hipGetLastError()));
        (void)hipGetLastError();
        EG_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        hipGraphExecDestroy(fg.exec);
        fg.exec = nullptr;
        if (fg.exec2) hipGraphExecDestroy(fg.exec2);
        fg.exec2 = nullptr;
        return EG_OK;
      }
    }
  }
  EG_HIP_CHECK(hipGraphLaunch(exec, ctx->stream));
  EG_HIP_CHECK(hipEventRecord(fg.done[t], ctx->stream));
  fg.launched[t] = true;
  fg.turn = fg.exec2 ? 1 - t : 0;
  *done = true;
  return EG_OK;
}

}  // namespace

extern "C" {

// fit (model.nim:413-454): one epoch of mini-batches.  The reference slices the host tensors
// (viewFirst) and uploads every batch with a blocking write before it enqueues the kernels
// (model.nim:364-368); here the data set is uploaded once in pieces on a second stream, piece k+1
// while the batches of piece k run, and every batch is one segment-copy launch (the batch's rows
// of every input -> that input's fixed staging buffer, so the captured launch sequence replays
// unchanged) plus one graph launch.  Nothing in the loop waits for the device.
static int model_fit(eg_model* m, const char* target, int n_inputs, const char* const* names, const float* const* data,
                     const int* on_device, const int* ranks, const int64_t* shapes8, int64_t batch_size, bool f64) {
  EG_REQUIRE(m && target, EG_ERR_INVALID, "NULL model or target");
  EG_REQUIRE(m->f64 == f64, EG_ERR_INVALID, "eg_model_fit%s: the model computes in %s (the T of compile[T])", f64 ? "_f64" : "",
             m->f64 ? "float64" : "float32");
  // model.nim:417-421
  EG_REQUIRE(n_inputs > 0, EG_ERR_RUNTIME,
             "Model.fit requires at least one input tensor. Use Model.apply instead if the target has zero inputs.");
  EG_REQUIRE(n_inputs <= 8, EG_ERR_INVALID, "Model.fit takes at most 8 inputs here");
  EG_REQUIRE(names && data && on_device && ranks && shapes8, EG_ERR_INVALID, "NULL argument");
  EG_REQUIRE(m->targets.count(target), EG_ERR_RUNTIME, "%s is not a target of the model", target);
  EG_REQUIRE(batch_size > 0, EG_ERR_INVALID, "batch size must be positive");
  struct Column {
    BoundInput* in;
    const float* data;
    bool device;
    long rows, row_floats;
    std::vector<long> batch_shape;
  };
  eg_model_clear_inputs(m);  // only the arguments of this call are bound (model.nim:438-447)
  std::vector<Column> cols((size_t)n_inputs);
  for (int i = 0; i < n_inputs; ++i) {
    EG_REQUIRE(names[i], EG_ERR_INVALID, "NULL input name");
    auto it = m->prog.inputs.find(names[i]);
    EG_REQUIRE(it != m->prog.inputs.end(), EG_ERR_RUNTIME, "%s is not an input to the model", names[i]);
    EG_REQUIRE(ranks[i] >= 1 && ranks[i] <= 8, EG_ERR_INVALID, "input %s needs a leading batch dimension", names[i]);
    Column& c = cols[(size_t)i];
    c.in = &m->inputs[it->second];
    c.data = data[i];
    c.device = on_device[i] != 0;
    c.rows = shapes8[i * 8];
    c.row_floats = 1;
    c.batch_shape.assign(1, (long)batch_size);
    for (int d = 1; d < ranks[i]; ++d) {
      EG_REQUIRE(shapes8[i * 8 + d] >= 0, EG_ERR_INVALID, "negative extent");
      c.row_floats *= shapes8[i * 8 + d];
      c.batch_shape.push_back(shapes8[i * 8 + d]);
    }
    c.row_floats *= m->esz;  // 4-byte units per row: a float64 model's rows are twice as long (host/model_types.hpp)
    EG_REQUIRE(c.data || c.rows * c.row_floats == 0, EG_ERR_INVALID, "NULL data for input %s", names[i]);
  }
  const long batch_count = cols[0].rows / batch_size;  // model.nim:434: the ragged tail is dropped
  for (auto& c : cols)
    EG_REQUIRE(c.rows >= batch_count * batch_size, EG_ERR_SHAPE, "an input has fewer rows (%ld) than the first one uses (%ld)",
               c.rows, batch_count * (long)batch_size);
  m->epoch += 1;  // model.nim:436
  if (batch_count == 0) return EG_OK;
  int rc = eg::set_device(m->ctx);
  if (rc) return rc;
  hipStream_t stream = m->ctx->stream;

  // ---- staging buffers of one batch (what the captured kernels read)
  m->inputs_gen++;
  for (auto& c : cols) {
    BoundInput& b = *c.in;
    const long count = batch_size * c.row_floats;
    const long elems = count / m->esz;  // (owned_count is in elements, as bind_input keeps it)
    if (b.owned_count < elems || !b.owned) {
      EG_HIP_CHECK(hipStreamSynchronize(stream));
      if (b.owned) EG_HIP_CHECK(hipFree(b.owned));
      b.owned = nullptr;
      b.owned_count = 0;
      EG_HIP_CHECK(hipMalloc((void**)&b.owned, (size_t)(count > 0 ? count : 1) * sizeof(float)));
      b.owned_count = elems;
    }
    b.device = b.owned;
    b.shape = c.batch_shape;
    b.bound = true;
  }

  // ---- device copy of the host columns: as many batches per segment as fit, uploaded piecewise
  size_t host_row_bytes = 0;
  for (auto& c : cols)
    if (!c.device) host_row_bytes += (size_t)c.row_floats * sizeof(float);
  long seg_batches = batch_count, piece_batches = batch_count;
  if (host_row_bytes > 0) {
    size_t free_b = 0, total_b = 0;
    EG_HIP_CHECK(hipMemGetInfo(&free_b, &total_b));
    size_t have = 0;
    for (size_t v : m->fit_bytes) have += v;
    size_t budget = (free_b + have) / 2, piece_bytes = 8u << 20;  // ~8 MiB per upload
    if (const char* e = eg::sw::raw("EG_FIT_SEGMENT_BYTES")) budget = std::min<size_t>(budget, strtoull(e, nullptr, 10));
    if (const char* e = eg::sw::raw("EG_FIT_PIECE_BYTES")) piece_bytes = strtoull(e, nullptr, 10);
    const size_t batch_bytes = host_row_bytes * (size_t)batch_size;
    EG_REQUIRE(batch_bytes <= budget, EG_ERR_SIZE, "one batch (%zu bytes) does not fit the device", batch_bytes);
    seg_batches = std::min<long>(batch_count, (long)(budget / batch_bytes));
    piece_batches = std::max<long>(1, (long)(piece_bytes / batch_bytes));
    if (!m->copy_stream) EG_HIP_CHECK(hipStreamCreateWithFlags(&m->copy_stream, hipStreamNonBlocking));
    if (!m->copy_event) EG_HIP_CHECK(hipEventCreateWithFlags(&m->copy_event, hipEventDisableTiming));
    if (!m->main_event) EG_HIP_CHECK(hipEventCreateWithFlags(&m->main_event, hipEventDisableTiming));
    m->fit_data.resize(8, nullptr);
    m->fit_bytes.resize(8, 0);
    for (int i = 0; i < n_inputs; ++i) {
      Column& c = cols[(size_t)i];
      if (c.device) continue;
      const size_t need = (size_t)seg_batches * batch_size * c.row_floats * sizeof(float);
      if (m->fit_bytes[i] < need) {
        EG_HIP_CHECK(hipStreamSynchronize(stream));
        if (m->fit_data[i]) EG_HIP_CHECK(hipFree(m->fit_data[i]));
        m->fit_data[i] = nullptr;
        m->fit_bytes[i] = 0;
        EG_HIP_CHECK(hipMalloc((void**)&m->fit_data[i], need ? need : 4));
        m->fit_bytes[i] = need;
      }
    }
  }

  TargetState* ts = nullptr;
  Plan* plan = nullptr;
  std::vector<BoundInput*> col_inputs;
  std::vector<int> col_tids;
  for (int i = 0; i < n_inputs; ++i) {
    col_inputs.push_back(cols[(size_t)i].in);
    col_tids.push_back(m->prog.inputs.find(names[i])->second);
  }
  long group = fit_group_size(batch_size), single_batches = 0;
  for (long seg = 0; seg < batch_count; seg += seg_batches) {
    const long seg_end = std::min(batch_count, seg + seg_batches);
    // the segment buffer is about to be overwritten: the batches that read it must be done — the ones
    // of the previous segment, and (seg == 0) the ones a previous fit call left queued: this call
    // returns once its uploads are complete, not its kernels, so the next call's first upload would
    // otherwise land in rows that pending batches still read.  The copy stream waits; the host does not.
    if (host_row_bytes > 0) {
      EG_HIP_CHECK(hipEventRecord(m->main_event, stream));
      EG_HIP_CHECK(hipStreamWaitEvent(m->copy_stream, m->main_event, 0));
    }
    for (long piece = seg; piece < seg_end; piece += piece_batches) {
      const long piece_end = std::min(seg_end, piece + piece_batches);
      if (host_row_bytes > 0) {
        for (int i = 0; i < n_inputs; ++i) {
          Column& c = cols[(size_t)i];
          if (c.device) continue;
          const size_t off = (size_t)(piece - seg) * batch_size * c.row_floats;
          const size_t count = (size_t)(piece_end - piece) * batch_size * c.row_floats;
          if (count)
            EG_HIP_CHECK(hipMemcpyAsync(m->fit_data[i] + off, c.data + (size_t)piece * batch_size * c.row_floats,
                                        count * sizeof(float), hipMemcpyHostToDevice, m->copy_stream));
        }
        EG_HIP_CHECK(hipEventRecord(m->copy_event, m->copy_stream));
        EG_HIP_CHECK(hipStreamWaitEvent(stream, m->copy_event, 0));
      }
      auto rows_of = [&](long b) {  // the segment copy of batch b: its rows of every input -> the staging buffers
        eg::CopySegments cs = {};
        cs.n = n_inputs;
        for (int i = 0; i < n_inputs; ++i) {
          Column& c = cols[(size_t)i];
          cs.src[i] = c.device ? c.data + (size_t)b * batch_size * c.row_floats
                               : m->fit_data[i] + (size_t)(b - seg) * batch_size * c.row_floats;
          cs.dst[i] = c.in->owned;
          cs.count[i] = batch_size * c.row_floats;
        }
        return cs;
      };
      for (long b = piece; b < piece_end;) {
        // `group` batches as one graph launch, once the launch sequence has run twice by itself (kernels built, workspaces
        // grown, its own graph captured)
        if (group > 1 && single_batches >= 2 && piece_end - b >= group && plan) {
          bool done = false;
          rc = launch_group(m, *ts, *plan, group, b, rows_of, col_inputs, col_tids, &done);
          if (rc) return rc;
          if (done) {
            b += group;
            continue;
          }
          group = 1;  // (capture or node update unavailable: batch by batch from here on)
        }
        rc = eg::copy_segments(m->ctx, rows_of(b));
        if (rc) return rc;
        if (!plan) {
          rc = get_plan(m, target, &ts, &plan);
          if (rc) return rc;
        }
        rc = run_range(m, *ts, *plan, 0, (int)plan->launches.size(), true, 0);
        if (rc) return rc;
        ++single_batches;
        ++b;
      }
    }
  }
  // the caller may reuse its host arrays: the uploads (not the kernels) are complete on return
  if (host_row_bytes > 0) EG_HIP_CHECK(hipStreamSynchronize(m->copy_stream));
  return EG_OK;
}

int eg_model_fit(eg_model* m, const char* target, int n_inputs, const char* const* names, const float* const* data,
                 const int* on_device, const int* ranks, const int64_t* shapes8, int64_t batch_size) try {
  return model_fit(m, target, n_inputs, names, data, on_device, ranks, shapes8, batch_size, false);
}
EG_CATCH_ALL

int eg_model_fit_f64(eg_model* m, const char* target, int n_inputs, const char* const* names, const double* const* data,
                     const int* on_device, const int* ranks, const int64_t* shapes8, int64_t batch_size) try {
  return model_fit(m, target, n_inputs, names, reinterpret_cast<const float* const*>(data), on_device, ranks, shapes8, batch_size, true);
}
EG_CATCH_ALL


}  // extern "C"
