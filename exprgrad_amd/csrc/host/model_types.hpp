// Types and internal interfaces of the model layer (group 3 of the C ABI), shared by its translation
// units:
//   match.cpp          library patterns (contraction, bias, convolution and its gradients)
//   lower.cpp          static lowering of a target, inlining, generated sources
//   plan.cpp           per-shape plans: launch list, overwrite / accumulate, arena   (+ plan_check.cpp)
//   plan_groups.cpp    row / small / map fusion groups
//   plan_epilogue.cpp  contraction + consumer epilogues
//   plan_overlap.cpp   side-lane groups
//   run.cpp            launches, HIP graphs
//   fit.cpp            eg_model_fit
//   serialize.cpp      eg_model_save / eg_model_load (io/serialize.nim layout)
//   model_api.cpp      the remaining extern "C" entry points
#pragma once
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <set>
#include <sstream>
#include <string>
#include <vector>

#include "../eg_internal.hpp"
#include "dp_schedule.hpp"
#include "../kernels/gemm_fused.hpp"
#include "codegen.hpp"
#include "epilogue.hpp"
#include "kd.hpp"
#include "dp_schedule.hpp"
#include "rowfuse.hpp"

namespace eg {
namespace model {

using namespace eg::kd;
using eg::set_error;

struct GemmMatch {
  int a_read = 0, b_read = 0;  // indices into k.reads
  bool trans_a = false, trans_b = false;
  int li = 0, lj = 0, lk = 0;  // loop indices of m, n, k
};

struct ConvMatch {
  // which operand plays which part: -1 = the written tensor, 0 / 1 = k.reads[i]
  int out_op = -1, img_op = 0, flt_op = 1;
  bool batched = true;
  enum Role { Forward, GradImage, GradFilter } role = Forward;
};

enum class StepKind { Gemm, Conv, ConvGradImage, ConvGradFilter, Seed, GenericA, GenericB, RowFused, SmallFused, GemmFused, SampleFused };

struct Generic {
  GenericSource src;
  eg_kernel* handle = nullptr;
};

// Static (shape independent) lowering decision for one live kernel.
struct Lowered {
  StepKind kind = StepKind::GenericA;
  int all_index = 0;  // index into target.all
  GemmMatch gemm;
  int bias_tensor = 0;  // fused bias (0 = none)
  bool absorbed = false;  // this kernel was folded into the previous step
  bool inlined = false;   // an elementwise producer recomputed inside its consumers: never launched, never stored
  // Consumer inlining (inline_consumers): this kernel with the elementwise kernel at live position
  // `consumer` applied to every value before it is stored; used by a plan when the shapes allow it.
  int consumer = -1;
  std::unique_ptr<Kernel> with_consumer;
  Generic with_consumer_code;
  ConvMatch conv;
  Generic mode_a;
  std::map<int, Generic> mode_b;  // by tx
  bool b_capable = false;
};

// Concrete launch for one plan.
struct Launch {
  int lowered = 0;
  StepKind kind = StepKind::GenericA;
  bool accumulate = true;
  // Gemm / Conv
  long M = 0, N = 0, K = 0, lda = 0, ldb = 0, ldc = 0;
  int a_tensor = 0, b_tensor = 0, c_tensor = 0, bias_tensor = 0;
  bool trans_a = false, trans_b = false;
  long cN = 0, cH = 0, cW = 0, cC = 0, cF = 0, cFH = 0, cFW = 0;
  // Seed
  long count = 0;
  // Generic
  Generic* generic = nullptr;
  std::vector<long> params;
  long blocks_x = 1, blocks_y = 1;
  long partial_rows = 0, partial_cols = 0;  // mode B second stage
  std::vector<int> epoch_slots;             // params refreshed from Model.epoch at every launch
  int row_group = -1;                       // RowFused: index into Plan::row_groups
  int epilogue = -1;                        // GemmFused: index into Plan::epilogues
  // Batch pipeline (plan_pipeline.cpp): how this launch is cut when the batch runs in two halves.
  //   0 = not cut (not pipelinable), 1 = rows: the batch indexes the rows of A / C and of every
  //   [batch, ...] operand, 2 = reduction: the batch is the contraction's K (weight gradients), the
  //   second half accumulates onto the first.  RowFused launches are always cut by rows.
  int slice_mode = 0;
  bool heavy = false;                       //   long contraction: stays on the main lane, the rest goes to the side lane
  int ones_tensor = 0;                      // Gemm: the bias gradient that rides along as row M of [gW; gb] (fold_bias_gradients)
  int ones_lowered = -1;                    //   live position of the column-sum kernel it replaces
  int consumer = -1;                        // GenericA: live position of the elementwise consumer folded into it
  int vec_slot = -1;                        // GenericA: index of the Slot::Vec4 argument, if the kernel has one
  bool vec_ok = false;                      //   the shapes allow four elements per thread (pointers are checked per launch)
  long total_items = 0;                     //   independent iterations (grid = items or items / 4)
  // Row tail (plan_groups.cpp fuse_row_tails): a small / map group that directly follows a multi-block row group runs in
  // that group's last block.  Both launches stay in the list; when a range holds both, the row launch gets MODE 2 and the
  // tail launch is skipped; when a range ends between them (backward | update of the data-parallel step) both run.
  int tail_launch = -1;                     // RowFused: index of the launch its last block can run
  int tail_of = -1;                         // SmallFused: index of the RowFused launch that can run it
  // Slab fold (plan_groups.cpp fuse_slab_fold): the map group behind a sample group adds up the slab rows itself
  int fold_launch = -1;                     // SampleFused: index of the map group that folds its slab when a range holds both
  int fold_of = -1;                         // SmallFused: index of that sample group's launch
  // float64 programs: a generated kernel that is a convolution or one of its two gradients (match_conv) first offers itself
  // to the matrix-core kernels over double (kernels/conv2_band.cpp, kernels/conv2_direct.cpp; the conv fields above are
  // filled in); what those decline runs as the generated kernel.
  int conv_direct64 = 0;  // 1 forward, 2 image gradient, 3 filter gradient
};

// A run of per-sample kernels fused into one generated kernel (rowfuse.hpp), built per plan.
struct PlanRowGroup {
  RowGroup g;
  eg_kernel* handle = nullptr;
  float* partial = nullptr;  // [nblocks][g.red_total]
  int nblocks = 0;
  std::vector<int> red_tensors;  // reduction destinations, in segment order
  unsigned* counter = nullptr;   // in_kernel_finalize: arrival tickets of the blocks (zero between launches)
  int tail_group = -1;           // index into Plan::small_groups of the group the last block runs (MODE 2), or -1
};

// A run of per-sample kernels as one generated kernel with one block per sample (rowfuse.hpp "sample groups"), per plan.
struct PlanSampleGroup {
  SampleGroup g;
  eg_kernel* handle = nullptr;
  float* slab = nullptr;          // [B][g.slab_floats]: every sample's contribution to the tensors summed over the batch
  long bucket_base = 0;           // the slab rows mirror bucket floats [bucket_base, bucket_base + g.slab_floats)
  std::vector<int> positions;     // live positions of the member kernels, in order
  std::vector<int> sum_tensors;   // tensors summed over the batch (gradient bucket members)
};

// A run of small-tensor kernels (optimizer updates) fused into one single-block kernel.
struct PlanSmallGroup {
  SmallGroup g;
  eg_kernel* handle = nullptr;
};

// A contraction whose elementwise consumer runs as its epilogue (epilogue.hpp), built per plan.
struct PlanEpilogue {
  EpilogueSpec spec;
  bool store_c = false;                     // the contraction result itself is written too (something else reads it)
  bool pred_write = false;                  // ... as one predicate bit per element (Plan::predicated), not as values
  bool pred_whole_words = false;            // every tile is whole and leaves through LDS: whole 32-bit words are stored, no OR
  std::map<int, PredicateSpec> pred_reads;  // operands of the consumer that exist as predicate bits only
  Launch consumer;                          // the consumer as its own launch (split-K fallback)
  // Row product (fold_row_products): the next layer's narrow contraction computed on the consumer's rows in LDS.
  bool row_product = false;
  Launch product;                           // that contraction as its own launch (fallback; its tensors for the checks)
  std::string plain_struct_code;            // the functor without the row product
  std::map<std::string, eg_kernel*> built_plain;
  std::map<std::string, eg_kernel*> built;  // by template variant (tile shape, alignment class)
};

struct DevTensor {
  float* ptr = nullptr;
  long count = 0;
  std::vector<long> shape;
};

struct Plan {
  std::string key;
  int esz = 1;  // 4-byte units per element (eg_model::esz)
  Shapes shapes;
  std::vector<Launch> launches;
  int n_backward = 0;  // launches before the first parameter update
  // Result tensors that are a whole-tensor raw copy of another tensor (reshape, passes.nim:643-688,
  // and the gradient of one) share its storage instead of being copied: dest -> source.
  std::map<int, int> alias;
  // Predicate tensors (plan_epilogue.cpp): result tensors every reader of which only asks one yes / no question of
  // each element; stored as one bit per element (bit idx & 31 of 32-bit word idx >> 5, flat element index), in the
  // zeroed part of the arena.
  std::map<int, PredicateSpec> predicated;
  std::vector<int> random_tensors;   // TensorRandom tensors the live kernels read: refilled on every run
  std::map<int, long> arena_offset;  // result tensor -> float offset in the arena
  long arena_floats = 0;
  long zero_floats = 0;  // leading part of the arena that is zeroed before every run
  std::vector<int> bucket_zero;  // gradient-bucket tensors that need zeroing
  std::set<int> zero_extra;      // result tensors a fold made accumulate (row products): zeroed with the others
  std::set<int> pred_unzeroed;   // predicate tensors whose words are stored whole by their producer: not in the zero prefix
  float* arena = nullptr;
  // The launch sequence of a range (whole call / backward part / update part) is captured into a
  // HIP graph on its second execution and replayed afterwards: the small-batch targets are
  // launch-latency bound (19 kernels for the XOR step), a replay costs one submission.
  std::vector<std::unique_ptr<PlanRowGroup>> row_groups;
  std::vector<std::unique_ptr<PlanSmallGroup>> small_groups;
  std::unique_ptr<PlanSampleGroup> sample_group;  // at most one per plan
  std::vector<std::unique_ptr<PlanEpilogue>> epilogues;
  // generated kernels of this plan waiting for the one hiprtc program make_plan builds at its end
  struct PendingKernel {
    std::string name, source;
    eg_kernel** slot;
  };
  std::vector<PendingKernel> pending;
  // Overlap groups (plan_overlap): launches [first, big) run on the context's side lane while the
  // long contraction `big` runs on the main stream; both are joined before launch big + 1.
  struct Overlap {
    int first = 0, big = 0;
    // a row group in front of the overlap whose partial rows nothing up to and including the contraction needs: its fold
    // (row_finalize) is the side lane's first launch instead of the row kernel's tail on the main lane (-1: none)
    int deferred_row = -1;
  };
  int defer_finalize_of = -1;   // set while a launch sequence is being issued: the row group whose fold the next fork runs
  std::vector<Overlap> overlaps;
  // Batch pipeline: the backward range [0, n_backward) runs as two half batches, the long contractions
  // of both halves back to back on the main lane, everything between them on the side lane — the
  // bandwidth-bound launches of one half under the matrix work of the other (plan_pipeline.cpp).
  struct Pipeline {
    bool active = false;
    long batch = 0, half = 0;
  } pipe;
  struct Captured {
    hipGraphExec_t exec = nullptr;
    std::string key;  // everything baked into the captured kernel arguments
    int runs = 0;
    // The same facts as plain words, compared first: a replay of a launch-bound step (XOR: 14 us of kernels) should not
    // format a string per call.  stamp = eg_model::inputs_gen when `key` was formed (any binding change bumps it).
    uint64_t stamp = ~0ull;
    const void* ptrs[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    float grad_scale = 0;
    long epoch = 0;
  };
  Captured graphs[6];  // 0 whole call, 1 backward part, 2 update part; data-parallel split: 3 head, 4 side lane, 5 tail
  int active_begin = 0, active_end = 0;  // the launch range being issued (run_range_eager): decides MODE of a row group with a tail
  long epoch = 0;      // Model.epoch the plan was made under (part of the key when the program has epoch_in_setup)
};

struct TargetState {
  Target* target = nullptr;
  std::vector<Lowered> lowered;  // parallel to target->live
  std::map<std::string, std::unique_ptr<Plan>> plans;
  Plan* last = nullptr;
  // parameter gradients of this target, laid out back to back (data-parallel exchange bucket)
  std::vector<int> grad_tensors;       // tensor ids (GenGradient destinations of parameters)
  std::map<int, long> bucket_offset;   // tensor id -> float offset
  long bucket_floats = 0;
  float* bucket = nullptr;
  bool bucket_owned = false;
  uint64_t last_stamp = ~0ull;  // eg_model::inputs_gen for which `last` was looked up
  // exchange schedule of the data-parallel step per group, keyed by the group's rank-independent identity
  // (GradExchange::group): host/dp_schedule.hpp ScheduleTable
  eg::dp::ScheduleTable dp_schedules;
  long last_epoch = -1;         // ... and Model.epoch (compared when the program computes host values from epoch())
};

struct BoundInput {
  const float* device = nullptr;
  std::vector<long> shape;
  float* owned = nullptr;  // staging copy of a host input
  long owned_count = 0;
  bool bound = false;
};

}  // namespace model
}  // namespace eg

struct eg_model {
  eg_ctx* ctx = nullptr;
  eg::kd::Program prog;
  std::map<std::string, eg::model::TargetState> targets;
  std::map<int, eg::model::DevTensor> params;  // device-resident parameters (model.params)
  std::map<int, eg::model::BoundInput> inputs;
  // compile[float64] (model.nim:253-260): every tensor of the model holds doubles.  The planner's bookkeeping stays in
  // units of 4 bytes ("floats": arena and bucket offsets, zero ranges, `float*` handles): an element of such a model
  // occupies esz = 2 of them, and the handles are only ever cast (host/run.cpp), never indexed by element.
  bool f64 = false;
  int esz = 1;
  bool keep_values = false;  // eg_model_keep_values: plans keep result values where they would keep predicate bits
  uint64_t inputs_gen = 0;  // bumped by every change of `inputs` (bind, clear, fit): caches keyed on the bindings compare it
  float grad_scale = 1.0f;
  long epoch = 0;
  int kernel_serial = 0;
  std::string source_text;  // the kernel-description text this model was compiled from (eg_model_save)
  std::string plan_text;
  std::string launch_text;
  std::vector<eg::model::Generic*> pending;  // generated kernels not built yet (eg_model_compile builds them together)
  uint64_t* rng_state = nullptr;  // device: {seed, fills drawn so far} for the TensorRandom tensors (eg_fill_uniform)
  std::vector<eg_kernel*> kernels;
  // eg_model_fit: the data set's device copy (one buffer per input, grown on demand), uploaded on its own stream
  std::vector<float*> fit_data;
  std::vector<size_t> fit_bytes;
  hipStream_t copy_stream = nullptr;
  hipEvent_t copy_event = nullptr;
  hipEvent_t main_event = nullptr;  // recorded on the context's stream: "the batches queued so far are done"
  // eg_model_fit, small batches: `group` consecutive batches (segment copy + launch sequence each) captured as ONE graph; a
  // launch re-points the copy nodes at the group's rows.  Between two launches of a captured sequence the device idles for
  // ~8 us: a group pays that once.
  struct FitGraph {
    hipGraphExec_t exec = nullptr;
    // A second executable of the same graph: a group's copy nodes are re-pointed with hipGraphExecKernelNodeSetParams, and
    // HIP does not say that a launch still queued keeps the arguments it was launched with — so the executable being
    // updated is never the one launched last (A / B in turns), and its own previous launch has completed (an event).
    hipGraphExec_t exec2 = nullptr;
    hipEvent_t done[2] = {nullptr, nullptr};
    bool launched[2] = {false, false};
    int turn = 0;
    hipGraph_t graph = nullptr;          // kept: the copy nodes are addressed through it
    std::vector<hipGraphNode_t> copies;  // the group's segment-copy nodes, in batch order
    // Inputs read in place (round 5): when the step's only reader of the inputs is a sample group's kernel, a group's
    // batches are captured WITHOUT their segment copies — the kernel nodes read the batch's rows where they lie, and a
    // launch re-points those nodes' input arguments instead of the copy nodes (one launch less per batch).
    bool direct = false;
    std::vector<hipGraphNode_t> readers;  // direct: the group's sample-kernel nodes, in batch order
    std::string key;                     // everything baked into the captured kernel arguments
  } fit_graph;
};

namespace eg {
namespace model {

inline long prod(const std::vector<long>& s) {
  long p = 1;
  for (long v : s) p *= v;
  return p;
}

inline long align4(long n) { return (n + 3) & ~3L; }

// floats of arena storage a result tensor occupies in this plan
inline long storage_floats(const Plan& plan, int tid) {
  const long n = prod(plan.shapes.at(tid));
  return plan.predicated.count(tid) ? (n + 31) / 32 : n * plan.esz;
}


// While alive, the context's stream and scratch blocks are the side lane's.
struct LaneSwap {
  eg_ctx* c;
  explicit LaneSwap(eg_ctx* ctx) : c(ctx) { swap(); }
  ~LaneSwap() { swap(); }
  void swap() {
    std::swap(c->stream, c->side_stream);
    std::swap(c->workspace, c->side_workspace);
    std::swap(c->workspace_bytes, c->side_workspace_bytes);
    std::swap(c->aux, c->side_aux);
    std::swap(c->aux_bytes, c->side_aux_bytes);
    c->on_side_lane = !c->on_side_lane;
  }
};

// match.cpp
bool bare2(const Op& op, int& r0, int& r1);
int loop_index(const Kernel& k, int reg);
bool match_gemm(const Kernel& k, GemmMatch& m);
bool match_bias(const Kernel& k, int tensor);
bool match_conv(const Kernel& k, ConvMatch& m);
// lower.cpp
int build_generic(eg_model* m, Generic& g);
void inline_producers(eg_model* m, TargetState& ts);
void inline_consumers(eg_model* m, TargetState& ts);
int lower_target(eg_model* m, TargetState& ts);
int build_pending(eg_model* m);
void describe(eg_model* m);
// plan_groups.cpp
int form_row_groups(eg_model* m, TargetState& ts, Plan& plan, const std::vector<KernelInfo>& infos, const std::map<int, int>& first_writer, std::vector<int>& group_of);
constexpr int SAMPLE_GROUP_CODE = -1000000;  // group_of[] entry of a sample group's kernels
int form_sample_group(eg_model* m, TargetState& ts, Plan& plan, const std::vector<KernelInfo>& infos, const std::map<int, int>& first_writer,
                      std::vector<int>& group_of, std::set<int>& needs_zero);
int fuse_row_tails(eg_model* m, TargetState& ts, Plan& plan, const std::vector<KernelInfo>& infos);
bool row_tail_active(const Plan& plan, const Launch& row_launch);
int fuse_slab_fold(eg_model* m, TargetState& ts, Plan& plan, const std::vector<KernelInfo>& infos);
bool slab_fold_active(const Plan& plan, const Launch& sample_launch);
// plan_overlap.cpp
int ensure_side_lane(eg_ctx* ctx);
bool launch_tensors(const Plan& plan, const Launch& L, std::set<int>& reads, std::set<int>& writes);
void plan_overlap(eg_model* m, TargetState& ts, Plan& plan);
// plan_epilogue.cpp
int fuse_epilogues(eg_model* m, TargetState& ts, Plan& plan, const std::vector<KernelInfo>& infos);
// plan.cpp
int build_plan_kernels(eg_model* m, Plan& plan);
float* tensor_ptr(eg_model* m, TargetState& ts, Plan& plan, int tid);
bool full_cover(const Kernel& k, const KernelInfo& info, const std::vector<long>& shape);
void note_vec4(Launch& L, long total);
int fill_params(eg_model* m, const Kernel& k, const KernelInfo& info, const Shapes& shapes, const GenericSource& src, bool accumulate, long total, long rtotal, long chunk, std::vector<long>& out);
std::string shape_key(eg_model* m);
void release_plan(Plan& plan);
bool copy_can_alias(eg_model* m, TargetState& ts, const Kernel& k, const KernelInfo& info, const Shapes& shapes, int p);
int make_plan(eg_model* m, TargetState& ts, Plan& plan);
int get_plan(eg_model* m, const char* target, TargetState** ts_out, Plan** plan_out);
// run.cpp
int ensure_rng(eg_model* m, uint64_t seed = 0x5eed5eed5eed5eedULL, bool reseed = false);
int run_launch(eg_model* m, TargetState& ts, Plan& plan, Launch& L);
// hook: called on the side lane of the overlap group whose contraction is launch `big`, after the group's
// side launches and before the join (the data-parallel exchange of the gradients that are complete by then)
struct SideHook {
  int big = -1;
  int (*fn)(void* user) = nullptr;
  void* user = nullptr;
};
int run_range_eager(eg_model* m, TargetState& ts, Plan& plan, int begin, int end, bool zero, const SideHook* hook = nullptr);
// Which parts of the gradient bucket are complete before the last long contraction of the backward
// range finishes (early: exchanged on the side lane, under it) and which only after it (late).
// big < 0: no such contraction, `late` is the whole bucket.  Segments are (float offset, float count).
struct ExchangePlan {
  int big = -1;
  std::vector<std::pair<long, long>> early, late;
};
int plan_exchange(eg_model* m, TargetState& ts, Plan& plan, ExchangePlan& ex);
bool graphs_enabled();
int run_range(eg_model* m, TargetState& ts, Plan& plan, int begin, int end, bool zero, int slot);
// everything a captured launch sequence bakes into its kernel arguments (input bindings, workspaces, bucket, seed scale, epoch)
std::string capture_key(eg_model* m, TargetState& ts);
// plan_epilogue.cpp: a bias gradient `gb[x] ++= g[y,x]` next to the weight gradient `gW[it,x] ++= a[y,it] * g[y,x]`
// of the same layer becomes the last row of that contraction (a virtual row of ones in A) when gb lies
// directly behind gW in the gradient bucket; the column-sum launch disappears.
int fold_bias_gradients(eg_model* m, TargetState& ts, Plan& plan, const std::vector<KernelInfo>& infos);
int fold_row_products(eg_model* m, TargetState& ts, Plan& plan);
// plan_pipeline.cpp
void plan_pipeline(eg_model* m, TargetState& ts, Plan& plan);
// One half of a sliced launch: rows [row0, row0 + rows) of the batch; second = the later half (reductions accumulate).
struct Slice {
  long row0 = 0, rows = 0;
  bool second = false;
};
int run_launch_sliced(eg_model* m, TargetState& ts, Plan& plan, Launch& L, const Slice& sl);
int run_pipelined(eg_model* m, TargetState& ts, Plan& plan, const SideHook* hook);
// plan_check.cpp: invariants of a finished plan (read / write sets against the kernel list, arena
// layout, overlap groups).  Returns EG_OK or EG_ERR_RUNTIME with the violated invariant named.
int check_plan(eg_model* m, TargetState& ts, Plan& plan);

}  // namespace model
}  // namespace eg

// ---- helpers implemented next to the group-1 / group-2 code ------------------------------------
// No C++ exception may cross the C ABI (std::bad_alloc, a failed .at() on a malformed program):
// every entry point that touches the STL turns it into a status + eg_last_error().
#define EG_CATCH_ALL                                           \
  catch (const std::exception& e) {                            \
    eg::set_error("internal error: %s", e.what());             \
    return EG_ERR_RUNTIME;                                     \
  }                                                            \
  catch (...) {                                                \
    eg::set_error("internal error");                           \
    return EG_ERR_RUNTIME;                                     \
  }
