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
#include <cstdlib>
#include <cstring>
#include <memory>
#include <random>
#include <set>
#include <sstream>

#include "../eg_internal.hpp"
#include "../kernels/gemm_fused.hpp"
#include "codegen.hpp"
#include "epilogue.hpp"
#include "kd.hpp"
#include "rowfuse.hpp"

using namespace eg::kd;
using eg::set_error;

namespace {

struct GemmMatch {
  int a_read = 0, b_read = 0;  // indices into k.reads
  bool trans_a = false, trans_b = false;
  int li = 0, lj = 0, lk = 0;  // loop indices of m, n, k
};

bool bare2(const Op& op, int& r0, int& r1) {
  if (op.raw || op.dims.size() != 2) return false;
  r0 = op.dims[0].only_register();
  r1 = op.dims[1].only_register();
  return r0 && r1 && r0 != r1;
}

int loop_index(const Kernel& k, int reg) {
  for (size_t l = 0; l < k.loops.size(); ++l)
    if (k.loops[l].reg == reg) return (int)l;
  return -1;
}

// c[i,j] += a(i,k) * b(k,j)      base.nim:27-28 and its two derived forms (passes.nim:519-549)
bool match_gemm(const Kernel& k, GemmMatch& m) {
  if (!k.index_instrs.empty()) return false;
  if (k.instrs.size() != 1 || k.instrs[0].kind != IK::Mul || k.result != k.instrs[0].res) return false;
  if (k.reads.size() != 2 || k.loops.size() != 3 || !k.setup.empty()) return false;
  for (auto& lp : k.loops)
    if (lp.has_bounds) return false;
  const std::vector<int>& args = k.instrs[0].args;
  if (!((args[0] == k.reads[0].reg && args[1] == k.reads[1].reg) || (args[0] == k.reads[1].reg && args[1] == k.reads[0].reg)))
    return false;
  int wi, wj;
  if (!bare2(k.write, wi, wj)) return false;
  int kk = 0;
  for (auto& lp : k.loops)
    if (lp.reg != wi && lp.reg != wj) kk = lp.reg;
  if (!kk || loop_index(k, wi) < 0 || loop_index(k, wj) < 0) return false;
  int r[2][2];
  if (!bare2(k.reads[0], r[0][0], r[0][1]) || !bare2(k.reads[1], r[1][0], r[1][1])) return false;
  auto is = [](const int* p, int x, int y) { return (p[0] == x && p[1] == y) || (p[0] == y && p[1] == x); };
  for (int a = 0; a < 2; ++a) {
    const int b = 1 - a;
    if (is(r[a], wi, kk) && is(r[b], kk, wj)) {
      m.a_read = a;
      m.b_read = b;
      m.trans_a = r[a][0] == kk;
      m.trans_b = r[b][0] == wj;
      m.li = loop_index(k, wi);
      m.lj = loop_index(k, wj);
      m.lk = loop_index(k, kk);
      return true;
    }
  }
  return false;
}

// out[y,x] += bias[x] on the tensor the preceding contraction wrote   dnn.nim:22-24
bool match_bias(const Kernel& k, int tensor) {
  if (!k.index_instrs.empty()) return false;
  if (!k.instrs.empty() || k.reads.size() != 1 || k.result != k.reads[0].reg || k.write.tensor != tensor) return false;
  if (k.loops.size() != 2 || !k.setup.empty()) return false;
  for (auto& lp : k.loops)
    if (lp.has_bounds) return false;
  int wi, wj;
  if (!bare2(k.write, wi, wj)) return false;
  const Op& b = k.reads[0];
  return !b.raw && b.dims.size() == 1 && b.dims[0].only_register() == wj;
}

struct ConvMatch {
  // which operand plays which part: -1 = the written tensor, 0 / 1 = k.reads[i]
  int out_op = -1, img_op = 0, flt_op = 1;
  bool batched = true;
  enum Role { Forward, GradImage, GradFilter } role = Forward;
};

// out[n,y,x,f] += img[n,y+dy,x+dx,c] * flt[f,dy,dx,c]   dnn.nim:45-49 (4-D) / conv2.nim:128-132 (3-D)
// and the two kernels derive (passes.nim:383-549) makes of it, which are the same loop nest with
// another of the three tensors written:
//   gimg[n,y+dy,x+dx,c] += gout[n,y,x,f] * flt[f,dy,dx,c]      gflt[f,dy,dx,c] += gout[n,y,x,f] * img[n,y+dy,x+dx,c]
bool match_conv(const Kernel& k, ConvMatch& m) {
  if (!k.index_instrs.empty()) return false;
  if (k.instrs.size() != 1 || k.instrs[0].kind != IK::Mul || k.result != k.instrs[0].res || k.reads.size() != 2) return false;
  if (!k.setup.empty() || k.write.raw || k.reads[0].raw || k.reads[1].raw) return false;
  const std::vector<int>& args = k.instrs[0].args;
  if (!((args[0] == k.reads[0].reg && args[1] == k.reads[1].reg) || (args[0] == k.reads[1].reg && args[1] == k.reads[0].reg)))
    return false;
  for (auto& lp : k.loops)
    if (lp.has_bounds) return false;
  const Op* ops[3] = {&k.write, &k.reads[0], &k.reads[1]};  // operand index + 1
  static const int perms[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};
  for (auto& pm : perms) {
    const Op& out = *ops[pm[0]];
    const Op& img = *ops[pm[1]];
    const Op& flt = *ops[pm[2]];
    const size_t nd = out.dims.size();
    if ((nd != 4 && nd != 3) || img.dims.size() != nd || flt.dims.size() != 4) continue;
    const bool batched = nd == 4;
    if (k.loops.size() != (batched ? 7u : 6u)) continue;
    std::vector<int> w;
    for (auto& d : out.dims) w.push_back(d.only_register());
    const int off = batched ? 1 : 0;
    const int n = batched ? w[0] : 0, y = w[off], x = w[off + 1], f = w[off + 2];
    if (!y || !x || !f || (batched && !n)) continue;
    const int ff = flt.dims[0].only_register(), dy = flt.dims[1].only_register(), dx = flt.dims[2].only_register(),
              c = flt.dims[3].only_register();
    if (ff != f || !dy || !dx || !c) continue;
    std::set<int> all = {y, x, f, dy, dx, c};
    if (batched) all.insert(n);
    if (all.size() != (batched ? 7u : 6u)) continue;
    auto pair_sum = [](const Lin& l, int p, int q) {
      return l.constant == 0 && l.factors.size() == 2 && l.factor_of(p) == 1 && l.factor_of(q) == 1;
    };
    if (batched && img.dims[0].only_register() != n) continue;
    if (!pair_sum(img.dims[off], y, dy) || !pair_sum(img.dims[off + 1], x, dx) || img.dims[off + 2].only_register() != c)
      continue;
    m.out_op = pm[0] - 1;
    m.img_op = pm[1] - 1;
    m.flt_op = pm[2] - 1;
    m.batched = batched;
    m.role = pm[0] == 0 ? ConvMatch::Forward : (pm[1] == 0 ? ConvMatch::GradImage : ConvMatch::GradFilter);
    return true;
  }
  return false;
}

enum class StepKind { Gemm, Conv, ConvGradImage, ConvGradFilter, Seed, GenericA, GenericB, RowFused, SmallFused, GemmFused };

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
  int consumer = -1;                        // GenericA: live position of the elementwise consumer folded into it
  int vec_slot = -1;                        // GenericA: index of the Slot::Vec4 argument, if the kernel has one
  bool vec_ok = false;                      //   the shapes allow four elements per thread (pointers are checked per launch)
  long total_items = 0;                     //   independent iterations (grid = items or items / 4)
};

// A run of per-sample kernels fused into one generated kernel (rowfuse.hpp), built per plan.
struct PlanRowGroup {
  RowGroup g;
  eg_kernel* handle = nullptr;
  float* partial = nullptr;  // [nblocks][g.red_total]
  int nblocks = 0;
  std::vector<int> red_tensors;  // reduction destinations, in segment order
};

// A run of small-tensor kernels (optimizer updates) fused into one single-block kernel.
struct PlanSmallGroup {
  SmallGroup g;
  eg_kernel* handle = nullptr;
};

// A contraction whose elementwise consumer runs as its epilogue (epilogue.hpp), built per plan.
struct PlanEpilogue {
  EpilogueSpec spec;
  Launch consumer;                          // the consumer as its own launch (split-K fallback)
  std::map<std::string, eg_kernel*> built;  // by template variant (tile shape, alignment class)
};

struct DevTensor {
  float* ptr = nullptr;
  long count = 0;
  std::vector<long> shape;
};

struct Plan {
  std::string key;
  Shapes shapes;
  std::vector<Launch> launches;
  int n_backward = 0;  // launches before the first parameter update
  // Result tensors that are a whole-tensor raw copy of another tensor (reshape, passes.nim:643-688,
  // and the gradient of one) share its storage instead of being copied: dest -> source.
  std::map<int, int> alias;
  std::vector<int> random_tensors;   // TensorRandom tensors the live kernels read: refilled on every run
  std::map<int, long> arena_offset;  // result tensor -> float offset in the arena
  long arena_floats = 0;
  long zero_floats = 0;  // leading part of the arena that is zeroed before every run
  std::vector<int> bucket_zero;  // gradient-bucket tensors that need zeroing
  float* arena = nullptr;
  // The launch sequence of a range (whole call / backward part / update part) is captured into a
  // HIP graph on its second execution and replayed afterwards: the small-batch targets are
  // launch-latency bound (19 kernels for the XOR step), a replay costs one submission.
  std::vector<std::unique_ptr<PlanRowGroup>> row_groups;
  std::vector<std::unique_ptr<PlanSmallGroup>> small_groups;
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
  };
  std::vector<Overlap> overlaps;
  struct Captured {
    hipGraphExec_t exec = nullptr;
    std::string key;  // everything baked into the captured kernel arguments
    int runs = 0;
  };
  Captured graphs[3];
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
};

struct BoundInput {
  const float* device = nullptr;
  std::vector<long> shape;
  float* owned = nullptr;  // staging copy of a host input
  long owned_count = 0;
  bool bound = false;
};

}  // namespace

struct eg_model {
  eg_ctx* ctx = nullptr;
  Program prog;
  std::map<std::string, TargetState> targets;
  std::map<int, DevTensor> params;  // device-resident parameters (model.params)
  std::map<int, BoundInput> inputs;
  float grad_scale = 1.0f;
  long epoch = 0;
  int kernel_serial = 0;
  std::string plan_text;
  std::string launch_text;
  std::vector<Generic*> pending;  // generated kernels not built yet (eg_model_compile builds them together)
  uint64_t* rng_state = nullptr;  // device: {seed, fills drawn so far} for the TensorRandom tensors (eg_fill_uniform)
  std::vector<eg_kernel*> kernels;
  // eg_model_fit: the data set's device copy (one buffer per input, grown on demand), uploaded on its own stream
  std::vector<float*> fit_data;
  std::vector<size_t> fit_bytes;
  hipStream_t copy_stream = nullptr;
  hipEvent_t copy_event = nullptr;
};

namespace {

long prod(const std::vector<long>& s) {
  long p = 1;
  for (long v : s) p *= v;
  return p;
}

long align4(long n) { return (n + 3) & ~3L; }

int build_generic(eg_model* m, Generic& g) {
  int rc = eg_kernel_compile(m->ctx, g.src.name.c_str(), g.src.source.c_str(), &g.handle);
  if (rc) {
    std::string msg = eg_last_error();
    set_error("%s\n--- generated source ---\n%s", msg.c_str(), g.src.source.c_str());
    return rc;
  }
  m->kernels.push_back(g.handle);
  return EG_OK;
}

// Producer inlining.  A unary elementwise kernel `T{it} ++= f(S{it})` whose result is read only by
// generated kernels (not by a contraction / convolution the library runs) is recomputed inside
// those consumers: every read `T[index]` becomes `f(S[index])`, the producer is never launched and T
// never touches memory (conv2 -> leakyRelu -> maxpool2: the activation disappears into the pooling
// kernel and into maxpool2's hand-written gradient).  Same operations in the same order per element:
// results are bit-identical.  The reference's CPU target gets a similar effect from fuseLoops
// (passes.nim:1929-2004); its GPU target launches every kernel.
void inline_producers(eg_model* m, TargetState& ts) {
  static const bool off = [] {
    const char* e = getenv("EG_NO_INLINE");
    return e && e[0] && e[0] != '0';
  }();
  if (off) return;
  Target& t = *ts.target;
  const Program& prog = m->prog;
  auto is_library = [&](const Kernel& k) {
    GemmMatch g;
    ConvMatch c;
    return k.is_seed || match_gemm(k, g) || match_conv(k, c);
  };
  for (size_t p = 0; p < t.live.size(); ++p) {
    if (ts.lowered[p].absorbed || (int)p == t.first_update) continue;
    const Kernel P = t.all[t.live[p]];  // copy: the consumers below are edited in place
    // ---- a pure unary map over whole tensors?
    if (P.loops.size() != 1 || !P.index_instrs.empty() || !P.setup.empty() || P.is_seed || P.instrs.size() > 32) continue;
    if (P.reads.empty() || !P.write.raw || P.write.dims.size() != 1) continue;
    const int it = P.loops[0].reg;
    if (P.loops[0].has_bounds || P.write.dims[0].only_register() != it) continue;
    const int T = P.write.tensor, S = P.reads[0].tensor;
    bool pure = prog.tensors[T].kind == TK::Result && T != S && T != t.output && !ts.bucket_offset.count(T);
    for (auto& rd : P.reads)
      if (rd.tensor != S || !rd.raw || rd.dims.size() != 1 || rd.dims[0].only_register() != it) pure = false;
    for (auto& ins : P.instrs) {
      if (ins.kind == IK::Shape || ins.kind == IK::Len || ins.kind == IK::ShapeLen || ins.kind == IK::Epoch) pure = false;
      for (int a : ins.args)
        if (a == it) pure = false;  // the value depends on the position
    }
    if (P.result == it) pure = false;
    // T must be shaped like S for an index into T to address the same element of S
    auto sc = prog.shape_copy.find(T);
    if (prog.shape_dims.count(T) || (sc != prog.shape_copy.end() ? sc->second != S : P.reads.size() != 1)) pure = false;
    if (!pure) continue;
    // ---- every other kernel: nobody else writes T, S is final, all readers of T are generated kernels
    std::vector<size_t> consumers;
    bool ok = true;
    for (size_t q = 0; q < t.live.size() && ok; ++q) {
      if (q == p) continue;
      const Kernel& K = t.all[t.live[q]];
      if (K.write.tensor == T) ok = false;
      if (q > p && K.write.tensor == S) ok = false;
      bool reads_t = false;
      for (auto& rd : K.reads) reads_t = reads_t || rd.tensor == T;
      if (!reads_t) continue;
      if (q < p || ts.lowered[q].absorbed || is_library(K) || K.instrs.size() + P.instrs.size() * K.reads.size() > 96) ok = false;
      if ((t.first_update >= 0) && ((int)p < t.first_update) != ((int)q < t.first_update)) ok = false;  // stay on one side
      consumers.push_back(q);
    }
    if (!ok || consumers.empty()) continue;
    // ---- rewrite the consumers
    for (size_t q : consumers) {
      Kernel& K = t.all[t.live[q]];
      std::vector<Op> reads;
      std::vector<Instr> prefix;
      for (auto& rd : K.reads) {
        if (rd.tensor != T) {
          reads.push_back(rd);
          continue;
        }
        // one load of S at the same index, then P's instructions with fresh registers
        std::map<int, int> rename;
        Op load = rd;
        load.tensor = S;
        load.reg = K.alloc();
        for (auto& pr : P.reads) rename[pr.reg] = load.reg;
        reads.push_back(load);
        for (auto& ins : P.instrs) {
          Instr c = ins;
          c.res = K.alloc();
          rename[ins.res] = c.res;
          for (int& a : c.args) a = rename.count(a) ? rename[a] : a;
          prefix.push_back(c);
        }
        // the old data register of the T read now names the recomputed value (0 + f, as P stored it)
        Instr zero, sum;
        zero.kind = IK::Scalar;
        zero.lit = 0.0;
        zero.res = K.alloc();
        sum.kind = IK::Add;
        sum.args = {zero.res, rename.count(P.result) ? rename[P.result] : P.result};
        sum.res = rd.reg;
        prefix.push_back(zero);
        prefix.push_back(sum);
      }
      K.reads.swap(reads);
      K.instrs.insert(K.instrs.begin(), prefix.begin(), prefix.end());
    }
    ts.lowered[p].absorbed = true;
    ts.lowered[p].inlined = true;
    ts.lowered[p].all_index = t.live[p];
  }
}

// Consumer inlining, the mirror image of inline_producers.  A generated kernel P without a reduction
// (every iteration writes its own element: maxpool2's hand-written gradient, upsample2, a binary
// map) followed directly by an elementwise kernel `U{it} ++= g(T{it}, V{it}...)` that is the only
// reader of P's result T: P stores g(value, V...) into U right away and T never exists.  The
// combined kernel is generated here; a plan uses it only when P covers T completely and T, U and
// the V have one shape (make_plan), otherwise both kernels run as they are.
void inline_consumers(eg_model* m, TargetState& ts) {
  static const bool off = [] {
    const char* e = getenv("EG_NO_INLINE");
    return e && e[0] && e[0] != '0';
  }();
  if (off) return;
  Target& t = *ts.target;
  const Program& prog = m->prog;
  const int n = (int)t.live.size();
  for (int p = 0; p < n; ++p) {
    Lowered& lo = ts.lowered[p];
    if (lo.absorbed || lo.kind != StepKind::GenericA) continue;
    int q = p + 1;
    while (q < n && ts.lowered[q].absorbed) ++q;
    if (q >= n || ts.lowered[q].kind != StepKind::GenericA || q == t.first_update) continue;
    if (t.first_update >= 0 && (p < t.first_update) != (q < t.first_update)) continue;
    const Kernel& P = t.all[t.live[p]];
    const Kernel& C = t.all[t.live[q]];
    // ---- P: one element per iteration
    std::vector<int> indep, red;
    bool scatter = false;
    split_loops(P, indep, red, scatter);
    if (!red.empty() || scatter || P.is_seed || P.gen != Gen::None) continue;
    const int T = P.write.tensor, U = C.write.tensor;
    if (prog.tensors[T].kind != TK::Result || T == t.output || ts.bucket_offset.count(T) || T == U) continue;
    // ---- C: a map over whole tensors that reads T
    if (C.loops.size() != 1 || !C.index_instrs.empty() || !C.setup.empty() || C.is_seed || C.gen != Gen::None) continue;
    if (C.loops[0].has_bounds || C.instrs.size() + P.instrs.size() > 96) continue;
    const int it = C.loops[0].reg;
    bool ok = C.write.raw && C.write.dims.size() == 1 && C.write.dims[0].only_register() == it && C.result != it;
    bool reads_t = false;
    for (auto& rd : C.reads) {
      if (!rd.raw || rd.dims.size() != 1 || rd.dims[0].only_register() != it || rd.tensor == U) ok = false;
      reads_t = reads_t || rd.tensor == T;
    }
    for (auto& ins : C.instrs) {
      if (ins.kind == IK::Shape || ins.kind == IK::Len || ins.kind == IK::ShapeLen || ins.kind == IK::Epoch) ok = false;
      for (int a : ins.args)
        if (a == it) ok = false;
    }
    for (auto& rd : P.reads)
      if (rd.tensor == U) ok = false;
    if (!ok || !reads_t) continue;
    // ---- nobody else writes or reads T
    for (int s2 = 0; s2 < n && ok; ++s2) {
      if (s2 == p || s2 == q) continue;
      const Kernel& K = t.all[t.live[s2]];
      if (K.write.tensor == T) ok = false;
      for (auto& rd : K.reads)
        if (rd.tensor == T) ok = false;
    }
    if (!ok) continue;
    // ---- P + C
    std::unique_ptr<Kernel> F(new Kernel(P));
    std::map<int, int> rename;
    Instr zero, value;  // what C would have loaded: 0 + P's value, as P stored it
    zero.kind = IK::Scalar;
    zero.lit = 0.0;
    zero.res = F->alloc();
    value.kind = IK::Add;
    value.args = {zero.res, P.result};
    value.res = F->alloc();
    F->instrs.push_back(zero);
    F->instrs.push_back(value);
    for (auto& rd : C.reads) {
      if (rd.tensor == T) {
        rename[rd.reg] = value.res;
        continue;
      }
      Op op = P.write;  // same element as the one P writes
      op.tensor = rd.tensor;
      op.reg = F->alloc();
      rename[rd.reg] = op.reg;
      F->reads.push_back(op);
    }
    for (auto& ins : C.instrs) {
      Instr c = ins;
      c.res = F->alloc();
      rename[ins.res] = c.res;
      for (int& a : c.args) a = rename.count(a) ? rename[a] : a;
      F->instrs.push_back(c);
    }
    F->result = rename.count(C.result) ? rename[C.result] : C.result;
    F->write.tensor = U;
    char name[64];
    snprintf(name, sizeof(name), "eg_k%d_ac", m->kernel_serial++);
    if (generate_mode_a(*F, name, lo.with_consumer_code.src) != EG_OK) {
      eg::clear_error();
      continue;
    }
    lo.consumer = q;
    lo.with_consumer = std::move(F);
    m->pending.push_back(&lo.with_consumer_code);
  }
}

int lower_target(eg_model* m, TargetState& ts) {
  Target& t = *ts.target;
  ts.lowered.clear();
  ts.lowered.resize(t.live.size());
  inline_producers(m, ts);
  for (size_t p = 0; p < t.live.size(); ++p) {
    Lowered& lo = ts.lowered[p];
    lo.all_index = t.live[p];
    const Kernel& k = t.all[lo.all_index];
    if (lo.absorbed) continue;
    if (k.is_seed) {
      lo.kind = StepKind::Seed;
      continue;
    }
    if (match_gemm(k, lo.gemm)) {
      lo.kind = StepKind::Gemm;
      // dense = contraction followed by the bias kernel on the same tensor (dnn.nim:21-24):
      // fold it into the epilogue.  Not across the backward/update boundary.
      if (p + 1 < t.live.size() && (int)(p + 1) != t.first_update) {
        const Kernel& nk = t.all[t.live[p + 1]];
        if (match_bias(nk, k.write.tensor) && nk.write.dims[1].only_register() &&
            k.write.dims.size() == 2) {
          lo.bias_tensor = nk.reads[0].tensor;
          ts.lowered[p + 1].absorbed = true;
          ts.lowered[p + 1].all_index = t.live[p + 1];
        }
      }
      continue;
    }
    if (match_conv(k, lo.conv)) {
      lo.kind = lo.conv.role == ConvMatch::Forward     ? StepKind::Conv
                : lo.conv.role == ConvMatch::GradImage ? StepKind::ConvGradImage
                                                       : StepKind::ConvGradFilter;
      continue;
    }
    lo.kind = StepKind::GenericA;
    lo.b_capable = split_reduction_capable(k);
    char name[64];
    snprintf(name, sizeof(name), "eg_k%d_a", m->kernel_serial++);
    int rc = generate_mode_a(k, name, lo.mode_a.src);
    if (rc) return rc;
    m->pending.push_back(&lo.mode_a);  // built together with the model's other generated kernels
  }
  inline_consumers(m, ts);
  return EG_OK;
}

// Device-resident generator state of the model's TensorRandom tensors: {seed, fills drawn so far}.
int ensure_rng(eg_model* m, uint64_t seed = 0x5eed5eed5eed5eedULL, bool reseed = false) {
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

// The kernels generated for one plan (row / small / map groups, split reductions): one program.
int build_plan_kernels(eg_model* m, Plan& plan) {
  if (plan.pending.empty()) return EG_OK;
  std::string source;
  std::vector<std::string> names;
  for (auto& pk : plan.pending) {
    if (*pk.slot) continue;  // the same split-reduction kernel requested twice
    bool dup = false;
    for (auto& n : names) dup = dup || n == pk.name;
    if (dup) continue;
    source += pk.source + "\n";
    names.push_back(pk.name);
  }
  std::vector<eg_kernel*> built;
  int rc = names.empty() ? EG_OK : eg::kernels_compile_batch(m->ctx, "eg_plan_kernels", source.c_str(), names, built);
  if (rc) {  // name the culprit
    eg::clear_error();
    for (auto& pk : plan.pending) {
      if (*pk.slot) continue;
      rc = eg_kernel_compile(m->ctx, pk.name.c_str(), pk.source.c_str(), pk.slot);
      if (rc) {
        std::string msg = eg_last_error();
        set_error("%s\n--- generated source ---\n%s", msg.c_str(), pk.source.c_str());
        return rc;
      }
      m->kernels.push_back(*pk.slot);
    }
  } else {
    for (size_t i = 0; i < built.size(); ++i) {
      m->kernels.push_back(built[i]);
      for (auto& pk : plan.pending)
        if (pk.name == names[i]) *pk.slot = built[i];
    }
  }
  plan.pending.clear();
  return EG_OK;
}

// All template-A kernels of the model in one hiprtc program (seconds -> fractions of a second for a
// 40-kernel network); if the program fails to build, build one by one to name the culprit.
int build_pending(eg_model* m) {
  if (m->pending.empty()) return EG_OK;
  std::string source;
  std::vector<std::string> names;
  for (Generic* g : m->pending) {
    source += g->src.source + "\n";
    names.push_back(g->src.name);
  }
  std::vector<eg_kernel*> built;
  int rc = eg::kernels_compile_batch(m->ctx, "eg_model_kernels", source.c_str(), names, built);
  if (rc) {
    eg::clear_error();
    for (Generic* g : m->pending) {
      rc = build_generic(m, *g);
      if (rc) return rc;
    }
  } else {
    for (size_t i = 0; i < built.size(); ++i) {
      m->pending[i]->handle = built[i];
      m->kernels.push_back(built[i]);
    }
  }
  m->pending.clear();
  return EG_OK;
}

float* tensor_ptr(eg_model* m, TargetState& ts, Plan& plan, int tid) {
  for (auto al = plan.alias.find(tid); al != plan.alias.end(); al = plan.alias.find(tid)) tid = al->second;
  const TensorDef& d = m->prog.tensors[tid];
  if (d.kind == TK::Param || d.kind == TK::Cache) return m->params[tid].ptr;
  if (d.kind == TK::Input) {
    auto it = m->inputs.find(tid);
    return (it == m->inputs.end() || !it->second.bound) ? nullptr : const_cast<float*>(it->second.device);
  }
  auto b = ts.bucket_offset.find(tid);
  if (b != ts.bucket_offset.end()) return ts.bucket + b->second;
  auto a = plan.arena_offset.find(tid);
  if (a != plan.arena_offset.end()) return plan.arena + a->second;
  return nullptr;
}

// write covers the whole tensor with plain stores?
bool full_cover(const Kernel& k, const KernelInfo& info, const std::vector<long>& shape) {
  std::set<int> seen;
  if (k.write.raw) {
    if (k.write.dims.size() != 1) return false;
    const int r = k.write.dims[0].only_register();
    const int l = r ? loop_index(k, r) : -1;
    return l >= 0 && info.bounds[l].first == 0 && info.bounds[l].second == prod(shape);
  }
  if (k.write.dims.size() != shape.size()) return false;
  for (size_t d = 0; d < shape.size(); ++d) {
    const Lin& lin = k.write.dims[d];
    const int r = lin.only_register();
    if (r) {
      const int l = loop_index(k, r);
      if (l < 0 || seen.count(r) || info.bounds[l].first != 0 || info.bounds[l].second != shape[d]) return false;
      seen.insert(r);
    } else if (!(lin.factors.empty() && lin.constant == 0 && shape[d] == 1)) {
      return false;
    }
  }
  return true;
}

// Remember where a generated kernel takes its four-elements-per-thread flag and what fill_params decided.
void note_vec4(Launch& L, long total) {
  L.total_items = total;
  L.vec_slot = -1;
  for (size_t i = 0; i < L.generic->src.slots.size(); ++i)
    if (L.generic->src.slots[i].kind == Slot::Vec4) L.vec_slot = (int)i;
  L.vec_ok = L.vec_slot >= 0 && L.params[L.vec_slot] != 0;
}

int fill_params(eg_model* m, const Kernel& k, const KernelInfo& info, const Shapes& shapes, const GenericSource& src,
                bool accumulate, long total, long rtotal, long chunk, std::vector<long>& out) {
  out.clear();
  for (const Slot& s : src.slots) {
    long v = 0;
    switch (s.kind) {
      case Slot::Accumulate: v = accumulate ? 1 : 0; break;
      case Slot::Total: v = total; break;
      case Slot::RTotal: v = rtotal; break;
      case Slot::Chunk: v = chunk; break;
      case Slot::LoopStart: v = info.bounds[s.a].first; break;
      case Slot::LoopExtent: v = info.bounds[s.a].second - info.bounds[s.a].first; break;
      case Slot::Stride: {
        const Op& op = s.a < (int)k.reads.size() ? k.reads[s.a] : k.write;
        const std::vector<long>& shp = shapes.at(op.tensor);
        long stride = 1;
        for (int d = (int)shp.size() - 1; d > s.b; --d) stride *= shp[d];
        v = stride;
        break;
      }
      case Slot::SetupVal: v = info.vals.at(k.setup[s.a].res); break;
      case Slot::Vec4: {
        static const bool off = getenv("EG_NO_VEC4") != nullptr;
        const int l = src.indep.empty() ? -1 : src.indep.back();
        v = !off && l >= 0 && info.bounds[l].first == 0 && info.bounds[l].second % 4 == 0 && total % 4 == 0 && total > 0;
        auto rows_of_four = [&](int tensor) {
          auto sh = shapes.find(tensor);
          return sh != shapes.end() && !sh->second.empty() && sh->second.back() % 4 == 0;
        };
        for (auto& rd : k.reads) v = v && rows_of_four(rd.tensor);
        v = v && rows_of_four(k.write.tensor);
        break;
      }
      case Slot::Narrow: {
        static const bool off = getenv("EG_NO_NARROW_INDEX") != nullptr;
        const long lim = 1L << 31;
        v = !off && total < lim && rtotal < lim;
        auto small = [&](int tensor) {
          auto sh = shapes.find(tensor);
          return sh != shapes.end() && prod(sh->second) < lim;
        };
        for (auto& rd : k.reads) v = v && small(rd.tensor);
        v = v && small(k.write.tensor);
        for (auto& b : info.bounds) v = v && b.first > -lim && b.second < lim;
        break;
      }
      case Slot::InstrVal: {
        const Instr& ins = k.instrs[s.a];
        if (ins.kind == IK::Epoch) {
          v = m->epoch;
        } else {
          const std::vector<long>& shp = shapes.at(ins.tensor);
          if (ins.kind == IK::Len) v = prod(shp);
          else if (ins.kind == IK::ShapeLen) v = (long)shp.size();
          else {
            int d = ins.dim < 0 ? ins.dim + (int)shp.size() : ins.dim;
            if (d < 0 || d >= (int)shp.size()) {
              set_error("shape()[%d] out of range for a rank-%zu tensor", ins.dim, shp.size());
              return EG_ERR_SHAPE;
            }
            v = shp[d];
          }
        }
        break;
      }
      default: break;
    }
    out.push_back(v);
  }
  // 32-bit copies of the arguments must be exact too
  for (size_t i = 0; i < src.slots.size(); ++i)
    if (src.slots[i].kind == Slot::Narrow)
      for (long v : out)
        if (v >= (1L << 31) || v < -(1L << 31)) out[i] = 0;
  return EG_OK;
}

std::string shape_key(eg_model* m) {
  std::ostringstream os;
  for (auto& in : m->inputs) {
    if (!in.second.bound) continue;
    os << in.first << ":";
    for (long s : in.second.shape) os << s << ",";
    os << ";";
  }
  os << "e" << 0;
  return os.str();
}

bool row_fusion_enabled() {
  static const bool on = [] {
    const char* e = getenv("EG_NO_ROWFUSE");
    return !(e && e[0] && e[0] != '0');
  }();
  return on;
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
    int rc = generate_row_group(m->prog, t.all, infos, shapes, g);
    if (rc) return rc;
    plan.pending.push_back({g.name, g.source, &pg->handle});
    pg->nblocks = (int)((B + 255) / 256);
    if (g.red_total > 0) {
      EG_HIP_CHECK(hipSetDevice(m->ctx->device));
      EG_HIP_CHECK(hipMalloc((void**)&pg->partial, (size_t)pg->nblocks * g.red_total * sizeof(float)));
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

// ---- side lane --------------------------------------------------------------------------------
int ensure_side_lane(eg_ctx* ctx) {
  if (ctx->side_stream) return EG_OK;
  EG_HIP_CHECK(hipSetDevice(ctx->device));
  EG_HIP_CHECK(hipStreamCreateWithFlags(&ctx->side_stream, hipStreamNonBlocking));
  EG_HIP_CHECK(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
  EG_HIP_CHECK(hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
  return EG_OK;
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
  }
};

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
    const char* e = getenv("EG_NO_OVERLAP");
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
    static const bool debug = getenv("EG_DEBUG_OVERLAP") != nullptr;
    if (debug) fprintf(stderr, "[eg] overlap: contraction %d (%.1f GFLOP) takes launches [%d, %d)\n", j, flops / 1e9, first, j);
    if (first < j && ensure_side_lane(m->ctx) == EG_OK) plan.overlaps.push_back({first, j});
  }
}

// Is live kernel p a whole-tensor raw copy `dst{it} ++= src{it}` whose destination can simply share
// the source's storage?  (reshape and its gradient.)  Requires: dst is written by this kernel only,
// src is complete by then (no later writer), same element count, dst not in the gradient bucket.
bool copy_can_alias(eg_model* m, TargetState& ts, const Kernel& k, const KernelInfo& info, const Shapes& shapes, int p) {
  static const bool off = [] {
    const char* e = getenv("EG_NO_ALIAS");
    return e && e[0] && e[0] != '0';
  }();
  if (off || !info.ok) return false;
  if (k.reads.size() != 1 || !k.instrs.empty() || !k.index_instrs.empty() || k.loops.size() != 1) return false;
  const Op& rd = k.reads[0];
  if (k.result != rd.reg || !k.write.raw || !rd.raw || k.write.dims.size() != 1 || rd.dims.size() != 1) return false;
  const int it = k.loops[0].reg;
  if (k.write.dims[0].only_register() != it || rd.dims[0].only_register() != it) return false;
  const int dst = k.write.tensor, src = rd.tensor;
  if (dst == src || ts.bucket_offset.count(dst) || m->prog.tensors[dst].kind != TK::Result) return false;
  const long n = prod(shapes.at(dst));
  if (info.bounds[0].first != 0 || info.bounds[0].second != n || prod(shapes.at(src)) != n) return false;
  const Target& t = *ts.target;
  for (size_t q = 0; q < t.live.size(); ++q) {
    const Kernel& o = t.all[t.live[q]];
    if ((int)q != p && o.write.tensor == dst) return false;   // another contribution to dst
    if ((int)q > p && o.write.tensor == src) return false;    // src still changes after the copy
  }
  return true;
}

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
    G.kind = StepKind::GemmFused;
    G.epilogue = (int)plan.epilogues.size();
    plan.epilogues.push_back(std::move(pe));
    plan.launches.erase(plan.launches.begin() + j);
    if (plan.n_backward > (int)j) plan.n_backward--;
  }
  return EG_OK;
}

int make_plan(eg_model* m, TargetState& ts, Plan& plan) {
  Target& t = *ts.target;
  Shapes& shapes = plan.shapes;
  for (auto& in : m->inputs) {
    if (!in.second.bound) continue;
    const TensorDef& d = m->prog.tensors[in.first];
    if (d.has_shape && !d.shape.empty()) {  // staticShapeMismatch (tests/test_errors.nim:56-59)
      bool ok = d.shape.size() == in.second.shape.size();
      for (size_t i = 0; ok && i < d.shape.size(); ++i)
        if (d.shape[i] >= 0 && d.shape[i] != in.second.shape[i]) ok = false;
      if (!ok) {
        set_error("input \"%s\" does not match its static shape", d.name.c_str());
        return EG_ERR_SHAPE;
      }
    }
    shapes[in.first] = in.second.shape;
  }
  for (auto& p : m->params) shapes[p.first] = p.second.shape;

  // shape inference over EVERY kernel (eliminated ones included: the reference collects shape
  // constraints before dead kernels are dropped, model.nim:46-77)
  std::vector<KernelInfo> infos(t.all.size());
  std::set<int> live_set(t.live.begin(), t.live.end());
  for (size_t i = 0; i < t.all.size(); ++i) {
    const Kernel& k = t.all[i];
    bool ready = true;
    int missing = 0;
    for (auto& r : k.reads)  // TensorRandom: shaped like the tensor `rand` was given (parser.nim:378-383)
      if (m->prog.tensors[r.tensor].kind == TK::Random && !shapes.count(r.tensor)) {
        auto sc = m->prog.shape_copy.find(r.tensor);
        if (sc != m->prog.shape_copy.end() && shapes.count(sc->second)) shapes[r.tensor] = shapes[sc->second];
      }
    for (auto& r : k.reads)
      if (!shapes.count(r.tensor)) {
        ready = false;
        missing = r.tensor;
      }
    if (ready)
      for (auto& s : k.setup)  // (bounds that name the written tensor's own shape are resolved by infer_kernel)
        if (s.tensor && s.tensor != k.write.tensor && !shapes.count(s.tensor)) {
          ready = false;
          missing = s.tensor;
        }
    if (!ready) {
      if (live_set.count((int)i)) {
        const TensorDef& d = m->prog.tensors[missing];
        if (d.kind == TK::Input) {
          set_error("input \"%s\" of target \"%s\" was not provided", d.name.c_str(), t.name.c_str());
          return EG_ERR_RUNTIME;
        }
        set_error("the shape of tensor %d is under-constrained", missing);
        return EG_ERR_SHAPE;
      }
      continue;
    }
    int rc = infer_kernel(m->prog, k, shapes, m->epoch, infos[i]);
    if (rc) {
      if (live_set.count((int)i)) return rc;
      eg::clear_error();
    }
  }

  // which result tensors does the live list write; who writes first
  std::map<int, int> first_writer;  // tensor -> position in live
  std::vector<int> result_tensors;
  for (size_t p = 0; p < t.live.size(); ++p) {
    if (ts.lowered[p].inlined) continue;  // its tensor is never materialised
    const Kernel& k = t.all[t.live[p]];
    const int wt = k.write.tensor;
    if (m->prog.tensors[wt].kind == TK::Result && !first_writer.count(wt)) {
      first_writer[wt] = (int)p;
      result_tensors.push_back(wt);
    }
  }
  if (t.output && m->prog.tensors[t.output].kind == TK::Result && !first_writer.count(t.output)) {
    if (!shapes.count(t.output)) {
      set_error("the shape of the output of target \"%s\" is under-constrained", t.name.c_str());
      return EG_ERR_SHAPE;
    }
    result_tensors.push_back(t.output);  // never written: stays zero
  }

  // ---- row fusion (rowfuse.hpp): runs of per-sample kernels become one generated kernel each
  std::vector<int> group_of(t.live.size(), -1);
  plan.row_groups.clear();
  int rc_groups = form_row_groups(m, ts, plan, infos, first_writer, group_of);
  if (rc_groups) return rc_groups;

  // decide overwrite vs accumulate per launch; collect tensors that must be zeroed
  std::set<int> needs_zero;
  plan.launches.clear();
  plan.n_backward = -1;
  std::set<int> folded;  // consumers that run inside the kernel before them
  for (size_t p = 0; p < t.live.size(); ++p) {
    if ((int)p == t.first_update) plan.n_backward = (int)plan.launches.size();
    if (folded.count((int)p)) continue;
    if (group_of[p] <= -2) {
      const int wt = t.all[t.live[p]].write.tensor;  // small groups always accumulate
      if (m->prog.tensors[wt].kind == TK::Result && first_writer[wt] == (int)p) needs_zero.insert(wt);
      if (p == 0 || group_of[p - 1] != group_of[p]) {
        Launch L;
        L.lowered = (int)p;
        L.kind = StepKind::SmallFused;
        L.row_group = -2 - group_of[p];
        plan.launches.push_back(L);
      }
      continue;
    }
    if (group_of[p] >= 0) {
      if (p == 0 || group_of[p - 1] != group_of[p]) {  // first kernel of the group: one launch for all
        Launch L;
        L.lowered = (int)p;
        L.kind = StepKind::RowFused;
        L.row_group = group_of[p];
        L.accumulate = false;
        plan.launches.push_back(L);
      }
      continue;
    }
    Lowered& lo = ts.lowered[p];
    if (lo.absorbed) continue;
    const Kernel& k = t.all[lo.all_index];
    const KernelInfo& info = infos[lo.all_index];
    const int wt = k.write.tensor;
    const bool is_result = m->prog.tensors[wt].kind == TK::Result;
    const bool first = is_result && first_writer[wt] == (int)p;
    const std::vector<long>& wshape = shapes.at(wt);
    if (lo.kind == StepKind::GenericA && first && copy_can_alias(m, ts, k, info, shapes, (int)p)) {
      plan.alias[wt] = k.reads[0].tensor;
      continue;
    }
    if (lo.consumer >= 0 && group_of[lo.consumer] == -1 && first && info.ok && full_cover(k, info, wshape)) {
      // consumer inlining: P covers T, and T, U and the consumer's other operands have one shape
      const int q = lo.consumer;
      const Kernel& C = t.all[t.live[q]];
      const KernelInfo& cinfo = infos[t.live[q]];
      const Kernel& F = *lo.with_consumer;
      const int U = C.write.tensor;
      const long count = prod(wshape);
      bool same = cinfo.ok && cinfo.bounds[0].first == 0 && cinfo.bounds[0].second == count;
      auto matches = [&](int tensor) {
        auto sh = shapes.find(tensor);
        if (sh == shapes.end()) return false;
        return k.write.raw ? prod(sh->second) == count : sh->second == wshape;
      };
      same = same && matches(U);
      for (auto& rd : C.reads) same = same && matches(rd.tensor);
      if (same) {
        const bool u_result = m->prog.tensors[U].kind == TK::Result;
        const bool u_first = u_result && first_writer[U] == q;
        Launch L;
        L.lowered = (int)p;
        L.kind = StepKind::GenericA;
        L.generic = &lo.with_consumer_code;
        L.consumer = q;
        L.blocks_x = (count + 255) / 256;
        int rc = fill_params(m, F, info, shapes, lo.with_consumer_code.src, !u_first, count, 1, 0, L.params);
        if (rc) return rc;
        note_vec4(L, count);
        for (size_t si = 0; si < L.generic->src.slots.size(); ++si) {
          const Slot& sl = L.generic->src.slots[si];
          if (sl.kind == Slot::InstrVal && F.instrs[sl.a].kind == IK::Epoch) L.epoch_slots.push_back((int)si);
        }
        L.c_tensor = U;
        L.accumulate = !u_first;
        plan.launches.push_back(L);
        folded.insert(q);
        continue;
      }
    }
    Launch L;
    L.lowered = (int)p;
    L.kind = lo.kind;
    if (lo.kind == StepKind::Seed) {
      L.count = prod(wshape);
      L.c_tensor = wt;
      L.accumulate = false;
      plan.launches.push_back(L);
      continue;
    }
    bool overwrite = first && full_cover(k, info, wshape);
    if (lo.kind == StepKind::Gemm) {
      const GemmMatch& g = lo.gemm;
      const Op& A = k.reads[g.a_read];
      const Op& B = k.reads[g.b_read];
      L.M = info.bounds[g.li].second;
      L.N = info.bounds[g.lj].second;
      L.K = info.bounds[g.lk].second;
      const std::vector<long>& as = shapes.at(A.tensor);
      const std::vector<long>& bs = shapes.at(B.tensor);
      // loop bounds come from the first tensor that names the iterator; the other operands must agree
      const long a_m = g.trans_a ? as[1] : as[0], a_k = g.trans_a ? as[0] : as[1];
      const long b_k = g.trans_b ? bs[1] : bs[0], b_n = g.trans_b ? bs[0] : bs[1];
      if (a_m < L.M || a_k < L.K || b_k < L.K || b_n < L.N || wshape[0] < L.M || wshape[1] < L.N) {
        set_error("contraction operands have inconsistent shapes ([%ld,%ld] x [%ld,%ld])", a_m, a_k, b_k, b_n);
        return EG_ERR_SHAPE;
      }
      L.lda = as[1];
      L.ldb = bs[1];
      L.ldc = wshape[1];
      L.a_tensor = A.tensor;
      L.b_tensor = B.tensor;
      L.c_tensor = wt;
      L.trans_a = g.trans_a;
      L.trans_b = g.trans_b;
      L.bias_tensor = lo.bias_tensor;
    } else if (lo.kind == StepKind::Conv || lo.kind == StepKind::ConvGradImage || lo.kind == StepKind::ConvGradFilter) {
      auto operand = [&](int which) -> const Op& { return which < 0 ? k.write : k.reads[which]; };
      const Op& img = operand(lo.conv.img_op);
      const Op& flt = operand(lo.conv.flt_op);
      const Op& out = operand(lo.conv.out_op);
      const std::vector<long>& is = shapes.at(img.tensor);
      const std::vector<long>& fs = shapes.at(flt.tensor);
      const std::vector<long>& os = shapes.at(out.tensor);
      const int off = lo.conv.batched ? 1 : 0;
      L.cN = lo.conv.batched ? is[0] : 1;
      L.cH = is[off];
      L.cW = is[off + 1];
      L.cC = is[off + 2];
      L.cF = fs[0];
      L.cFH = fs[1];
      L.cFW = fs[2];
      if (fs[3] != L.cC) {
        set_error("conv2: image has %ld channels, filters have %ld", L.cC, fs[3]);
        return EG_ERR_SHAPE;
      }
      if (lo.kind != StepKind::Conv) {
        // the gradient kernels cover their destination completely only if the three shapes are
        // the ones of a valid convolution
        const bool ok = os.size() == is.size() && (!lo.conv.batched || os[0] == L.cN) && os[off] == L.cH - L.cFH + 1 &&
                        os[off + 1] == L.cW - L.cFW + 1 && os[off + 2] == L.cF;
        if (!ok) {
          set_error("conv2 gradient: output gradient shape does not match image and filter shapes");
          return EG_ERR_SHAPE;
        }
        if (first) overwrite = true;
      }
      // a_tensor: image (forward, filter gradient) or filters (image gradient); b_tensor: filters
      // (forward) or the output gradient
      if (lo.kind == StepKind::Conv) {
        L.a_tensor = img.tensor;
        L.b_tensor = flt.tensor;
      } else if (lo.kind == StepKind::ConvGradFilter) {
        L.a_tensor = img.tensor;
        L.b_tensor = out.tensor;
      } else {
        L.a_tensor = flt.tensor;
        L.b_tensor = out.tensor;
      }
      L.c_tensor = wt;
    } else {
      // generic: choose the template
      std::vector<int> indep, red;
      bool scatter;
      split_loops(k, indep, red, scatter);
      long total = 1, rtotal = 1;
      for (int l : indep) total *= info.bounds[l].second - info.bounds[l].first;
      for (int l : red) rtotal *= info.bounds[l].second - info.bounds[l].first;
      if (total < 0) total = 0;
      if (rtotal < 0) rtotal = 0;
      const bool use_b = lo.b_capable && !scatter && rtotal >= 2048 && total <= 8192 && total * 64 <= rtotal &&
                         full_cover(k, info, wshape);
      if (scatter) overwrite = false;
      if (use_b) {
        int tx = 1;
        while (tx < total && tx < 64) tx <<= 1;
        Generic& g = lo.mode_b[tx];
        if (!g.handle) {
          char name[64];
          snprintf(name, sizeof(name), "eg_k%d_b%d", m->kernel_serial++, tx);
          int rc = generate_mode_b(k, name, tx, g.src);
          if (rc) return rc;
          plan.pending.push_back({g.src.name, g.src.source, &g.handle});
        }
        const int ty = 256 / tx;
        const long col_tiles = (total + tx - 1) / tx;
        long nchunks = (4L * m->ctx->compute_units + col_tiles - 1) / col_tiles;
        const long max_chunks = (rtotal + ty * 8 - 1) / (ty * 8);
        if (nchunks > max_chunks) nchunks = max_chunks;
        if (nchunks < 1) nchunks = 1;
        const long chunk = (rtotal + nchunks - 1) / nchunks;
        nchunks = (rtotal + chunk - 1) / chunk;
        L.kind = StepKind::GenericB;
        L.generic = &g;
        L.blocks_x = nchunks;
        L.blocks_y = col_tiles;
        L.partial_rows = nchunks;
        L.partial_cols = total;
        int rc = fill_params(m, k, info, shapes, g.src, !overwrite, total, rtotal, chunk, L.params);
        if (rc) return rc;
      } else {
        L.kind = StepKind::GenericA;
        L.generic = &lo.mode_a;
        L.blocks_x = (total + 255) / 256;
        int rc = fill_params(m, k, info, shapes, lo.mode_a.src, !overwrite, total, rtotal, 0, L.params);
        if (rc) return rc;
        note_vec4(L, total);
      }
      for (size_t si = 0; si < L.generic->src.slots.size(); ++si) {
        const Slot& sl = L.generic->src.slots[si];
        if (sl.kind == Slot::InstrVal && k.instrs[sl.a].kind == IK::Epoch) L.epoch_slots.push_back((int)si);
      }
      L.c_tensor = wt;
    }
    L.accumulate = !overwrite;
    if (is_result && first && !overwrite) needs_zero.insert(wt);
    plan.launches.push_back(L);
  }
  if (plan.n_backward < 0) plan.n_backward = (int)plan.launches.size();
  if (t.output && m->prog.tensors[t.output].kind == TK::Result && !first_writer.count(t.output)) needs_zero.insert(t.output);
  {
    int rc = fuse_epilogues(m, ts, plan, infos);
    if (rc) return rc;
  }

  plan_overlap(m, ts, plan);
  {
    int rc = build_plan_kernels(m, plan);
    if (rc) return rc;
  }
  // arena layout: tensors that need zeroing first (one memset), then the rest
  plan.arena_offset.clear();
  plan.bucket_zero.clear();
  long off = 0;
  for (int pass = 0; pass < 2; ++pass) {
    for (int tid : result_tensors) {
      if (plan.alias.count(tid)) continue;  // lives in its source's storage
      if (ts.bucket_offset.count(tid)) {
        if (pass == 0 && needs_zero.count(tid)) plan.bucket_zero.push_back(tid);
        continue;
      }
      const bool z = needs_zero.count(tid) != 0;
      if ((pass == 0) != z) continue;
      plan.arena_offset[tid] = off;
      off += align4(prod(shapes.at(tid)));
    }
    if (pass == 0) plan.zero_floats = off;
  }
  plan.random_tensors.clear();
  for (int p : t.live)
    for (auto& rd : t.all[p].reads)
      if (m->prog.tensors[rd.tensor].kind == TK::Random && !plan.arena_offset.count(rd.tensor)) {
        plan.arena_offset[rd.tensor] = off;
        off += align4(prod(shapes.at(rd.tensor)));
        plan.random_tensors.push_back(rd.tensor);
      }
  if (!plan.random_tensors.empty()) {
    int rc = ensure_rng(m);
    if (rc) return rc;
  }
  plan.arena_floats = off;
  if (off > 0) {
    EG_HIP_CHECK(hipSetDevice(m->ctx->device));
    EG_HIP_CHECK(hipMalloc((void**)&plan.arena, (size_t)off * sizeof(float)));
  }
  return EG_OK;
}

int run_launch(eg_model* m, TargetState& ts, Plan& plan, Launch& L) {
  eg_ctx* ctx = m->ctx;
  switch (L.kind) {
    case StepKind::Seed:
      return eg_fill_f32(ctx, L.count, m->grad_scale, tensor_ptr(m, ts, plan, L.c_tensor));
    case StepKind::Gemm:
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
      if (f.splits > 1) {  // cannot happen for the shapes the plan was made for; stay correct anyway
        rc = eg_sgemm(ctx, L.trans_a, L.trans_b, L.M, L.N, L.K, tensor_ptr(m, ts, plan, L.a_tensor), L.lda,
                      tensor_ptr(m, ts, plan, L.b_tensor), L.ldb, tensor_ptr(m, ts, plan, L.c_tensor), L.ldc, 0, bias);
        if (rc) return rc;
        return run_launch(m, ts, plan, pe.consumer);
      }
      const std::string variant = eg::gemm::fused_variant(f);
      eg_kernel*& handle = pe.built[variant];
      if (!handle) {
        const std::string name = "eg_gemm_epi" + std::to_string(m->kernel_serial++);
        const std::string src = eg::gemm::fused_source(f, pe.spec.struct_code, pe.spec.struct_name, name);
        rc = eg_kernel_compile(ctx, name.c_str(), src.c_str(), &handle);
        if (rc) {
          std::string msg = eg_last_error();
          set_error("%s\n--- generated epilogue (%s) ---\n%s", msg.c_str(), variant.c_str(), pe.spec.struct_code.c_str());
          handle = nullptr;
          return rc;
        }
        m->kernels.push_back(handle);
      }
      void* operands[eg::gemm::MAX_EPILOGUE_OPERANDS] = {};
      for (size_t o = 0; o < pe.spec.operands.size(); ++o) operands[o] = tensor_ptr(m, ts, plan, pe.spec.operands[o]);
      eg::gemm::set_epilogue_operands(f, operands, (int)pe.spec.operands.size(), m->grad_scale, m->epoch);
      void* args[] = {f.args};
      return eg::kernel_launch_raw(handle, f.grid, 1, 1, (unsigned)f.nt, args);
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
      return eg::kernel_launch_raw(sg.handle, (unsigned)sg.g.blocks, 1, 1, 256, args.data());
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
      int rc = eg::kernel_launch_raw(pg.handle, (unsigned)pg.nblocks, 1, 1, 256, args.data());
      if (rc) return rc;
      if (pg.g.red_total > 0) {
        eg::RowFinalizeArgs fa = {};
        fa.nseg = (int)pg.red_tensors.size();
        for (int s = 0; s < fa.nseg; ++s) {
          const RowGroupTensor& gt = pg.g.tensors.at(pg.red_tensors[s]);
          fa.dst[s] = tensor_ptr(m, ts, plan, pg.red_tensors[s]);
          fa.offset[s] = (int)gt.red_offset;
          fa.accumulate[s] = gt.accumulate ? 1 : 0;
        }
        fa.offset[fa.nseg] = (int)pg.g.red_total;
        return eg::row_finalize(ctx, pg.partial, pg.nblocks, (int)pg.g.red_total, fa);
      }
      return EG_OK;
    }
    case StepKind::GenericA:
    case StepKind::GenericB: {
      if (L.blocks_x <= 0 || L.blocks_y <= 0) return EG_OK;
      const GenericSource& src = L.generic->src;
      std::vector<void*> args;
      std::vector<float*> ptrs;
      ptrs.reserve(src.tensor_args.size() + 1);
      float* partial = nullptr;
      float* scratch = nullptr;
      if (L.kind == StepKind::GenericB) {
        const long pfloats = (L.partial_rows * L.partial_cols + 3) & ~3L;
        const long sfloats = eg::colsum_scratch_floats(ctx, L.partial_rows, L.partial_cols);
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
      if (L.kind == StepKind::GenericB)
        return eg::colsum_with_scratch(ctx, L.partial_rows, L.partial_cols, partial,
                                       tensor_ptr(m, ts, plan, L.c_tensor), L.accumulate, scratch);
      return EG_OK;
    }
  }
  return EG_OK;
}

int get_plan(eg_model* m, const char* target, TargetState** ts_out, Plan** plan_out) {
  EG_REQUIRE(m && target, EG_ERR_INVALID, "NULL model or target");
  auto it = m->targets.find(target);
  // model.nim:395-396
  EG_REQUIRE(it != m->targets.end(), EG_ERR_RUNTIME, "%s is not a target of the model", target);
  TargetState& ts = it->second;
  const std::string key = shape_key(m);
  auto p = ts.plans.find(key);
  if (p == ts.plans.end()) {
    std::unique_ptr<Plan> plan(new Plan());
    plan->key = key;
    int rc = make_plan(m, ts, *plan);
    if (rc) {
      if (plan->arena) hipFree(plan->arena);
      return rc;
    }
    p = ts.plans.emplace(key, std::move(plan)).first;
  }
  ts.last = p->second.get();
  *ts_out = &ts;
  *plan_out = p->second.get();
  return EG_OK;
}

int run_range_eager(eg_model* m, TargetState& ts, Plan& plan, int begin, int end, bool zero) {
  if (zero) {
    // allocShapes / zeroResultTensor (model.nim:318, 383): results start from zero on every call
    if (plan.zero_floats > 0)
      EG_HIP_CHECK(hipMemsetAsync(plan.arena, 0, (size_t)plan.zero_floats * sizeof(float), m->ctx->stream));
    for (int tid : plan.bucket_zero)
      EG_HIP_CHECK(hipMemsetAsync(ts.bucket + ts.bucket_offset[tid], 0, (size_t)prod(plan.shapes.at(tid)) * sizeof(float),
                                  m->ctx->stream));
    // fresh random tensors for this call (model.nim:310-314 does it on the host); the draw counter
    // is bumped on the device so a captured sequence advances on every replay
    for (size_t r = 0; r < plan.random_tensors.size(); ++r) {
      const int tid = plan.random_tensors[r];
      const TensorDef& d = m->prog.tensors[tid];
      int rc = eg_fill_uniform(m->ctx, prod(plan.shapes.at(tid)), (float)d.lo, (float)d.hi, m->rng_state, (uint64_t)tid,
                               plan.arena + plan.arena_offset[tid]);
      if (rc) return rc;
    }
    if (!plan.random_tensors.empty()) {
      int rc = eg_rng_advance(m->ctx, m->rng_state);
      if (rc) return rc;
    }
  }
  size_t next_overlap = 0;
  for (int i = begin; i < end; ++i) {
    while (next_overlap < plan.overlaps.size() && plan.overlaps[next_overlap].first < i) ++next_overlap;
    if (next_overlap < plan.overlaps.size() && plan.overlaps[next_overlap].first == i &&
        plan.overlaps[next_overlap].big < end) {
      // fork: the side lane takes launches [i, big) after everything issued so far, the main stream
      // goes on with the contraction; join before whatever follows
      const int big = plan.overlaps[next_overlap].big;
      eg_ctx* ctx = m->ctx;
      EG_HIP_CHECK(hipEventRecord(ctx->ev_fork, ctx->stream));
      EG_HIP_CHECK(hipStreamWaitEvent(ctx->side_stream, ctx->ev_fork, 0));
      {
        LaneSwap lane(ctx);
        for (int s2 = i; s2 < big; ++s2) {
          int rc = run_launch(m, ts, plan, plan.launches[s2]);
          if (rc) return rc;
        }
        EG_HIP_CHECK(hipEventRecord(ctx->ev_join, ctx->stream));  // (the side stream, while swapped)
      }
      int rc = run_launch(m, ts, plan, plan.launches[big]);
      if (rc) return rc;
      EG_HIP_CHECK(hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
      i = big;
      continue;
    }
    int rc = run_launch(m, ts, plan, plan.launches[i]);
    if (rc) return rc;
  }
  return EG_OK;
}

bool graphs_enabled() {
  static const bool on = [] {
    const char* e = getenv("EG_NO_GRAPH");
    return !(e && e[0] && e[0] != '0');
  }();
  return on;
}

// slot: 0 = whole call, 1 = backward part, 2 = update part
int run_range(eg_model* m, TargetState& ts, Plan& plan, int begin, int end, bool zero, int slot) {
  int rc = eg::set_device(m->ctx);
  if (rc) return rc;
  Plan::Captured& cap = plan.graphs[slot];
  if (!graphs_enabled() || end - begin < 2) return run_range_eager(m, ts, plan, begin, end, zero);
  // every pointer / scalar that ends up in a kernel argument
  std::ostringstream key;
  for (auto& in : m->inputs)
    if (in.second.bound) key << in.first << "=" << (const void*)in.second.device << ";";
  key << "w" << m->ctx->workspace << "x" << m->ctx->aux << "s" << m->ctx->side_workspace << "y" << m->ctx->side_aux << "b"
      << (void*)ts.bucket << "g" << m->grad_scale << "e" << m->epoch;
  const std::string k = key.str();
  if (cap.exec && cap.key == k) {
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
  static const bool debug = getenv("EG_DEBUG_GRAPH") != nullptr;
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
  EG_HIP_CHECK(hipGraphLaunch(cap.exec, m->ctx->stream));
  return EG_OK;
}

void describe(eg_model* m) {
  std::ostringstream os;
  for (auto& kv : m->targets) {
    TargetState& ts = kv.second;
    os << "target " << kv.first << " (" << ts.target->live.size() << " kernels)\n";
    for (size_t p = 0; p < ts.lowered.size(); ++p) {
      const Lowered& lo = ts.lowered[p];
      const Kernel& k = ts.target->all[lo.all_index];
      const char* kind = lo.absorbed ? "fused-into-previous"
                         : lo.kind == StepKind::Gemm ? (lo.bias_tensor ? "gemm+bias" : "gemm")
                         : lo.kind == StepKind::Conv ? "conv2"
                         : lo.kind == StepKind::ConvGradImage ? "conv2-grad-image"
                         : lo.kind == StepKind::ConvGradFilter ? "conv2-grad-filter"
                         : lo.kind == StepKind::Seed ? "seed-fill"
                         : (lo.b_capable ? "generic(map|split-reduce)" : "generic(map)");
      os << "  [" << p << "] " << kind;
      if (lo.kind == StepKind::Gemm) os << (lo.gemm.trans_a ? " T" : " N") << (lo.gemm.trans_b ? "T" : "N");
      os << " : " << to_text(k) << "\n";
    }
  }
  m->plan_text = os.str();
}

}  // namespace

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

extern "C" {

int eg_model_compile(eg_ctx* ctx, const char* program_text, eg_model** out) try {
  EG_REQUIRE(ctx && program_text && out, EG_ERR_INVALID, "eg_model_compile: NULL argument");
  std::unique_ptr<eg_model> m(new eg_model());
  m->ctx = ctx;
  int rc = parse(program_text, m->prog);
  if (rc) return rc;
  rc = compile_program(m->prog);
  if (rc) return rc;
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
    if (dt.count > 0) {
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
        off += align4(prod(m->prog.tensors[k.gen_tensor].shape));
      }
    ts.bucket_floats = off;
    if (off > 0) {
      EG_HIP_CHECK(hipMalloc((void**)&ts.bucket, (size_t)off * sizeof(float)));
      EG_HIP_CHECK(hipMemset(ts.bucket, 0, (size_t)off * sizeof(float)));
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
  for (auto& kv : m->targets) {
    for (auto& p : kv.second.plans) {
      for (auto& g : p.second->graphs)
        if (g.exec) hipGraphExecDestroy(g.exec);
      for (auto& rg : p.second->row_groups)
        if (rg->partial) hipFree(rg->partial);
      if (p.second->arena) hipFree(p.second->arena);
    }
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
        if (L.kind == StepKind::GemmFused) {
          const PlanEpilogue& pe = *plan.epilogues[L.epilogue];
          os << " | consumer kernel " << pe.consumer.lowered << " operands";
          for (int t : pe.spec.operands) os << " t" << t;
        }
        break;
      case StepKind::Conv: os << "conv2 -> t" << L.c_tensor << (L.accumulate ? " accumulate" : ""); break;
      case StepKind::ConvGradImage: os << "conv2-grad-image -> t" << L.c_tensor << (L.accumulate ? " accumulate" : ""); break;
      case StepKind::ConvGradFilter: os << "conv2-grad-filter -> t" << L.c_tensor << (L.accumulate ? " accumulate" : ""); break;
      case StepKind::GenericA:
        os << "generated(map) kernel " << L.lowered << " -> t" << L.c_tensor;
        if (L.consumer >= 0) os << " (with its consumer, kernel " << L.consumer << ")";
        break;
      case StepKind::GenericB: os << "generated(split-reduce) kernel " << L.lowered << " -> t" << L.c_tensor; break;
      case StepKind::RowFused: {
        const PlanRowGroup& pg = *plan.row_groups[L.row_group];
        os << "row-fused " << pg.g.kernel_index.size() << " kernels";
        break;
      }
      case StepKind::SmallFused: {
        const PlanSmallGroup& sg = *plan.small_groups[L.row_group];
        os << (sg.g.blocks > 1 ? "map-fused " : "small-fused ") << sg.g.kernel_index.size() << " kernels";
        break;
      }
    }
    for (auto& ov : plan.overlaps)
      if ((int)i >= ov.first && (int)i < ov.big) os << "   || side lane, next to launch " << ov.big;
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

int eg_model_param_write(eg_model* m, int tensor_id, const float* host, int64_t count) try {
  EG_REQUIRE(m && host, EG_ERR_INVALID, "eg_model_param_write: NULL argument");
  auto it = m->params.find(tensor_id);
  EG_REQUIRE(it != m->params.end(), EG_ERR_INVALID, "tensor %d is not a parameter", tensor_id);
  EG_REQUIRE(count == it->second.count, EG_ERR_SIZE, "parameter %d has %ld elements, got %ld", tensor_id,
             it->second.count, (long)count);
  if (count == 0) return EG_OK;
  EG_HIP_CHECK(hipSetDevice(m->ctx->device));
  EG_HIP_CHECK(hipMemcpyAsync(it->second.ptr, host, (size_t)count * sizeof(float), hipMemcpyHostToDevice, m->ctx->stream));
  EG_HIP_CHECK(hipStreamSynchronize(m->ctx->stream));
  return EG_OK;
}
EG_CATCH_ALL

int eg_model_param_read(eg_model* m, int tensor_id, float* host, int64_t count) try {
  EG_REQUIRE(m && host, EG_ERR_INVALID, "eg_model_param_read: NULL argument");
  auto it = m->params.find(tensor_id);
  EG_REQUIRE(it != m->params.end(), EG_ERR_INVALID, "tensor %d is not a parameter", tensor_id);
  EG_REQUIRE(count == it->second.count, EG_ERR_SIZE, "parameter %d has %ld elements, got %ld", tensor_id,
             it->second.count, (long)count);
  if (count == 0) return EG_OK;
  EG_HIP_CHECK(hipSetDevice(m->ctx->device));
  EG_HIP_CHECK(hipMemcpyAsync(host, it->second.ptr, (size_t)count * sizeof(float), hipMemcpyDeviceToHost, m->ctx->stream));
  EG_HIP_CHECK(hipStreamSynchronize(m->ctx->stream));
  return EG_OK;
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
  if (count) *count = it->second.bucket_floats;
  return EG_OK;
}
EG_CATCH_ALL

int eg_model_bind_grad_bucket(eg_model* m, const char* target, float* device_ptr, int64_t count) try {
  EG_REQUIRE(m && target && device_ptr, EG_ERR_INVALID, "NULL argument");
  auto it = m->targets.find(target);
  EG_REQUIRE(it != m->targets.end(), EG_ERR_RUNTIME, "%s is not a target of the model", target);
  TargetState& ts = it->second;
  EG_REQUIRE(count >= ts.bucket_floats, EG_ERR_SIZE, "gradient bucket needs %ld floats, got %ld", ts.bucket_floats,
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
                      const int64_t* shape) {
  EG_REQUIRE(m && name, EG_ERR_INVALID, "NULL argument");
  auto it = m->prog.inputs.find(name);
  // model.nim:358-359
  EG_REQUIRE(it != m->prog.inputs.end(), EG_ERR_RUNTIME, "%s is not an input to the model", name);
  EG_REQUIRE(rank >= 0 && rank <= 8 && (rank == 0 || shape), EG_ERR_INVALID, "bad input rank");
  BoundInput& b = m->inputs[it->second];
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
      EG_HIP_CHECK(hipMalloc((void**)&b.owned, (size_t)(count > 0 ? count : 1) * sizeof(float)));
      b.owned_count = count;
    }
    if (count > 0) {
      // blocking H2D on every call, as the reference does (model.nim:364-368 -> cl.nim:111-116)
      EG_HIP_CHECK(hipMemcpyAsync(b.owned, host, (size_t)count * sizeof(float), hipMemcpyHostToDevice, m->ctx->stream));
      EG_HIP_CHECK(hipStreamSynchronize(m->ctx->stream));
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

int eg_model_clear_inputs(eg_model* m) try {
  EG_REQUIRE(m, EG_ERR_INVALID, "NULL model");
  // Host-staged inputs keep their staging buffer (reused by the next host bind); only the
  // bindings are forgotten.  No synchronisation: nothing is freed.
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

// fit (model.nim:413-454): one epoch of mini-batches.  The reference slices the host tensors
// (viewFirst) and uploads every batch with a blocking write before it enqueues the kernels
// (model.nim:364-368); here the data set is uploaded once in pieces on a second stream, piece k+1
// while the batches of piece k run, and every batch is one segment-copy launch (the batch's rows
// of every input -> that input's fixed staging buffer, so the captured launch sequence replays
// unchanged) plus one graph launch.  Nothing in the loop waits for the device.
int eg_model_fit(eg_model* m, const char* target, int n_inputs, const char* const* names, const float* const* data,
                 const int* on_device, const int* ranks, const int64_t* shapes8, int64_t batch_size) try {
  EG_REQUIRE(m && target, EG_ERR_INVALID, "NULL model or target");
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
  for (auto& c : cols) {
    BoundInput& b = *c.in;
    const long count = batch_size * c.row_floats;
    if (b.owned_count < count || !b.owned) {
      EG_HIP_CHECK(hipStreamSynchronize(stream));
      if (b.owned) EG_HIP_CHECK(hipFree(b.owned));
      b.owned = nullptr;
      b.owned_count = 0;
      EG_HIP_CHECK(hipMalloc((void**)&b.owned, (size_t)(count > 0 ? count : 1) * sizeof(float)));
      b.owned_count = count;
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
    if (const char* e = getenv("EG_FIT_SEGMENT_BYTES")) budget = std::min<size_t>(budget, strtoull(e, nullptr, 10));
    if (const char* e = getenv("EG_FIT_PIECE_BYTES")) piece_bytes = strtoull(e, nullptr, 10);
    const size_t batch_bytes = host_row_bytes * (size_t)batch_size;
    EG_REQUIRE(batch_bytes <= budget, EG_ERR_SIZE, "one batch (%zu bytes) does not fit the device", batch_bytes);
    seg_batches = std::min<long>(batch_count, (long)(budget / batch_bytes));
    piece_batches = std::max<long>(1, (long)(piece_bytes / batch_bytes));
    if (!m->copy_stream) EG_HIP_CHECK(hipStreamCreateWithFlags(&m->copy_stream, hipStreamNonBlocking));
    if (!m->copy_event) EG_HIP_CHECK(hipEventCreateWithFlags(&m->copy_event, hipEventDisableTiming));
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
  for (long seg = 0; seg < batch_count; seg += seg_batches) {
    const long seg_end = std::min(batch_count, seg + seg_batches);
    // the segment buffer is about to be overwritten: the batches that read it must be done
    if (seg > 0 && host_row_bytes > 0) EG_HIP_CHECK(hipStreamSynchronize(stream));
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
      for (long b = piece; b < piece_end; ++b) {
        eg::CopySegments cs;
        cs.n = n_inputs;
        for (int i = 0; i < n_inputs; ++i) {
          Column& c = cols[(size_t)i];
          const float* base = c.device ? c.data + (size_t)b * batch_size * c.row_floats
                                       : m->fit_data[i] + (size_t)(b - seg) * batch_size * c.row_floats;
          cs.src[i] = base;
          cs.dst[i] = c.in->owned;
          cs.count[i] = batch_size * c.row_floats;
        }
        rc = eg::copy_segments(m->ctx, cs);
        if (rc) return rc;
        if (!plan) {
          rc = get_plan(m, target, &ts, &plan);
          if (rc) return rc;
        }
        rc = run_range(m, *ts, *plan, 0, (int)plan->launches.size(), true, 0);
        if (rc) return rc;
      }
    }
  }
  // the caller may reuse its host arrays: the uploads (not the kernels) are complete on return
  if (host_row_bytes > 0) EG_HIP_CHECK(hipStreamSynchronize(m->copy_stream));
  return EG_OK;
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

static int read_tensor(eg_model* m, TargetState& ts, int tid, float* host, int64_t count) {
  EG_REQUIRE(ts.last && host, EG_ERR_INVALID, "nothing to read");
  auto s = ts.last->shapes.find(tid);
  EG_REQUIRE(s != ts.last->shapes.end(), EG_ERR_INVALID, "tensor %d has no shape in the last run", tid);
  const long n = prod(s->second);
  EG_REQUIRE(count == n, EG_ERR_SIZE, "Buffer size is not equal to target size (%ld vs %ld)", n, (long)count);
  if (n == 0) return EG_OK;
  float* p = tensor_ptr(m, ts, *ts.last, tid);
  EG_REQUIRE(p, EG_ERR_INVALID, "tensor %d was not materialised by the last run", tid);
  EG_HIP_CHECK(hipSetDevice(m->ctx->device));
  EG_HIP_CHECK(hipMemcpyAsync(host, p, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, m->ctx->stream));
  EG_HIP_CHECK(hipStreamSynchronize(m->ctx->stream));
  return EG_OK;
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

int eg_model_tensor_ptr(eg_model* m, const char* target, int tensor_id, float** device_ptr, int64_t* count) try {
  EG_REQUIRE(m, EG_ERR_INVALID, "NULL model");
  TargetState* ts = last_target(m, target);
  EG_REQUIRE(ts && ts->last, EG_ERR_RUNTIME, "target has not been run");
  auto s = ts->last->shapes.find(tensor_id);
  EG_REQUIRE(s != ts->last->shapes.end(), EG_ERR_INVALID, "tensor %d has no shape in the last run", tensor_id);
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

int eg_model_set_seed(eg_model* m, uint64_t seed) try {
  EG_REQUIRE(m, EG_ERR_INVALID, "NULL model");
  return ensure_rng(m, seed, true);
}
EG_CATCH_ALL

}  // extern "C"
