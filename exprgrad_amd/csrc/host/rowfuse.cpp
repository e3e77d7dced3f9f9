#include "rowfuse.hpp"

#include <algorithm>
#include <cstdio>
#include <regex>
#include <set>

#include "../eg_internal.hpp"
#include "codegen.hpp"

namespace eg {
namespace kd {

namespace {

constexpr long MAX_INNER = 64;     // floats of one tensor row kept in registers
constexpr long SMALL_MAX = 4096;   // a "small" tensor (parameters, their gradients, scalars)
constexpr long MAX_WORK = 2048;    // unrolled iterations of one kernel per sample

long prodv(const std::vector<long>& s, size_t from = 0) {
  long p = 1;
  for (size_t i = from; i < s.size(); ++i) p *= s[i];
  return p;
}

bool lin_has(const Lin& l, int reg) { return l.factor_of(reg) != 0; }

bool op_has(const Op& op, int reg) {
  for (auto& d : op.dims)
    if (lin_has(d, reg)) return true;
  return false;
}

std::string lin_text(const Lin& l, const std::map<int, std::string>& subst) {
  std::string s = std::to_string(l.constant) + "L";
  for (auto& f : l.factors) {
    auto it = subst.find(f.first);
    const std::string var = it != subst.end() ? it->second : "r" + std::to_string(f.first);
    s += " + " + std::to_string(f.second) + "L * " + var;
  }
  return "(" + s + ")";
}

}  // namespace

RowKernelInfo analyse_row_kernel(const Program& prog, const Kernel& k, const KernelInfo& info, const Shapes& shapes,
                                 long B) {
  RowKernelInfo r;
  if (!info.ok || B <= 0 || !k.index_instrs.empty()) return r;
  std::vector<const Op*> ops;
  for (auto& rd : k.reads) ops.push_back(&rd);
  ops.push_back(&k.write);
  for (const Op* op : ops)
    if (!shapes.count(op->tensor)) return r;
  auto small = [&](const Op* op) { return prodv(shapes.at(op->tensor)) <= SMALL_MAX; };
  auto state = [&](const Op* op) {
    const TK kind = prog.tensors[op->tensor].kind;
    return kind == TK::Param || kind == TK::Cache;
  };

  long other_work = 1;
  for (size_t l = 0; l < k.loops.size(); ++l) other_work *= std::max(0L, info.bounds[l].second - info.bounds[l].first);

  // the gradLoss seed and similar: no batch loop at all, everything small
  if (k.is_seed) {
    bool all_small = true;
    for (const Op* op : ops) all_small = all_small && small(op);
    if (all_small && other_work <= 64) {
      r.ok = true;
      r.small_only = true;
      r.work = other_work;
    }
    return r;
  }
  if (!k.setup.empty()) return r;

  for (size_t l = 0; l < k.loops.size(); ++l) {
    const int y = k.loops[l].reg;
    const long lo = info.bounds[l].first, hi = info.bounds[l].second;
    if (lo != 0) continue;
    bool ok = true, any = false;
    bool raw = false;
    long inner = 0;
    // a raw iterator ({it}) shows in raw ops only; [B,1] tensors make its extent equal B as well
    bool any_raw = false;
    for (const Op* op : ops)
      if (op_has(*op, y) && op->raw) any_raw = true;
    if (hi == B && !any_raw) {
      for (const Op* op : ops) {
        if (!op_has(*op, y)) {
          if (!small(op)) ok = false;
          continue;
        }
        any = true;
        const std::vector<long>& shp = shapes.at(op->tensor);
        if (op->raw || state(op) || shp.empty() || shp[0] != B || op->dims[0].only_register() != y ||
            prodv(shp, 1) > MAX_INNER)
          ok = false;
        for (size_t d = 1; ok && d < op->dims.size(); ++d)
          if (lin_has(op->dims[d], y)) ok = false;
      }
    } else if (any_raw && hi >= B && hi % B == 0 && hi / B <= MAX_INNER) {
      // raw iterator over B * S elements: it = y * S + j
      raw = true;
      inner = hi / B;
      for (const Op* op : ops) {
        if (!op_has(*op, y)) {
          if (!small(op)) ok = false;
          continue;
        }
        any = true;
        const std::vector<long>& shp = shapes.at(op->tensor);
        if (!op->raw || state(op) || op->dims.size() != 1 || op->dims[0].only_register() != y || shp.empty() ||
            shp[0] != B || prodv(shp) != hi)
          ok = false;
      }
    } else {
      continue;
    }
    if (!ok || !any) continue;
    // the iterator must not be used as a value by the expression (it only indexes)
    for (auto& ins : k.instrs)
      for (int a : ins.args)
        if (a == y) ok = false;
    if (!ok) continue;
    const long span = hi - lo;
    long work = span > 0 ? other_work / span : 0;
    if (raw) work *= inner;
    if (work > MAX_WORK) continue;
    r.ok = true;
    r.row_loop = (int)l;
    r.raw = raw;
    r.inner = inner;
    r.work = work;
    return r;
  }
  return r;
}

namespace {

struct GroupEmitter {
  const Program& prog;
  const Shapes& shapes;
  RowGroup& g;
  std::string code;

  std::string tname(int t) const { return "t" + std::to_string(t); }

  // text of the element a tensor op refers to
  std::string element(const Op& op, const RowKernelInfo& ri, const Kernel& k, const std::map<int, std::string>& subst,
                      const std::string& raw_j) {
    const RowGroupTensor& gt = g.tensors.at(op.tensor);
    const std::vector<long>& shp = shapes.at(op.tensor);
    const int yreg = ri.row_loop >= 0 ? k.loops[ri.row_loop].reg : 0;
    const bool row_op = ri.row_loop >= 0 && op_has(op, yreg);
    std::string idx;
    if (row_op) {
      if (ri.raw) {
        idx = raw_j;
      } else {
        long stride = 1;
        std::vector<std::string> terms;
        for (size_t d = shp.size(); d-- > 1;) {
          terms.push_back(std::to_string(stride) + "L * " + lin_text(op.dims[d], subst));
          stride *= shp[d];
        }
        idx = "0L";
        for (auto& t : terms) idx += " + " + t;
      }
      if (gt.role == RowGroupTensor::RowLocal) return "L" + std::to_string(op.tensor) + "[" + idx + "]";
      return tname(op.tensor) + "[y * " + std::to_string(gt.inner) + "L + " + idx + "]";
    }
    // small tensor
    if (op.raw) {
      idx = lin_text(op.dims[0], subst);
    } else {
      long stride = 1;
      idx = "0L";
      for (size_t d = shp.size(); d-- > 0;) {
        idx += " + " + std::to_string(stride) + "L * " + lin_text(op.dims[d], subst);
        stride *= shp[d];
      }
    }
    switch (gt.role) {
      case RowGroupTensor::SmallLocal: return "S" + std::to_string(op.tensor) + "[" + idx + "]";
      case RowGroupTensor::Reduction: return "R" + std::to_string(op.tensor) + "[" + idx + "]";
      default: return tname(op.tensor) + "[" + idx + "]";
    }
  }

  void emit_kernel(const Kernel& k, const KernelInfo& info, const RowKernelInfo& ri, int serial) {
    const std::vector<Ty> ty = infer_types(k);
    std::map<int, std::string> subst;
    std::string raw_j;
    code += "  if (active) {  // kernel " + std::to_string(serial) + ": " + to_text(k).substr(0, 90) + "\n";
    for (auto& s : k.setup) {  // host-evaluated values become literals (shapes are fixed for this build)
      code += "    const long r" + std::to_string(s.res) + " = " + std::to_string(info.vals.at(s.res)) + "L;\n";
    }
    int depth = 0;
    for (size_t l = 0; l < k.loops.size(); ++l) {
      const std::string r = "r" + std::to_string(k.loops[l].reg);
      if ((int)l == ri.row_loop) {
        if (ri.raw) {
          raw_j = "j" + std::to_string(serial);
          code += "    _Pragma(\"unroll\") for (long " + raw_j + " = 0; " + raw_j + " < " + std::to_string(ri.inner) + "L; ++" +
                  raw_j + ") {\n";
          ++depth;
        }
        continue;  // the batch iterator is the thread's row
      }
      code += "    _Pragma(\"unroll\") for (long " + r + " = " + std::to_string(info.bounds[l].first) + "L; " + r + " < " +
              std::to_string(info.bounds[l].second) + "L; ++" + r + ") {\n";
      ++depth;
    }
    for (auto& rd : k.reads)
      code += "      const float r" + std::to_string(rd.reg) + " = " + element(rd, ri, k, subst, raw_j) + ";\n";
    for (auto& ins : k.instrs) {
      const Ty t = ty[ins.res];
      const char* ctype = t == Ty::Scalar ? "float" : (t == Ty::Index ? "long" : "bool");
      std::string special;
      if (ins.kind == IK::Epoch) {
        special = "EP";
      } else if (ins.kind == IK::Shape || ins.kind == IK::Len || ins.kind == IK::ShapeLen) {
        const std::vector<long>& shp = shapes.at(ins.tensor);
        long v = 0;
        if (ins.kind == IK::Len) v = prodv(shp);
        else if (ins.kind == IK::ShapeLen) v = (long)shp.size();
        else {
          int d = ins.dim < 0 ? ins.dim + (int)shp.size() : ins.dim;
          v = (d >= 0 && d < (int)shp.size()) ? shp[d] : 0;
        }
        special = std::to_string(v) + "L";
      }
      std::string e = instr_expression(ins, special, "r");
      if (k.is_seed && ins.kind == IK::Scalar) e = "GS";  // gradLoss{i} = 1, times the data-parallel scale
      code += std::string("      const ") + ctype + " r" + std::to_string(ins.res) + " = " + e + ";\n";
    }
    const std::string w = element(k.write, ri, k, subst, raw_j);
    code += "      " + w + " = " + w + " + r" + std::to_string(k.result) + ";\n";
    for (int d = 0; d < depth; ++d) code += "    }\n";
    code += "  }\n";
  }
};

}  // namespace

static std::string small_kernel_body(const Kernel& k, const KernelInfo& info, const Shapes& shapes, const std::string& prefix,
                                     int serial, bool barrier = true);

int generate_row_group(const Program& prog, const std::vector<Kernel>& all, const std::vector<KernelInfo>& infos,
                       const Shapes& shapes, RowGroup& g) {
  GroupEmitter em{prog, shapes, g, {}};
  em.code.clear();
  // pointer arguments: every tensor that touches memory
  g.ptr_args.clear();
  for (auto& kv : g.tensors) {
    const RowGroupTensor& t = kv.second;
    const bool mem = t.role == RowGroupTensor::RowExternal || t.role == RowGroupTensor::SmallExternal ||
                     (t.role == RowGroupTensor::RowLocal && (t.load_first || t.store));
    if (mem) g.ptr_args.push_back(kv.first);
  }
  // Large batches with little per-thread state: cap the kernel at 96 registers (5 waves per SIMD) so its
  // waves fit next to a long contraction's on the same SIMD (a 256x256 tile leaves 128 of 512 registers
  // per lane): on the side lane of the batch pipeline a 7 us row group took 53 us waiting for whole CUs.
  long state = 0;
  for (auto& kv : g.tensors)
    if (kv.second.role != RowGroupTensor::RowExternal && kv.second.role != RowGroupTensor::SmallExternal) state += kv.second.inner;
  const bool slim = g.B >= 4096 && state <= 64;
  std::string sig = std::string("extern \"C\" __global__ void __launch_bounds__(256") + (slim ? ", 5" : "") + ") " + g.name +
                    "(float* __restrict__ partial";
  // a tensor the tail kernels touch is also reachable through u<t> (and written there): its t<t> must not promise
  // that nobody else modifies it
  std::set<int> tail_touched;
  if (g.in_kernel_finalize && g.red_total > 0)
    for (int ki : g.tail_kernels) {
      tail_touched.insert(all[ki].write.tensor);
      for (auto& rd : all[ki].reads) tail_touched.insert(rd.tensor);
    }
  for (int t : g.ptr_args) {
    const RowGroupTensor& gt = g.tensors.at(t);
    sig += gt.role == RowGroupTensor::RowLocal ? ", float* t" : tail_touched.count(t) ? ", const float* t" : ", const float* __restrict__ t";
    sig += std::to_string(t);
  }
  sig += ", long B, float GS, long EP";
  // One block (a batch of at most 256 rows): the totals go straight to their destinations.  In a captured graph a
  // dependent launch costs ~4.5 us whatever it does, and row_finalize of one partial row does nothing but copy
  // (p + 0 + 0 + 0 in its tree: the same value).
  g.single_block = g.B <= 256 && g.red_total > 0 && eg::sw::raw("EG_NO_ROW_DIRECT") == nullptr;
  if (g.single_block) g.in_kernel_finalize = false;
  if (g.red_total <= 0) g.in_kernel_finalize = false;
  if (!g.in_kernel_finalize) g.tail_kernels.clear();
  if (g.single_block || g.in_kernel_finalize)
    for (auto& kv : g.tensors)
      if (kv.second.role == RowGroupTensor::Reduction) sig += ", float* d" + std::to_string(kv.first);
  g.tail_ptr_args.clear();
  if (g.in_kernel_finalize) {
    sig += ", unsigned* counter, long MODE";
    std::set<int> touched;
    for (int ki : g.tail_kernels) {
      touched.insert(all[ki].write.tensor);
      for (auto& rd : all[ki].reads) touched.insert(rd.tensor);
    }
    g.tail_ptr_args.assign(touched.begin(), touched.end());
    // NOT __restrict__: the same tensors are reachable through d<t> (the totals just written) and t<t> (parameters the
    // tail overwrites) in this kernel; the barriers between those accesses order them, the qualifier would deny them
    for (int t : g.tail_ptr_args) sig += ", float* u" + std::to_string(t);
  }
  sig += ")";

  std::string& c = em.code;
  // in_kernel_finalize: the samples are walked with a grid stride, so that a launch may use FEWER blocks than B / 256 (a
  // row group with a tail: 64 blocks — fewer arrivals at the ticket counter, fewer partial rows for the last block);
  // with ceil(B / 256) blocks the loop runs once and every value is what the one-sample-per-thread form computes.
  const bool strided = g.in_kernel_finalize;
  if (!strided) c += "  const long y = (long)blockIdx.x * 256 + threadIdx.x;\n  const bool active = y < B;\n";
  std::string init;  // per-sample state: (re)initialised for every sample
  for (auto& kv : g.tensors) {
    const RowGroupTensor& t = kv.second;
    const std::string id = std::to_string(kv.first);
    if (t.role == RowGroupTensor::RowLocal) {
      c += "  float L" + id + "[" + std::to_string(t.inner) + "];\n";
      init += "  _Pragma(\"unroll\") for (int j = 0; j < " + std::to_string(t.inner) + "; ++j) L" + id + "[j] = ";
      init += t.load_first ? "active ? t" + id + "[y * " + std::to_string(t.inner) + "L + j] : 0.0f;\n" : "0.0f;\n";
    } else if (t.role == RowGroupTensor::SmallLocal || t.role == RowGroupTensor::Reduction) {
      const char* p = t.role == RowGroupTensor::SmallLocal ? "S" : "R";
      c += std::string("  float ") + p + id + "[" + std::to_string(t.inner) + "];\n";
      std::string z = "  _Pragma(\"unroll\") for (int j = 0; j < " + std::to_string(t.inner) + "; ++j) " + p + id + "[j] = 0.0f;\n";
      if (t.role == RowGroupTensor::Reduction) c += z;  // batch totals: once
      else init += z;
    }
  }
  if (strided) c += "  auto one_sample = [&](const long y) {\n  const bool active = true;\n";
  c += init;
  for (size_t i = 0; i < g.kernel_index.size(); ++i)
    em.emit_kernel(all[g.kernel_index[i]], infos[g.kernel_index[i]], g.infos[i], (int)i);
  // rows that are needed after the group
  for (auto& kv : g.tensors) {
    const RowGroupTensor& t = kv.second;
    if (t.role != RowGroupTensor::RowLocal || !t.store) continue;
    const std::string id = std::to_string(kv.first);
    c += "  if (active) { _Pragma(\"unroll\") for (int j = 0; j < " + std::to_string(t.inner) + "; ++j) t" + id + "[y * " +
         std::to_string(t.inner) + "L + j] = L" + id + "[j]; }\n";
  }
  if (strided) {
    c += "  };\n";
    const long per_trip = g.grid_blocks * 256;
    const long trips = per_trip > 0 && g.B % per_trip == 0 ? g.B / per_trip : 0;
    if (trips >= 2 && trips <= 8) {
      // the launch this kernel was generated for: every thread has exactly `trips` samples, no bounds test between them
      c += "  if (gridDim.x == " + std::to_string(g.grid_blocks) + " && B == " + std::to_string(g.B) + "L) {\n";
      c += "    _Pragma(\"unroll\") for (int trip = 0; trip < " + std::to_string(trips) + "; ++trip) one_sample((long)blockIdx.x * 256 + threadIdx.x + (long)trip * " +
           std::to_string(per_trip) + "L);\n  } else {\n";
      c += "    for (long y = (long)blockIdx.x * 256 + threadIdx.x; y < B; y += (long)gridDim.x * 256) one_sample(y);\n  }\n";
    } else {
      c += "  for (long y = (long)blockIdx.x * 256 + threadIdx.x; y < B; y += (long)gridDim.x * 256) one_sample(y);\n";
    }
  }
  // batch reductions: wave shuffles, then the four wave totals through LDS, one partial row per block
  if (g.red_total > 0) {
    const std::string E = std::to_string(g.red_total);
    c += "  __shared__ float red[4 * " + E + "];\n";
    c += "  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;\n";
    for (auto& kv : g.tensors) {
      const RowGroupTensor& t = kv.second;
      if (t.role != RowGroupTensor::Reduction) continue;
      const std::string id = std::to_string(kv.first);
      // The butterfly runs over ALL the values of a step together (round 6, EG_ROW_TRACE: one value after the other — the
      // guarded store behind each kept the compiler from interleaving them — every one of the 6 x 17 shuffles of the XOR
      // step's totals waited out the LDS crossbar's latency by itself: 7 700 cycles, and again in the last block's fold;
      // together 6.5 of the kernel's 11.8 us).  Same additions per value, same order.
      for (int off = 32; off >= 1; off >>= 1)
        c += "  _Pragma(\"unroll\") for (int j = 0; j < " + std::to_string(t.inner) + "; ++j) R" + id + "[j] += eg_xor_lane<" + std::to_string(off) +
             ">(R" + id + "[j]);\n";
      c += "  if (lane == 0) {\n    _Pragma(\"unroll\") for (int j = 0; j < " + std::to_string(t.inner) + "; ++j) red[wave * " + E + " + " +
           std::to_string(t.red_offset) + " + j] = R" + id + "[j];\n  }\n";
    }
    c += "  __syncthreads();\n";
    if (g.single_block) {
      for (auto& kv : g.tensors) {
        const RowGroupTensor& t = kv.second;
        if (t.role != RowGroupTensor::Reduction) continue;
        const std::string id = std::to_string(kv.first), off = std::to_string(t.red_offset);
        c += "  for (int j = threadIdx.x; j < " + std::to_string(t.inner) + "; j += 256) {\n";
        c += "    const int e = " + off + " + j;\n";
        c += "    const float s = (red[e] + red[" + E + " + e]) + (red[2 * " + E + " + e] + red[3 * " + E + " + e]);\n";
        c += std::string("    d") + id + "[j] = " + (t.accumulate ? "d" + id + "[j] + s" : std::string("s")) + ";\n  }\n";
      }
    } else {
      // A block's partial row: whole 16-byte groups at a stride of ES floats.  With the in-kernel fold the groups go out as
      // ONE `global_store_dwordx4 sc0 sc1` each and come back as `global_load_dwordx4 sc0 sc1` (round 6, EG_ROW_TRACE: as
      // 4-byte accesses — every one a transaction of its own on the fabric — the last block of the XOR step waited 3.6 us for
      // its 17 stores to drain and 3.8 us for 64 x 17 loads: 7.4 of the kernel's 11.8 us).
      const std::string ES = std::to_string(g.red_stride()), G4 = std::to_string(g.red_stride() / 4);
      if (g.in_kernel_finalize) {
        c += "  typedef float f4_ __attribute__((ext_vector_type(4)));\n";
        c += "  if (threadIdx.x < " + G4 + ") {\n    f4_ tv;\n";
        c += "    _Pragma(\"unroll\") for (int k = 0; k < 4; ++k) {\n      const int e = 4 * threadIdx.x + k;\n";
        c += "      tv[k] = e < " + E + " ? (red[e] + red[" + E + " + e]) + (red[2 * " + E + " + e] + red[3 * " + E + " + e]) : 0.0f;\n    }\n";
        c += "    float* const dst = partial + (long)blockIdx.x * " + ES + " + 4 * threadIdx.x;\n";
        c += "    asm volatile(\"global_store_dwordx4 %0, %1, off sc0 sc1\" : : \"v\"(dst), \"v\"(tv) : \"memory\");\n  }\n";
      } else {
        c += "  for (int e = threadIdx.x; e < " + E + "; e += 256) {\n";
        c += "    const float total = (red[e] + red[" + E + " + e]) + (red[2 * " + E + " + e] + red[3 * " + E + " + e]);\n";
        c += "    partial[(long)blockIdx.x * " + ES + " + e] = total;\n  }\n";
      }
      if (g.in_kernel_finalize) {
        // The last block to arrive folds the partial rows.  No agent-scope fences (each costs ~1.7 us on MI355X, and
        // every block would pay one): the partial rows go out as write-through stores (system scope: sc0 sc1) and are
        // read back with loads of the same scope, which bypass the non-coherent caches on both sides
        // (MI355X_MICROARCH.md, "Workgroup dispatch ... inter-workgroup visibility": sc0 sc1 stores and loads on both
        // sides are a valid hand-off; the stores are drained — vmcnt(0) — before the block takes its ticket).
        // Sums in the order of row_finalize_kernel (reduce.hip): thread t of 256 adds rows t, t + 256, ...; xor-shuffle
        // tree per wave; ((w0 + w1) + w2) + w3 — the same value to the bit for the same number of blocks.
        // This hand-off is OUTSIDE the HIP / LLVM memory model (relaxed atomics + an explicit vmcnt drain instead of a
        // release / acquire pair on the ticket): it holds on gfx9-family ISAs, where stores count in vmcnt and sc0 sc1
        // accesses go to memory.  The library is built for gfx950 only; the generated text refuses anything else.
        c += "#if !defined(__gfx950__)\n#error \"row-tail hand-off relies on gfx950 cache-bypass stores and vmcnt store counting\"\n#endif\n";
        c += "  if (MODE != 0) {\n";
        c += "    __shared__ int s_last;\n";
        c += "    asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");\n";
        c += "    __syncthreads();\n";
        c += "    if (threadIdx.x == 0) {\n";
        c += "      const unsigned ticket = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);\n";
        c += "      s_last = ticket == gridDim.x - 1 ? 1 : 0;\n";
        c += "      if (s_last) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch\n";
        c += "    }\n";
        c += "    __syncthreads();\n";
        c += "    if (s_last) {\n";
        c += "      const int NB = (int)gridDim.x;\n";
        c += "      float acc[" + E + "];\n";
        c += "      _Pragma(\"unroll\") for (int e = 0; e < " + E + "; ++e) acc[e] = 0.0f;\n";
        c += "      for (int b = threadIdx.x; b < NB; b += 256) {\n";
        c += "        const float* const src = partial + (long)b * " + ES + ";\n        f4_ q_[" + G4 + "];\n";
        {
          std::string tie;
          for (long q = 0; q < g.red_stride() / 4; ++q) {
            c += "        asm volatile(\"global_load_dwordx4 %0, %1, off sc0 sc1\" : \"=v\"(q_[" + std::to_string(q) + "]) : \"v\"(src + " +
                 std::to_string(4 * q) + ") : \"memory\");\n";
            tie += std::string(q ? ", " : "") + "\"+v\"(q_[" + std::to_string(q) + "])";
          }
          // (the compiler does not count loads issued from inline assembly: the wait is explicit and tied to the registers)
          c += "        asm volatile(\"s_waitcnt vmcnt(0)\" : " + tie + " : : \"memory\");\n";
        }
        c += "        _Pragma(\"unroll\") for (int e = 0; e < " + E + "; ++e) acc[e] += q_[e >> 2][e & 3];\n      }\n";
        for (int off = 32; off >= 1; off >>= 1)
          c += "      _Pragma(\"unroll\") for (int e = 0; e < " + E + "; ++e) acc[e] += eg_xor_lane<" + std::to_string(off) + ">(acc[e]);\n";
        c += "      if (lane == 0) {\n        _Pragma(\"unroll\") for (int e = 0; e < " + E + "; ++e) red[wave * " + E + " + e] = acc[e];\n      }\n";
        c += "      __syncthreads();\n";
        for (auto& kv : g.tensors) {
          const RowGroupTensor& t = kv.second;
          if (t.role != RowGroupTensor::Reduction) continue;
          const std::string id = std::to_string(kv.first), off = std::to_string(t.red_offset);
          c += "      for (int j = threadIdx.x; j < " + std::to_string(t.inner) + "; j += 256) {\n";
          c += "        const int e = " + off + " + j;\n";
          c += "        const float s = ((red[e] + red[" + E + " + e]) + red[2 * " + E + " + e]) + red[3 * " + E + " + e];\n";
          c += std::string("        d") + id + "[j] = " + (t.accumulate ? "d" + id + "[j] + s" : std::string("s")) + ";\n      }\n";
        }
        if (!g.tail_kernels.empty()) {
          c += "      if (MODE == 2) {  // the kernels that follow the group, on the totals just written\n";
          c += "      __syncthreads();\n";
          // Kernels none of which touches what another one writes (one gradientDescent kernel per parameter,
          // base.nim:37-38) need no barrier between them: a thread's loads of all of them can be in flight together —
          // one memory round trip for the tail instead of one per kernel.
          bool independent = true;
          for (size_t i = 0; i < g.tail_kernels.size() && independent; ++i)
            for (size_t j = 0; j < g.tail_kernels.size() && independent; ++j) {
              if (i == j) continue;
              const Kernel &a = all[g.tail_kernels[i]], &b = all[g.tail_kernels[j]];
              if (a.write.tensor == b.write.tensor) independent = false;
              for (auto& rd : b.reads)
                if (rd.tensor == a.write.tensor) independent = false;
            }
          for (size_t i = 0; i < g.tail_kernels.size(); ++i)
            c += small_kernel_body(all[g.tail_kernels[i]], infos[g.tail_kernels[i]], shapes, "u", (int)i, !independent);
          c += "      }\n";
        }
        c += "    }\n  }\n";
        // EG_ROW_TRACE=1 (detector): the last block to arrive prints where ITS time went — cycles since its own start at:
        // samples done, partial row stored and drained, ticket taken, partial rows of all blocks read, totals and tail done
        if (eg::sw::raw("EG_ROW_TRACE") != nullptr) {
          auto insert_before = [&](const std::string& anchor, const std::string& text, size_t from) {
            const size_t at = c.find(anchor, from);
            if (at == std::string::npos) return std::string::npos;
            c.insert(at, text);
            return at + text.size() + anchor.size();
          };
          auto stamp = [](int k) { return "  tr_[" + std::to_string(k) + "] = __builtin_readcyclecounter();\n"; };
          c = "  long long tr_[8];\n" + stamp(0) + c;
          size_t pos = insert_before("  __shared__ float red[", stamp(1), 0);
          if (pos != std::string::npos) pos = insert_before("  if (threadIdx.x < ", stamp(6), pos);                  // wave totals in LDS
          if (pos != std::string::npos) pos = insert_before("    asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");\n", stamp(7), pos);   // store issued
          if (pos != std::string::npos) pos = insert_before("    if (threadIdx.x == 0) {\n      const unsigned ticket", stamp(2), pos);
          if (pos != std::string::npos) pos = insert_before("    if (s_last) {\n", stamp(3), pos);
          if (pos != std::string::npos) pos = insert_before("      __syncthreads();\n", stamp(4), pos);
          const size_t end = c.rfind("    }\n  }\n");
          if (pos != std::string::npos && end != std::string::npos)
            c.insert(end, stamp(5) + "      if (threadIdx.x == 0) printf(\"[eg] " + g.name +
                              " last block (%d of %d): samples %lld, wave totals %lld, store issued %lld, drained %lld, ticket %lld, partials in %lld, done %lld cycles\\n\", "
                              "(int)blockIdx.x, (int)gridDim.x, tr_[1] - tr_[0], tr_[6] - tr_[0], tr_[7] - tr_[0], tr_[2] - tr_[0], tr_[3] - tr_[0], tr_[4] - tr_[0], tr_[5] - tr_[0]);\n");
        }
      }
    }
  }
  // eg_xor_lane<OFF>(v): the value of lane (l ^ OFF) — what __shfl_xor(v, OFF, 64) returns through the LDS crossbar, here as
  // DPP moves for OFF < 16 (quad permutations for 1 and 2; for 4 and 8 two row shifts whose bank masks pick, per group of four
  // lanes, the one that comes from the right side): four of a butterfly's six steps become vector instructions without a
  // trip to LDS.  Same partner lanes, so the same sums to the bit (tests/test_gpu_row_tail.py holds the in-kernel fold
  // against row_finalize_kernel, which shuffles).
  static const char* kXorLane =
      "#ifndef EG_XOR_LANE\n#define EG_XOR_LANE\n"
      "template <int OFF> __device__ __forceinline__ float eg_xor_lane(float v) {\n"
      "  const int b = __builtin_bit_cast(int, v);\n"
      "  if constexpr (OFF == 1) return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(b, b, 0xB1, 0xF, 0xF, false));\n"
      "  else if constexpr (OFF == 2) return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(b, b, 0x4E, 0xF, 0xF, false));\n"
      "  else if constexpr (OFF == 4) {\n"
      "    int t = __builtin_amdgcn_update_dpp(b, b, 0x104, 0xF, 0x5, false);   // row_shl:4 into lanes 0-3, 8-11 of a row\n"
      "    t = __builtin_amdgcn_update_dpp(t, b, 0x114, 0xF, 0xA, false);       // row_shr:4 into lanes 4-7, 12-15\n"
      "    return __builtin_bit_cast(float, t);\n"
      "  } else if constexpr (OFF == 8) {\n"
      "    int t = __builtin_amdgcn_update_dpp(b, b, 0x108, 0xF, 0x3, false);   // row_shl:8 into lanes 0-7\n"
      "    t = __builtin_amdgcn_update_dpp(t, b, 0x118, 0xF, 0xC, false);       // row_shr:8 into lanes 8-15\n"
      "    return __builtin_bit_cast(float, t);\n"
      "  } else return __shfl_xor(v, OFF, 64);\n"
      "}\n#endif\n";
  g.source = (eg::sw::raw("EG_NO_DPP_BUTTERFLY") ? std::string("#ifndef EG_XOR_LANE\n#define EG_XOR_LANE\ntemplate <int OFF> __device__ __forceinline__ float "
                                                               "eg_xor_lane(float v) { return __shfl_xor(v, OFF, 64); }\n#endif\n")
                                                 : std::string(kXorLane)) +
             sig + " {\n" + c + "}\n";
  return EG_OK;
}

// ---------------------------------------------------------------------------------- small groups

bool is_small_kernel(const Program& prog, const Kernel& k, const KernelInfo& info, const Shapes& shapes) {
  if (!info.ok || !k.index_instrs.empty()) return false;
  if (!k.setup.empty() && !k.is_seed) return false;
  std::vector<const Op*> ops;
  for (auto& rd : k.reads) ops.push_back(&rd);
  ops.push_back(&k.write);
  for (const Op* op : ops) {
    auto it = shapes.find(op->tensor);
    if (it == shapes.end() || prodv(it->second) > SMALL_MAX) return false;
  }
  long work = 1;
  for (size_t l = 0; l < k.loops.size(); ++l) work *= std::max(0L, info.bounds[l].second - info.bounds[l].first);
  if (work > 65536) return false;
  std::vector<int> indep, red;
  bool scatter;
  split_loops(k, indep, red, scatter);
  return !scatter;
}

// One small kernel as a loop of a single 256-thread block over its independent iterations, reductions serial per
// thread, `__syncthreads()` behind it: the body of a small group, and of the tail a row group's last block runs
// (pointer names <prefix><tensor id>).
static std::string small_kernel_body(const Kernel& k, const KernelInfo& info, const Shapes& shapes, const std::string& prefix,
                                     int serial, bool barrier) {
  std::string c;
  const std::map<int, std::string> no_subst;
  {
    const std::vector<Ty> ty = infer_types(k);
    std::vector<int> indep, red;
    bool scatter;
    split_loops(k, indep, red, scatter);
    long total = 1;
    for (int l : indep) total *= std::max(0L, info.bounds[l].second - info.bounds[l].first);
    auto element = [&](const Op& op) {
      const std::vector<long>& shp = shapes.at(op.tensor);
      std::string idx;
      if (op.raw) {
        idx = lin_text(op.dims[0], no_subst);
      } else {
        long stride = 1;
        idx = "0L";
        for (size_t d = shp.size(); d-- > 0;) {
          idx += " + " + std::to_string(stride) + "L * " + lin_text(op.dims[d], no_subst);
          stride *= shp[d];
        }
      }
      return prefix + std::to_string(op.tensor) + "[" + idx + "]";
    };
    c += "  // kernel " + std::to_string(serial) + ": " + to_text(k).substr(0, 90) + "\n";
    c += "  for (long idx = threadIdx.x; idx < " + std::to_string(total) + "L; idx += 256) {\n";
    for (auto& s : k.setup) c += "    const long r" + std::to_string(s.res) + " = " + std::to_string(info.vals.at(s.res)) + "L;\n";
    c += "    long rem = idx;\n";
    for (size_t i = indep.size(); i-- > 0;) {
      const int l = indep[i];
      const long ext = info.bounds[l].second - info.bounds[l].first;
      c += "    const long r" + std::to_string(k.loops[l].reg) + " = " + std::to_string(info.bounds[l].first) + "L + rem % " +
           std::to_string(ext) + "L; rem /= " + std::to_string(ext) + "L;\n";
    }
    c += "    float acc = 0.0f;\n";
    for (int l : red) {
      const std::string r = "r" + std::to_string(k.loops[l].reg);
      c += "    for (long " + r + " = " + std::to_string(info.bounds[l].first) + "L; " + r + " < " +
           std::to_string(info.bounds[l].second) + "L; ++" + r + ") {\n";
    }
    for (auto& rd : k.reads) c += "      const float r" + std::to_string(rd.reg) + " = " + element(rd) + ";\n";
    for (auto& ins : k.instrs) {
      const Ty t = ty[ins.res];
      const char* ctype = t == Ty::Scalar ? "float" : (t == Ty::Index ? "long" : "bool");
      std::string special;
      if (ins.kind == IK::Epoch) {
        special = "EP";
      } else if (ins.kind == IK::Shape || ins.kind == IK::Len || ins.kind == IK::ShapeLen) {
        const std::vector<long>& shp = shapes.at(ins.tensor);
        long v = 0;
        if (ins.kind == IK::Len) v = prodv(shp);
        else if (ins.kind == IK::ShapeLen) v = (long)shp.size();
        else {
          int d = ins.dim < 0 ? ins.dim + (int)shp.size() : ins.dim;
          v = (d >= 0 && d < (int)shp.size()) ? shp[d] : 0;
        }
        special = std::to_string(v) + "L";
      }
      std::string e = instr_expression(ins, special, "r");
      if (k.is_seed && ins.kind == IK::Scalar) e = "GS";
      c += std::string("      const ") + ctype + " r" + std::to_string(ins.res) + " = " + e + ";\n";
    }
    c += "      acc = acc + r" + std::to_string(k.result) + ";\n";
    for (size_t i = 0; i < red.size(); ++i) c += "    }\n";
    const std::string w = element(k.write);
    c += "    " + w + " = " + w + " + acc;\n  }\n";
    if (barrier) c += "  __syncthreads();\n";
  }
  return c;
}

int generate_small_group(const Program& prog, const std::vector<Kernel>& all, const std::vector<KernelInfo>& infos,
                         const Shapes& shapes, SmallGroup& g) {
  (void)prog;
  std::set<int> written, touched;
  for (int ki : g.kernel_index) {
    written.insert(all[ki].write.tensor);
    touched.insert(all[ki].write.tensor);
    for (auto& rd : all[ki].reads) touched.insert(rd.tensor);
  }
  g.ptr_args.assign(touched.begin(), touched.end());
  std::string sig = "extern \"C\" __global__ void __launch_bounds__(256) " + g.name + "(";
  for (size_t i = 0; i < g.ptr_args.size(); ++i) {
    const int t = g.ptr_args[i];
    sig += (i ? ", " : "") + std::string(written.count(t) ? "float* t" : "const float* t") + std::to_string(t);
  }
  sig += std::string(g.ptr_args.empty() ? "" : ", ") + "float GS, long EP)";
  std::string c;
  for (size_t gi = 0; gi < g.kernel_index.size(); ++gi)
    c += small_kernel_body(all[g.kernel_index[gi]], infos[g.kernel_index[gi]], shapes, "t", (int)gi);
  g.source = sig + " {\n" + c + "}\n";
  return EG_OK;
}

bool is_map_kernel(const Program& prog, const Kernel& k, const KernelInfo& info, const Shapes& shapes, long& count) {
  (void)prog;
  if (!info.ok || !k.index_instrs.empty() || k.loops.size() != 1 || k.gen != Gen::None) return false;
  if (!k.setup.empty() && !k.is_seed) return false;
  const int it = k.loops[0].reg;
  std::vector<const Op*> ops;
  for (auto& rd : k.reads) ops.push_back(&rd);
  ops.push_back(&k.write);
  auto ws = shapes.find(k.write.tensor);
  if (ws == shapes.end()) return false;
  count = prodv(ws->second);
  if (count <= 0 || info.bounds[0].first != 0 || info.bounds[0].second != count) return false;
  for (const Op* op : ops) {
    auto sh = shapes.find(op->tensor);
    if (sh == shapes.end() || prodv(sh->second) != count) return false;
    if (!op->raw || op->dims.size() != 1 || op->dims[0].only_register() != it) return false;
  }
  return true;
}

int generate_map_group(const Program& prog, const std::vector<Kernel>& all, const std::vector<KernelInfo>& infos,
                       const Shapes& shapes, SmallGroup& g) {
  std::set<int> written, touched;
  std::vector<long> counts;  // distinct element counts, in order of first appearance = the segments
  std::vector<long> of_kernel;
  for (int ki : g.kernel_index) {
    long n = 0;
    if (!is_map_kernel(prog, all[ki], infos[ki], shapes, n)) {
      set_error("internal: kernel %d is not an elementwise map", ki);
      return EG_ERR_INVALID;
    }
    of_kernel.push_back(n);
    if (std::find(counts.begin(), counts.end(), n) == counts.end()) counts.push_back(n);
    written.insert(all[ki].write.tensor);
    touched.insert(all[ki].write.tensor);
    for (auto& rd : all[ki].reads) touched.insert(rd.tensor);
  }
  for (auto& f : g.fold_offset) written.insert(f.first);  // (the folded total is stored where the gradient lives)
  g.ptr_args.assign(touched.begin(), touched.end());
  std::string sig = "extern \"C\" __global__ void __launch_bounds__(256) " + g.name + "(";
  for (size_t i = 0; i < g.ptr_args.size(); ++i) {
    const int t = g.ptr_args[i];
    sig += (i ? ", " : "") + std::string(written.count(t) ? "float* t" : "const float* t") + std::to_string(t);
  }
  sig += std::string(g.ptr_args.empty() ? "" : ", ") + "float GS, long EP";
  if (!g.fold_offset.empty()) sig += ", const float* __restrict__ slab, long FOLD";
  sig += ")";
  std::string c = "  const long block = blockIdx.x;\n";
  long first_block = 0;
  for (size_t seg = 0; seg < counts.size(); ++seg) {
    const long n = counts[seg], nblocks = (n + 255) / 256;
    c += std::string(seg ? "  else if" : "  if") + " (block < " + std::to_string(first_block + nblocks) + "L) {  // " +
         std::to_string(n) + " elements\n";
    c += "    const long idx = (block - " + std::to_string(first_block) + "L) * 256 + threadIdx.x;\n";
    c += "    if (idx < " + std::to_string(n) + "L) {\n";
    for (auto& f : g.fold_offset) {
      if (prodv(shapes.at(f.first)) != n || !touched.count(f.first)) continue;
      bool in_segment = false;
      for (size_t gi = 0; gi < g.kernel_index.size(); ++gi)
        if (of_kernel[gi] == n)
          for (auto& rd : all[g.kernel_index[gi]].reads) in_segment = in_segment || rd.tensor == f.first;
      if (!in_segment) continue;
      c += "      if (FOLD) {  // this element's sum over the batch: the samples' contributions in sample order\n";
      c += "        float total = 0.0f;\n";
      c += "        for (long s = 0; s < " + std::to_string(g.fold_rows) + "L; ++s) total = total + slab[s * " + std::to_string(g.fold_row_floats) +
           "L + " + std::to_string(f.second) + "L + idx];\n";
      c += "        t" + std::to_string(f.first) + "[idx] = total;\n      }\n";
    }
    for (size_t gi = 0; gi < g.kernel_index.size(); ++gi) {
      if (of_kernel[gi] != n) continue;
      const Kernel& k = all[g.kernel_index[gi]];
      const KernelInfo& info = infos[g.kernel_index[gi]];
      const std::vector<Ty> ty = infer_types(k);
      c += "      {  // " + to_text(k).substr(0, 100) + "\n";
      for (auto& s : k.setup) c += "        const long r" + std::to_string(s.res) + " = " + std::to_string(info.vals.at(s.res)) + "L;\n";
      c += "        const long r" + std::to_string(k.loops[0].reg) + " = idx;\n";
      for (auto& rd : k.reads) c += "        const float r" + std::to_string(rd.reg) + " = t" + std::to_string(rd.tensor) + "[idx];\n";
      for (auto& ins : k.instrs) {
        const Ty t = ty[ins.res];
        const char* ctype = t == Ty::Scalar ? "float" : (t == Ty::Index ? "long" : "bool");
        std::string special;
        if (ins.kind == IK::Epoch) {
          special = "EP";
        } else if (ins.kind == IK::Shape || ins.kind == IK::Len || ins.kind == IK::ShapeLen) {
          const std::vector<long>& shp = shapes.at(ins.tensor);
          long v = 0;
          if (ins.kind == IK::Len) v = prodv(shp);
          else if (ins.kind == IK::ShapeLen) v = (long)shp.size();
          else {
            int d = ins.dim < 0 ? ins.dim + (int)shp.size() : ins.dim;
            v = (d >= 0 && d < (int)shp.size()) ? shp[d] : 0;
          }
          special = std::to_string(v) + "L";
        }
        std::string e = instr_expression(ins, special, "r");
        if (k.is_seed && ins.kind == IK::Scalar) e = "GS";
        c += std::string("        const ") + ctype + " r" + std::to_string(ins.res) + " = " + e + ";\n";
      }
      const std::string w = "t" + std::to_string(k.write.tensor) + "[idx]";
      c += "        " + w + " = " + w + " + (0.0f + r" + std::to_string(k.result) + ");\n      }\n";
    }
    c += "    }\n  }\n";
    first_block += nblocks;
  }
  g.blocks = first_block;
  g.source = sig + " {\n" + c + "}\n";
  return EG_OK;
}


// ---------------------------------------------------------------------------------- sample groups

SampleKernelInfo analyse_sample_kernel(const Program& prog, const Kernel& k, const KernelInfo& info, const Shapes& shapes, long B) {
  (void)prog;
  SampleKernelInfo r;
  if (!info.ok || B <= 0 || k.gen != Gen::None) return r;
  std::vector<const Op*> ops;
  for (auto& rd : k.reads) ops.push_back(&rd);
  ops.push_back(&k.write);
  for (const Op* op : ops)
    if (!shapes.count(op->tensor)) return r;
  long all_work = 1;
  for (size_t l = 0; l < k.loops.size(); ++l) all_work *= std::max(0L, info.bounds[l].second - info.bounds[l].first);
  if (k.is_seed) {
    if (all_work > 4096 || !k.index_instrs.empty()) return r;
    r.ok = true;
    r.seed = true;
    r.work = all_work;
    return r;
  }
  auto used_as_value = [&](int y) {
    for (auto& ins : k.instrs)
      for (int a : ins.args)
        if (a == y) return true;
    for (auto& ins : k.index_instrs)
      for (int a : ins.args)
        if (a == y) return true;
    return false;
  };
  for (size_t l = 0; l < k.loops.size(); ++l) {
    const int y = k.loops[l].reg;
    const long lo = info.bounds[l].first, hi = info.bounds[l].second;
    if (lo != 0 || used_as_value(y)) continue;
    bool any_raw = false, any = false, ok = true;
    for (const Op* op : ops)
      if (op_has(*op, y) && op->raw) any_raw = true;
    bool raw = false;
    long inner = 0;
    if (hi == B && !any_raw) {
      for (const Op* op : ops) {
        if (!op_has(*op, y)) continue;
        any = true;
        const std::vector<long>& shp = shapes.at(op->tensor);
        if (op->raw || shp.empty() || shp[0] != B || op->dims.empty() || op->dims[0].only_register() != y) ok = false;
        for (size_t d = 1; ok && d < op->dims.size(); ++d)
          if (lin_has(op->dims[d], y)) ok = false;
      }
    } else if (any_raw && hi >= B && hi % B == 0) {
      raw = true;
      inner = hi / B;
      for (const Op* op : ops) {
        if (!op_has(*op, y)) continue;
        any = true;
        const std::vector<long>& shp = shapes.at(op->tensor);
        if (!op->raw || op->dims.size() != 1 || op->dims[0].only_register() != y || shp.empty() || shp[0] != B || prodv(shp) != hi)
          ok = false;
      }
      if (!op_has(k.write, y)) ok = false;
    } else {
      continue;
    }
    if (!ok || !any) continue;
    r.ok = true;
    r.batch_loop = (int)l;
    r.raw = raw;
    r.inner = inner;
    r.reduced = !op_has(k.write, y);
    r.work = raw ? inner : all_work / B;
    return r;
  }
  return r;
}

namespace {

// literal element offset of a tensor op (extents are literals in a per-plan kernel)
std::string literal_element(const Op& op, const Shapes& shapes, const std::string& name, long local_inner = 0) {
  const std::vector<long>& shp = shapes.at(op.tensor);
  const std::map<int, std::string> no_subst;
  std::string idx;
  if (op.raw) {
    idx = lin_text(op.dims[0], no_subst);
  } else {
    long stride = 1;
    idx = "0L";
    for (size_t d = shp.size(); d-- > 0;) {
      idx += " + " + std::to_string(stride) + "L * " + lin_text(op.dims[d], no_subst);
      stride *= shp[d];
    }
  }
  // a block's own slice of a [B, ...] tensor (kept in LDS): the same element, counted from the start of sample n
  if (local_inner > 0) idx = "(" + idx + ") - n * " + std::to_string(local_inner) + "L";
  return name + "[" + idx + "]";
}

std::string sample_instr(const Kernel& k, const Instr& ins, const std::vector<Ty>& ty, const Shapes& shapes) {
  const Ty t = ty[ins.res];
  const char* ctype = t == Ty::Scalar ? "float" : (t == Ty::Index ? "long" : "bool");
  std::string special;
  if (ins.kind == IK::Epoch) {
    special = "EP";
  } else if (ins.kind == IK::Shape || ins.kind == IK::Len || ins.kind == IK::ShapeLen) {
    const std::vector<long>& shp = shapes.at(ins.tensor);
    long v = 0;
    if (ins.kind == IK::Len) v = prodv(shp);
    else if (ins.kind == IK::ShapeLen) v = (long)shp.size();
    else {
      int d = ins.dim < 0 ? ins.dim + (int)shp.size() : ins.dim;
      v = (d >= 0 && d < (int)shp.size()) ? shp[d] : 0;
    }
    special = std::to_string(v) + "L";
  }
  std::string e = instr_expression(ins, special, "r");
  if (k.is_seed && ins.kind == IK::Scalar) e = "GS";
  return std::string("        const ") + ctype + " r" + std::to_string(ins.res) + " = " + e + ";\n";
}

}  // namespace

int generate_sample_group(const Program& prog, const std::vector<Kernel>& all, const std::vector<KernelInfo>& infos,
                          const Shapes& shapes, SampleGroup& g) {
  (void)prog;
  std::set<int> touched, written;
  for (size_t i = 0; i < g.kernel_index.size(); ++i) {
    const Kernel& k = all[g.kernel_index[i]];
    for (auto& rd : k.reads) touched.insert(rd.tensor);
    if (!g.infos[i].reduced) {
      touched.insert(k.write.tensor);
      written.insert(k.write.tensor);
    }
  }
  g.ptr_args.clear();
  for (int t : touched)
    if (!g.lds.count(t)) g.ptr_args.push_back(t);
  const std::string NT = std::to_string(g.threads);
  std::string sig = "extern \"C\" __global__ void __launch_bounds__(" + NT + ") " + g.name + "(float* __restrict__ slab";
  // (a staged parameter's argument is g<id>; t<id> names its copy in LDS, so the members' text does not change)
  for (int t : g.ptr_args) sig += std::string(written.count(t) ? ", float* t" : (g.staged.count(t) ? ", const float* __restrict__ g" : ", const float* t")) + std::to_string(t);
  sig += ", float GS, long EP)";
  std::string c = "  const long n = blockIdx.x;  // this block's sample\n";
  // Staged parameters: ALL loads first (literal trip counts, one register each), then the stores.  As one copy loop per
  // parameter the compiler emitted load - wait - store round trips one after the other (rolled loops for the longer ones):
  // ten dependent trips to L2 in front of the first member, 5 - 6 us of a 23 us kernel (ISA of the batch-32 fit step).
  if (!g.staged.empty()) {
    std::string loads, stores;
    long nv = 0;
    for (auto& kv : g.staged) {
      const std::string id = std::to_string(kv.first);
      c += "  __shared__ __attribute__((aligned(16))) float t" + id + "[" + std::to_string(kv.second) + "];\n";
      for (long base = 0; base < kv.second; base += g.threads, ++nv) {
        const std::string v = "sv" + std::to_string(nv), i = "threadIdx.x + " + std::to_string(base);
        const bool whole = base + g.threads <= kv.second;
        const std::string guard = whole ? "" : "if (threadIdx.x < " + std::to_string(kv.second - base) + ") ";
        loads += "    float " + v + " = 0.0f; " + guard + v + " = g" + id + "[" + i + "];\n";
        stores += "    " + guard + "t" + id + "[" + i + "] = " + v + ";\n";
      }
    }
    c += "  {\n" + loads + stores + "  }\n";
  }
  for (auto& kv : g.lds) c += "  __shared__ __attribute__((aligned(16))) float t" + std::to_string(kv.first) + "[" + std::to_string(kv.second) + "];\n";
  for (int t : g.lds_zero)
    c += "  for (int i = threadIdx.x; i < " + std::to_string(g.lds.at(t)) + "; i += " + NT + ") t" + std::to_string(t) + "[i] = 0.0f;\n";
  // ONE barrier behind the prologue (staged copies, zeroed tensors, the four zeros of the image-gradient gathers); it goes
  // too when the first member touches none of that (the copy of a sample's image: its loads then travel with the staged ones)
  c += "/*zeros4*/";
  size_t prologue_barrier = std::string::npos;
  if (!g.staged.empty() || !g.lds_zero.empty()) {
    prologue_barrier = c.size();
    c += "  __syncthreads();\n";
  }
  auto local = [&](int tensor) -> long {
    auto it = g.lds.find(tensor);
    return it == g.lds.end() ? 0 : it->second;
  };
  if (g.slab_floats > 0) c += "  float* const row = slab + n * " + std::to_string(g.slab_floats) + "L;\n";
  std::set<int> slab_seen;
  long scratch_floats = 0;
  bool need_dummy = false;    // a slot per thread that row blocks store the lanes outside their tensor to
  std::vector<size_t> member_start;   // offset of every emitted member's text in c
  bool need_zeros4 = false;   // four zeros in LDS: what an image-gradient member's gather reads outside the output
  // EG_SAMPLE_STOP=<k> (tuning aid): the kernel ends behind member k — wrong numbers, the time of the first k + 1 members
  const long stop = eg::sw::integer("EG_SAMPLE_STOP", -1);
  for (size_t gi = 0; gi < g.kernel_index.size(); ++gi) {
    if (stop >= 0 && (long)gi > stop) break;
    const Kernel& k = all[g.kernel_index[gi]];
    const KernelInfo& info = infos[g.kernel_index[gi]];
    const SampleKernelInfo& si = g.infos[gi];
    const std::vector<Ty> ty = infer_types(k);
    member_start.push_back(c.size());
    c += "  {  // kernel " + std::to_string(gi) + ": " + to_text(k).substr(0, 100) + "\n";
    if (si.conv_role != 0) {
      // ---- matrix-core convolution members (round 6).  The scalar members spend their time in LDS reads (two per multiply-add:
      // conv1 forward 7.5 us, its filter gradient 12.3, conv2's three 3.5 + 5.4 + 4.9 of the 31 us kernel at batch 32).  Here a
      // member is a handful of v_mfma_f32_16x16x4_f32 per wave: the SMALL operand (filter bank / its transpose) sits in
      // registers as B fragments for the whole member, the other fragment is ONE gathered element per lane and instruction
      // (window element, output gradient), the eight waves share row blocks (forward, image gradient) or the pixel range
      // (filter gradient: the waves' accumulator blocks meet in LDS in wave order — a fixed order).  Padding lanes multiply by
      // an exact 0.0f from the small operand (or read an element that is part of the true sum), never uninitialised memory.
      const std::vector<long>&is = shapes.at(si.conv_img), &os = shapes.at(si.conv_out), &fs = shapes.at(si.conv_flt);
      const long H = is[1], W = is[2], C = is[3], Ho = os[1], Wo = os[2], F = os[3], FH = fs[1], FW = fs[2];
      (void)H;
      const long K = FH * FW * C, P = Ho * Wo, Q = is[1] * is[2];
      const long NW = g.threads / 64;
      auto S = [](long v) { return std::to_string(v); };
      auto at = [&](int tensor, const std::string& idx) {   // element idx of this sample's slice (the filter bank: of the bank)
        const std::string name = "t" + std::to_string(tensor);
        if (tensor == si.conv_flt) return name + "[" + idx + "]";
        const long inner = prodv(shapes.at(tensor)) / std::max(1L, shapes.at(tensor)[0]);
        return local(tensor) ? name + "[" + idx + "]" : name + "[n * " + S(inner) + "L + " + idx + "]";
      };
      c += "    typedef float mf4 __attribute__((ext_vector_type(4)));\n";
      c += "    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, l4 = lane >> 4;\n";
      // Shapes of the loops below (round 6, after the cycle stamps of EG_SAMPLE_TRACE):
      //  * a wave's row blocks are a lambda called with a LITERAL trip count (the whole trips; the ragged one under a
      //    wave-uniform guard) so that the gathers of block i + 1 are in flight under the MFMAs of block i — the rolled
      //    `for (pb = wave; ...)` exposed the LDS latency and the whole dependent MFMA chain of every block; k-steps
      //    alternate between two accumulators (one chain of KS dependent MFMAs becomes two of KS / 2);
      //  * gathers of FOUR k-values per lane and instruction: with the channels (forward) / the filters (image gradient) a
      //    multiple of 4 and the gathered tensor in LDS, the k index is permuted so that lane group l4 holds
      //    k = 16 g + 4 l4 + j in the j-th MFMA of group g — four consecutive channels of ONE tap, one ds_read_b128, one
      //    address, one bounds test.  Any bijection of k is the same sum; the B fragments use the same one;
      //  * a forward member's row block of 16 output pixels is 16 consecutive pixels of the row-major image or — when the
      //    output tiles exactly into bw x (16 / bw) patches — such a patch, and a wave then takes whole ROWS of patches:
      //    patch row and column are literals at every call, every gather and store a per-lane base plus a literal
      //    offset, no pixel of a block lies past the end (the store tests the filter only).
      // A row block's store without a branch when the destination lives in LDS: lanes outside the tensor write their value
      // to a slot of `dummy_` of their own instead (a guarded store is a basic-block boundary the compiler moves no gather
      // across).
      auto guarded_store = [&](int tensor, const std::string& cond, const std::string& idx, const std::string& value) {
        const std::string o = at(tensor, idx);
        if (!local(tensor))
          return "          if (" + cond + ") " + o + " = " + (g.overwrite[gi] ? std::string("0.0f") : o) + " + " + value + ";\n";
        need_dummy = true;   // (not `scratch`: a member behind an elided barrier may be using that)
        std::string d = "          float* const dst_ = (" + cond + ") ? &" + o + " : &dummy_[threadIdx.x];\n";
        d += "          *dst_ = " + (g.overwrite[gi] ? std::string("0.0f") : std::string("*dst_")) + " + " + value + ";\n";
        return d;
      };
      auto trips = [&](const std::string& fn, long units) {   // calls fn(unit) for unit = wave, wave + NW, ...
        std::string d;
        const long whole = units / NW, ragged = units % NW;
        if (whole > 0) d += "    _Pragma(\"unroll\") for (int it_ = 0; it_ < " + S(whole) + "; ++it_) " + fn + "(wave + it_ * " + S(NW) + ");\n";
        if (ragged > 0) d += "    if (wave < " + S(ragged) + ") " + fn + "(wave + " + S(whole * NW) + ");\n";
        return d;
      };
      std::string fw_head, fw_store, fw_pre, fw_calls;
      long fw_blocks = 1;   // row blocks per call of the member's lambda
      if (si.conv_role == 1) {
        long bw = 0;
        for (long cand : {16L, 8L, 4L, 2L})
          if (!bw && Wo % cand == 0 && Ho % (16 / cand) == 0) bw = cand;
        const std::string value = "(acc[bi][0][nb][j] + acc[bi][1][nb][j])";
        if (bw) {
          // patch row = wave + NW * ri
          const long bh = 16 / bw, nbc = Wo / bw, nbr = Ho / bh, whole = nbr / NW, ragged = nbr % NW;
          fw_pre = "    const int wu = __builtin_amdgcn_readfirstlane(wave);\n";
          fw_blocks = nbc <= 4 ? nbc : 1;
          if (whole > 0)
            fw_calls += "    _Pragma(\"unroll\") for (int ri = 0; ri < " + S(whole) + "; ++ri) _Pragma(\"unroll\") for (int bc = 0; bc < " + S(nbc) +
                        "; bc += " + S(fw_blocks) + ") block(ri * " + S(nbc) + " + bc);\n";
          if (ragged > 0)
            fw_calls += "    if (wu < " + S(ragged) + ") { _Pragma(\"unroll\") for (int bc = 0; bc < " + S(nbc) + "; bc += " + S(fw_blocks) + ") block(" +
                        S(whole * nbc) + " + bc); }\n";
          fw_head = "      const int br = wu + " + S(NW) + " * (pb / " + S(nbc) + "), bc = pb % " + S(nbc) + ";\n";
          fw_head += "      const int poff = ((br * " + S(bh) + " + l15 / " + S(bw) + ") * " + S(W) + " + bc * " + S(bw) + " + l15 % " + S(bw) + ") * " + S(C) + ";\n";
          fw_store = "          const int m_ = 4 * l4 + j, f = 16 * nb + l15;\n";
          fw_store += guarded_store(si.conv_out, "f < " + S(F),
                                    "((br * " + S(bh) + " + m_ / " + S(bw) + ") * " + S(Wo) + " + bc * " + S(bw) + " + m_ % " + S(bw) + ") * " + S(F) + " + f", value);
        } else {
          fw_head = "      int p = 16 * pb + l15;\n      if (p > " + S(P - 1) + ") p = " + S(P - 1) + ";\n";
          fw_head += "      const int poff = ((p / " + S(Wo) + ") * " + S(W) + " + p % " + S(Wo) + ") * " + S(C) + ";\n";
          fw_store = "          const int pr = 16 * pb + 4 * l4 + j, f = 16 * nb + l15;\n";
          fw_store += guarded_store(si.conv_out, "pr < " + S(P) + " && f < " + S(F), "pr * " + S(F) + " + f", value);
          fw_calls = trips("block", (P + 15) / 16);
        }
      }
      if (si.conv_role == 1 && C % 4 == 0 && local(si.conv_img)) {
        const long G = (K + 15) / 16, NB = (F + 15) / 16;
        c += "    float bf[" + S(NB) + "][" + S(4 * G) + "];\n    int toff[" + S(G) + "];\n";
        c += "    _Pragma(\"unroll\") for (int g4 = 0; g4 < " + S(G) + "; ++g4) {\n";
        c += "      const int t0 = 16 * g4 + 4 * l4, tc = t0 < " + S(K) + " ? t0 : 0;\n";
        c += "      toff[g4] = ((tc / " + S(FW * C) + ") * " + S(W) + " + (tc / " + S(C) + ") % " + S(FW) + ") * " + S(C) + " + tc % " + S(C) + ";\n";
        c += "      _Pragma(\"unroll\") for (int nb = 0; nb < " + S(NB) + "; ++nb) {\n        const int f = 16 * nb + l15;\n";
        // (columns f >= F of B are never stored: they read filter F - 1 instead of a masked zero; only k past the end is masked)
        c += "        const int fc = f < " + S(F) + " ? f : " + S(F - 1) + ";\n";
        if (K % 16 == 0) {
          c += "        _Pragma(\"unroll\") for (int j = 0; j < 4; ++j) bf[nb][4 * g4 + j] = " + at(si.conv_flt, "fc * " + S(K) + " + t0 + j") + ";\n      }\n    }\n";
        } else {
          c += "        const bool in_ = t0 < " + S(K) + ";\n";
          c += "        _Pragma(\"unroll\") for (int j = 0; j < 4; ++j) {\n          const float bv = " + at(si.conv_flt, "fc * " + S(K) + " + (in_ ? t0 : 0) + j") +
               ";\n          bf[nb][4 * g4 + j] = in_ ? bv : 0.0f;\n        }\n      }\n    }\n";
        }
        // A call covers fw_blocks row blocks (a whole row of patches, or one block): first the gathers and MFMAs of ALL of
        // them — one straight-line run the scheduler can interleave: block by block, every block's LDS latency and dependent
        // MFMA chain stood in line (conv1 forward: 4 300 cycles for 42 MFMAs per wave whatever was trimmed around them) —
        // then all their stores.
        c += fw_pre + "    auto block = [&](const int pb0) {\n";
        c += "      mf4 acc[" + S(fw_blocks) + "][2][" + S(NB) + "];\n";
        c += "      _Pragma(\"unroll\") for (int bi = 0; bi < " + S(fw_blocks) + "; ++bi) {\n      const int pb = pb0 + bi;\n" + fw_head;
        c += "      _Pragma(\"unroll\") for (int nb = 0; nb < " + S(NB) + "; ++nb) acc[bi][0][nb] = acc[bi][1][nb] = mf4{0.0f, 0.0f, 0.0f, 0.0f};\n";
        c += "      _Pragma(\"unroll\") for (int g4 = 0; g4 < " + S(G) + "; ++g4) {\n";
        c += "        const mf4 a4 = *reinterpret_cast<const mf4*>(&" + at(si.conv_img, "poff + toff[g4]") + ");\n";
        c += "        _Pragma(\"unroll\") for (int j = 0; j < 4; ++j)\n          _Pragma(\"unroll\") for (int nb = 0; nb < " + S(NB) +
             "; ++nb) acc[bi][j & 1][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[j], bf[nb][4 * g4 + j], acc[bi][j & 1][nb], 0, 0, 0);\n      }\n";
        c += "      }\n      _Pragma(\"unroll\") for (int bi = 0; bi < " + S(fw_blocks) + "; ++bi) {\n      const int pb = pb0 + bi;\n" + fw_head + "      (void)poff;\n";
        c += "      _Pragma(\"unroll\") for (int nb = 0; nb < " + S(NB) + "; ++nb)\n        _Pragma(\"unroll\") for (int j = 0; j < 4; ++j) {\n";
        c += fw_store + "        }\n      }\n    };\n";
        c += fw_calls;
      } else if (si.conv_role == 3 && F % 4 == 0 && local(si.conv_out)) {
        const long KD = FH * FW * F, G = (KD + 15) / 16, NB = (C + 15) / 16, QB = (Q + 15) / 16;
        // (F a multiple of 16: the tap of group g4 is a literal)
        const std::string tap_s = F % 16 == 0 ? "g4 / " + S(F / 16) : "k0 / " + S(F);
        const std::string tap_f = F % 16 == 0 ? "16 * (g4 % " + S(F / 16) + ") + 4 * l4" : "k0 % " + S(F);
        c += "    float bf[" + S(NB) + "][" + S(4 * G) + "];\n";
        c += "    _Pragma(\"unroll\") for (int g4 = 0; g4 < " + S(G) + "; ++g4) {\n      const int k0 = 16 * g4 + 4 * l4, s = " + tap_s + ", f0 = " + tap_f + ";\n";
        c += "      _Pragma(\"unroll\") for (int nb = 0; nb < " + S(NB) + "; ++nb) {\n        const int ch = 16 * nb + l15;\n";
        c += "        const int cc = ch < " + S(C) + " ? ch : " + S(C - 1) + ";   // (columns ch >= C are never stored)\n";
        if (KD % 16 == 0) {
          c += "        _Pragma(\"unroll\") for (int j = 0; j < 4; ++j) bf[nb][4 * g4 + j] = " + at(si.conv_flt, "(f0 + j) * " + S(K) + " + s * " + S(C) + " + cc") + ";\n      }\n    }\n";
        } else {
          c += "        const bool in_ = k0 < " + S(KD) + ";\n";
          c += "        _Pragma(\"unroll\") for (int j = 0; j < 4; ++j) {\n          const float bv = " +
               at(si.conv_flt, "(in_ ? (f0 + j) * " + S(K) + " + s * " + S(C) + " + cc : 0)") + ";\n          bf[nb][4 * g4 + j] = in_ ? bv : 0.0f;\n        }\n      }\n    }\n";
        }
        c += "    auto block = [&](const int qb) {\n";
        c += "      int q = 16 * qb + l15;\n      if (q > " + S(Q - 1) + ") q = " + S(Q - 1) + ";\n      const int qy = q / " + S(W) + ", qx = q % " + S(W) + ";\n";
        c += "      mf4 acc[2][" + S(NB) + "];\n      _Pragma(\"unroll\") for (int nb = 0; nb < " + S(NB) + "; ++nb) acc[0][nb] = acc[1][nb] = mf4{0.0f, 0.0f, 0.0f, 0.0f};\n";
        c += "      _Pragma(\"unroll\") for (int g4 = 0; g4 < " + S(G) + "; ++g4) {\n";
        c += "        const int k0 = 16 * g4 + 4 * l4, s = " + tap_s + ", f0 = " + tap_f + ", y = qy - s / " + S(FW) + ", x = qx - s % " + S(FW) + ";\n";
        c += "        const bool ok = k0 < " + S(KD) + " && y >= 0 && y < " + S(Ho) + " && x >= 0 && x < " + S(Wo) + ";\n";
        // (outside the output: the lane reads four zeros kept in LDS — one select of the address instead of four of the values)
        need_zeros4 = true;
        c += "        const float* const ap = ok ? &" + at(si.conv_out, "(y * " + S(Wo) + " + x) * " + S(F) + " + f0") + " : zeros4_;\n";
        c += "        const mf4 a4 = *reinterpret_cast<const mf4*>(ap);\n";
        c += "        _Pragma(\"unroll\") for (int j = 0; j < 4; ++j) {\n          const float a = a4[j];\n";
        c += "          _Pragma(\"unroll\") for (int nb = 0; nb < " + S(NB) +
             "; ++nb) acc[j & 1][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bf[nb][4 * g4 + j], acc[j & 1][nb], 0, 0, 0);\n        }\n      }\n";
        c += "      _Pragma(\"unroll\") for (int nb = 0; nb < " + S(NB) + "; ++nb)\n        _Pragma(\"unroll\") for (int j = 0; j < 4; ++j) {\n";
        c += "          const int qr = 16 * qb + 4 * l4 + j, ch = 16 * nb + l15;\n";
        c += guarded_store(si.conv_img, "qr < " + S(Q) + " && ch < " + S(C), "qr * " + S(C) + " + ch", "(acc[0][nb][j] + acc[1][nb][j])") + "        }\n    };\n";
        c += trips("block", QB);
      } else if (si.conv_role == 1) {   // out[p, f] (+)= sum_t img[pix(p) + tap(t)] * flt[f, t]
        const long KS = (K + 3) / 4, NB = (F + 15) / 16;
        c += "    float bf[" + S(NB) + "][" + S(KS) + "];\n    int toff[" + S(KS) + "];\n";
        c += "    _Pragma(\"unroll\") for (int ks = 0; ks < " + S(KS) + "; ++ks) {\n";
        c += "      const int t = 4 * ks + l4, tc = t < " + S(K) + " ? t : 0;\n";
        c += "      toff[ks] = ((tc / " + S(FW * C) + ") * " + S(W) + " + (tc / " + S(C) + ") % " + S(FW) + ") * " + S(C) + " + tc % " + S(C) + ";\n";
        c += "      _Pragma(\"unroll\") for (int nb = 0; nb < " + S(NB) + "; ++nb) {\n        const int f = 16 * nb + l15;\n";
        c += "        const bool in_ = t < " + S(K) + ";\n        const float bv = " + at(si.conv_flt, "(f < " + S(F) + " ? f : " + S(F - 1) + ") * " + S(K) + " + tc") + ";\n";
        c += "        bf[nb][ks] = in_ ? bv : 0.0f;   // (columns f >= F are never stored: they repeat filter F - 1)\n      }\n    }\n";
        // A call covers fw_blocks row blocks (a whole row of patches, or one block): first the gathers and MFMAs of ALL of
        // them — one straight-line run the scheduler can interleave: block by block, every block's LDS latency and dependent
        // MFMA chain stood in line (conv1 forward: 4 300 cycles for 42 MFMAs per wave whatever was trimmed around them) —
        // then all their stores.
        c += fw_pre + "    auto block = [&](const int pb0) {\n";
        c += "      mf4 acc[" + S(fw_blocks) + "][2][" + S(NB) + "];\n";
        c += "      _Pragma(\"unroll\") for (int bi = 0; bi < " + S(fw_blocks) + "; ++bi) {\n      const int pb = pb0 + bi;\n" + fw_head;
        c += "      _Pragma(\"unroll\") for (int nb = 0; nb < " + S(NB) + "; ++nb) acc[bi][0][nb] = acc[bi][1][nb] = mf4{0.0f, 0.0f, 0.0f, 0.0f};\n";
        c += "      _Pragma(\"unroll\") for (int ks = 0; ks < " + S(KS) + "; ++ks) {\n";
        c += "        const float a = " + at(si.conv_img, "poff + toff[ks]") + ";\n";
        c += "        _Pragma(\"unroll\") for (int nb = 0; nb < " + S(NB) + "; ++nb) acc[bi][ks & 1][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bf[nb][ks], acc[bi][ks & 1][nb], 0, 0, 0);\n      }\n";
        c += "      }\n      _Pragma(\"unroll\") for (int bi = 0; bi < " + S(fw_blocks) + "; ++bi) {\n      const int pb = pb0 + bi;\n" + fw_head + "      (void)poff;\n";
        c += "      _Pragma(\"unroll\") for (int nb = 0; nb < " + S(NB) + "; ++nb)\n        _Pragma(\"unroll\") for (int j = 0; j < 4; ++j) {\n";
        c += fw_store + "        }\n      }\n    };\n";
        c += fw_calls;
      } else if (si.conv_role == 3) {   // gimg[q, ch] (+)= sum_{s, f} gout[pixel(q) - tap(s), f] * flt[f, s, ch]
        const long KD = FH * FW * F, KS = (KD + 3) / 4, NB = (C + 15) / 16, QB = (Q + 15) / 16;
        // With F a multiple of 4 the four lanes groups of a k-step share their tap: tap and window offset of k-step ks are
        // literals (the bounds test is per tap, not per k-step, and the gather's address is the pixel's base + a literal).
        const bool f4 = F % 4 == 0;
        const std::string tap_s = f4 ? "ks / " + S(F / 4) : "kk / " + S(F), tap_f = f4 ? "4 * (ks % " + S(F / 4) + ") + l4" : "kk % " + S(F);
        c += "    float bf[" + S(NB) + "][" + S(KS) + "];\n";
        c += "    _Pragma(\"unroll\") for (int ks = 0; ks < " + S(KS) + "; ++ks) {\n      const int kk = 4 * ks + l4, s = " + tap_s + ", f = " + tap_f + ";\n";
        c += "      _Pragma(\"unroll\") for (int nb = 0; nb < " + S(NB) + "; ++nb) {\n        const int ch = 16 * nb + l15;\n";
        c += "        const bool in_ = kk < " + S(KD) + ";\n        const float bv = " + at(si.conv_flt, "(in_ ? f * " + S(K) + " + s * " + S(C) + " + (ch < " + S(C) + " ? ch : " + S(C - 1) + ") : 0)") + ";\n";
        c += "        bf[nb][ks] = in_ ? bv : 0.0f;   // (columns ch >= C are never stored)\n      }\n    }\n";
        c += "    auto block = [&](const int qb) {\n";
        c += "      int q = 16 * qb + l15;\n      if (q > " + S(Q - 1) + ") q = " + S(Q - 1) + ";\n      const int qy = q / " + S(W) + ", qx = q % " + S(W) + ";\n";
        c += "      mf4 acc[2][" + S(NB) + "];\n      _Pragma(\"unroll\") for (int nb = 0; nb < " + S(NB) + "; ++nb) acc[0][nb] = acc[1][nb] = mf4{0.0f, 0.0f, 0.0f, 0.0f};\n";
        c += "      _Pragma(\"unroll\") for (int ks = 0; ks < " + S(KS) + "; ++ks) {\n";
        c += "        const int kk = 4 * ks + l4, s = " + tap_s + ", f = " + tap_f + ", y = qy - s / " + S(FW) + ", x = qx - s % " + S(FW) + ";\n";
        c += "        const bool ok = kk < " + S(KD) + " && y >= 0 && y < " + S(Ho) + " && x >= 0 && x < " + S(Wo) + ";\n";
        c += "        const int go = ok ? (y * " + S(Wo) + " + x) * " + S(F) + " + f : 0;\n";
        c += "        const float av = " + at(si.conv_out, "go") + ";\n        const float a = ok ? av : 0.0f;\n";
        c += "        _Pragma(\"unroll\") for (int nb = 0; nb < " + S(NB) + "; ++nb) acc[ks & 1][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bf[nb][ks], acc[ks & 1][nb], 0, 0, 0);\n      }\n";
        c += "      _Pragma(\"unroll\") for (int nb = 0; nb < " + S(NB) + "; ++nb)\n        _Pragma(\"unroll\") for (int j = 0; j < 4; ++j) {\n";
        c += "          const int qr = 16 * qb + 4 * l4 + j, ch = 16 * nb + l15;\n";
        c += guarded_store(si.conv_img, "qr < " + S(Q) + " && ch < " + S(C), "qr * " + S(C) + " + ch", "(acc[0][nb][j] + acc[1][nb][j])") + "        }\n    };\n";
        c += trips("block", QB);
      } else {   // gflt[f, t] (+)= sum_p gout[p, f] * img[pix(p) + tap(t)]
        const long MB = (F + 15) / 16, NB = (K + 15) / 16, PS = (P + 3) / 4;
        const bool first = slab_seen.count(k.write.tensor) == 0;   // the first contribution of this block to that slab range
        const long off = g.slab_offset.at(k.write.tensor);
        c += "    int toff[" + S(NB) + "];\n";
        c += "    _Pragma(\"unroll\") for (int nb = 0; nb < " + S(NB) + "; ++nb) {\n      int t = 16 * nb + l15;\n      if (t > " + S(K - 1) + ") t = " + S(K - 1) + ";\n";
        c += "      toff[nb] = ((t / " + S(FW * C) + ") * " + S(W) + " + (t / " + S(C) + ") % " + S(FW) + ") * " + S(C) + " + t % " + S(C) + ";\n    }\n";
        c += "    mf4 acc[" + S(MB) + "][" + S(NB) + "];\n";
        c += "    _Pragma(\"unroll\") for (int mb = 0; mb < " + S(MB) + "; ++mb) _Pragma(\"unroll\") for (int nb = 0; nb < " + S(NB) + "; ++nb) acc[mb][nb] = mf4{0.0f, 0.0f, 0.0f, 0.0f};\n";
        // Output rows a multiple of 4 pixels wide: a wave takes WHOLE ROWS (row = wave + NW * ri) and walks them in literal
        // steps of 4 pixels, so every gather is a per-lane base (computed once) plus a literal offset — the linear walk
        // `p = 4 ps + l4` paid a division by Wo, two multiplies and three additions per step and lane (the members are bound
        // by instruction issue: EG_SAMPLE_TRACE, 5 000 cycles for 18 steps of 2 MFMAs).
        // (rows not a multiple of 4 wide but even, an even number of them: the four pixels of a step are a 2 x 2 tile)
        const long th = Wo % 4 == 0 ? 1 : 2, tw = 4 / th, trows = Ho / th;
        const bool by_rows = Wo % tw == 0 && Ho % th == 0 && (trows / NW + 1) * (Wo / tw) <= 40;
        if (by_rows) {
          // (rows go to the waves from the LAST one down: the row-block members in front of this one — no barrier in between
          // when they are independent — give their ragged extra block to the first waves)
          c += "    const int wu = " + S(NW - 1) + " - __builtin_amdgcn_readfirstlane(wave), lr = l4 / " + S(tw) + ", lc = l4 % " + S(tw) + ";\n";
          c += "    const int pbase = ((wu * " + S(th) + " + lr) * " + S(W) + " + lc) * " + S(C) + ";\n";
          c += "    int abase[" + S(MB) + "];\n    _Pragma(\"unroll\") for (int mb = 0; mb < " + S(MB) + "; ++mb) {\n      const int f = 16 * mb + l15;\n";
          c += "      abase[mb] = ((wu * " + S(th) + " + lr) * " + S(Wo) + " + lc) * " + S(F) + " + (f < " + S(F) + " ? f : 0);\n    }\n";
          c += "    auto rowstep = [&](const int ri) {\n      _Pragma(\"unroll\") for (int cg = 0; cg < " + S(Wo / tw) + "; ++cg) {\n";
          c += "        float a[" + S(MB) + "];\n        _Pragma(\"unroll\") for (int mb = 0; mb < " + S(MB) + "; ++mb) a[mb] = " +
               at(si.conv_out, "abase[mb] + (" + S(NW * th * Wo) + " * ri + " + S(tw) + " * cg) * " + S(F)) + ";\n";
          c += "        _Pragma(\"unroll\") for (int nb = 0; nb < " + S(NB) + "; ++nb) {\n          const float b = " +
               at(si.conv_img, "pbase + toff[nb] + (" + S(NW * th * W) + " * ri + " + S(tw) + " * cg) * " + S(C)) + ";\n";
          c += "          _Pragma(\"unroll\") for (int mb = 0; mb < " + S(MB) + "; ++mb) acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mb], b, acc[mb][nb], 0, 0, 0);\n        }\n      }\n    };\n";
          const long whole = trows / NW, ragged = trows % NW;
          if (whole > 0) c += "    _Pragma(\"unroll\") for (int ri = 0; ri < " + S(whole) + "; ++ri) rowstep(ri);\n";
          if (ragged > 0) c += "    if (wu < " + S(ragged) + ") rowstep(" + S(whole) + ");\n";
        }
        c += "    auto step = [&](const int ps) {\n";
        c += "      const int p = 4 * ps + l4, pc = p < " + S(P) + " ? p : " + S(P - 1) + ";\n";
        c += "      const int poff = ((pc / " + S(Wo) + ") * " + S(W) + " + pc % " + S(Wo) + ") * " + S(C) + ";\n";
        // (rows f >= F of the result are never stored: they repeat filter 0; pixels past the end are masked, if there are any)
        c += "      float a[" + S(MB) + "];\n      _Pragma(\"unroll\") for (int mb = 0; mb < " + S(MB) + "; ++mb) {\n        const int f = 16 * mb + l15, fc = f < " + S(F) + " ? f : 0;\n";
        c += "        const float av = " + at(si.conv_out, "pc * " + S(F) + " + fc") + ";\n        a[mb] = " + (P % 4 == 0 ? std::string("av") : "p < " + S(P) + " ? av : 0.0f") + ";\n      }\n";
        c += "      _Pragma(\"unroll\") for (int nb = 0; nb < " + S(NB) + "; ++nb) {\n        const float b = " + at(si.conv_img, "poff + toff[nb]") + ";\n";
        c += "        _Pragma(\"unroll\") for (int mb = 0; mb < " + S(MB) + "; ++mb) acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mb], b, acc[mb][nb], 0, 0, 0);\n      }\n    };\n";
        // (more than 24 whole trips: the unrolled body would not fit the instruction cache's reach; a rolled loop of 4)
        if (by_rows) {
          c += "    (void)step;\n";
        } else if (PS / NW <= 24) {
          c += trips("step", PS);
        } else {
          c += "    _Pragma(\"unroll 4\") for (int ps = wave; ps < " + S(PS) + "; ps += " + S(NW) + ") step(ps);\n";
        }
        // the waves' accumulator blocks meet in LDS — as many of the MB x NB at a time as 16 KB of scratch hold (the planner's
        // LDS budget leaves that much: plan_groups.cpp) — and are added in wave order
        const long BL = MB * NB, per_round = std::max(1L, 4096 / (NW * 256));
        for (long b0 = 0; b0 < BL; b0 += per_round) {
          const long nb_round = std::min(per_round, BL - b0);
          if (b0 > 0) c += "    __syncthreads();\n";
          for (long b = b0; b < b0 + nb_round; ++b)
            c += "    _Pragma(\"unroll\") for (int j = 0; j < 4; ++j) scratch[(" + S(b - b0) + " * " + S(NW) + " + wave) * 256 + (4 * l4 + j) * 16 + l15] = acc[" +
                 S(b / NB) + "][" + S(b % NB) + "][j];\n";
          c += "    __syncthreads();\n";
          c += "    _Pragma(\"unroll\") for (int e0 = 0; e0 < " + S(nb_round * 256) + "; e0 += " + NT + ") {\n      const int e = e0 + threadIdx.x, bl = " + S(b0) +
               " + (e >> 8), el = e & 255;\n";
          c += "      if (e < " + S(nb_round * 256) + ") {\n        float s = 0.0f;\n        _Pragma(\"unroll\") for (int w = 0; w < " + S(NW) +
               "; ++w) s = s + scratch[((e >> 8) * " + S(NW) + " + w) * 256 + el];\n";
          c += "        const int f = 16 * (bl / " + S(NB) + ") + (el >> 4), t = 16 * (bl % " + S(NB) + ") + (el & 15);\n";
          const std::string o = "row[" + S(off) + " + f * " + S(K) + " + t]";
          c += "        if (f < " + S(F) + " && t < " + S(K) + ") " + o + " = " + (first ? std::string("0.0f") : o) + " + s;\n      }\n    }\n";
        }
        scratch_floats = std::max(scratch_floats, NW * 256 * std::min(per_round, BL));
        slab_seen.insert(k.write.tensor);
      }
      c += "  }\n  __syncthreads();\n";
      continue;
    }
    if (si.gather) {
      // gimg[n, Y, X, c] = sum over (dy, dx, f) of gout[n, Y - dy, X - dx, f] * flt[f, dy, dx, c] where the output pixel exists
      const std::vector<long>&gi_s = shapes.at(si.g_img), &go_s = shapes.at(si.g_out), &fl_s = shapes.at(si.g_flt);
      const long H = gi_s[1], W = gi_s[2], C = gi_s[3], Ho = go_s[1], Wo = go_s[2], F = go_s[3], FH = fl_s[1], FW = fl_s[2];
      auto L = [](long v) { return std::to_string(v) + "L"; };
      const std::string img = "t" + std::to_string(si.g_img), out = "t" + std::to_string(si.g_out), flt = "t" + std::to_string(si.g_flt);
      c += "    for (long idx = threadIdx.x; idx < " + L(H * W * C) + "; idx += " + NT + ") {\n";
      c += "      const long ch = idx % " + L(C) + ", X = (idx / " + L(C) + ") % " + L(W) + ", Y = idx / " + L(C * W) + ";\n";
      c += "      float acc = 0.0f;\n";
      c += "      for (long dy = 0; dy < " + L(FH) + "; ++dy) {\n        const long y = Y - dy;\n        if (y < 0 || y >= " + L(Ho) + ") continue;\n";
      c += "        for (long dx = 0; dx < " + L(FW) + "; ++dx) {\n          const long x = X - dx;\n          if (x < 0 || x >= " + L(Wo) + ") continue;\n";
      c += "          const float* go = " + out + " + ((" + (local(si.g_out) ? std::string("0L") : std::string("n")) + " * " + L(Ho) + " + y) * " + L(Wo) + " + x) * " + L(F) + ";\n";
      c += "          const float* fl = " + flt + " + (dy * " + L(FW) + " + dx) * " + L(C) + " + ch;\n";
      c += "          for (long f = 0; f < " + L(F) + "; ++f) acc = acc + go[f] * fl[f * " + L(FH * FW * C) + "];\n        }\n      }\n";
      const std::string w = img + "[" + (local(si.g_img) ? std::string("0L") : std::string("n")) + " * " + L(H * W * C) + " + idx]";
      c += "      " + w + " = " + (g.overwrite[gi] ? std::string("0.0f") : w) + " + acc;\n    }\n  }\n  __syncthreads();\n";
      continue;
    }
    std::vector<int> indep, red;
    bool scatter;
    split_loops(k, indep, red, scatter);
    auto drop = [&](std::vector<int>& v, int l) { v.erase(std::remove(v.begin(), v.end(), l), v.end()); };
    long total = 1;
    std::string fix;  // the batch iterator of this block
    if (si.seed) {
      // every block writes the same values: no batch loop
    } else if (si.raw) {
      drop(indep, si.batch_loop);
      drop(red, si.batch_loop);
      total = si.inner;
      fix = "      const long r" + std::to_string(k.loops[si.batch_loop].reg) + " = n * " + std::to_string(si.inner) + "L + idx;\n";
    } else {
      drop(indep, si.batch_loop);
      drop(red, si.batch_loop);
      fix = "      const long r" + std::to_string(k.loops[si.batch_loop].reg) + " = n;\n";
    }
    if (!si.raw)
      for (int l : indep) total *= std::max(0L, info.bounds[l].second - info.bounds[l].first);
    if (si.raw && !indep.empty()) {
      set_error("internal: raw sample kernel with further independent loops");
      return EG_ERR_INVALID;
    }
    // where the value goes
    std::string w;
    bool plain = g.overwrite[gi] != 0;
    if (si.reduced) {
      const long off = g.slab_offset.at(k.write.tensor);
      const std::string e = literal_element(k.write, shapes, "row");
      w = e.substr(0, 4) + std::to_string(off) + "L + " + e.substr(4);   // row[<off> + index]
      plain = slab_seen.count(k.write.tensor) == 0;  // the first contribution of this block to that tensor
    } else {
      w = literal_element(k.write, shapes, "t" + std::to_string(k.write.tensor), local(k.write.tensor));
    }
    long rtotal = 1;
    for (int l : red) rtotal *= std::max(0L, info.bounds[l].second - info.bounds[l].first);
    // Register blocking.  A convolution member spends its time in LDS reads — two per multiply-add, 921 KB per sample for
    // 115 K multiply-adds against 128 B per clock and CU — not in arithmetic.  A thread therefore owns R consecutive
    // values of the fastest independent iterator (R <= 8 dividing its extent) and walks the reduction once for all of
    // them: the unrolled copies share every load that does not depend on that iterator (the compiler merges them).
    int blk = -1;
    long R = 1;
    constexpr bool no_block = false;
    if (!no_block && !scatter && !si.raw && !si.seed && !red.empty() && rtotal >= 4) {
      // NOT the fastest iterator: consecutive lanes walk that one, so that a wave's LDS reads fall into consecutive banks
      // (blocking it measured 47 us against 32 for the whole kernel: eight-way bank conflicts); the next one up —
      // `x` of out[n, y, x, f], `dx` of gflt[f, dy, dx, c] — is where the window operand and the other operand repeat.
      // EG_SAMPLE_BLOCK_POS (tuning aid): which iterator, counted from the fastest one that has an extent (0), is blocked
      constexpr long block_pos = 1;
      long seen = 0;
      for (size_t i = indep.size(); i-- > 0 && blk < 0;) {
        const long ext = info.bounds[indep[i]].second - info.bounds[indep[i]].first;
        if (ext < 2) continue;
        if (seen++ < block_pos) continue;
        // R by what a thread ends up doing, not "as large as divides": the threads run ceil(items / threads) rounds of R
        // multiply-adds with R loads of every read that moves with the iterator and one of every read that does not
        // (4 608 outputs on 512 threads: R = 8 is 2 rounds of 17 with 7 of 8 waves idle in the second, R = 3 is 3 of 7);
        // with few items the outermost reduction iterator is split over T threads instead (below).
        const int breg_c = k.loops[indep[i]].reg;
        long nd = 0, ns = 0;
        for (auto& rd : k.reads) (op_has(rd, breg_c) ? nd : ns) += 1;
        const long outer = red.empty() ? 1 : std::max(1L, info.bounds[red[0]].second - info.bounds[red[0]].first);
        double best_cost = 1e30;
        for (long r = 1; r <= std::min<long>(ext, 8); ++r) {
          if (ext % r != 0) continue;
          const long it = total / r;
          double rounds = (double)((it + g.threads - 1) / g.threads);
          if (it * 2 <= g.threads && rtotal >= 16 && outer >= 2) {  // the split-reduction path
            const long t = std::max(1L, std::min<long>(g.threads / it, std::min<long>(outer, 64)));
            rounds = (double)((outer + t - 1) / t) / (double)outer;
          }
          const double cost = rounds * (double)(r * (1 + nd) + ns);
          if (cost < best_cost - 1e-9) {
            best_cost = cost;
            R = r;
          }
        }
        if (R > 1) blk = (int)i;
        break;
      }
    }
    const long items = total / R;
    const std::string RS = std::to_string(R);
    const std::string breg = blk >= 0 ? "r" + std::to_string(k.loops[indep[blk]].reg) : "";
    auto decode_indep = [&](const std::string& from, const std::string& ind) {
      std::string d;
      for (auto& s : k.setup) d += ind + "const long r" + std::to_string(s.res) + " = " + std::to_string(info.vals.at(s.res)) + "L;\n";
      if (si.raw) {
        d += ind + "const long r" + std::to_string(k.loops[si.batch_loop].reg) + " = n * " + std::to_string(si.inner) + "L + " + from + ";\n";
        return d;
      }
      if (!si.seed) d += ind + "const long r" + std::to_string(k.loops[si.batch_loop].reg) + " = n;\n";
      d += ind + "long rem = " + from + ";\n";
      for (size_t i = indep.size(); i-- > 0;) {
        const int l = indep[i];
        const long ext = info.bounds[l].second - info.bounds[l].first;
        const std::string lo = std::to_string(info.bounds[l].first) + "L";
        if ((int)i == blk) {
          d += ind + "const long " + breg + "_0 = " + lo + " + (rem % " + std::to_string(ext / R) + "L) * " + RS + "L; rem /= " +
               std::to_string(ext / R) + "L;\n";
        } else {
          d += ind + "const long r" + std::to_string(k.loops[l].reg) + " = " + lo + " + rem % " + std::to_string(ext) + "L; rem /= " +
               std::to_string(ext) + "L;\n";
        }
      }
      d += ind + "(void)rem;\n";
      return d;
    };
    auto term = [&](const std::string& ind) {  // loads + expression of one point of the loop nest
      std::string d;
      for (auto& ins : k.index_instrs) d += ind + "const long r" + std::to_string(ins.res) + " = " + instr_expression(ins, "0L", "r") + ";\n";
      for (auto& rd : k.reads)
        d += ind + "const float r" + std::to_string(rd.reg) + " = " + literal_element(rd, shapes, "t" + std::to_string(rd.tensor), local(rd.tensor)) + ";\n";
      for (auto& ins : k.instrs) d += sample_instr(k, ins, ty, shapes);
      return d;
    };
    // one point of the reduction for the thread's R values: acc[u] += term(u)
    auto accumulate = [&](const std::string& ind) {
      std::string d;
      if (blk < 0) {
        d += term(ind);
        d += ind + "acc[0] = acc[0] + r" + std::to_string(k.result) + ";\n";
        return d;
      }
      d += ind + "_Pragma(\"unroll\")\n" + ind + "for (int u = 0; u < " + RS + "; ++u) {\n";
      d += ind + "  const long " + breg + " = " + breg + "_0 + u;\n";
      d += term(ind + "  ");
      d += ind + "  acc[u] = acc[u] + r" + std::to_string(k.result) + ";\n" + ind + "}\n";
      return d;
    };
    // store acc[u] for the thread's R values
    auto store = [&](const std::string& ind) {
      std::string d;
      if (blk < 0) return ind + w + " = " + (plain ? std::string("0.0f") : w) + " + acc[0];\n";
      d += ind + "_Pragma(\"unroll\")\n" + ind + "for (int u = 0; u < " + RS + "; ++u) {\n";
      d += ind + "  const long " + breg + " = " + breg + "_0 + u;\n";
      d += ind + "  " + w + " = " + (plain ? std::string("0.0f") : w) + " + acc[u];\n" + ind + "}\n";
      return d;
    };
    // Few outputs, long reductions (a dense layer's 10 outputs of 400 terms each, a first-layer filter gradient's 200
    // outputs of 576): one thread per output would leave most of the block idle for hundreds of serial iterations.
    // T threads share an item: thread `part` takes the values part, part + T, ... of the OUTERMOST reduction iterator
    // (the inner ones stay plain nested loops), the T partial sums meet in LDS and are added in the order of `part`
    // (fixed: run-to-run identical).
    long T = 1;
    const long outer_ext = red.empty() ? 0 : std::max(0L, info.bounds[red[0]].second - info.bounds[red[0]].first);
    if (!scatter && !si.raw && items > 0 && items * 2 <= g.threads && rtotal >= 16 && outer_ext >= 2) {
      T = std::min<long>(g.threads / items, std::min<long>(outer_ext, 64));
      if (T < 2) T = 1;
      // a power of two: the T threads of an item are consecutive lanes of ONE wave and their partial sums meet in a
      // butterfly of shuffles (log2 T steps, every lane ends with the same sum) instead of LDS, a block barrier and one
      // thread adding T values one after the other (the dense member of the fashion_mnist step: 51 dependent additions)
      if (T >= 2) {
        long p2 = 2;
        while (p2 * 2 <= T) p2 *= 2;
        T = p2;
      }
    }
    auto inner_loops = [&](size_t from, const std::string& ind) {  // reduction loops red[from ...], innermost unrolled
      std::string d;
      for (size_t i = from; i < red.size(); ++i) {
        const int l = red[i];
        const long ext = info.bounds[l].second - info.bounds[l].first;
        const std::string r = "r" + std::to_string(k.loops[l].reg);
        // loads of several iterations in flight; EG_SAMPLE_UNROLL_MACS (tuning aid): outer reduction loops are unrolled too
        // while the unrolled body stays below that many multiply-adds (copies that read the same element merge)
        constexpr long unroll_macs = 0;
        long inner = R;
        for (size_t j = i; j < red.size(); ++j) inner *= std::max(1L, info.bounds[red[j]].second - info.bounds[red[j]].first);
        if (i + 1 == red.size()) d += ind + (ext * R <= 64 ? "_Pragma(\"unroll\")\n" : "_Pragma(\"unroll 4\")\n");
        else if (unroll_macs > 0 && inner <= unroll_macs) d += ind + "_Pragma(\"unroll\")\n";
        d += ind + "for (long " + r + " = " + std::to_string(info.bounds[l].first) + "L; " + r + " < " + std::to_string(info.bounds[l].second) +
             "L; ++" + r + ") {\n";
      }
      return d;
    };
    const std::string zero_acc = "      float acc[" + RS + "];\n      _Pragma(\"unroll\") for (int u = 0; u < " + RS + "; ++u) acc[u] = 0.0f;\n";
    if (T > 1) {
      const std::string TS = std::to_string(T);
      c += "    {\n      const long out = threadIdx.x / " + TS + "L, part = threadIdx.x % " + TS + "L;\n" + zero_acc;
      c += "      if (out < " + std::to_string(items) + "L) {\n";
      c += decode_indep("out", "        ");
      {
        const int l = red[0];
        const std::string r = "r" + std::to_string(k.loops[l].reg);
        // literal trip count for the whole trips (their loads go out together), the ragged last one guarded
        const long ext0 = info.bounds[l].second - info.bounds[l].first, whole = ext0 / T, ragged = ext0 % T;
        std::string body = inner_loops(1, "          ") + accumulate("          ");
        for (size_t i = 1; i < red.size(); ++i) body += "          }\n";
        if (whole <= 16) {
          if (whole > 0)
            c += "        _Pragma(\"unroll\") for (int it_ = 0; it_ < " + std::to_string(whole) + "; ++it_) {\n          const long " + r + " = " +
                 std::to_string(info.bounds[l].first) + "L + part + it_ * " + TS + "L;\n" + body + "        }\n";
          if (ragged > 0)
            c += "        if (part < " + std::to_string(ragged) + "L) {\n          const long " + r + " = " +
                 std::to_string(info.bounds[l].first + whole * T) + "L + part;\n" + body + "        }\n";
        } else {
          c += "        for (long " + r + " = " + std::to_string(info.bounds[l].first) + "L + part; " + r + " < " +
               std::to_string(info.bounds[l].second) + "L; " + r + " += " + TS + "L) {\n" + body + "        }\n";
        }
      }
      c += "      }\n";
      c += "      _Pragma(\"unroll\") for (int m_ = " + std::to_string(T / 2) + "; m_ >= 1; m_ >>= 1)\n        _Pragma(\"unroll\") for (int u = 0; u < " + RS +
           "; ++u) acc[u] = acc[u] + __shfl_xor(acc[u], m_, " + TS + ");\n";
      c += "      if (out < " + std::to_string(items) + "L && part == 0) {\n";
      c += decode_indep("out", "        ");
      c += store("        ");
      c += "      }\n    }\n  }\n  __syncthreads();\n";
      if (si.reduced) slab_seen.insert(k.write.tensor);
      continue;
    }
    std::string body = decode_indep("idx", "      "), body_store;
    if (scatter) {
      // the element depends on the reduction iterators: add term by term (the destination starts from zero)
      body += inner_loops(0, "      ");
      body += term("        ");
      body += "        " + w + " = " + w + " + r" + std::to_string(k.result) + ";\n";
      for (size_t i = 0; i < red.size(); ++i) body += "      }\n";
    } else {
      body += zero_acc;
      body += inner_loops(0, "      ");
      body += accumulate("        ");
      for (size_t i = 0; i < red.size(); ++i) body += "      }\n";
      body_store = store("      ");
    }
    // The whole trips of the item loop with a literal trip count (a thread's start depends on threadIdx.x, so the
    // compiler cannot count the trips of `for (idx = threadIdx.x; idx < items; idx += threads)` and leaves it rolled: every
    // trip a read - compute - write round trip to LDS; unrolled, the reads of all trips go out together), the ragged last
    // trip computes element 0 again in the threads past the end and guards only its STORE (a guarded trip is a branch the
    // compiler does not move loads across: the 784-float copy of a sample's image was two dependent trips to memory).  A
    // scatter adds onto elements other trips may touch: it keeps the rolled loop.
    const long whole_trips = items / g.threads, ragged_items = items % g.threads;
    if (!scatter && whole_trips <= 12 && rtotal * whole_trips <= 256) {
      if (ragged_items > 0)
        c += "    {\n      const bool ok_ = threadIdx.x < " + std::to_string(ragged_items) + ";\n      const long idx = ok_ ? threadIdx.x + " +
             std::to_string(whole_trips * g.threads) + "L : 0L;\n" + body + "      if (ok_) {\n" + body_store + "      }\n    }\n";
      if (whole_trips > 0)
        c += "    _Pragma(\"unroll\") for (int it_ = 0; it_ < " + std::to_string(whole_trips) + "; ++it_) {\n      const long idx = threadIdx.x + it_ * " +
             NT + "L;\n" + body + body_store + "    }\n";
    } else {
      c += "    for (long idx = threadIdx.x; idx < " + std::to_string(items) + "L; idx += " + NT + ") {\n" + body + body_store + "    }\n";
    }
    c += "  }\n  __syncthreads();\n";
    if (si.reduced) slab_seen.insert(k.write.tensor);
  }
  // 32-bit index arithmetic where it is exact: every tensor (and the slab) has fewer than 2^31 elements and no member
  // computes with Index VALUES (`toScalar(i * 100000)`: only addressing is known to fit — the rule of Slot::Narrow,
  // codegen.hpp).  64-bit divisions and multiply-adds per element were most of a convolution member's time.
  // ---- barriers between INDEPENDENT members go.  Member b needs no barrier in front of it when, for every member a since the
  // last barrier that stays, a's result is neither read nor written by b, b's result is not read by a, and at most one of
  // them uses `scratch`.  Then the waves that are done with a (a row-block member with 9 blocks for 8 waves leaves seven
  // waves waiting for the ninth block) start on b; the barrier behind b orders both against what follows.
  if (eg::sw::raw("EG_SAMPLE_KEEP_BARRIERS") == nullptr) {
    const std::string barrier = "  __syncthreads();\n";
    struct Use {
      std::set<int> reads, writes;
      bool scratch = false;
    };
    std::vector<Use> use(member_start.size());
    for (size_t gi = 0; gi < member_start.size(); ++gi) {
      const Kernel& k = all[g.kernel_index[gi]];
      for (auto& rd : k.reads) use[gi].reads.insert(rd.tensor);
      use[gi].writes.insert(k.write.tensor);
      if (!g.overwrite[gi]) use[gi].reads.insert(k.write.tensor);
      const size_t end = gi + 1 < member_start.size() ? member_start[gi + 1] : c.size();
      const std::string text = c.substr(member_start[gi], end - member_start[gi]);
      use[gi].scratch = text.find("scratch[") != std::string::npos;
      if (text.find("zeros4_") != std::string::npos) use[gi].reads.insert(-4);   // (the prologue writes them)
    }
    Use prologue;
    for (auto& kv : g.staged) prologue.writes.insert(kv.first);
    for (int t : g.lds_zero) prologue.writes.insert(t);
    prologue.writes.insert(-4);
    auto conflict = [](const Use& a, const Use& b) {   // b behind a without a barrier
      for (int w : a.writes)
        if (b.reads.count(w) || b.writes.count(w)) return true;
      for (int w : b.writes)
        if (a.reads.count(w)) return true;
      return a.scratch && b.scratch;
    };
    std::vector<size_t> erase;   // offsets of the barriers that go
    std::vector<const Use*> since;
    if (!member_start.empty()) {
      if (prologue_barrier != std::string::npos && !conflict(prologue, use[0])) {
        erase.push_back(prologue_barrier);
        since.push_back(&prologue);
      }
      since.push_back(&use[0]);
    }
    for (size_t b = 1; b < member_start.size(); ++b) {
      bool independent = member_start[b] >= barrier.size() && c.compare(member_start[b] - barrier.size(), barrier.size(), barrier) == 0;
      for (const Use* a : since)
        if (independent && conflict(*a, use[b])) independent = false;
      if (independent) {
        erase.push_back(member_start[b] - barrier.size());
      } else {
        since.clear();
      }
      since.push_back(&use[b]);
    }
    for (size_t i = erase.size(); i-- > 0;) c.erase(erase[i], barrier.size());
  }
  if (need_dummy) c = "  __shared__ float dummy_[" + NT + "];\n" + c;
  {
    const size_t at = c.find("/*zeros4*/");
    std::string init;
    if (need_zeros4) {
      init = "  __shared__ __attribute__((aligned(16))) float zeros4_[4];\n  if (threadIdx.x < 4) zeros4_[threadIdx.x] = 0.0f;\n";
      if (g.staged.empty() && g.lds_zero.empty()) init += "  __syncthreads();\n";   // (no prologue barrier to ride on)
    }
    if (at != std::string::npos) c.replace(at, 10, init);
  }
  if (scratch_floats > 0) c = "  __shared__ float scratch[" + std::to_string(scratch_floats) + "];\n" + c;
  bool narrow = eg::sw::raw("EG_NO_NARROW_INDEX") == nullptr && g.B * std::max(1L, g.slab_floats) < (1L << 31);
  for (int t : touched) narrow = narrow && prodv(shapes.at(t)) < (1L << 31);
  for (int ki : g.kernel_index) {
    const Kernel& k = all[ki];
    const std::vector<Ty> ty = infer_types(k);
    for (auto& ins : k.instrs) {
      const bool index_typed = ins.res > 0 && ins.res < (int)ty.size() && ty[ins.res] == Ty::Index;
      const bool derived = ins.kind != IK::Shape && ins.kind != IK::Len && ins.kind != IK::ShapeLen && ins.kind != IK::Epoch;
      if (index_typed && derived) narrow = false;
    }
  }
  if (narrow) {
    c = std::regex_replace(c, std::regex("\\blong\\b"), "int");
    c = std::regex_replace(c, std::regex("\\b([0-9]+)L\\b"), "$1");
  }
  // EG_SAMPLE_TRACE=1 (detector): block 0 stamps the cycle counter behind every member's barrier and prints the stamps
  // (cycles since the first) at the end — where a sample kernel's time goes, without the dead-code elimination that makes
  // EG_SAMPLE_STOP's differences hard to read (a member whose result never leaves LDS disappears with its producers).
  if (eg::sw::raw("EG_SAMPLE_TRACE") != nullptr) {
    // (EG_SAMPLE_TRACE=100 + k: member k runs twice — idempotent when it overwrites its result — so that the stamps show
    // what its second, instruction-cache-warm execution costs)
    const long repeat = eg::sw::integer("EG_SAMPLE_TRACE", 1) - 100;
    if (repeat >= 0) {
      const size_t from = c.find("  {  // kernel " + std::to_string(repeat) + ":");
      const size_t to = from == std::string::npos ? from : c.find("  {  // kernel " + std::to_string(repeat + 1) + ":", from);
      if (from != std::string::npos && to != std::string::npos) {
        std::string again = c.substr(from, to - from);
        if (again.find("\n  __syncthreads();\n") == std::string::npos) again += "  __syncthreads();\n";
        c.replace(from, to - from, "  _Pragma(\"nounroll\") for (int rep_ = 0; rep_ < 2; ++rep_) {\n" + again + "  }\n");
      }
    }
    const std::string stamp = "  if (threadIdx.x == 0 && blockIdx.x == 0 && trn_ < 48) tr_[trn_++] = __builtin_readcyclecounter();\n";
    std::string traced = "  long long tr_[48]; int trn_ = 0;\n" + stamp;
    size_t pos = 0;
    const std::string barrier = "\n  __syncthreads();\n";
    while (true) {
      const size_t at = c.find(barrier, pos);
      if (at == std::string::npos) break;
      traced += c.substr(pos, at + barrier.size() - pos) + stamp;
      pos = at + barrier.size();
    }
    traced += c.substr(pos);
    traced += "  if (threadIdx.x == 0 && blockIdx.x == 0) {\n    printf(\"[eg] " + g.name + " cycles behind each member barrier:\");\n"
              "    for (int k_ = 1; k_ < trn_; ++k_) printf(\" %d:%lld\", k_, tr_[k_] - tr_[0]);\n    printf(\"\\n\");\n  }\n";
    c = traced;
  }
  g.source = sig + " {\n" + c + "}\n";
  return EG_OK;
}

}  // namespace kd
}  // namespace eg
