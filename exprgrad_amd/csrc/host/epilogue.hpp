// Epilogue fusion: an elementwise kernel that consumes the output of a contraction runs on the
// accumulator registers of the matrix kernel instead of as its own launch.
//
// Role in the reference: fuseLoops (passes.nim:1929-2004) merges `dense` and the activation that
// follows it into one loop nest on the CPU target; its GPU target launches every kernel separately
// (llvmgen.nim:455-500).  Here, for `h = x * W + b; a{it} ++= f(h{it})` the contraction writes h
// (only if something later reads it) and a from the same registers; for the backward pair
// `ga = gz * W2^T; gh{it} ++= select(..h{it}.., ga{it}, ..)` ga never exists in memory.
// The generated functor is spliced into gemm_block<..., Epi> (kernels/gemm_fused.hpp).
#pragma once
#include <map>
#include <string>
#include <vector>

#include "kd.hpp"

namespace eg {
namespace kd {

// A yes / no question about a tensor element: cmp(literal, x) or cmp(x, literal) with cmp in {le, lt, eq} — the only
// way some tensors are ever read after the launch that produces them (relu's `inp >= 0`, dnn.nim:26-27, in its derived
// gradient).  Such a tensor is stored as one bit per element (plan_epilogue.cpp: predicate tensors).
struct PredicateSpec {
  IK kind = IK::Le;
  double literal = 0;
  bool literal_first = true;
  bool operator==(const PredicateSpec& o) const { return kind == o.kind && literal == o.literal && literal_first == o.literal_first; }
};

// Is every use of `tensor` in `k` the same comparison against a literal?  (The value itself may not flow anywhere else.)
bool only_predicate_uses(const Kernel& k, int tensor, PredicateSpec& out);

struct EpilogueSpec {
  std::string struct_name;     // "EgEpi"
  std::string struct_code;     // the functor definition
  std::vector<int> operands;   // tensor ids bound to a.epi[i], in order
};

// Can `k` (with inferred bounds `info`) run as the epilogue of a contraction that produces
// `c_tensor` = [M, N] (dense rows)?  Requirements: no reduction, every operand is addressed by the
// same flat element index, that index enumerates 0 .. M*N-1 exactly once, the kernel reads
// c_tensor and writes a different tensor it does not read.
bool epilogue_capable(const Kernel& k, const KernelInfo& info, const Shapes& shapes, int c_tensor, long M, long N);

// The line of every generated functor that says "no row product" (gemm_f32_mfma.hpp, RD_N); fold_row_products
// (plan_epilogue.cpp) replaces it.
extern const char* const kNoRowProduct;

// store_c: also write the contraction result itself (it is read again later).
// accumulate: the consumer adds to its destination instead of overwriting it.
// pred_reads: operands of `k` that exist as predicate bits only (tensor -> the question they answer).
// pred_write: the contraction result itself is stored as predicate bits answering *pred_write (then store_c is false).
int generate_epilogue(const Kernel& k, const KernelInfo& info, const Shapes& shapes, int c_tensor, bool store_c,
                      bool accumulate, EpilogueSpec& out, const std::map<int, PredicateSpec>* pred_reads = nullptr,
                      const PredicateSpec* pred_write = nullptr);

}  // namespace kd
}  // namespace eg
