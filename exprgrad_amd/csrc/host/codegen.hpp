// HIP source generation for kernel descriptions that have no hand-written library kernel.
//
// Role in the reference: clgen.nim:74-257 (`toCl`) — turn one lowered kernel into device source.
// This is not a translation of it: instead of one work-item per point of the (up to 3) hoisted
// independent loops with every reduction left serial (passes.nim:2438-2514), the generated kernel
// picks one of two hand-written templates and splices the scalar expression into it:
//   mode A  one thread per point of ALL independent loops (flattened, write-coalesced order),
//           reduction loops run inside the thread with a register accumulator;
//   mode B  (few outputs, long reduction: bias gradients, scalar losses) the flattened reduction
//           space is split over thread rows and blocks, folded through LDS, and finished by the
//           library's deterministic column-sum — the lowering the reference lacks entirely
//           (passes.nim:2411-2524 emits no InstrGpu when no loop is independent).
// All shape-dependent quantities are kernel arguments (as in clgen.nim:249-256: tensors, then
// `long` index registers), so one build serves every input shape.
#pragma once
#include <string>
#include <vector>

#include "kd.hpp"

namespace eg {
namespace kd {

struct Slot {
  enum Kind : int { Accumulate, Total, RTotal, Chunk, LoopStart, LoopExtent, Stride, SetupVal, InstrVal, GradScaleBits,
                    Narrow /* 1: every operand has < 2^31 elements, 32-bit index arithmetic is exact */,
                    Vec4 /* 1: four elements of the fastest iterator per thread, 16-byte loads and stores */ };
  Kind kind;
  int a = 0, b = 0;  // LoopStart/LoopExtent: loop index; Stride: op index (reads..., write last), dim;
                     // SetupVal: index into k.setup; InstrVal: index into k.instrs
};

struct GenericSource {
  std::string name;
  std::string source;
  std::vector<int> tensor_args;  // tensor ids in argument order (mode A: written tensor first;
                                 // mode B: the partial buffer comes first and is not listed)
  std::vector<Slot> slots;       // the `long` arguments that follow the tensors
  std::vector<int> indep;        // loop indices, slowest -> fastest varying in the thread decode
  std::vector<int> red;          // reduction loop indices, outer -> inner
  bool scatter = false;          // the write index involves a non-independent iterator
  int tx = 0, ty = 0;            // mode B block shape
};

// Which loops are independent (identifyIndependent, passes.nim:1774-1782), in write-dim order.
void split_loops(const Kernel& k, std::vector<int>& indep, std::vector<int>& red, bool& scatter);

// True if mode B's flat output index equals the flat index of the written tensor for every
// shape (all write dims bare distinct iterators, or a single constant element).
bool split_reduction_capable(const Kernel& k);

// C float literal that parses back to exactly (float)v.
std::string f32_literal(double v);
// One scalar instruction as a C expression over variables `<prefix><register id>`; `special`
// is the text used for the host-evaluated builtins (shape / len / shapelen / epoch).
std::string instr_expression(const Instr& ins, const std::string& special, const std::string& prefix, bool f64 = false);

int generate_mode_a(const Kernel& k, const std::string& name, GenericSource& out);
int generate_mode_b(const Kernel& k, const std::string& name, int tx, GenericSource& out);

}  // namespace kd
}  // namespace eg
