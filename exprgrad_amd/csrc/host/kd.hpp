// Kernel-description IR: the information content of exprgrad's `Program` before `generate`
// (ir.nim:211-270), in the minimal form the GPU backend needs.  Text grammar: DESIGN.md.
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <utility>
#include <vector>

namespace eg {
namespace kd {

// ir.nim:51-76 (the scalar subset that can appear in a kernel expression)
enum class IK : uint8_t {
  Index, Scalar, Boolean,
  Add, Sub, Mul, Div, IndexDiv, Mod, Wrap, Negate, Sin, Cos, Exp, Pow, Sqrt, Log, Log10, Log2, Ln,
  Eq, Lt, Le, And, Or, Select, ToScalar, ToIndex, Shape, Len, ShapeLen, Epoch
};
enum class Ty : uint8_t { None, Scalar, Index, Boolean };

const char* ik_name(IK k);
bool ik_from_name(const std::string& s, IK& out);

// ir.nim:120-123 after foldLinearIndices: constant + sum(factor * register)
struct Lin {
  long constant = 0;
  std::vector<std::pair<int, long>> factors;  // (register, factor), factor != 0
  int only_register() const;                  // passes.nim:968-972
  long factor_of(int reg) const;
  bool operator==(const Lin& o) const;
};

struct Instr {
  IK kind = IK::Scalar;
  int res = 0;
  std::vector<int> args;
  double lit = 0;  // Index / Scalar / Boolean literal
  int tensor = 0;  // Shape / Len / ShapeLen
  int dim = 0;     // Shape
};

// ir.nim:166-173
struct Op {
  int tensor = 0;
  int reg = 0;
  bool raw = false;
  std::vector<Lin> dims;
};

// ir.nim:136-150 (bounds only)
struct Loop {
  int reg = 0;
  std::string name;
  bool has_bounds = false;
  Lin start, stop;
};

enum class Gen : uint8_t { None, Backwards, Gradient };  // ir.nim:196-204

// ir.nim:211-230
struct Kernel {
  int nregs = 0;
  std::vector<Instr> setup;  // host-evaluated: shape()/len() terms of explicit loop bounds
  std::vector<Loop> loops;
  // Index instructions evaluated inside the loop nest before the reads: the non-affine parts of
  // tensor indices (`y div 2`; LinearIndex.setup in the reference, ir.nim:120-123).  Their result
  // registers appear in the Lin dims of the operands like iterators do.
  std::vector<Instr> index_instrs;
  std::vector<Op> reads;
  std::vector<Instr> instrs;
  int result = 0;
  Op write;
  // KernelGradient(isCustom) ir.nim:203-209: gradient kernels written by the user (maxpool2,
  // dnn.nim:59-71), used by generate instead of derive.  In them tensor id -t stands for the
  // gradient tensor of t.
  bool has_custom_grad = false;
  std::vector<Kernel> custom_grad;
  Gen gen = Gen::None;
  int gen_tensor = 0;  // Backwards: the loss; Gradient: differentiate with respect to this tensor
  int gen_dest = 0;    // Gradient: tensor that receives the gradient
  bool is_seed = false;  // the gradLoss{i} = 1 kernel (passes.nim:575-606)
  bool f64 = false;      // the program computes in float64 (compile[float64], model.nim:253-260): set on every kernel by compile_program
  int alloc() { return ++nregs; }
};

enum class TK : uint8_t { Input, Param, Result, Cache, Random };  // ir.nim:232-245

struct TensorDef {
  TK kind = TK::Result;
  std::string name;
  bool has_shape = false;
  std::vector<long> shape;  // static shape; -1 = unknown extent
  double lo = 0, hi = 0;    // TensorParam.initRange / TensorRandom.randomRange
};

struct Target {
  std::string name;
  int output = 0;
  std::vector<Kernel> source;  // as parsed (with generator placeholders)
  std::vector<Kernel> all;     // after generate: every kernel, used for shape inference
  std::vector<int> live;       // indices into `all` that survive deadKernelElim, in order
  int first_update = -1;       // position in `live` of the first kernel that writes a parameter
};

struct Program {
  std::vector<TensorDef> tensors;  // 1-based: tensors[0] is unused
  std::map<int, int> shape_copy;   // dest -> src      (ShapeCopy, ir.nim:186-187)
  std::map<int, std::vector<Lin>> shape_dims;  // dest -> dims (ShapeDims)
  std::map<int, std::vector<Instr>> shape_setup;  // dest -> host instructions defining the registers of shape_dims
  std::vector<Target> targets;
  std::map<std::string, int> inputs;
  // Some host-evaluated value (a kernel's setup, a shape constraint) is epoch() or computed from it: such values are
  // fixed when a plan is made (loop bounds, kernel arguments, literals of generated code), so plans are keyed by the
  // epoch as well as by the input shapes (plan.cpp shape_key) — `x[epoch() mod n]` reads another row every epoch.
  bool epoch_in_setup = false;
  // Scalar type of the program, from the header line (`kd 1 f32` / `kd 1 f64`): the T of compile[T] (model.nim:253-260,
  // toScalarType -> Scalar32 / Scalar64).  Every tensor of a program has it.
  bool f64 = false;
  Target* find_target(const std::string& name);
  int alloc_tensor(TK kind, const std::string& name);
};

// Parse the text form.  Returns EG_OK or sets the thread-local error.
int parse(const char* text, Program& out);

// generate (passes.nim:558-640) + deadKernelElim (passes.nim:331-350) for every target.
int compile_program(Program& prog);

// Register types (inferTypes restated): indexed by register id, 0 unused.
std::vector<Ty> infer_types(const Kernel& k);

// ---- run-time shape inference (inferLoopBounds passes.nim:986-1010; shape constraints
//      passes.nim:1059-1095 solved as a forward walk over the kernel list) -------------------
struct KernelInfo {
  bool ok = false;
  std::vector<std::pair<long, long>> bounds;  // per loop, [start, stop)
  std::map<int, long> vals;                   // host-evaluated registers (setup)
};
using Shapes = std::map<int, std::vector<long>>;
// Infers loop bounds of `k` and, if still unknown, the shape of the tensor it writes.
// Returns EG_OK, or EG_ERR_SHAPE with the error text set.
int infer_kernel(const Program& prog, const Kernel& k, Shapes& shapes, long epoch, KernelInfo& out);

std::string to_text(const Kernel& k);

}  // namespace kd
}  // namespace eg
