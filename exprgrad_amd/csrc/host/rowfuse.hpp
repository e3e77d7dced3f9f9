// Row fusion: run a chain of per-sample kernels as ONE generated kernel, one thread per sample.
//
// Role in the reference: fuseLoops (passes.nim:1929-2004, 2526-2549) merges consecutive kernels
// that share their leading loop — on the CPU target dense + bias + activation become one `y`
// loop.  The reference's GPU target does not benefit (each lowered kernel is still its own
// launch, llvmgen.nim:455-500).  Here the idea is taken further for the launch-bound small-width
// chains of the hot path (the whole forward + backward of the XOR net: 15 kernels over [B,4] /
// [B,1] tensors; the softmax / cross-entropy gradient chain of the dense net over [B,10]):
//   * a kernel is a "row kernel" if it has one loop over the batch B that indexes dimension 0 of
//     every batch-major tensor it touches ([B, inner...], inner size <= 64), all its other loops
//     are short, and everything else it touches is small (parameters, their gradients);
//   * a run of consecutive row kernels is specialised for the current shapes (all extents become
//     literals, loops fully unroll) and emitted as one kernel: thread y keeps the rows of the
//     intermediate tensors in registers, tensors needed later are stored once, tensors that are
//     dead after the run never touch memory;
//   * writes that reduce over the batch (bias / weight gradients, losses) accumulate in
//     registers, are folded across the block with wave shuffles + LDS, and one small `finalize`
//     launch adds the per-block partials into the destination tensors in fixed order
//     (deterministic, no float atomics).
#pragma once
#include <map>
#include <set>
#include <string>
#include <vector>

#include "kd.hpp"

namespace eg {
namespace kd {

struct RowKernelInfo {
  bool ok = false;
  int row_loop = -1;     // index into k.loops of the batch iterator (or of the raw iterator)
  bool raw = false;      // the batch iterator is a raw `{it}` index over B * inner elements
  long inner = 0;        // raw: inner size S (it = y * S + j)
  bool small_only = false;  // touches only small tensors and has no batch loop (the gradLoss seed)
  long work = 0;         // unrolled iterations per sample
};

// Can `k` run as part of a row group over batch size B?
RowKernelInfo analyse_row_kernel(const Program& prog, const Kernel& k, const KernelInfo& info, const Shapes& shapes,
                                 long B);

struct RowGroupTensor {
  int tensor = 0;
  enum Role { RowLocal, RowExternal, SmallExternal, SmallLocal, Reduction } role = RowExternal;
  long inner = 0;          // elements per row (row tensors) or total elements (small tensors)
  bool load_first = false;  // RowLocal: written before the group too, start from memory
  bool store = false;       // RowLocal: needed after the group
  bool accumulate = false;  // Reduction: add to the destination instead of overwriting it
  long red_offset = 0;      // Reduction: column offset inside the partial rows
};

struct RowGroup {
  std::vector<int> kernel_index;  // indices into target.all, in execution order
  std::vector<RowKernelInfo> infos;
  std::map<int, RowGroupTensor> tensors;
  long B = 0;
  long red_total = 0;  // partial row length
  // floats between two blocks' partial rows: whole 16-byte groups (the hand-off stores and loads them as such)
  long red_stride() const { return (red_total + 3) & ~3L; }
  std::string name, source;
  std::vector<int> ptr_args;  // tensor ids in pointer-argument order (after `partial`)
  // B <= 256: the one block adds its totals to their destinations itself (arguments d<id> behind `epoch`, one per
  // reduction in segment order) — no second launch; EG_NO_ROW_DIRECT=1: partial row + row_finalize as for many blocks
  bool single_block = false;
  // More than one block (round 5): the LAST block to arrive (a ticket counter) folds the partial rows itself, in the order
  // of row_finalize_kernel, instead of a second launch doing it; and it can go on with a run of small kernels that
  // directly follows the group — the optimizer updates of a small net (the whole XOR step becomes ONE launch).
  // Arguments behind `epoch`: d<id> per reduction, unsigned* counter, long MODE (0: partial rows only, the caller
  // launches row_finalize; 1: fold in the kernel; 2: fold, then the tail), u<id> per tail tensor.
  bool in_kernel_finalize = false;
  // Round 6: the grid the launch will use when it is known at generation time (fuse_row_tails: a row group with a tail runs on
  // few, long blocks).  With B a multiple of grid_blocks * 256 the strided sample loop is emitted with a literal trip count and
  // fully unrolled: the loads of all of a thread's samples are in flight together instead of one memory round trip per sample
  // (four in a row at the XOR step).  The generic loop stays in the kernel for any other grid / B.
  long grid_blocks = 0;
  std::vector<int> tail_kernels;   // indices into target.all, in execution order
  std::vector<int> tail_ptr_args;  // tensor ids of the tail kernels, in pointer-argument order
};

// Emit the fused kernel.  Arguments of the generated kernel:
//   (float* partial, float* / const float* t<ids>..., long B, float grad_scale, long epoch[, float* d<ids>...: single_block
//    or in_kernel_finalize][, unsigned* counter, long MODE, float* u<ids>...: in_kernel_finalize])
int generate_row_group(const Program& prog, const std::vector<Kernel>& all, const std::vector<KernelInfo>& infos,
                       const Shapes& shapes, RowGroup& group);

// ---- small-kernel fusion -------------------------------------------------------------------------
// Consecutive kernels that touch only small tensors (the optimizer updates: gradientDescent's
// `param{it} ++= -grad{it} * rate` per parameter, adam's m / v / param chain, base.nim:37-53) run as
// ONE single-block kernel: each kernel's independent iterations are strided over the 256 threads,
// `__syncthreads()` separates the kernels.  Four launches of ~5 us each become one.
bool is_small_kernel(const Program& prog, const Kernel& k, const KernelInfo& info, const Shapes& shapes);

struct SmallGroup {
  std::vector<int> kernel_index;  // indices into target.all, in execution order
  std::string name, source;
  std::vector<int> ptr_args;      // tensor ids in pointer-argument order
  long blocks = 1;                // grid size (1 for a small group, sum over the segments of a map group)
  // Map groups only (round 5): gradient tensors whose sum over the batch the group takes from a sample group's slab
  // itself — thread idx adds rows 0 .. fold_rows - 1 of its element in that order, stores the total where the gradient
  // lives (other readers find it there) and goes on with its kernels: the slab_sum launch between the sample kernel and
  // the optimizer disappears.  Arguments behind `epoch`: const float* slab, long FOLD (0: the gradients are in place).
  std::map<int, long> fold_offset;  // tensor -> float offset inside a slab row
  long fold_rows = 0, fold_row_floats = 0;
};

// Arguments of the generated kernel: (float* / const float* t<ids>..., float grad_scale, long epoch).
// Every kernel accumulates into its destination (the caller zeroes first-written results).
int generate_small_group(const Program& prog, const std::vector<Kernel>& all, const std::vector<KernelInfo>& infos,
                         const Shapes& shapes, SmallGroup& group);

// ---- map groups ------------------------------------------------------------------------------------
// Consecutive raw elementwise kernels over whole tensors of ANY size (`p{it} ++= -g{it} * rate` for
// every parameter: gradientDescent, base.nim:37-38; adam's m / v / p chains) run as one launch: the
// kernels are grouped into segments by element count, a block belongs to one segment (block ranges
// are literals in the generated source) and a thread runs, for its element, every kernel of the
// segment in order.  Kernels of different segments cannot share a tensor (each covers its tensors
// completely), so segments are independent.  Same kernel arguments as a small group.
bool is_map_kernel(const Program& prog, const Kernel& k, const KernelInfo& info, const Shapes& shapes, long& count);
int generate_map_group(const Program& prog, const std::vector<Kernel>& all, const std::vector<KernelInfo>& infos,
                       const Shapes& shapes, SmallGroup& group);

// ---- sample groups (round 5) -----------------------------------------------------------------------------
// A whole training step at a SMALL batch is a chain of dependent launches, each at the ~4.5 us floor of a dependent
// kernel whatever it computes (Model.fit at the reference's default batch of 32, model.nim:413-454 with the network of
// examples/fashion_mnist/fashion_mnist.nim:39-57: 16 launches, 70 us).  But every kernel of the forward and backward
// pass works sample by sample — only the parameter gradients add over the batch.  A sample group runs a run of such
// kernels as ONE generated kernel with ONE BLOCK PER SAMPLE: the block walks the kernel list in order, its 256 threads
// share the independent iterations of a kernel restricted to the block's sample (reductions serial per thread,
// `__syncthreads()` between kernels; the tensors stay where they are, in L2-resident global memory), and a kernel that
// reduces over the batch writes its sample's contribution into row `sample` of a slab, which one deterministic
// slab_sum launch folds into the gradient bucket afterwards.  The reference fuses loops for its CPU target only
// (fuseLoops, passes.nim:1929-2004); its GPU target launches every kernel (llvmgen.nim:455-500).
struct SampleKernelInfo {
  bool ok = false;
  int batch_loop = -1;   // index into k.loops of the loop that walks the samples (-1: the seed, no such loop)
  bool raw = false;      // that loop is a raw iterator over B * inner elements (it = sample * inner + j)
  long inner = 0;
  bool reduced = false;  // the write is not indexed by the batch loop: a sum over the samples (parameter gradients)
  bool seed = false;     // gradLoss{i} = 1 (passes.nim:575-606): every block writes the same value
  long work = 0;         // loop iterations per sample
  // conv2's image gradient — a scatter in the derived loop nest (gimg[n, y + dy, x + dx, c] += ...) — as a gather over
  // the image gradient's own elements (tensor ids; filled in by the planner from match_conv)
  bool gather = false;
  int g_img = 0, g_out = 0, g_flt = 0;
  // Round 6: a batched conv2 member (forward, filter gradient or image gradient; planner: match_conv) on the matrix cores —
  // 16 x 16 x 4 instructions whose fragments are gathered from the block's LDS-resident tensors (generate_sample_group,
  // "matrix-core convolution members").  0 = the generic loop nest (or `gather`), 1 = forward, 2 = filter gradient,
  // 3 = image gradient; tensor ids of image, output (or its gradient) and filter bank.
  int conv_role = 0;
  int conv_img = 0, conv_out = 0, conv_flt = 0;
};

SampleKernelInfo analyse_sample_kernel(const Program& prog, const Kernel& k, const KernelInfo& info, const Shapes& shapes, long B);

struct SampleGroup {
  std::vector<int> kernel_index;  // indices into target.all, in execution order
  std::vector<SampleKernelInfo> infos;
  std::vector<char> overwrite;    // per member: the write is a plain store (first writer, whole tensor) instead of +=
  long B = 0;
  std::map<int, long> slab_offset;  // tensors summed over the batch -> float offset inside a slab row
  long slab_floats = 0;             // length of a slab row (a multiple of 4)
  std::string name, source;
  std::vector<int> ptr_args;        // tensor ids in pointer-argument order (behind `slab`)
  // Tensors that exist inside the group only (written and read by members, sample by sample): a block keeps its sample's
  // slice in LDS instead of global memory — an L2 round trip (~0.5 - 1 us) at the head of every one of the ~25 kernels of
  // a step is what the first version of this kernel spent most of its 85 us on.  tensor id -> floats per sample.
  std::map<int, long> lds;
  std::set<int> lds_zero;           // of those: accumulated into before being written whole (start from zero)
  int threads = 256;                // block size (a multiple of 64)
  // Round 6: small PARAMETER tensors the members read (filter banks, a dense layer's weights) are copied into LDS once, at the
  // kernel's start: a member that walked them with dependent loads paid an L2 round trip per iteration (the 400-term dense
  // layer of the fashion_mnist network: eight of them in a row).  tensor id -> floats.
  std::map<int, long> staged;
};

// Arguments of the generated kernel: (float* slab, float* t<ids>..., float grad_scale, long epoch); grid = B blocks of `threads`.
int generate_sample_group(const Program& prog, const std::vector<Kernel>& all, const std::vector<KernelInfo>& infos,
                          const Shapes& shapes, SampleGroup& group);

}  // namespace kd
}  // namespace eg
