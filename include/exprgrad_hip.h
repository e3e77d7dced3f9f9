/*
 * exprgrad_hip.h — C ABI of libexprgrad_hip.so, the MI355X (gfx950) kernel backend for
 * exprgrad's compiled tensor hot path.
 *
 * This header is the drop-in boundary. Every entry point names the reference interface it
 * replaces (paths relative to the exprgrad repository):
 *
 *   group 1  "runtime"  — 1:1 with the proc set every GPU runtime must provide,
 *                         exprgrad/runtimes/gpu.nim:24-52 (implemented for OpenCL in
 *                         exprgrad/runtimes/cl.nim:45-207).
 *   group 2  "library"  — hand-written CDNA4 kernels for the kernel classes the reference
 *                         lowers from `++=` statements (exprgrad/layers/base.nim:19-67,
 *                         exprgrad/layers/dnn.nim:19-100) and that clgen.nim:217-257 would
 *                         otherwise emit as OpenCL text.
 *   group 3  "model"    — kernel-description programs: what model.nim:215-251 (newModel),
 *                         model.nim:392-454 (call/apply/fit) and the three JIT builtins
 *                         model.nim:148-172 do for a CompileGpu target.
 *
 * Conventions
 *   - plain C: opaque handles, raw pointers, sizes; no C++ / torch types.
 *   - every function returns an int status (EG_OK == 0); on failure a thread-local message is
 *     available from eg_last_error().  The library never aborts or exits
 *     (cl.nim:41-43 raises GpuError; a binding re-raises from the status).
 *   - tensors are dense row-major float32, last dimension contiguous (tensors.nim:20-25);
 *     all extents / indices are signed 64 bit (wrappers/llvm.nim:295-303, clgen.nim:59).
 *   - one in-order HIP stream per context (cl.nim:92: one in-order command queue); copies are
 *     blocking, fills and launches are asynchronous (cl.nim:111-131, 190-207).
 *   - device pointers passed to group 2 may come from eg_buf_ptr() or from any other HIP
 *     allocation on the context's device (e.g. a torch tensor's data_ptr()).
 */
#ifndef EXPRGRAD_HIP_H
#define EXPRGRAD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EG_OK 0
#define EG_ERR_INVALID 1     /* bad argument / handle / shape                          */
#define EG_ERR_HIP 2         /* HIP runtime or driver error (message carries the code) */
#define EG_ERR_COMPILE 3     /* hiprtc build failure (message carries the build log)   */
#define EG_ERR_SIZE 4        /* host<->device size mismatch (cl.nim:112-113, 134-142)  */
#define EG_ERR_UNSUPPORTED 5 /* kernel description the backend has no lowering for     */
#define EG_ERR_RUNTIME 6     /* unknown target / input (model.nim:358-359, 395-396)    */
#define EG_ERR_SHAPE 7       /* shape inference failure (passes.nim ShapeError)        */

typedef struct eg_ctx eg_ctx;       /* GpuContext  (cl.nim:24-27)  */
typedef struct eg_buf eg_buf;       /* GpuBuffer   (cl.nim:29-32)  */
typedef struct eg_kernel eg_kernel; /* GpuKernel   (cl.nim:38-40)  */
typedef struct eg_dp eg_dp;         /* one rank of a data-parallel group (no reference counterpart) */
typedef struct eg_model eg_model;   /* Model[T] restricted to its GpuModel (model.nim:21-43) */

/* Thread-local text of the last failure on this thread ("" if none). cl.nim:41-43. */
const char* eg_last_error(void);
/* ABI version of this header: major*1000 + minor. */
int eg_version(void);

/* ------------------------------------------------------------------ group 1: runtime -- */

/* listDevices(): gpu.nim:34, cl.nim:64-66. */
int eg_device_count(int* count);
/* name/vendor/version/isGpu(device): gpu.nim:35-38, cl.nim:75-81. Any out pointer may be NULL. */
int eg_device_info(int device, char* name, size_t name_cap, char* vendor, size_t vendor_cap,
                   char* version, size_t version_cap, int* is_gpu);
/* Static facts used for roofline reporting (not in the reference). */
int eg_device_props(int device, int* compute_units, int* clock_khz, int64_t* hbm_bytes,
                    char* arch, size_t arch_cap);
/* Which compiler builds kernel text at run time (compile(ctx, name, source): gpu.nim:48, cl.nim:149-179, and every
 * generated kernel of a model, model.nim:209-213): "hiprtc <major>.<minor> <path of the library>; cache <dir>|off".
 * The library opens hiprtc itself, preferring the ROCm release it was built with over whatever the process has
 * loaded (EG_HIPRTC_LIB overrides); code objects are cached on disk (EG_KERNEL_CACHE, EG_NO_KERNEL_CACHE). */
int eg_compiler_info(char* text, size_t cap);
/* Run-time compilations of this process: served from the on-disk cache / compiled, and the seconds spent compiling. */
int eg_kernel_cache_stats(int64_t* hits, int64_t* misses, double* compile_seconds);
/* Environment switches (csrc/switches.cpp): the library honours exactly the names of ONE table — every optimisation can
 * be turned off (class "execution"), detectors and dumps ("detector"), the data-parallel and run-time-compiler settings,
 * and measurement aids ("tuning") that are read only under EG_TUNING=1.  The environment is read once, at first use;
 * eg_switches_reload re-reads it (a host that changes a variable between two calls).  eg_switch_table writes
 * "<name>\t<class>\t<purpose>\n" per switch into text (at most cap - 1 bytes) and returns the length of the whole table.
 * No reference counterpart (its only switches are compile-time defines: -d:opencl, -d:exprgrad_fast_math). */
int eg_switches_reload(void);
int64_t eg_switch_table(char* text, size_t cap);

/* newGpuContext(device): gpu.nim:39, cl.nim:83-93. Creates one non-blocking in-order stream. */
int eg_ctx_create(int device, eg_ctx** out);
/* Same, but launches go to a stream owned by the caller (hipStream_t as void*; NULL = the
 * legacy default stream).  Lets a host that already owns streams (torch) stay in order. */
int eg_ctx_create_on_stream(int device, void* hip_stream, eg_ctx** out);
int eg_ctx_destroy(eg_ctx* ctx);
/* Block until everything queued on the context's stream has finished. */
int eg_ctx_sync(eg_ctx* ctx);
/* hipStream_t of the context, as void*. */
void* eg_ctx_stream(eg_ctx* ctx);
int eg_ctx_device(eg_ctx* ctx);

/* allocBuffer(ctx, size): gpu.nim:41, cl.nim:101-106. */
int eg_buf_alloc(eg_ctx* ctx, size_t bytes, eg_buf** out);
/* Wrap device memory owned by the caller (never freed by the library). */
int eg_buf_wrap(eg_ctx* ctx, void* device_ptr, size_t bytes, eg_buf** out);
/* dealloc(buffer): cl.nim:108-109 (the reference never calls it; here the caller must). */
int eg_buf_free(eg_buf* buf);
size_t eg_buf_size(const eg_buf* buf);
void* eg_buf_ptr(const eg_buf* buf);
/* write(buffer, data, size): gpu.nim:42, cl.nim:111-116. Blocking; bytes must equal the buffer size. */
int eg_buf_write(eg_buf* buf, const void* host, size_t bytes);
/* readInto(buffer, data): gpu.nim:45-46, cl.nim:128-139. Blocking; bytes must equal the buffer size. */
int eg_buf_read(eg_buf* buf, void* host, size_t bytes);
/* Pinned (page-locked) host memory: a write / read with such an array is a direct DMA at PCIe speed,
 * and a result tensor allocated here and recycled by the host never page-faults inside the copy (the
 * reference allocates a fresh host tensor per call, gpu.nim:68-70; reading 64 MiB into fresh pageable
 * memory costs more than the 4096^3 product that filled it).  No reference counterpart. */
int eg_host_alloc(size_t bytes, void** out);
int eg_host_free(void* p);
/* fill(buffer, value): gpu.nim:44, cl.nim:122-126. Asynchronous; pattern_bytes in {1,2,4,8}. */
int eg_buf_fill(eg_buf* buf, const void* pattern, size_t pattern_bytes);

/* compile(ctx, name, source): gpu.nim:48-49, cl.nim:149-179.  `source` is HIP C++ text holding
 * an `extern "C" __global__` function called `name`; built with hiprtc for the context's
 * device.  On failure the build log is in eg_last_error() (cl.nim:163-171). */
int eg_kernel_compile(eg_ctx* ctx, const char* name, const char* source, eg_kernel** out);
int eg_kernel_free(eg_kernel* kernel);
/* arg(kernel, index, buffer): gpu.nim:51, cl.nim:186-188.  Arguments are sticky like
 * clSetKernelArg: the JIT stub sets them, then launches (llvmgen.nim:461-500). */
int eg_kernel_set_arg_buf(eg_kernel* kernel, int index, eg_buf* buf);
/* arg[T](kernel, index, value): gpu.nim:50, cl.nim:181-184 (T = int / float32 / float64). */
int eg_kernel_set_arg_i64(eg_kernel* kernel, int index, int64_t value);
int eg_kernel_set_arg_f32(eg_kernel* kernel, int index, float value);
int eg_kernel_set_arg_f64(eg_kernel* kernel, int index, double value);
/* run(kernel, groupSize, localSize): gpu.nim:52, cl.nim:190-207.  dims in 1..3, dimension 0
 * is the fastest varying (passes.nim:2439-2448). groups[d] work-groups of local[d] threads. */
int eg_kernel_launch(eg_kernel* kernel, int dims, const int64_t* groups, const int64_t* local);

/* ------------------------------------------------------------------ group 2: library -- */
/* All functions enqueue on the context's stream and return immediately.  `accumulate` != 0
 * means `out += value` (the `++=` of parser.nim:677-681 on an already written tensor),
 * 0 means `out = value` (first writer, passes.nim:888-897). */

/* Contraction  C[m,n] (+)= sum_k opA(m,k) * opB(k,n)  (+ bias[n]).
 *   trans_a == 0: A is [M,K] row-major (lda >= K);  != 0: A is [K,M] (lda >= M)
 *   trans_b == 0: B is [K,N] row-major (ldb >= N);  != 0: B is [N,K] (ldb >= K)
 * Covers matmul (base.nim:27-28), dense forward with its bias kernel folded in
 * (dnn.nim:19-24), and the two derived gradient contractions of passes.nim:519-549:
 *   gradA[y,it] += g[y,x]*B[it,x]  -> trans_b = 1      gradB[it,x] += A[y,it]*g[y,x] -> trans_a = 1
 * bias may be NULL. */
int eg_sgemm(eg_ctx* ctx, int trans_a, int trans_b, int64_t M, int64_t N, int64_t K,
             const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc,
             int accumulate, const float* bias);

/* out[y,x] (+)= bias[x]   — dense's second kernel, dnn.nim:22-24. out is [rows, cols]. */
int eg_bias_add(eg_ctx* ctx, int64_t rows, int64_t cols, const float* bias, float* out,
                int accumulate);
/* out[x] (+)= sum_y in[y,x]  — derived bias gradient (Appendix A.1 G5/G2 of SURVEY.md). */
int eg_colsum(eg_ctx* ctx, int64_t rows, int64_t cols, const float* in, float* out,
              int accumulate);
/* out[y] (+)= sum_x in[y,x]  — softmax.sums without the exp (dnn.nim:90-92 shape). */
int eg_rowsum(eg_ctx* ctx, int64_t rows, int64_t cols, const float* in, float* out,
              int accumulate);
/* out[0] (+)= sum_i in[i]    — scalar losses, base.nim:57-67 (no GPU lowering in the reference). */
int eg_sum(eg_ctx* ctx, int64_t n, const float* in, float* out, int accumulate);
/* y[i] += alpha * x[i]       — gradientDescent, base.nim:37-38 (alpha = -rate). */
int eg_axpy(eg_ctx* ctx, int64_t n, float alpha, const float* x, float* y);
/* out[i] = value             — result zeroing (model.nim:318, 383), gradLoss = 1 (passes.nim:575-606). */
int eg_fill_f32(eg_ctx* ctx, int64_t n, float value, float* out);

/* Uniform fill in [lo, hi) of a TensorRandom tensor (`rand`, parser.nim:732-736; the reference
 * fills these on the host and uploads them on every call, model.nim:310-314).  Counter-based:
 * element i = hash(state[0] = seed, state[1] = fills drawn so far, stream, i).  `state` is two
 * uint64 in DEVICE memory; eg_rng_advance increments state[1] on the stream, so a captured launch
 * sequence draws fresh numbers on every replay. */
int eg_fill_uniform(eg_ctx* ctx, int64_t n, float lo, float hi, const uint64_t* state, uint64_t stream, float* out);
int eg_rng_advance(eg_ctx* ctx, uint64_t* state);

/* Fixed-function elementwise maps of the layer library (raw-indexed `{it}` kernels). */
enum eg_map_op {
  EG_MAP_IDENTITY = 0,
  EG_MAP_RELU = 1,        /* select(x >= 0, x, 0)                      dnn.nim:26-27  */
  EG_MAP_LEAKY_RELU = 2,  /* select(x >= 0, 1, p) * x                  dnn.nim:29-30  */
  EG_MAP_SIGMOID = 3,     /* 1 / (1 + exp(-x))                         dnn.nim:32-33  */
  EG_MAP_TANH = 4,        /* (e^x - e^-x) / (e^x + e^-x)               dnn.nim:35-40  */
  EG_MAP_SCALE = 5,       /* x * p                                     base.nim:24    */
  EG_MAP_SIN = 6,         /* sin(x)                                    dnn.nim:42-43  */
  EG_MAP_XOR_LEAKY = 7,   /* select(x <= 0, p*x, x)   xor_from_scratch.nim:22          */
  EG_MAP_EXP = 8
};
/* out[i] (+)= f(in[i]; param). */
int eg_map(eg_ctx* ctx, int op, int64_t n, const float* in, float* out, float param,
           int accumulate);
/* gin[i] (+)= gout[i] * f'(in[i]; param) — the kernel passes.nim:519-549 derives from the map
 * above (recomputing from the forward input, no saved activation). */
int eg_map_grad(eg_ctx* ctx, int op, int64_t n, const float* in, const float* gout, float* gin,
                float param, int accumulate);

/* Direct (im2col-free) convolution, valid padding, stride 1 (dnn.nim:45-49):
 *   out[n,y,x,f] (+)= sum_{dy,dx,c} img[n,y+dy,x+dx,c] * flt[f,dy,dx,c]
 * img [N,H,W,C], flt [F,FH,FW,C], out [N,H-FH+1,W-FW+1,F], all dense row-major. */
int eg_conv2_nhwc(eg_ctx* ctx, int64_t N, int64_t H, int64_t W, int64_t C, int64_t F, int64_t FH,
                  int64_t FW, const float* img, const float* flt, float* out, int accumulate);

/* The two gradients derive (passes.nim:383-549) produces for conv2 (dnn.nim:45-49), which the
 * reference runs as the forward loop nest with the roles of the tensors exchanged:
 *   grad_filter: gflt[f,dy,dx,c]     (+)= sum_{n,y,x}   gout[n,y,x,f] * img[n,y+dy,x+dx,c]
 *   grad_image:  gimg[n,y+dy,x+dx,c] (+)= sum_{f}       gout[n,y,x,f] * flt[f,dy,dx,c]
 * img / gimg [N,H,W,C], flt / gflt [F,FH,FW,C], gout [N,H-FH+1,W-FW+1,F]; accumulate = 0 overwrites
 * the whole destination. */
int eg_conv2_nhwc_grad_filter(eg_ctx* ctx, int64_t N, int64_t H, int64_t W, int64_t C, int64_t F, int64_t FH,
                              int64_t FW, const float* img, const float* gout, float* gflt, int accumulate);
int eg_conv2_nhwc_grad_image(eg_ctx* ctx, int64_t N, int64_t H, int64_t W, int64_t C, int64_t F, int64_t FH,
                             int64_t FW, const float* flt, const float* gout, float* gimg, int accumulate);

/* float64 forms of the library kernels — what a `compile[float64]` model (model.nim:253-260: every kernel of the
 * program instantiated over Scalar64) runs on; same conventions as their float32 namesakes above.
 *   eg_dgemm: `v_mfma_f64_16x16x4_f64` tiles (exact float64 multiply-adds in k order; sliced products are summed in a
 *   fixed order), eg_colsum_f64 / eg_fill_f64 / eg_fill_uniform_f64: as eg_colsum / eg_fill_f32 / eg_fill_uniform. */
int eg_dgemm(eg_ctx* ctx, int trans_a, int trans_b, int64_t M, int64_t N, int64_t K, const double* A, int64_t lda,
             const double* B, int64_t ldb, double* C, int64_t ldc, int accumulate, const double* bias);
int eg_colsum_f64(eg_ctx* ctx, int64_t rows, int64_t cols, const double* in, double* out, int accumulate);
int eg_fill_f64(eg_ctx* ctx, int64_t n, double value, double* out);
int eg_fill_uniform_f64(eg_ctx* ctx, int64_t n, double lo, double hi, const uint64_t* state, uint64_t stream, double* out);

/* ------------------------------------------------------------------ group 3: model ---- */
/* A program is the text form of exprgrad's `Program` (ir.nim:247-270) before `generate`:
 * tensors, targets, and per target the ordered list of `++=` kernel descriptions
 * (loops + reads + expr + write, ir.nim:211-230; computed index parts of an operand —
 * LinearIndex.setup, ir.nim:120-123 — as `idx` instructions; a customGrad block, ir.nim:203-209,
 * as nested kernels) plus the GenBackwards / GenGradient placeholders (ir.nim:196-204); reshape
 * (GenReshape) arrives as the copy kernel + shape constraint generate makes of it
 * (passes.nim:643-688).  Grammar: DESIGN.md "Kernel-description text".
 *
 * eg_model_compile does what newModel does for a CompileGpu target (model.nim:232-251):
 * autodiff + dead-kernel elimination, pattern-match every kernel to a library kernel or
 * generate HIP source for it, hiprtc-build, allocate parameters. */
int eg_model_compile(eg_ctx* ctx, const char* program_text, eg_model** out);
int eg_model_free(eg_model* model);

/* Introspection (model.emitIr, model.nim:262-264: the lowered plan as text). Returned pointer
 * is owned by the model and valid until the next call of the same function. */
const char* eg_model_plan_text(eg_model* model);
/* The launch sequence of the plan the target last ran with (shape dependent: fusion groups,
 * contractions with a fused epilogue, graph ranges), one line per launch; "" before the first
 * run.  Same ownership rule. */
const char* eg_model_launch_text(eg_model* model, const char* target);
/* Number of kernels in a target after autodiff + elimination, or -1. */
int eg_model_kernel_count(eg_model* model, const char* target);

/* Parameter / cache access (Model.params, model.nim:37-38).  Parameters live on the device;
 * these copy (blocking).  `name` is the tensor name or "#<id>". */
int eg_model_tensor_count(eg_model* model);
int eg_model_param_info(eg_model* model, int tensor_id, int* kind, int* rank, int64_t* shape8,
                        char* name, size_t name_cap);
int eg_model_param_write(eg_model* model, int tensor_id, const float* host, int64_t count);
int eg_model_param_read(eg_model* model, int tensor_id, float* host, int64_t count);
/* Device pointer of a parameter or of its gradient in `target` (NULL if absent). For the
 * data-parallel exchange step: gradients are laid out back to back in one flat bucket. */
int eg_model_grad_bucket(eg_model* model, const char* target, float** device_ptr,
                         int64_t* count);
/* Use caller-owned device memory (>= the bucket's count floats) for the gradients of `target`,
 * e.g. a torch tensor that torch.distributed can all-reduce in place. */
int eg_model_bind_grad_bucket(eg_model* model, const char* target, float* device_ptr,
                              int64_t count);
int eg_model_param_ptr(eg_model* model, int tensor_id, float** device_ptr, int64_t* count);

/* Bind an input (writeInput, model.nim:357-368).  Host form copies H2D (blocking, as the
 * reference does on every call); device form borrows the pointer. */
int eg_model_set_input_host(eg_model* model, const char* name, const float* host, int rank,
                            const int64_t* shape);
int eg_model_set_input_device(eg_model* model, const char* name, const float* device_ptr,
                              int rank, const int64_t* shape);

/* Forget every bound input (a later run of a target that needs one then fails with
 * EG_ERR_RUNTIME, like a call without that argument). */
int eg_model_clear_inputs(eg_model* model);

/* call (model.nim:392-406): infer shapes from the bound inputs, (re)allocate and zero the
 * target's result tensors, run its kernel list.  Asynchronous. */
int eg_model_run(eg_model* model, const char* target);
/* Split form used by the data-parallel step: everything up to (not including) the optimizer
 * kernels, then the optimizer kernels.  run == run_backward + run_update. */
int eg_model_run_backward(eg_model* model, const char* target);
int eg_model_run_update(eg_model* model, const char* target);
/* fit (model.nim:413-454): Model.epoch += 1, then the target once per mini-batch of `batch_size`
 * leading rows of every input (batchCount = rows of the first input div batch_size; the tail is
 * dropped).  data[i] is a host array (on_device[i] == 0: uploaded once, piecewise, overlapped
 * with the batches already queued) or a device array used in place; shapes8 holds 8 extents per
 * input.  Asynchronous with respect to the kernels; the host arrays are free on return.
 * EG_ERR_RUNTIME with the reference's message when n_inputs == 0.
 * After fit the model's bound inputs are UNSPECIFIED (the reference leaves the last batch's views bound,
 * model.nim:441-446): when the step's kernels read the batches where they lie in the data set, nothing is copied into
 * the inputs' staging buffers.  Bind inputs again before the next eg_model_run. */
int eg_model_fit(eg_model* model, const char* target, int n_inputs, const char* const* names, const float* const* data,
                 const int* on_device, const int* ranks, const int64_t* shapes8, int64_t batch_size);
/* Scale applied to the seed gradient gradLoss (passes.nim:594-596) — B_local/B_global for
 * batch-mean losses under data parallelism.  Default 1. */
int eg_model_set_grad_scale(eg_model* model, float scale);

/* readOutput (model.nim:370-376): shape of / blocking copy of the target's output tensor. */
int eg_model_output_shape(eg_model* model, const char* target, int* rank, int64_t* shape8);
int eg_model_read_output(eg_model* model, const char* target, float* host, int64_t count);
/* Any tensor of the last run of `target`, by id (debugging / parity tests). */
int eg_model_tensor_shape(eg_model* model, const char* target, int tensor_id, int* rank,
                          int64_t* shape8);
int eg_model_read_tensor(eg_model* model, const char* target, int tensor_id, float* host,
                         int64_t count);
/* (eg_model_read_tensor and eg_model_tensor_ptr refuse — EG_ERR_INVALID — a tensor whose values the last run's plan
 * never stored: predicate bits, tensors that lived in the LDS of a sample group; see eg_model_keep_values.) */
int eg_model_tensor_ptr(eg_model* model, const char* target, int tensor_id, float** device_ptr,
                        int64_t* count);
/* Intermediates are an implementation matter of a plan: a tensor every reader of which is fused away may never exist,
 * and one of which every reader only asks a yes / no question (relu's gradient, dnn.nim:26-27) may exist as one bit
 * per element — eg_model_read_tensor then refuses it, whatever the shapes made the planner decide.  on != 0: from the
 * next run on, every plan of this model keeps the VALUES of its result tensors where it would have kept bits (the
 * reference's own behaviour: every kernel's output is a tensor, model.nim:295-300); slower, for debugging and parity
 * tests.  The switch is part of the plan key: turning it back off returns to the plans made before. */
int eg_model_keep_values(eg_model* model, int on);

/* Model.epoch (model.nim:39, bumped by fit at model.nim:436). */
int eg_model_set_epoch(eg_model* model, int64_t epoch);
int64_t eg_model_epoch(eg_model* model);
/* Seed of the model's random tensors (`rand` / dropout; the reference draws them from Nim's global
 * generator, `randomize(seed)`): same seed, same call sequence -> same numbers.  Resets the draw counter. */
int eg_model_set_seed(eg_model* model, uint64_t seed);

/* save / loadModel (io/serialize.nim:344-379) in the reference's byte layout: integers little
 * endian, `int` = 8 bytes, bool = 1 byte, string / seq = int64 length + items, Table = int64 count +
 * (key, value) pairs, Tensor = bool isNil + seq[int] shape + elements (serialize.nim:21-75).
 *
 * state = the `params` and `caches` tables (Table[TensorId, Tensor[float32]]) exactly as
 * serialize.nim:348-349 writes them, read from / written to the DEVICE copies (the reference's GPU
 * path never copies trained parameters back, model.nim:326-345).  A Nim host stores isNil and its
 * `Program` itself and appends these bytes; on load it reads its Program, compiles, and hands the
 * rest of the stream to eg_model_load_state (*consumed = bytes used). */
int eg_model_state_bytes(eg_model* model, size_t* bytes);
int eg_model_store_state(eg_model* model, void* buf, size_t cap, size_t* written);
int eg_model_load_state(eg_model* model, const void* buf, size_t bytes, size_t* consumed);
/* Whole-file form for hosts without the Nim front-end: isNil, the program field holding the
 * kernel-description text as a serialize.nim string, params, caches, and behind them one int64 with
 * Model.epoch (which the reference forgets).  eg_model_load = read + eg_model_compile + load_state. */
int eg_model_save(eg_model* model, const char* path);
int eg_model_load(eg_ctx* ctx, const char* path, eg_model** out);
/* The kernel-description text the model was compiled from. */
const char* eg_model_source_text(eg_model* model);

/* compile[float64] (model.nim:253-260: toScalarType(T) -> Scalar64, every tensor a Tensor[float64]).  The scalar type
 * of a program is in its header line (`kd 1 f32` / `kd 1 f64`).  A float64 model runs its contractions on the float64
 * matrix cores (eg_dgemm) and everything else as generated kernels over `double`; the float32 fusion passes stay out of
 * it.  The entry points above that carry `float*` arrays refuse a float64 model (EG_ERR_INVALID) — a Tensor[float32]
 * does not type-check against a Model[float64] in the reference either — and these, their twins, refuse a float32 one.
 * Entry points that only hand out device pointers (eg_model_param_ptr, eg_model_tensor_ptr, eg_model_grad_bucket /
 * eg_model_bind_grad_bucket) return / take the address of DOUBLES for such a model, typed `float*`; their counts are
 * elements.  eg_model_save / eg_model_load / *_state store 8-byte elements (serialize.nim:35).  eg_model_step_dp
 * all-reduces the bucket as float64. */
int eg_model_scalar_bytes(eg_model* model); /* 4 or 8 */
int eg_model_param_write_f64(eg_model* model, int tensor_id, const double* host, int64_t count);
int eg_model_param_read_f64(eg_model* model, int tensor_id, double* host, int64_t count);
int eg_model_set_input_host_f64(eg_model* model, const char* name, const double* host, int rank, const int64_t* shape);
int eg_model_set_input_device_f64(eg_model* model, const char* name, const double* device_ptr, int rank,
                                  const int64_t* shape);
int eg_model_read_output_f64(eg_model* model, const char* target, double* host, int64_t count);
int eg_model_read_tensor_f64(eg_model* model, const char* target, int tensor_id, double* host, int64_t count);
int eg_model_fit_f64(eg_model* model, const char* target, int n_inputs, const char* const* names,
                     const double* const* data, const int* on_device, const int* ranks, const int64_t* shapes8,
                     int64_t batch_size);

/* ---------------------------------------------------------------------------------------------
 * Group 4 — data-parallel exchange (BASELINE.json north_star: "training-step batches
 * shard data-parallel across the 8 GPUs of one node with an RCCL all-reduce of parameter gradients
 * over xGMI before the gradientDescent optim step").  The reference has no multi-device code.
 * One process per GPU: rank 0 draws an id and hands its 128 bytes to the other ranks (file,
 * socket, environment — the host's business), every rank joins with its own context; the
 * collective runs on the context's stream, after the backward kernels, before the optimizer's.
 * librccl.so is opened on first use.
 * ------------------------------------------------------------------------------------------- */
int eg_dp_unique_id(void* id128);                     /* ncclGetUniqueId: 128 opaque bytes */
/* Blocks until every rank has joined; gives up with EG_ERR_RUNTIME after EG_DP_INIT_TIMEOUT_S seconds (default 180). */
int eg_dp_init(eg_ctx* ctx, const void* id128, int rank, int world, eg_dp** out);
int eg_dp_free(eg_dp* dp);
int eg_dp_rank(const eg_dp* dp);
int eg_dp_world(const eg_dp* dp);
/* What RCCL itself reports for the communicator (ncclCommCount / ncclCommUserRank), -1 if it cannot say: the figure a
 * benchmark line should carry as "ranks the collective ran over". */
int eg_dp_rccl_count(const eg_dp* dp);
int eg_dp_rccl_rank(const eg_dp* dp);
/* eg_model_step_dp may exchange the gradients that are complete early under the backward pass's last long contraction
 * (two all-reduce calls per step instead of one).  enabled = 0 keeps the single exchange after the backward pass
 * (initial value: 1, or 0 with EG_DP_NO_SPLIT=1); lets a host time both forms in one run.  Every rank must choose the same. */
int eg_dp_set_split(eg_dp* dp, int enabled);
/* In-place SUM over the ranks of `count` floats at `device_buf`; asynchronous on the stream. */
int eg_dp_allreduce_sum_f32(eg_dp* dp, float* device_buf, int64_t count);
int eg_dp_allreduce_sum_f64(eg_dp* dp, double* device_buf, int64_t count);
/* One training step on this rank's shard of the batch (inputs bound with eg_model_set_input_*):
 * eg_model_run_backward | all-reduce of the gradient bucket | eg_model_run_update.  mean != 0 for
 * losses that divide by the batch (mse, crossEntropy; base.nim:57-67): the seed gradient is scaled
 * by 1 / world so the summed gradients are those of the full batch; 0 for sum-type losses.
 * The model must have been compiled on dp's context (EG_ERR_INVALID otherwise). */
int eg_model_step_dp(eg_model* model, const char* target, eg_dp* dp, int mean);
/* How many all-reduce calls the last eg_model_step_dp issued: 1 = the whole bucket after the backward
 * pass; more = the gradients that were complete early went out on the side lane, under the last long
 * contraction (EG_DP_NO_SPLIT=1 forces 1). */
int eg_dp_last_pieces(const eg_dp* dp);

#ifdef __cplusplus
}
#endif
#endif /* EXPRGRAD_HIP_H */
