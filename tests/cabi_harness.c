/* A host that is neither Python nor C++: plain C (what Nim compiles to) driving libexprgrad_hip.so through
 * include/exprgrad_hip.h only — the calls nim/exprgrad/runtimes/hip.nim and hipmodel.nim make, in their order.
 *
 *   cabi_harness runtime                      group 1: devices, context, buffers (write / read / fill with the size
 *                                             checks of cl.nim:111-146), compile(name, source) + sticky arguments +
 *                                             launch (cl.nim:149-207), build log on a broken source (cl.nim:163-171)
 *   cabi_harness jit                          group 1 the way the reference's JIT launch stub uses it (llvmgen.nim:455-500):
 *                                             four of the reference's own lowered kernels (tests/cache, the .ir goldens) as the HIP
 *                                             text the clgen.nim patch emits, set-tensor / set-index / run, known answers
 *   cabi_harness model <program.kd> <case>    group 3: eg_model_compile -> param_write -> set_input_host -> run ->
 *                                             output_shape / read_output (Model.call, model.nim:392-406), run again
 *                                             for `apply`, param_read; values compared with the case file
 *
 * Case file (written by tests/test_gpu_cabi_harness.py from tests/golden/handwritten.json), one record per line:
 *   tol <t> | epoch <n> | param <id> <count> v... | input <name> <rank> d... v...
 *   call <target> <n names> name... <count> expected... | apply <target> | expect <id> <count> v...
 * Build: gcc -std=c99 -Iinclude tests/cabi_harness.c -Lexprgrad_amd/lib -lexprgrad_hip -Wl,-rpath,$PWD/exprgrad_amd/lib -lm
 * Exit status 0 = every comparison held.  No GPU: every entry point fails with an error text, the harness exits 2. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "exprgrad_hip.h"

#define CHECK(call)                                                                         \
  do {                                                                                      \
    int status_ = (call);                                                                   \
    if (status_ != 0) {                                                                     \
      fprintf(stderr, "%s:%d: %s -> %d: %s\n", __FILE__, __LINE__, #call, status_, eg_last_error()); \
      exit(2);                                                                              \
    }                                                                                       \
  } while (0)

static int failures = 0;

static void expect(int ok, const char* what) {
  if (!ok) {
    fprintf(stderr, "FAILED: %s\n", what);
    ++failures;
  }
}

static int close_enough(const float* got, const double* want, long n, double tol, const char* what) {
  double scale = 1e-30, worst = 0;
  for (long i = 0; i < n; ++i)
    if (fabs(want[i]) > scale) scale = fabs(want[i]);
  for (long i = 0; i < n; ++i) {
    const double d = fabs((double)got[i] - want[i]);
    if (!(d <= worst)) worst = d; /* (NaN-proof) */
  }
  const int ok = tol == 0.0 ? worst == 0.0 : worst <= tol * scale;
  if (!ok) fprintf(stderr, "FAILED: %s: max |got - want| = %.3e, allowed %.3e\n", what, worst, tol * scale);
  return ok;
}

static int close_enough_f64(const double* got, const double* want, long n, double tol, const char* what) {
  double scale = 1e-30, worst = 0;
  for (long i = 0; i < n; ++i)
    if (fabs(want[i]) > scale) scale = fabs(want[i]);
  for (long i = 0; i < n; ++i) {
    const double d = fabs(got[i] - want[i]);
    if (!(d <= worst)) worst = d;
  }
  const int ok = tol == 0.0 ? worst == 0.0 : worst <= tol * scale;
  if (!ok) fprintf(stderr, "FAILED: %s: max |got - want| = %.3e, allowed %.3e\n", what, worst, tol * scale);
  return ok;
}

/* ---------------------------------------------------------------------------------------- group 1 */
static const char* kSource =
    "extern \"C\" __global__ void scale_shift(float* x, long n, float a, float b) {\n"
    "  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;\n"
    "  if (i < n) x[i] = a * x[i] + b;\n"
    "}\n";

static int run_runtime(void) {
  int n_dev = 0;
  CHECK(eg_device_count(&n_dev));
  expect(n_dev >= 1, "at least one device");
  char name[256], vendor[256], version[256], compiler[512];
  int is_gpu = 0;
  CHECK(eg_device_info(0, name, sizeof name, vendor, sizeof vendor, version, sizeof version, &is_gpu));
  expect(is_gpu == 1, "device 0 is a GPU");
  CHECK(eg_compiler_info(compiler, sizeof compiler));
  printf("device 0: %s (%s, %s); run-time compiler: %s\n", name, vendor, version, compiler);

  eg_ctx* ctx = NULL;
  CHECK(eg_ctx_create(0, &ctx));
  enum { N = 1000 };
  float host[N], back[N];
  for (int i = 0; i < N; ++i) host[i] = (float)i * 0.5f;
  eg_buf* buf = NULL;
  CHECK(eg_buf_alloc(ctx, sizeof host, &buf));
  CHECK(eg_buf_write(buf, host, sizeof host));
  expect(eg_buf_write(buf, host, sizeof host - 4) != 0, "a write of another size than the buffer is refused (cl.nim:112-113)");
  expect(strlen(eg_last_error()) > 0, "... with an error text");

  eg_kernel* kernel = NULL;
  CHECK(eg_kernel_compile(ctx, "scale_shift", kSource, &kernel));
  CHECK(eg_kernel_set_arg_buf(kernel, 0, buf));
  CHECK(eg_kernel_set_arg_i64(kernel, 1, (int64_t)N));
  CHECK(eg_kernel_set_arg_f32(kernel, 2, 2.0f));
  CHECK(eg_kernel_set_arg_f32(kernel, 3, 1.0f));
  int64_t groups[1] = {(N + 63) / 64}, local[1] = {64};
  CHECK(eg_kernel_launch(kernel, 1, groups, local));
  CHECK(eg_kernel_set_arg_f32(kernel, 3, -1.0f)); /* arguments are sticky: only the changed one is set again */
  CHECK(eg_kernel_launch(kernel, 1, groups, local));
  CHECK(eg_buf_read(buf, back, sizeof back));
  int same = 1;
  for (int i = 0; i < N; ++i) same = same && back[i] == 2.0f * (2.0f * host[i] + 1.0f) - 1.0f;
  expect(same, "two launches with sticky arguments");

  const float seven = 7.0f;
  CHECK(eg_buf_fill(buf, &seven, sizeof seven));
  CHECK(eg_buf_read(buf, back, sizeof back));
  same = 1;
  for (int i = 0; i < N; ++i) same = same && back[i] == 7.0f;
  expect(same, "fill with a 4-byte pattern (cl.nim:122-126)");

  eg_kernel* broken = NULL;
  expect(eg_kernel_compile(ctx, "nope", "extern \"C\" __global__ void nope() { this is not HIP }", &broken) != 0,
         "a broken source does not compile");
  expect(strstr(eg_last_error(), "error") != NULL, "... and the error text carries the build log (cl.nim:163-171)");

  CHECK(eg_kernel_free(kernel));
  CHECK(eg_buf_free(buf));
  CHECK(eg_ctx_destroy(ctx));
  return failures;
}


/* ------------------------------------------------------------------- group 1, as the reference's JIT stub drives it
 * Level 1 of INTEGRATION.md: the reference keeps generating kernel TEXT (clgen.nim) and its JIT-compiled host code calls
 * setGpuKernelTensor / setGpuKernelIndex / runGpuKernel (llvmgen.nim:455-500 -> model.nim:148-172 -> runtimes/hip.nim ->
 * eg_kernel_set_arg_buf / _i64 / eg_kernel_launch).  The four sources below are the reference's own lowered GPU kernels
 * — tests/cache/{matmul_basic, matmul_schedule_tiled16, relu_basic, conv1_basic}.ir, register names kept — written as
 * the HIP C++ text the clgen.nim patch of nim/PATCHES.md section 5 emits for them (get_local_id(d) -> threadIdx,
 * get_group_id(d) -> blockIdx, __local -> __shared__, barrier -> __syncthreads, float literals with an f suffix; the
 * kernel is always called cl_kernel, llvmgen.nim:458; closure tensors in id order, then the Index registers).  Launch
 * geometry as builtinRunGpuKernel computes it (model.nim:150-164): groups = ceil(global / local) per dimension, local =
 * 16 (ir.nim:283).  Expected values: the host loops the reference's own tests compare with (test_gpu.nim:57-69,
 * 190-209, 238-246), on inputs that make float32 exact. */
static const char* kMatmulBasic =
    "extern \"C\" __global__ void cl_kernel(float* tensor0, float* tensor1, float* tensor2) {\n"
    "  long reg21 = threadIdx.x;\n  long reg22 = blockIdx.x;\n  long reg25 = threadIdx.y;\n  long reg26 = blockIdx.y;\n"
    "  long reg3 = ((reg22 * 16) + reg21);\n"
    "  long reg0 = ((reg26 * 16) + reg25);\n"
    "  long reg10 = (reg0 * 64);\n"
    "  long reg17 = (reg3 + (reg0 * 64));\n"
    "  for(long reg1 = 0; reg1 < 64; ++reg1) {\n"
    "    tensor0[reg17] += (tensor1[(reg1 + reg10)] * tensor2[(reg3 + (reg1 * 64))])/* tensor0 [64, 64] */;\n"
    "  } // false\n"
    "}\n";

static const char* kMatmulTiled16 =
    "extern \"C\" __global__ void cl_kernel(float* tensor0, float* tensor1, float* tensor2) {\n"
    "  long reg14 = threadIdx.x;\n  long reg64 = blockIdx.x;\n  long reg15 = threadIdx.y;\n  long reg66 = blockIdx.y;\n"
    "  long reg11 = (reg64 * 16);\n"
    "  long reg3 = (reg11 + reg14);\n"
    "  long reg10 = (reg66 * 16);\n"
    "  long reg0 = (reg10 + reg15);\n"
    "  __shared__ float reg12[256];\n"
    "  long reg18 = ((reg15 * 16) + reg14);\n"
    "  long reg24 = ((reg10 + reg15) * 64);\n"
    "  __shared__ float reg13[256];\n"
    "  long reg29 = ((reg15 * 16) + reg14);\n"
    "  long reg31 = (reg11 + reg14);\n"
    "  long reg45 = (((reg10 * -1) + reg0) * 16);\n"
    "  long reg49 = (reg3 + (reg11 * -1));\n"
    "  long reg58 = (reg3 + (reg0 * 64));\n"
    "  for(long reg9 = 0; reg9 < 64; reg9 += 16) {\n"
    "    __syncthreads();\n"
    "    reg12[reg18] = tensor1[((reg9 + reg14) + reg24)];\n"
    "    reg13[reg29] = tensor2[(reg31 + ((reg15 + reg9) * 64))];\n"
    "    __syncthreads();\n"
    "    long reg38 = (reg9 * -1);\n"
    "    long reg51 = (reg9 * -1);\n"
    "    for(long reg1 = reg9; reg1 < (reg9 + 16); ++reg1) {\n"
    "      tensor0[reg58] += (reg12[((reg1 + reg38) + reg45)] * reg13[(reg49 + ((reg1 + reg51) * 16))])/* tensor0 [64, 64] */;\n"
    "    } // false\n"
    "  } // false\n"
    "}\n";

static const char* kReluBasic =
    "extern \"C\" __global__ void cl_kernel(float* tensor0, float* tensor1, long reg9) {\n"
    "  long reg11 = threadIdx.x;\n  long reg12 = blockIdx.x;\n"
    "  long reg1 = ((reg12 * 16) + reg11);\n"
    "  if ((reg1 < reg9)) {\n"
    "    float reg2 = tensor1[reg1];\n"
    "    tensor0[reg1] = ((0.0f < reg2) ? reg2 : (0.01f * reg2))/* tensor0 [-1] */;\n"
    "  }\n"
    "}\n";

static const char* kConv1Basic =
    "extern \"C\" __global__ void cl_kernel(float* tensor0, float* tensor1, float* tensor2) {\n"
    "  long reg11 = threadIdx.x;\n  long reg12 = blockIdx.x;\n"
    "  long reg0 = ((reg12 * 16) + reg11);\n"
    "  for(long reg1 = 0; reg1 < 5; ++reg1) {\n"
    "    tensor0[reg0] += (tensor1[(reg1 + reg0)] * tensor2[reg1])/* tensor0 [64] */;\n"
    "  } // false\n"
    "}\n";

/* builtinRunGpuKernel, model.nim:150-164 */
static void run_gpu_kernel(eg_kernel* kernel, int work_dims, const int64_t* global_size, const int64_t* local_size) {
  int64_t groups[3];
  for (int d = 0; d < work_dims; ++d) groups[d] = (global_size[d] + local_size[d] - 1) / local_size[d];
  CHECK(eg_kernel_launch(kernel, work_dims, groups, local_size));
}

static eg_buf* upload(eg_ctx* ctx, const float* data, size_t count) {
  eg_buf* b = NULL;
  CHECK(eg_buf_alloc(ctx, count * sizeof(float), &b));
  if (data) {
    CHECK(eg_buf_write(b, data, count * sizeof(float)));
  } else { /* allocShapes zero-fills every result tensor before a call (model.nim:318): the kernels accumulate */
    const float zero = 0.0f;
    CHECK(eg_buf_fill(b, &zero, sizeof zero));
  }
  return b;
}

static int run_jit(void) {
  eg_ctx* ctx = NULL;
  CHECK(eg_ctx_create(0, &ctx));
  static float a[64 * 64], b[64 * 64], want[64 * 64], got[64 * 64];
  for (int i = 0; i < 64 * 64; ++i) {
    a[i] = (float)((i * 7 + 3) % 17) / 8.0f;   /* multiples of 1/8 below 2.2: 64-term sums of products are exact in float32 */
    b[i] = (float)((i * 5 + 1) % 13) / 8.0f - 0.75f;
  }
  for (int y = 0; y < 64; ++y)                  /* tensors.nim:248-256, the `*` the reference's test compares with */
    for (int x = 0; x < 64; ++x) {
      float acc = 0.0f;
      for (int it = 0; it < 64; ++it) acc += a[y * 64 + it] * b[it * 64 + x];
      want[y * 64 + x] = acc;
    }
  const int64_t local2[2] = {16, 16}, global2[2] = {64, 64};
  const char* sources[2] = {kMatmulBasic, kMatmulTiled16};
  const char* names[2] = {"matmul_basic.ir", "matmul_schedule_tiled16.ir (shared cache + barriers)"};
  for (int v = 0; v < 2; ++v) {
    eg_buf *ta = upload(ctx, a, 64 * 64), *tb = upload(ctx, b, 64 * 64), *tc = upload(ctx, NULL, 64 * 64);
    eg_kernel* k = NULL;
    CHECK(eg_kernel_compile(ctx, "cl_kernel", sources[v], &k));
    CHECK(eg_kernel_set_arg_buf(k, 0, tc)); /* setGpuKernelTensor per closure tensor (llvmgen.nim:462-470) */
    CHECK(eg_kernel_set_arg_buf(k, 1, ta));
    CHECK(eg_kernel_set_arg_buf(k, 2, tb));
    run_gpu_kernel(k, 2, global2, local2);
    CHECK(eg_buf_read(tc, got, sizeof got));
    int same = 1;
    for (int i = 0; i < 64 * 64; ++i) same = same && got[i] == want[i];
    expect(same, names[v]);
    CHECK(eg_kernel_free(k));
    CHECK(eg_buf_free(ta));
    CHECK(eg_buf_free(tb));
    CHECK(eg_buf_free(tc));
  }
  { /* relu/basic (test_gpu.nim:238-246): extent 6 on a work-group of 16 -> the bounds `if`; reg9 = len(x) as an Index argument */
    const float x[6] = {1, 2, -1, -2, 0, 3}, y[6] = {1, 2, -0.01f, -0.02f, 0, 3};
    float out[6];
    eg_buf *tx = upload(ctx, x, 6), *ty = upload(ctx, NULL, 6);
    eg_kernel* k = NULL;
    CHECK(eg_kernel_compile(ctx, "cl_kernel", kReluBasic, &k));
    CHECK(eg_kernel_set_arg_buf(k, 0, ty));
    CHECK(eg_kernel_set_arg_buf(k, 1, tx));
    CHECK(eg_kernel_set_arg_i64(k, 2, 6)); /* setGpuKernelIndex (llvmgen.nim:471-481) */
    const int64_t local1[1] = {16}, global1[1] = {6};
    run_gpu_kernel(k, 1, global1, local1);
    CHECK(eg_buf_read(ty, out, sizeof out));
    int same = 1;
    for (int i = 0; i < 6; ++i) same = same && out[i] == y[i];
    expect(same, "relu_basic.ir");
    CHECK(eg_kernel_free(k));
    CHECK(eg_buf_free(tx));
    CHECK(eg_buf_free(ty));
  }
  { /* conv1/basic (test_gpu.nim:190-209): image 68, filter 5 -> 64 outputs */
    float image[68], filter[5], res_want[64], res[64];
    for (int i = 0; i < 68; ++i) image[i] = (float)((i * 11 + 2) % 9) / 4.0f;
    for (int i = 0; i < 5; ++i) filter[i] = (float)(i - 2) / 2.0f;
    for (int x = 0; x < 64; ++x) {
      res_want[x] = 0.0f;
      for (int dx = 0; dx < 5; ++dx) res_want[x] += image[x + dx] * filter[dx];
    }
    eg_buf *ti = upload(ctx, image, 68), *tf = upload(ctx, filter, 5), *tr = upload(ctx, NULL, 64);
    eg_kernel* k = NULL;
    CHECK(eg_kernel_compile(ctx, "cl_kernel", kConv1Basic, &k));
    CHECK(eg_kernel_set_arg_buf(k, 0, tr));
    CHECK(eg_kernel_set_arg_buf(k, 1, ti));
    CHECK(eg_kernel_set_arg_buf(k, 2, tf));
    const int64_t local1[1] = {16}, global1[1] = {64};
    run_gpu_kernel(k, 1, global1, local1);
    CHECK(eg_buf_read(tr, res, sizeof res));
    int same = 1;
    for (int i = 0; i < 64; ++i) same = same && res[i] == res_want[i];
    expect(same, "conv1_basic.ir");
    CHECK(eg_kernel_free(k));
    CHECK(eg_buf_free(ti));
    CHECK(eg_buf_free(tf));
    CHECK(eg_buf_free(tr));
  }
  CHECK(eg_ctx_destroy(ctx));
  printf("jit stub path: 4 reference kernels through eg_kernel_compile / sticky arguments / eg_kernel_launch\n");
  return failures;
}

/* ---------------------------------------------------------------------------------------- group 3 */
static char* read_file(const char* path) {
  FILE* fp = fopen(path, "rb");
  if (!fp) {
    fprintf(stderr, "cannot open %s\n", path);
    exit(2);
  }
  fseek(fp, 0, SEEK_END);
  long n = ftell(fp);
  fseek(fp, 0, SEEK_SET);
  char* text = (char*)malloc((size_t)n + 1);
  if (fread(text, 1, (size_t)n, fp) != (size_t)n) exit(2);
  text[n] = 0;
  fclose(fp);
  return text;
}

#define MAX_INPUTS 8
struct input {
  char name[64];
  int rank;
  int64_t shape[8];
  float* data;
  double* data64; /* the same values as read (a float64 model's input) */
  long count;
};

static double* read_doubles(FILE* fp, long n) {
  double* v = (double*)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
  for (long i = 0; i < n; ++i)
    if (fscanf(fp, "%lf", &v[i]) != 1) {
      fprintf(stderr, "case file: number expected\n");
      exit(2);
    }
  return v;
}

static int run_model(const char* kd_path, const char* case_path) {
  char* text = read_file(kd_path);
  eg_ctx* ctx = NULL;
  eg_model* model = NULL;
  CHECK(eg_ctx_create(0, &ctx));
  CHECK(eg_model_compile(ctx, text, &model));
  /* compile[float64] (header line `kd 1 f64`): the same walk through the _f64 twins of the typed entry points */
  const int f64 = eg_model_scalar_bytes(model) == 8;
  FILE* fp = fopen(case_path, "r");
  if (!fp) {
    fprintf(stderr, "cannot open %s\n", case_path);
    return 2;
  }
  struct input inputs[MAX_INPUTS];
  int n_inputs = 0;
  double tol = 1e-6;
  char word[64];
  while (fscanf(fp, "%63s", word) == 1) {
    if (!strcmp(word, "tol")) {
      if (fscanf(fp, "%lf", &tol) != 1) return 2;
    } else if (!strcmp(word, "epoch")) {
      long e;
      if (fscanf(fp, "%ld", &e) != 1) return 2;
      CHECK(eg_model_set_epoch(model, e));
    } else if (!strcmp(word, "param") || !strcmp(word, "expect")) {
      const int is_param = word[0] == 'p';
      int id;
      long count;
      if (fscanf(fp, "%d %ld", &id, &count) != 2) return 2;
      double* v = read_doubles(fp, count);
      float* f = (float*)malloc(sizeof(float) * (size_t)(count > 0 ? count : 1));
      if (is_param && f64) {
        CHECK(eg_model_param_write_f64(model, id, v, count));
      } else if (is_param) {
        for (long i = 0; i < count; ++i) f[i] = (float)v[i];
        CHECK(eg_model_param_write(model, id, f, count));
      } else {
        char what[96];
        snprintf(what, sizeof what, "tensor %d after the step", id);
        if (f64) {
          double* d = (double*)malloc(sizeof(double) * (size_t)(count > 0 ? count : 1));
          CHECK(eg_model_param_read_f64(model, id, d, count));
          if (!close_enough_f64(d, v, count, tol, what)) ++failures;
          free(d);
        } else {
          CHECK(eg_model_param_read(model, id, f, count));
          if (!close_enough(f, v, count, tol, what)) ++failures;
        }
      }
      free(f);
      free(v);
    } else if (!strcmp(word, "input")) {
      if (n_inputs == MAX_INPUTS) return 2;
      struct input* in = &inputs[n_inputs++];
      if (fscanf(fp, "%63s %d", in->name, &in->rank) != 2) return 2;
      in->count = 1;
      for (int d = 0; d < in->rank; ++d) {
        long s;
        if (fscanf(fp, "%ld", &s) != 1) return 2;
        in->shape[d] = s;
        in->count *= s;
      }
      double* v = read_doubles(fp, in->count);
      in->data = (float*)malloc(sizeof(float) * (size_t)(in->count > 0 ? in->count : 1));
      for (long i = 0; i < in->count; ++i) in->data[i] = (float)v[i];
      in->data64 = v;
    } else if (!strcmp(word, "call") || !strcmp(word, "apply")) {
      const int is_call = word[0] == 'c';
      char target[64];
      if (fscanf(fp, "%63s", target) != 1) return 2;
      /* Model.call binds its arguments, nothing else (model.nim:400-402): an input that is not named is not bound */
      CHECK(eg_model_clear_inputs(model));
      int n_names = n_inputs;
      char names[MAX_INPUTS][64];
      if (is_call) {
        if (fscanf(fp, "%d", &n_names) != 1) return 2;
        for (int i = 0; i < n_names; ++i)
          if (fscanf(fp, "%63s", names[i]) != 1) return 2;
      } else {
        for (int i = 0; i < n_inputs; ++i) strcpy(names[i], inputs[i].name);
      }
      for (int i = 0; i < n_names; ++i)
        for (int j = 0; j < n_inputs; ++j)
          if (!strcmp(names[i], inputs[j].name)) {
            if (f64) CHECK(eg_model_set_input_host_f64(model, inputs[j].name, inputs[j].data64, inputs[j].rank, inputs[j].shape));
            else CHECK(eg_model_set_input_host(model, inputs[j].name, inputs[j].data, inputs[j].rank, inputs[j].shape));
          }
      CHECK(eg_model_run(model, target));
      if (is_call) {
        long count;
        if (fscanf(fp, "%ld", &count) != 1) return 2;
        double* want = read_doubles(fp, count);
        int rank = 0;
        int64_t shape8[8] = {0};
        CHECK(eg_model_output_shape(model, target, &rank, shape8));
        long have = 1;
        for (int d = 0; d < rank; ++d) have *= shape8[d];
        char what[96];
        snprintf(what, sizeof what, "output of target %s", target);
        if (have != count) {
          fprintf(stderr, "FAILED: %s has %ld elements, %ld expected\n", what, have, count);
          ++failures;
        } else if (f64) {
          double* got = (double*)malloc(sizeof(double) * (size_t)(count > 0 ? count : 1));
          CHECK(eg_model_read_output_f64(model, target, got, count));
          if (!close_enough_f64(got, want, count, tol, what)) ++failures;
          free(got);
        } else {
          float* got = (float*)malloc(sizeof(float) * (size_t)(count > 0 ? count : 1));
          CHECK(eg_model_read_output(model, target, got, count));
          if (!close_enough(got, want, count, tol, what)) ++failures;
          free(got);
        }
        free(want);
      } else {
        CHECK(eg_ctx_sync(ctx));
      }
    } else {
      fprintf(stderr, "case file: unknown record '%s'\n", word);
      return 2;
    }
  }
  fclose(fp);
  expect(eg_model_run(model, "no-such-target") == EG_ERR_RUNTIME, "an unknown target is a RuntimeError (model.nim:395-396)");
  CHECK(eg_model_free(model));
  CHECK(eg_ctx_destroy(ctx));
  free(text);
  return failures;
}

int main(int argc, char** argv) {
  int rc;
  if (argc >= 2 && !strcmp(argv[1], "runtime")) {
    rc = run_runtime();
  } else if (argc >= 2 && !strcmp(argv[1], "jit")) {
    rc = run_jit();
  } else if (argc >= 4 && !strcmp(argv[1], "model")) {
    rc = run_model(argv[2], argv[3]);
  } else {
    fprintf(stderr, "usage: %s runtime | jit | model <program.kd> <case file>\n", argv[0]);
    return 2;
  }
  printf(rc == 0 ? "ok\n" : "%d comparison(s) failed\n", rc);
  return rc == 0 ? 0 : 1;
}
