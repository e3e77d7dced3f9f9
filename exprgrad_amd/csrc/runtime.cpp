// Group 1 of the C ABI: the GPU runtime exprgrad expects from a backend.
//
// Replaces exprgrad/runtimes/cl.nim (OpenCL) behind the proc set listed in
// exprgrad/runtimes/gpu.nim:24-52.  HIP mapping:
//   platform/device enumeration (cl.nim:45-81)      -> hipGetDeviceCount / hipGetDeviceProperties
//   context + one in-order queue (cl.nim:83-93)     -> one non-blocking hipStream_t per eg_ctx
//   createBuffer / blocking write / read / fill     -> hipMalloc / hipMemcpyAsync+sync / hipMemsetD*Async
//   clCreateProgramWithSource + clBuildProgram      -> hiprtc (source text -> code object) + hipModuleLoadData
//   clSetKernelArg (sticky) + clEnqueueNDRangeKernel-> stored argument block + hipModuleLaunchKernel

#include <cstdlib>
#include <cstring>
#include <memory>

#include "eg_internal.hpp"

namespace eg {
static thread_local std::string g_error;

void set_error(const char* fmt, ...) {
  char stack[2048];
  va_list ap;
  va_start(ap, fmt);
  int n = vsnprintf(stack, sizeof(stack), fmt, ap);
  va_end(ap);
  if (n < (int)sizeof(stack)) {
    g_error.assign(stack, n < 0 ? 0 : n);
    return;
  }
  std::vector<char> heap(n + 1);
  va_start(ap, fmt);
  vsnprintf(heap.data(), heap.size(), fmt, ap);
  va_end(ap);
  g_error.assign(heap.data(), n);
}
void clear_error() { g_error.clear(); }

// EG_POISON=1 (debugging aid): every scratch block is filled with NaN bit patterns right before a
// library call uses it, and models do the same with the parts of their result arena that the
// kernels are supposed to overwrite completely.  A read of memory nobody wrote then shows up as NaN
// in the result instead of as whatever an earlier launch left there.
bool poison_enabled() {
  static const bool on = [] {
    const char* e = eg::sw::raw("EG_POISON");
    return e && e[0] && e[0] != '0';
  }();
  return on;
}

static int poison_block(eg_ctx* ctx, void* p, size_t bytes) {
  if (!poison_enabled() || !p || !bytes) return EG_OK;
  EG_HIP_CHECK(hipMemsetAsync(p, 0xFF, bytes, ctx->stream));
  return EG_OK;
}

int ensure_workspace(eg_ctx* ctx, size_t bytes) {
  if (bytes <= ctx->workspace_bytes) return poison_block(ctx, ctx->workspace, ctx->workspace_bytes);
  EG_HIP_CHECK(hipSetDevice(ctx->device));
  // Kernels already queued may still be using the old block.
  EG_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  if (ctx->workspace) EG_HIP_CHECK(hipFree(ctx->workspace));
  ctx->workspace = nullptr;
  ctx->workspace_bytes = 0;
  size_t want = bytes + bytes / 4;
  EG_HIP_CHECK(hipMalloc(&ctx->workspace, want));
  ctx->workspace_bytes = want;
  return poison_block(ctx, ctx->workspace, ctx->workspace_bytes);
}

int ensure_aux(eg_ctx* ctx, size_t bytes) {
  if (bytes <= ctx->aux_bytes) return poison_block(ctx, ctx->aux, ctx->aux_bytes);
  EG_HIP_CHECK(hipSetDevice(ctx->device));
  EG_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  if (ctx->aux) EG_HIP_CHECK(hipFree(ctx->aux));
  ctx->aux = nullptr;
  ctx->aux_bytes = 0;
  size_t want = bytes + bytes / 4;
  EG_HIP_CHECK(hipMalloc(&ctx->aux, want));
  ctx->aux_bytes = want;
  return poison_block(ctx, ctx->aux, ctx->aux_bytes);
}
}  // namespace eg

using eg::set_error;

// A loaded code object; several kernels built in one hiprtc program share it.
struct eg_module {
  int device = 0;
  hipModule_t handle = nullptr;
  ~eg_module() {
    if (handle) {
      hipSetDevice(device);
      hipModuleUnload(handle);
    }
  }
};

struct eg_kernel {
  eg_ctx* ctx = nullptr;
  std::shared_ptr<eg_module> module;
  hipFunction_t fn = nullptr;
  std::string name;
  // Sticky arguments, 8 bytes each (pointers, int64, double) or 4 (float).
  struct Arg {
    unsigned char bytes[8];
    int size = 0;
  };
  std::vector<Arg> args;
};

namespace eg {
void* kernel_function(eg_kernel* kernel) { return kernel ? reinterpret_cast<void*>(kernel->fn) : nullptr; }

int kernel_launch_raw(eg_kernel* kernel, unsigned gx, unsigned gy, unsigned gz, unsigned block, void** args) {
  EG_REQUIRE(kernel, EG_ERR_INVALID, "kernel_launch_raw: kernel is NULL");
  if (gx == 0 || gy == 0 || gz == 0) return EG_OK;
  EG_HIP_CHECK(hipSetDevice(kernel->ctx->device));
  EG_HIP_CHECK(hipModuleLaunchKernel(kernel->fn, gx, gy, gz, block, 1, 1, 0, kernel->ctx->stream, args, nullptr));
  return EG_OK;
}
}  // namespace eg

namespace eg {
// One hiprtc program, one code object, one handle per kernel name (a model's generated kernels are
// built together: the per-program overhead of hiprtc dominates small kernels).
int kernels_compile_batch(eg_ctx* ctx, const char* label, const char* source, const std::vector<std::string>& names,
                          std::vector<eg_kernel*>& out) {
  out.clear();
  EG_HIP_CHECK(hipSetDevice(ctx->device));
  // which hiprtc compiles it and the on-disk cache of code objects: rtc.cpp
  std::vector<char> code;
  int rc = rtc::compile(label, source, ctx->arch, code);
  if (rc) return rc;
  if (const char* dump = eg::sw::raw("EG_DUMP_CODE")) {  // debugging aid: the code object, for llvm-objdump -d
    if (FILE* fp = fopen((std::string(dump) + "/" + label + ".co").c_str(), "wb")) {
      fwrite(code.data(), 1, code.size(), fp);
      fclose(fp);
    }
    if (FILE* fp = fopen((std::string(dump) + "/" + label + ".hip").c_str(), "wb")) {  // ... and the text it was built from
      fwrite(source, 1, strlen(source), fp);
      fclose(fp);
    }
  }

  std::shared_ptr<eg_module> mod(new eg_module());
  mod->device = ctx->device;
  EG_HIP_CHECK(hipModuleLoadData(&mod->handle, code.data()));
  std::vector<std::unique_ptr<eg_kernel>> built;
  for (auto& name : names) {
    std::unique_ptr<eg_kernel> k(new eg_kernel());
    k->ctx = ctx;
    k->name = name;
    k->module = mod;
    hipError_t e = hipModuleGetFunction(&k->fn, mod->handle, name.c_str());
    if (e != hipSuccess) {
      set_error("kernel '%s' not found in compiled module: %s", name.c_str(), hipGetErrorString(e));
      return EG_ERR_COMPILE;
    }
    built.push_back(std::move(k));
  }
  for (auto& k : built) out.push_back(k.release());
  return EG_OK;
}
}  // namespace eg

extern "C" {

const char* eg_last_error(void) { return eg::g_error.c_str(); }
int eg_version(void) { return 1000; }

int eg_device_count(int* count) {
  EG_REQUIRE(count, EG_ERR_INVALID, "eg_device_count: count is NULL");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e == hipErrorNoDevice) {
    *count = 0;
    return EG_OK;
  }
  EG_HIP_CHECK(e);
  *count = n;
  return EG_OK;
}

static void copy_str(char* dst, size_t cap, const std::string& s) {
  if (!dst || cap == 0) return;
  size_t n = s.size() < cap - 1 ? s.size() : cap - 1;
  memcpy(dst, s.data(), n);
  dst[n] = 0;
}

int eg_device_info(int device, char* name, size_t name_cap, char* vendor, size_t vendor_cap,
                   char* version, size_t version_cap, int* is_gpu) {
  hipDeviceProp_t p;
  EG_HIP_CHECK(hipGetDeviceProperties(&p, device));
  copy_str(name, name_cap, p.name);
  copy_str(vendor, vendor_cap, "Advanced Micro Devices, Inc.");
  int rt = 0;
  EG_HIP_CHECK(hipRuntimeGetVersion(&rt));
  char buf[128];
  snprintf(buf, sizeof(buf), "HIP %d (%s)", rt, p.gcnArchName);
  copy_str(version, version_cap, buf);
  if (is_gpu) *is_gpu = 1;
  return EG_OK;
}

int eg_device_props(int device, int* compute_units, int* clock_khz, int64_t* hbm_bytes, char* arch,
                    size_t arch_cap) {
  hipDeviceProp_t p;
  EG_HIP_CHECK(hipGetDeviceProperties(&p, device));
  if (compute_units) *compute_units = p.multiProcessorCount;
  if (clock_khz) *clock_khz = p.clockRate;
  if (hbm_bytes) *hbm_bytes = (int64_t)p.totalGlobalMem;
  copy_str(arch, arch_cap, p.gcnArchName);
  return EG_OK;
}

static int ctx_init(int device, eg_ctx* ctx) {
  int n = 0;
  EG_HIP_CHECK(hipGetDeviceCount(&n));
  // cl.nim:95-99: "Unable to find device".
  EG_REQUIRE(n > 0, EG_ERR_HIP, "Unable to find device");
  EG_REQUIRE(device >= 0 && device < n, EG_ERR_INVALID, "device %d out of range (have %d)", device, n);
  EG_HIP_CHECK(hipSetDevice(device));
  hipDeviceProp_t p;
  EG_HIP_CHECK(hipGetDeviceProperties(&p, device));
  ctx->device = device;
  ctx->compute_units = p.multiProcessorCount;
  ctx->arch = p.gcnArchName;
  return EG_OK;
}

int eg_ctx_create(int device, eg_ctx** out) {
  EG_REQUIRE(out, EG_ERR_INVALID, "eg_ctx_create: out is NULL");
  std::unique_ptr<eg_ctx> ctx(new eg_ctx());
  int rc = ctx_init(device, ctx.get());
  if (rc) return rc;
  EG_HIP_CHECK(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
  ctx->owns_stream = true;
  *out = ctx.release();
  return EG_OK;
}

int eg_ctx_create_on_stream(int device, void* hip_stream, eg_ctx** out) {
  EG_REQUIRE(out, EG_ERR_INVALID, "eg_ctx_create_on_stream: out is NULL");
  std::unique_ptr<eg_ctx> ctx(new eg_ctx());
  int rc = ctx_init(device, ctx.get());
  if (rc) return rc;
  ctx->stream = (hipStream_t)hip_stream;
  ctx->owns_stream = false;
  *out = ctx.release();
  return EG_OK;
}

int eg_ctx_destroy(eg_ctx* ctx) {
  if (!ctx) return EG_OK;
  hipSetDevice(ctx->device);
  hipStreamSynchronize(ctx->stream);
  if (ctx->workspace) hipFree(ctx->workspace);
  if (ctx->aux) hipFree(ctx->aux);
  if (ctx->side_stream) hipStreamSynchronize(ctx->side_stream);
  if (ctx->side_workspace) hipFree(ctx->side_workspace);
  if (ctx->side_aux) hipFree(ctx->side_aux);
  if (ctx->ev_fork) hipEventDestroy(ctx->ev_fork);
  if (ctx->ev_join) hipEventDestroy(ctx->ev_join);
  for (hipEvent_t e : ctx->pipe_events) hipEventDestroy(e);
  if (ctx->side_stream) hipStreamDestroy(ctx->side_stream);
  eg::host_stager_free(ctx->stager);
  if (ctx->ones) hipFree(ctx->ones);
  for (auto& kv : ctx->jit) delete kv.second;  // eg_kernel: the code object goes with it
  if (ctx->owns_stream && ctx->stream) hipStreamDestroy(ctx->stream);
  delete ctx;
  return EG_OK;
}

int eg_ctx_sync(eg_ctx* ctx) {
  EG_REQUIRE(ctx, EG_ERR_INVALID, "eg_ctx_sync: ctx is NULL");
  EG_HIP_CHECK(hipSetDevice(ctx->device));
  EG_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  return EG_OK;
}

void* eg_ctx_stream(eg_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }
int eg_ctx_device(eg_ctx* ctx) { return ctx ? ctx->device : -1; }

int eg_buf_alloc(eg_ctx* ctx, size_t bytes, eg_buf** out) {
  EG_REQUIRE(ctx && out, EG_ERR_INVALID, "eg_buf_alloc: NULL argument");
  EG_HIP_CHECK(hipSetDevice(ctx->device));
  std::unique_ptr<eg_buf> b(new eg_buf());
  b->ctx = ctx;
  b->bytes = bytes;
  b->owned = true;
  if (bytes > 0) EG_HIP_CHECK(hipMalloc(&b->ptr, bytes));
  *out = b.release();
  return EG_OK;
}

int eg_buf_wrap(eg_ctx* ctx, void* device_ptr, size_t bytes, eg_buf** out) {
  EG_REQUIRE(ctx && out, EG_ERR_INVALID, "eg_buf_wrap: NULL argument");
  eg_buf* b = new eg_buf();
  b->ctx = ctx;
  b->ptr = device_ptr;
  b->bytes = bytes;
  b->owned = false;
  *out = b;
  return EG_OK;
}

int eg_buf_free(eg_buf* buf) {
  if (!buf) return EG_OK;
  if (buf->owned && buf->ptr) {
    hipSetDevice(buf->ctx->device);
    // The stream may still reference the block.
    hipStreamSynchronize(buf->ctx->stream);
    hipError_t e = hipFree(buf->ptr);
    if (e != hipSuccess) {
      set_error("hipFree failed: %s", hipGetErrorString(e));
      delete buf;
      return EG_ERR_HIP;
    }
  }
  delete buf;
  return EG_OK;
}

size_t eg_buf_size(const eg_buf* buf) { return buf ? buf->bytes : 0; }
void* eg_buf_ptr(const eg_buf* buf) { return buf ? buf->ptr : nullptr; }

int eg_buf_write(eg_buf* buf, const void* host, size_t bytes) {
  EG_REQUIRE(buf, EG_ERR_INVALID, "eg_buf_write: buf is NULL");
  // cl.nim:112-113
  EG_REQUIRE(bytes == buf->bytes, EG_ERR_SIZE,
             "Attempted to write %zu bytes, but the size of the buffer is %zu bytes", bytes, buf->bytes);
  if (bytes == 0) return EG_OK;
  EG_REQUIRE(host, EG_ERR_INVALID, "eg_buf_write: host is NULL");
  return eg::copy_h2d(buf->ctx, buf->ptr, host, bytes);
}

int eg_buf_read(eg_buf* buf, void* host, size_t bytes) {
  EG_REQUIRE(buf, EG_ERR_INVALID, "eg_buf_read: buf is NULL");
  // cl.nim:134-135
  EG_REQUIRE(bytes == buf->bytes, EG_ERR_SIZE, "Buffer size is not equal to target size (%zu vs %zu)",
             buf->bytes, bytes);
  if (bytes == 0) return EG_OK;
  EG_REQUIRE(host, EG_ERR_INVALID, "eg_buf_read: host is NULL");
  return eg::copy_d2h(buf->ctx, host, buf->ptr, bytes);
}

int eg_buf_fill(eg_buf* buf, const void* pattern, size_t pattern_bytes) {
  EG_REQUIRE(buf && pattern, EG_ERR_INVALID, "eg_buf_fill: NULL argument");
  if (buf->bytes == 0) return EG_OK;
  EG_REQUIRE(buf->bytes % pattern_bytes == 0, EG_ERR_SIZE,
             "buffer size %zu is not a multiple of the fill pattern (%zu bytes)", buf->bytes, pattern_bytes);
  EG_HIP_CHECK(hipSetDevice(buf->ctx->device));
  hipStream_t s = buf->ctx->stream;
  switch (pattern_bytes) {
    case 1:
      EG_HIP_CHECK(hipMemsetD8Async((hipDeviceptr_t)buf->ptr, *(const unsigned char*)pattern, buf->bytes, s));
      break;
    case 2: {
      unsigned short v;
      memcpy(&v, pattern, 2);
      EG_HIP_CHECK(hipMemsetD16Async((hipDeviceptr_t)buf->ptr, v, buf->bytes / 2, s));
      break;
    }
    case 4: {
      int v;
      memcpy(&v, pattern, 4);
      EG_HIP_CHECK(hipMemsetD32Async((hipDeviceptr_t)buf->ptr, v, buf->bytes / 4, s));
      break;
    }
    case 8: {
      // No 64-bit memset in HIP: the two halves are equal for 0.0 (the only f64 fill the
      // reference issues, model.nim:318); otherwise go through the f32-pair kernel path.
      unsigned int lo, hi;
      memcpy(&lo, pattern, 4);
      memcpy(&hi, (const char*)pattern + 4, 4);
      EG_REQUIRE(lo == hi, EG_ERR_UNSUPPORTED, "8-byte fill pattern with distinct halves is not supported");
      EG_HIP_CHECK(hipMemsetD32Async((hipDeviceptr_t)buf->ptr, (int)lo, buf->bytes / 4, s));
      break;
    }
    default:
      set_error("eg_buf_fill: unsupported pattern size %zu", pattern_bytes);
      return EG_ERR_INVALID;
  }
  return EG_OK;
}

// ------------------------------------------------------------------ hiprtc kernels

int eg_kernel_compile(eg_ctx* ctx, const char* name, const char* source, eg_kernel** out) {
  EG_REQUIRE(ctx && name && source && out, EG_ERR_INVALID, "eg_kernel_compile: NULL argument");
  std::vector<eg_kernel*> built;
  int rc = eg::kernels_compile_batch(ctx, name, source, {std::string(name)}, built);
  if (rc) return rc;
  *out = built[0];
  return EG_OK;
}

int eg_kernel_free(eg_kernel* kernel) {
  if (!kernel) return EG_OK;
  hipSetDevice(kernel->ctx->device);
  hipStreamSynchronize(kernel->ctx->stream);
  delete kernel;  // the code object goes with the last kernel that shares it
  return EG_OK;
}

static int set_arg(eg_kernel* kernel, int index, const void* data, int size) {
  EG_REQUIRE(kernel, EG_ERR_INVALID, "kernel is NULL");
  EG_REQUIRE(index >= 0 && index < 256, EG_ERR_INVALID, "argument index %d out of range", index);
  if ((size_t)index >= kernel->args.size()) kernel->args.resize(index + 1);
  memcpy(kernel->args[index].bytes, data, size);
  kernel->args[index].size = size;
  return EG_OK;
}

int eg_kernel_set_arg_buf(eg_kernel* kernel, int index, eg_buf* buf) {
  EG_REQUIRE(buf, EG_ERR_INVALID, "eg_kernel_set_arg_buf: buf is NULL");
  void* p = buf->ptr;
  return set_arg(kernel, index, &p, sizeof(p));
}
int eg_kernel_set_arg_i64(eg_kernel* kernel, int index, int64_t value) {
  return set_arg(kernel, index, &value, sizeof(value));
}
int eg_kernel_set_arg_f32(eg_kernel* kernel, int index, float value) {
  return set_arg(kernel, index, &value, sizeof(value));
}
int eg_kernel_set_arg_f64(eg_kernel* kernel, int index, double value) {
  return set_arg(kernel, index, &value, sizeof(value));
}

int eg_kernel_launch(eg_kernel* kernel, int dims, const int64_t* groups, const int64_t* local) {
  EG_REQUIRE(kernel, EG_ERR_INVALID, "eg_kernel_launch: kernel is NULL");
  // cl.nim:191-194
  EG_REQUIRE(dims >= 1, EG_ERR_INVALID, "Group size must have at least one dimension");
  EG_REQUIRE(dims <= 3, EG_ERR_INVALID, "at most 3 launch dimensions (passes.nim:1801-1805)");
  EG_REQUIRE(groups && local, EG_ERR_INVALID, "eg_kernel_launch: NULL sizes");
  unsigned g[3] = {1, 1, 1}, l[3] = {1, 1, 1};
  for (int d = 0; d < dims; ++d) {
    EG_REQUIRE(groups[d] >= 0 && local[d] > 0, EG_ERR_INVALID, "bad launch size in dimension %d", d);
    if (groups[d] == 0) return EG_OK;  // empty range: nothing to do
    g[d] = (unsigned)groups[d];
    l[d] = (unsigned)local[d];
  }
  std::vector<void*> params(kernel->args.size());
  for (size_t i = 0; i < kernel->args.size(); ++i) {
    EG_REQUIRE(kernel->args[i].size > 0, EG_ERR_INVALID, "kernel '%s': argument %zu was never set",
               kernel->name.c_str(), i);
    params[i] = kernel->args[i].bytes;
  }
  EG_HIP_CHECK(hipSetDevice(kernel->ctx->device));
  EG_HIP_CHECK(hipModuleLaunchKernel(kernel->fn, g[0], g[1], g[2], l[0], l[1], l[2], 0, kernel->ctx->stream,
                                     params.data(), nullptr));
  return EG_OK;
}

}  // extern "C"
