// Run-time compilation of generated kernels: which hiprtc, and the on-disk code-object cache.
//
// The reference hands kernel text to the OpenCL driver (`clCreateProgramWithSource` + `clBuildProgram`,
// runtimes/cl.nim:149-179) every time a model is built (model.nim:209-213).  Here the text goes to hiprtc —
// but not to "whichever libhiprtc the process happens to resolve": inside a PyTorch-ROCm process that is the
// copy bundled with torch (ROCm 7.0 in this image), in a plain C / Nim host the system one (7.2), and the two
// compilers schedule the LDS-DMA loops of kernels/gemm_f32_mfma.hpp differently (the 7.0 one waits
// for a tile's DMA right after issuing it).  So the library opens hiprtc itself, by path, preferring the ROCm
// release it was built with (its own kernels and the generated ones then come from one compiler):
//     $EG_HIPRTC_LIB  >  <ROCm lib dir of the build>/libhiprtc.so.<major>  >  libhiprtc.so.<major>  >  libhiprtc.so
// dlopen by absolute path maps a second copy next to one that is already loaded under the same soname (glibc
// matches loaded objects by the name asked for, then by file identity), RTLD_LOCAL | RTLD_DEEPBIND keeps its
// symbols to itself; ROCm >= 6 hiprtc finds its comgr through its own RUNPATH ($ORIGIN).
// What was picked is reported by eg_compiler_info() and is part of every cache key.
//
// Cache: code objects under $EG_KERNEL_CACHE (default $XDG_CACHE_HOME/exprgrad_hip or ~/.cache/exprgrad_hip),
// one file per (source text, options, compiler identity); EG_NO_KERNEL_CACHE=1 switches it off.  Written to a
// temporary name and renamed, so concurrent processes (one per GPU) never see a partial file.
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <sys/types.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "eg_internal.hpp"

#ifndef EG_ROCM_LIBDIR
#define EG_ROCM_LIBDIR "/opt/rocm/lib"
#endif

namespace eg {
namespace rtc {

namespace {

typedef struct _hiprtcProgram* Program;
struct Api {
  void* handle = nullptr;
  int (*create)(Program*, const char*, const char*, int, const char**, const char**) = nullptr;
  int (*compile)(Program, int, const char**) = nullptr;
  int (*log_size)(Program, size_t*) = nullptr;
  int (*log)(Program, char*) = nullptr;
  int (*code_size)(Program, size_t*) = nullptr;
  int (*code)(Program, char*) = nullptr;
  int (*destroy)(Program*) = nullptr;
  const char* (*error_string)(int) = nullptr;
  int (*version)(int*, int*) = nullptr;
  std::string path;      // the file that was mapped
  std::string identity;  // path + size + mtime + version: part of every cache key
  int major = 0, minor = 0;
  std::string error;
};

Api g_api;
std::once_flag g_once;

bool try_open(const std::string& name) {
  void* h = dlopen(name.c_str(), RTLD_NOW | RTLD_LOCAL | RTLD_DEEPBIND);
  if (!h) return false;
  Api a;
  a.handle = h;
#define EG_SYM(field, sym)                                  \
  *reinterpret_cast<void**>(&a.field) = dlsym(h, sym);      \
  if (!a.field) {                                           \
    dlclose(h);                                             \
    return false;                                           \
  }
  EG_SYM(create, "hiprtcCreateProgram")
  EG_SYM(compile, "hiprtcCompileProgram")
  EG_SYM(log_size, "hiprtcGetProgramLogSize")
  EG_SYM(log, "hiprtcGetProgramLog")
  EG_SYM(code_size, "hiprtcGetCodeSize")
  EG_SYM(code, "hiprtcGetCode")
  EG_SYM(destroy, "hiprtcDestroyProgram")
  EG_SYM(error_string, "hiprtcGetErrorString")
  EG_SYM(version, "hiprtcVersion")
#undef EG_SYM
  Dl_info info;
  a.path = name;
  if (dladdr(reinterpret_cast<void*>(a.create), &info) && info.dli_fname) {
    char real[4096];
    a.path = realpath(info.dli_fname, real) ? real : info.dli_fname;
  }
  a.version(&a.major, &a.minor);
  struct stat st;
  char id[256] = "";
  if (stat(a.path.c_str(), &st) == 0) snprintf(id, sizeof(id), " size=%lld mtime=%lld", (long long)st.st_size, (long long)st.st_mtime);
  a.identity = a.path + id + " hiprtc=" + std::to_string(a.major) + "." + std::to_string(a.minor);
  g_api = a;
  return true;
}

void load() {
  std::vector<std::string> names;
  if (const char* e = eg::sw::raw("EG_HIPRTC_LIB"))
    if (e[0]) names.push_back(e);
  const std::string major = std::to_string(HIP_VERSION_MAJOR);
  names.push_back(std::string(EG_ROCM_LIBDIR) + "/libhiprtc.so." + major);
  names.push_back("libhiprtc.so." + major);
  names.push_back("libhiprtc.so");
  std::string tried;
  for (auto& n : names) {
    if (try_open(n)) return;
    const char* why = dlerror();
    tried += "\n  " + n + ": " + (why ? why : "missing hiprtc symbols");
  }
  g_api.error = "no usable libhiprtc found; tried:" + tried;
}

const Api* api() {
  std::call_once(g_once, load);
  return g_api.handle ? &g_api : nullptr;
}

// ---- cache --------------------------------------------------------------------------------------

std::atomic<long> g_hits{0}, g_misses{0};
std::atomic<long> g_compile_us{0};

uint64_t fnv1a(const void* data, size_t n, uint64_t h) {
  const unsigned char* p = static_cast<const unsigned char*>(data);
  for (size_t i = 0; i < n; ++i) {
    h ^= p[i];
    h *= 0x100000001b3ull;
  }
  return h;
}

bool cache_enabled() {
  static const bool on = [] {
    const char* e = eg::sw::raw("EG_NO_KERNEL_CACHE");
    return !(e && e[0] && e[0] != '0');
  }();
  return on;
}

// Directory of the cache, created on first use; empty when there is nowhere to put one.
const std::string& cache_dir() {
  static const std::string dir = [] {
    std::string d;
    if (const char* e = eg::sw::raw("EG_KERNEL_CACHE")) d = e;
    else if (const char* x = getenv("XDG_CACHE_HOME")) d = std::string(x) + "/exprgrad_hip";
    else if (const char* h = getenv("HOME")) d = std::string(h) + "/.cache/exprgrad_hip";
    if (d.empty()) return d;
    for (size_t i = 1; i <= d.size(); ++i)  // mkdir -p
      if (i == d.size() || d[i] == '/') mkdir(d.substr(0, i).c_str(), 0755);
    struct stat st;
    if (stat(d.c_str(), &st) != 0 || !S_ISDIR(st.st_mode) || access(d.c_str(), W_OK) != 0) d.clear();
    return d;
  }();
  return dir;
}

std::string cache_key(const Api& a, const char* source, const std::vector<std::string>& opts) {
  // two independent 64-bit FNV-1a passes over (compiler identity, options, source)
  uint64_t h1 = 0xcbf29ce484222325ull, h2 = 0x84222325cbf29ce4ull;
  auto feed = [&](const void* p, size_t n) {
    h1 = fnv1a(p, n, h1);
    h2 = fnv1a(p, n, h2 ^ 0x9e3779b97f4a7c15ull);
  };
  feed(a.identity.data(), a.identity.size() + 1);
  for (auto& o : opts) feed(o.data(), o.size() + 1);
  const size_t n = strlen(source);
  feed(&n, sizeof(n));
  feed(source, n);
  char name[64];
  snprintf(name, sizeof(name), "%016llx%016llx.co", (unsigned long long)h1, (unsigned long long)h2);
  return name;
}

// A cache entry is the code object followed by a 16-byte trailer {magic, length, FNV-1a of the bytes}: a short or
// damaged file (a writer that died, a disk that filled up) is recognised, removed and recompiled instead of being
// handed to hipModuleLoadData.
constexpr uint32_t kTrailerMagic = 0x45474b43u;  // "EGKC"

uint64_t content_hash(const char* p, size_t n) {
  uint64_t h = 1469598103934665603ull;
  for (size_t i = 0; i < n; ++i) h = (h ^ (unsigned char)p[i]) * 1099511628211ull;
  return h;
}

bool read_file(const std::string& path, std::vector<char>& out) {
  FILE* fp = fopen(path.c_str(), "rb");
  if (!fp) return false;
  fseek(fp, 0, SEEK_END);
  const long n = ftell(fp);
  fseek(fp, 0, SEEK_SET);
  bool ok = n > 20;
  if (ok) {
    out.resize((size_t)n);
    ok = fread(out.data(), 1, (size_t)n, fp) == (size_t)n;
  }
  fclose(fp);
  if (ok) {
    uint32_t magic, length;
    uint64_t sum;
    memcpy(&magic, out.data() + n - 16, 4);
    memcpy(&length, out.data() + n - 12, 4);
    memcpy(&sum, out.data() + n - 8, 8);
    ok = magic == kTrailerMagic && (long)length == n - 16 && memcmp(out.data(), "\x7f" "ELF", 4) == 0 &&
         sum == content_hash(out.data(), (size_t)length);
    if (ok) out.resize((size_t)length);
  }
  // An entry that does not validate is simply not served: the caller compiles and write_file_atomic's rename replaces it.
  // (Unlinking it here could remove a VALID entry another process renamed into place between this read and the unlink —
  // several ranks starting together would then recompile in turns)
  return ok;
}

void write_file_atomic(const std::string& path, const std::vector<char>& data) {
  // the temporary name is unique per writer (process, and a counter for the threads of one process compiling the same
  // source for different contexts), opened exclusively
  static std::atomic<unsigned> serial{0};
  char tmp[4200];
  snprintf(tmp, sizeof(tmp), "%s.%d.%u.tmp", path.c_str(), (int)getpid(), serial.fetch_add(1));
  const int fd = open(tmp, O_WRONLY | O_CREAT | O_EXCL, 0644);
  if (fd < 0) return;
  FILE* fp = fdopen(fd, "wb");
  if (!fp) {
    close(fd);
    unlink(tmp);
    return;
  }
  const uint32_t magic = kTrailerMagic, length = (uint32_t)data.size();
  const uint64_t sum = content_hash(data.data(), data.size());
  bool ok = fwrite(data.data(), 1, data.size(), fp) == data.size();
  ok = ok && fwrite(&magic, 4, 1, fp) == 1 && fwrite(&length, 4, 1, fp) == 1 && fwrite(&sum, 8, 1, fp) == 1;
  ok = (fclose(fp) == 0) && ok;
  if (!ok || rename(tmp, path.c_str()) != 0) unlink(tmp);
}

}  // namespace

// Source text -> code object for `arch`.  EG_OK, or EG_ERR_COMPILE with the build log in the error text
// (cl.nim:163-171 puts the build log into the exception).
int compile(const char* label, const char* source, const std::string& arch, std::vector<char>& code) {
  const Api* a = api();
  if (!a) {
    set_error("%s", g_api.error.c_str());
    return EG_ERR_COMPILE;
  }
  // Arch comes from the device (e.g. "gfx950:sramecc+:xnack-").  contract=off keeps the generated scalar code
  // inside the reference's no-fast-math arithmetic (wrappers/llvm.nim:486-491).
  const std::vector<std::string> opts = {"--offload-arch=" + arch, "-O3", "-ffp-contract=off", "-std=c++17"};
  std::string cached;
  if (cache_enabled() && !cache_dir().empty()) {
    cached = cache_dir() + "/" + cache_key(*a, source, opts);
    if (read_file(cached, code)) {
      ++g_hits;
      return EG_OK;
    }
  }
  const auto t0 = std::chrono::steady_clock::now();
  Program prog;
  int r = a->create(&prog, source, label, 0, nullptr, nullptr);
  if (r != 0) {
    set_error("hiprtcCreateProgram failed: %s", a->error_string(r));
    return EG_ERR_COMPILE;
  }
  std::vector<const char*> argv;
  for (auto& o : opts) argv.push_back(o.c_str());
  r = a->compile(prog, (int)argv.size(), argv.data());
  if (r != 0) {
    size_t n = 0;
    a->log_size(prog, &n);
    std::string log(n, '\0');
    if (n) a->log(prog, &log[0]);
    a->destroy(&prog);
    if (n > 1)
      set_error("Failed to build program: %s", log.c_str());
    else
      set_error("Failed to build program");
    return EG_ERR_COMPILE;
  }
  size_t n = 0;
  a->code_size(prog, &n);
  code.resize(n);
  a->code(prog, code.data());
  a->destroy(&prog);
  g_compile_us += std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
  ++g_misses;
  if (!cached.empty()) write_file_atomic(cached, code);
  return EG_OK;
}

std::string compiler_info() {
  const Api* a = api();
  if (!a) return "unavailable: " + g_api.error;
  std::string s = "hiprtc " + std::to_string(a->major) + "." + std::to_string(a->minor) + " " + a->path;
  s += cache_enabled() && !cache_dir().empty() ? "; cache " + cache_dir() : "; cache off";
  return s;
}

}  // namespace rtc
}  // namespace eg

extern "C" {

int eg_compiler_info(char* text, size_t cap) {
  EG_REQUIRE(text && cap > 0, EG_ERR_INVALID, "eg_compiler_info: no buffer");
  const std::string s = eg::rtc::compiler_info();
  snprintf(text, cap, "%s", s.c_str());
  return EG_OK;
}

int eg_kernel_cache_stats(int64_t* hits, int64_t* misses, double* compile_seconds) {
  if (hits) *hits = eg::rtc::g_hits.load();
  if (misses) *misses = eg::rtc::g_misses.load();
  if (compile_seconds) *compile_seconds = eg::rtc::g_compile_us.load() * 1e-6;
  return EG_OK;
}

}  // extern "C"
