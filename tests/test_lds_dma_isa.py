"""Instruction-level check of the LDS-DMA publish barrier, without a GPU.

The generated-epilogue contractions are compiled at run time by whatever hiprtc the process resolves — inside a
PyTorch-ROCm process that is the one bundled with torch (ROCm 7.0 here), not /opt/rocm's (7.2).  The two differ in
where they put the `s_waitcnt vmcnt(0)` that must precede the barrier publishing an LDS-DMA tile (`global_load_lds`, or
`buffer_load ... lds` for plain operands since round 3): the
bundled one used to put it BEHIND the barrier (stale tiles, the rare wrong update of round 1); the kernel header now
spells the wait out (`dma_publish_barrier`).  This test compiles the header with the bundled hiprtc (compilation needs
no device) and scans the disassembly: no `s_barrier` may follow a `global_load_lds` in fall-through order without an
`s_waitcnt vmcnt(0)` in between.  Checked against the header of commit 999e7c5 by hand: 1 and 3 violations there."""
import ctypes
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"

EPILOGUE = """
struct EgEpi {
  static constexpr bool ACTIVE = true;
  static constexpr int NX = 1;
  static constexpr bool STORE_C = false;
  static constexpr int OUT = 0;
  static constexpr int PRED = -1;
  static constexpr int RD_N = 0, RD_W = 0, RD_OUT = 0, RD_BIAS = -1, RD_LDW = 0, RD_LDO = 0;
  __device__ __forceinline__ static bool predicate(float) { return false; }
  __device__ __forceinline__ static void prefetch(const eg::gemm::GemmArgs&, long, float (&)[1]) {}
  __device__ __forceinline__ static void prefetch4(const eg::gemm::GemmArgs&, long, eg::gemm::f32x4 (&)[1]) {}
  __device__ __forceinline__ static float compute(const eg::gemm::GemmArgs&, long, float v, const float (&)[1]) {
    return v > 0.0f ? v : 0.0f;
  }
};
extern "C" __global__ __launch_bounds__(%(nt)d, %(waves)d) void k(eg::gemm::GemmArgs a) {
  eg::gemm::gemm_block<%(bm)d, %(bn)d, 16, %(wm)d, %(wn)d, %(akc)s, false, 4, %(edge)s, false, 0, true, EgEpi>(a);
}
"""


def bundled_hiprtc():
    try:
        import torch
    except Exception:  # noqa: BLE001
        return None
    lib = os.path.join(os.path.dirname(torch.__file__), "lib")
    rtc, comgr = os.path.join(lib, "libhiprtc.so"), os.path.join(lib, "libamd_comgr.so")
    if not (os.path.exists(rtc) and os.path.exists(comgr)):
        return None
    ctypes.CDLL(comgr, mode=ctypes.RTLD_GLOBAL)
    return ctypes.CDLL(rtc)


def compile_to_isa(rtc, source, tmp_path):
    prog = ctypes.c_void_p()
    assert rtc.hiprtcCreateProgram(ctypes.byref(prog), source.encode(), b"k", 0, None, None) == 0
    opts = [b"--offload-arch=gfx950", b"-O3", b"-ffp-contract=off", b"-std=c++17"]      # runtime.cpp: kernels_compile_batch
    status = rtc.hiprtcCompileProgram(prog, len(opts), (ctypes.c_char_p * len(opts))(*opts))
    if status != 0:
        n = ctypes.c_size_t()
        rtc.hiprtcGetProgramLogSize(prog, ctypes.byref(n))
        log = ctypes.create_string_buffer(n.value)
        rtc.hiprtcGetProgramLog(prog, log)
        raise AssertionError(log.value.decode()[-3000:])
    n = ctypes.c_size_t()
    rtc.hiprtcGetCodeSize(prog, ctypes.byref(n))
    code = ctypes.create_string_buffer(n.value)
    rtc.hiprtcGetCode(prog, code)
    path = os.path.join(tmp_path, "k.co")
    with open(path, "wb") as f:
        f.write(code.raw)
    return subprocess.run([OBJDUMP, "-d", path], capture_output=True, text=True, check=True).stdout


def unpublished_barriers(isa):
    pending, loads, bad = False, 0, []
    for line in isa.splitlines():
        ins = line.split()[0] if line.split() else ""
        if "global_load_lds" in line or ("buffer_load_" in line and " lds" in line):   # either LDS-DMA form
            pending, loads = True, loads + 1
        elif ins == "s_waitcnt" and "vmcnt(0)" in line:
            pending = False
        elif ins in ("s_branch", "s_endpgm", "s_setpc_b64"):
            pending = False     # what follows in the layout is not reached by falling through
        elif ins == "s_barrier" and pending:
            bad.append(line.strip())
            pending = False
    return loads, bad


@pytest.mark.parametrize("variant", [
    dict(bm=64, bn=64, wm=32, wn=32, nt=256, waves=4, akc="true", edge="false"),
    dict(bm=64, bn=64, wm=32, wn=32, nt=256, waves=3, akc="true", edge="true"),
    dict(bm=256, bn=256, wm=128, wn=64, nt=512, waves=2, akc="false", edge="true"),
], ids=["64x64-whole", "64x64-ragged", "256x256-ragged-tn"])
def test_every_barrier_after_an_lds_dma_load_is_preceded_by_its_wait(variant, tmp_path):
    rtc = bundled_hiprtc()
    if rtc is None or not os.path.exists(OBJDUMP):
        pytest.skip("no bundled hiprtc / llvm-objdump")
    with open(os.path.join(ROOT, "exprgrad_amd", "csrc", "kernels", "gemm_f32_mfma.hpp")) as f:
        header = f.read()
    isa = compile_to_isa(rtc, header + EPILOGUE % variant, str(tmp_path))
    loads, bad = unpublished_barriers(isa)
    assert loads >= 4, loads            # the LDS-DMA loops are in there
    assert not bad, bad


ROW_PRODUCT = """
struct EgEpi {
  static constexpr bool ACTIVE = true;
  static constexpr int NX = 1;
  static constexpr bool STORE_C = false;
  static constexpr int OUT = 0;
  static constexpr int PRED = 1;
  static constexpr int RD_N = 10, RD_W = 2, RD_OUT = 3, RD_BIAS = 4, RD_LDW = 10, RD_LDO = 10;
  __device__ __forceinline__ static bool predicate(float v) { return 0.0f <= v; }
  __device__ __forceinline__ static void prefetch(const eg::gemm::GemmArgs&, long, float (&)[1]) {}
  __device__ __forceinline__ static void prefetch4(const eg::gemm::GemmArgs&, long, eg::gemm::f32x4 (&)[1]) {}
  __device__ __forceinline__ static float compute(const eg::gemm::GemmArgs&, long, float v, const float (&)[1]) {
    return v > 0.0f ? v : 0.0f;
  }
};
extern "C" __global__ __launch_bounds__(512, 2) void k(eg::gemm::GemmArgs a) {
  eg::gemm::gemm_block<256, 256, 16, 128, 64, true, false, 4, false, false, 0, true, EgEpi>(a);
}
"""


def test_the_row_product_variant_builds_without_a_device(tmp_path):
    """The fused forward of a classifier head (predicate bits + the narrow next layer in the epilogue, gemm_f32_mfma.hpp
    RD_N) only exists as a hiprtc build at run time: compile it here, with the compiler the process resolves, so that a
    change that breaks it fails the CPU suite — and hold it to what the GPU relies on: no scratch, the 16 x 16 x 4 MFMAs
    and float atomics without return are there, LDS stays under the CU's 160 KiB, and the LDS-DMA publish rule holds."""
    rtc = bundled_hiprtc()
    if rtc is None or not os.path.exists(OBJDUMP):
        pytest.skip("no bundled hiprtc / llvm-objdump")
    with open(os.path.join(ROOT, "exprgrad_amd", "csrc", "kernels", "gemm_f32_mfma.hpp")) as f:
        header = f.read()
    isa = compile_to_isa(rtc, header + ROW_PRODUCT, str(tmp_path))
    loads, bad = unpublished_barriers(isa)
    assert loads >= 4 and not bad, (loads, bad)
    assert "v_mfma_f32_16x16x4" in isa and "v_mfma_f32_32x32x2" in isa
    assert "global_atomic_add_f32" in isa
    assert "scratch_" not in isa                      # no spill
    notes = subprocess.run([OBJDUMP.replace("llvm-objdump", "llvm-readelf"), "--notes", os.path.join(str(tmp_path), "k.co")],
                           capture_output=True, text=True).stdout
    lds = [int(line.split(":")[1]) for line in notes.splitlines() if ".group_segment_fixed_size" in line]
    assert lds and max(lds) <= 160 * 1024, lds


NARROW_K = """
struct EgEpi {
  static constexpr bool ACTIVE = true;
  static constexpr int NX = 1;
  static constexpr bool STORE_C = false;
  static constexpr int OUT = 0;
  static constexpr int PRED = -1;
  static constexpr int RD_N = 0, RD_W = 0, RD_OUT = 0, RD_BIAS = -1, RD_LDW = 0, RD_LDO = 0;
  __device__ __forceinline__ static bool predicate(float) { return false; }
  __device__ __forceinline__ static void prefetch(const eg::gemm::GemmArgs& a, long idx, float (&x)[1]) {
    x[0] = (float)((((const unsigned*)a.epi[1])[idx >> 5] >> (idx & 31)) & 1u);
  }
  __device__ __forceinline__ static void prefetch4(const eg::gemm::GemmArgs& a, long idx, eg::gemm::f32x4 (&x)[1]) {
    const unsigned w = ((const unsigned*)a.epi[1])[idx >> 5] >> (idx & 31);
    for (int e = 0; e < 4; ++e) x[0][e] = (float)((w >> e) & 1u);
  }
  __device__ __forceinline__ static float compute(const eg::gemm::GemmArgs&, long, float v, const float (&x)[1]) {
    return x[0] != 0.0f ? v : 0.0f;
  }
};
extern "C" __global__ __launch_bounds__(256) void k(eg::gemm::GemmArgs a) {
  eg::gemm::gemm_narrow_k_block<10, 128, true, true, EgEpi>(a);
}
"""


def test_the_narrow_k_variant_builds_without_a_device(tmp_path):
    """The activation-gradient product of a 10-class layer with relu's gradient in the epilogue runs as a streaming kernel
    on the vector ALUs (gemm_f32_mfma.hpp, gemm_narrow_k_block), built by hiprtc at run time only: compile it here — no
    scratch, no matrix instruction, 16-byte stores, and the K values of a row as scalar loads on the one-wave-per-row path."""
    rtc = bundled_hiprtc()
    if rtc is None or not os.path.exists(OBJDUMP):
        pytest.skip("no bundled hiprtc / llvm-objdump")
    with open(os.path.join(ROOT, "exprgrad_amd", "csrc", "kernels", "gemm_f32_mfma.hpp")) as f:
        header = f.read()
    isa = compile_to_isa(rtc, header + NARROW_K, str(tmp_path))
    assert "scratch_" not in isa and "v_mfma" not in isa
    assert "global_store_dwordx4" in isa and "s_load_dword" in isa
    assert "v_fma_f32" in isa or "v_fmac_f32" in isa or "v_pk_fma_f32" in isa
