"""CPU-side checks of the drop-in boundary: the shared library loads and exports exactly the
symbols include/exprgrad_hip.h declares (no compute: there is no GPU in the build container)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "exprgrad_hip.h")


def header_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(eg_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_four_groups():
    syms = header_symbols()
    for must in ("eg_ctx_create", "eg_buf_write", "eg_kernel_compile", "eg_kernel_launch", "eg_sgemm",
                 "eg_conv2_nhwc", "eg_model_compile", "eg_model_run", "eg_model_fit", "eg_dp_init",
                 "eg_dp_allreduce_sum_f32", "eg_model_step_dp"):
        assert must in syms


def test_library_exports_every_declared_symbol():
    from exprgrad_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH], text=True)
    exported = {line.split()[-1] for line in out.splitlines() if " T " in line}
    missing = [s for s in header_symbols() if s not in exported]
    assert not missing, f"declared in the header but not exported: {missing}"


def test_binding_matches_header():
    from exprgrad_amd import _lib
    assert _lib.declared_symbols() == header_symbols()
    handle = _lib.lib()  # sets argtypes for every symbol; raises if one is absent
    assert handle.eg_version() >= 1000


def test_no_device_is_an_error_not_a_fallback():
    """Without a GPU every entry point must fail loudly (GpuError), never compute on the CPU."""
    import ctypes
    import exprgrad_amd as eg
    from exprgrad_amd import _lib
    n = ctypes.c_int(-1)
    _lib.call("eg_device_count", ctypes.byref(n))
    if n.value > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(eg.GpuError):
        eg.newGpuContext()
    h = ctypes.c_void_p()
    rc = _lib.lib().eg_ctx_create(0, ctypes.byref(h))
    assert rc != 0 and _lib.last_error()


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure; nothing under exprgrad_amd/ may reference it."""
    pkg = os.path.join(ROOT, "exprgrad_amd")
    offenders = []
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                if re.search(r"\boracle\b|refcpu|librefcpu", text):
                    offenders.append(os.path.join(dirpath, f))
    assert not offenders, offenders


def test_null_handles_are_errors_with_a_message():
    """Argument validation needs no device: status != 0 and eg_last_error() says why (never a crash)."""
    import ctypes
    from exprgrad_amd import _lib
    lib = _lib.lib()
    null = ctypes.c_void_p()
    out = ctypes.c_void_p()
    cases = [
        lambda: lib.eg_model_run(null, b"t"),
        lambda: lib.eg_model_fit(null, b"t", 0, None, None, None, None, None, 4),
        lambda: lib.eg_dp_init(null, null, 0, 1, ctypes.byref(out)),
        lambda: lib.eg_dp_allreduce_sum_f32(null, null, 4),
        lambda: lib.eg_model_step_dp(null, b"t", null, 1),
        lambda: lib.eg_dp_unique_id(null),
        lambda: lib.eg_sgemm(null, 0, 0, 1, 1, 1, null, 1, null, 1, null, 1, 0, null),
    ]
    for i, case in enumerate(cases):
        assert case() != 0, i
        assert _lib.last_error(), i
    assert lib.eg_dp_free(null) == 0 and lib.eg_model_free(null) == 0      # freeing nothing is fine
    assert lib.eg_dp_world(null) == 0 and lib.eg_dp_rank(null) == -1


def test_switches_are_one_closed_table():
    """Every environment variable the library reads is a row of csrc/switches.cpp: no getenv() anywhere else (rtc.cpp's
    HOME / XDG_CACHE_HOME for the cache directory excepted), every name passed to eg::sw::raw / on / present / integer /
    real is in the table, every row of the table is read somewhere, and DESIGN.md documents every row."""
    import glob
    import re
    from exprgrad_amd import _lib
    table = _lib.switch_table()
    names = [t[0] for t in table]
    assert len(names) == len(set(names)) and all(len(t) == 3 and t[1] in ("execution", "data-parallel", "compiler", "detector", "tuning") for t in table)
    used = set()
    src = os.path.join(ROOT, "exprgrad_amd", "csrc")
    for path in glob.glob(os.path.join(src, "**", "*"), recursive=True):
        if not path.endswith((".cpp", ".hip", ".hpp")) or os.sep + "build" + os.sep in path:
            continue
        text = open(path).read()
        base = os.path.basename(path)
        raw_env = re.findall(r'(?<![A-Za-z_:])getenv\("([A-Z_0-9]+)"\)', text)
        if base == "rtc.cpp":
            assert sorted(raw_env) == ["HOME", "XDG_CACHE_HOME"], raw_env
        elif base not in ("switches.cpp", "switches.hpp"):
            assert not raw_env and "getenv(" not in text.replace("eg::sw::", ""), (base, raw_env)
        used.update(re.findall(r'sw::(?:raw|on|present|integer|real)\("([A-Z_0-9]+)"', text))
        used.update(re.findall(r'env_on\("([A-Z_0-9]+)"', text))
        used.update(re.findall(r'\{"(EG_[A-Z_0-9]+)", "EG_', text))          # a loop over several names
        used.update(re.findall(r', "(EG_[A-Z_0-9]+)"\}\)', text))
    assert used <= set(names), sorted(used - set(names))
    assert set(names) - used <= {"EG_TUNING"}, sorted(set(names) - used)
    design = open(os.path.join(ROOT, "DESIGN.md")).read()
    assert [n for n in names if n not in design] == []
