"""The C ABI driven from plain C (tests/cabi_harness.c, built with gcc against include/exprgrad_hip.h): what a Nim host
compiles to.  Group 1 (runtime) and, for every hand-written kernel description with a closed-form answer
(tests/golden/handwritten/*.kd — among them the text the Nim emitter must produce for examples/xor_from_scratch),
group 3: compile -> param_write -> set_input_host -> run -> read_output / param_read."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HERE = os.path.join(ROOT, "tests")


def build_harness(out_dir):
    exe = os.path.join(str(out_dir), "cabi_harness")
    lib_dir = os.path.join(ROOT, "exprgrad_amd", "lib")
    cmd = ["gcc", "-std=c99", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), os.path.join(HERE, "cabi_harness.c"),
           "-L" + lib_dir, "-lexprgrad_hip", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib", "-lm", "-o", exe]
    done = subprocess.run(cmd, capture_output=True, text=True)
    assert done.returncode == 0, done.stderr
    return exe


def case_file(case, path):
    """tests/golden/handwritten.json -> the record-per-line form the harness reads."""
    def numbers(spec):
        return " ".join(repr(float(v)) for v in spec["data"])

    def count(spec):
        n = 1
        for d in spec["shape"]:
            n *= d
        return n
    lines = ["tol %r" % case["tol"]]
    if "epoch" in case:
        lines.append("epoch %d" % case["epoch"])
    for tid, spec in case["params"].items():
        lines.append("param %s %d %s" % (tid, count(spec), numbers(spec)))
    for name, spec in case["inputs"].items():
        lines.append("input %s %d %s %s" % (name, len(spec["shape"]), " ".join(str(d) for d in spec["shape"]), numbers(spec)))
    for call in case.get("calls", []):
        names = call.get("inputs", list(case["inputs"]))
        lines.append("call %s %d %s %d %s" % (call["target"], len(names), " ".join(names), count(call["expect"]), numbers(call["expect"])))
    if "apply" in case:
        lines.append("apply %s" % case["apply"])
        for tid, spec in list(case["expect_params"].items()) + list(case["expect_caches"].items()):
            lines.append("expect %s %d %s" % (tid, count(spec), numbers(spec)))
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")


def test_the_harness_builds_against_the_public_header_alone(tmp_path):
    """No GPU needed: gcc -Wall -Werror on plain C, every eg_* it names resolved by the library."""
    build_harness(tmp_path)


@pytest.mark.gpu
def test_runtime_group_from_c(tmp_path):
    done = subprocess.run([build_harness(tmp_path), "runtime"], capture_output=True, text=True)
    assert done.returncode == 0, done.stdout + done.stderr
    assert "run-time compiler: hiprtc" in done.stdout, done.stdout


@pytest.mark.gpu
def test_the_references_own_lowered_kernels_through_the_jit_stub_calls(tmp_path):
    """Level 1 of INTEGRATION.md end to end: four kernels the reference's pipeline lowers for its GPU target
    (tests/cache/{matmul_basic, matmul_schedule_tiled16, relu_basic, conv1_basic}.ir), as the HIP text the clgen.nim
    patch of nim/PATCHES.md section 5 emits, compiled by eg_kernel_compile and driven with the calls the JIT launch stub
    makes (set tensor, set index, run with groups = ceil(global / local)); results equal the host loops the
    reference's tests compare with."""
    done = subprocess.run([build_harness(tmp_path), "jit"], capture_output=True, text=True)
    assert done.returncode == 0, done.stdout + done.stderr
    assert "4 reference kernels" in done.stdout


with open(os.path.join(HERE, "golden", "handwritten.json")) as _f:
    CASES = json.load(_f)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
def test_model_group_from_c(tmp_path, name):
    exe = build_harness(tmp_path)
    path = os.path.join(str(tmp_path), name + ".case")
    case_file(CASES[name], path)
    done = subprocess.run([exe, "model", os.path.join(HERE, "golden", "handwritten", name + ".kd"), path], capture_output=True, text=True)
    assert done.returncode == 0, done.stdout + done.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
def test_model_group_from_c_in_float64(tmp_path, name):
    """The same hand-derived programs with the header line `kd 1 f64` (compile[float64], model.nim:253-260) through the
    _f64 twins of the typed entry points, from plain C.  The closed-form answers were derived for the float32 program,
    whose constants are float32(0.1), float32(0.001) ...: the float64 program differs from them by those roundings (1e-7
    class), so the bound is the float32 case's, never looser than 1e-6; what this shows is that the double-typed walk
    (param_write_f64 -> set_input_host_f64 -> run -> read_output_f64 / param_read_f64) is wired through."""
    exe = build_harness(tmp_path)
    case = dict(CASES[name])
    case["tol"] = max(case["tol"], 1e-6)
    path = os.path.join(str(tmp_path), name + ".case")
    case_file(case, path)
    with open(os.path.join(HERE, "golden", "handwritten", name + ".kd")) as f:
        text = f.read()
    assert text.count("\nkd 1 f32\n") + text.startswith("kd 1 f32\n") == 1
    kd64 = os.path.join(str(tmp_path), name + "_f64.kd")
    with open(kd64, "w") as f:
        f.write(text.replace("kd 1 f32", "kd 1 f64", 1))
    done = subprocess.run([exe, "model", kd64, path], capture_output=True, text=True)
    assert done.returncode == 0, done.stdout + done.stderr
