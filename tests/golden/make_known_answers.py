#!/usr/bin/env python
"""Writes tests/golden/known_answers.json: inputs and expected outputs of the reference's own
known-answer tests for the compiled-tensor hot path (DATA only; transcribed by hand from the cited
lines of the exprgrad repository, closed forms evaluated exactly as the reference test evaluates
them: float32 arithmetic, glibc libm for sin/cos/exp/ln/pow — Nim's std/math calls the C library).

The reference itself (Nim + LLVM 13) cannot run in the build image, so these vectors — not
reference-generated tensors — are what pins the oracle.  Re-run: python tests/golden/make_known_answers.py
"""
import ctypes
import ctypes.util
import json
import os

import numpy as np

libm = ctypes.CDLL(ctypes.util.find_library("m"))
for fn in ("sinf", "cosf", "expf", "logf"):
    getattr(libm, fn).restype = ctypes.c_float
    getattr(libm, fn).argtypes = [ctypes.c_float]
libm.powf.restype = ctypes.c_float
libm.powf.argtypes = [ctypes.c_float, ctypes.c_float]

f32 = np.float32


def m1(fn, x):
    return np.array([getattr(libm, fn)(float(v)) for v in x], dtype=f32)


def powf(a, b):
    a, b = np.broadcast_arrays(np.asarray(a, dtype=f32), np.asarray(b, dtype=f32))
    return np.array([libm.powf(float(p), float(q)) for p, q in zip(a.ravel(), b.ravel())], dtype=f32).reshape(a.shape)


def T(data, shape=None):
    a = np.asarray(data, dtype=f32)
    if shape is not None:
        a = a.reshape(shape)
    return {"shape": list(a.shape), "data": [float(v) for v in a.ravel()]}


def linspace(lo, hi, n):
    # Tensor.linspace (tensors.nim): lo + (hi-lo) * i/(n-1) in float32
    return np.array([f32(lo) + f32(hi - lo) * f32(i) / f32(n - 1) for i in range(n)], dtype=f32)


cases = {}


def case(name, cite, calls):
    cases[name] = {"cite": cite, "calls": calls}


def call(target, inputs, expected, mode="exact", eps=None):
    c = {"target": target, "inputs": inputs, "expected": expected, "mode": mode}
    if eps is not None:
        c["eps"] = eps
    return c


x23 = [1, 2, 3, 4, 5, 6]
case("identity", "tests/test_model.nim:21-27", [call("y", {"x": T(x23, [2, 3])}, T(x23, [2, 3]))])
case("double", "tests/test_model.nim:29-35", [call("y", {"x": T(x23, [2, 3])}, T(np.array(x23) * 2, [2, 3]))])
case("matmul", "tests/test_model.nim:37-44; tests/test_tensors.nim:20-23,69",
     [call("c", {"a": T(x23, [2, 3]), "b": T(x23, [3, 2])}, T([22, 28, 49, 64], [2, 2]))])
case("matmulExample", "examples/matmul/matmul.nim:22-30",
     [call("c", {"a": T(x23, [3, 2]), "b": T(x23, [2, 3])}, T([9, 12, 15, 19, 26, 33, 29, 40, 51], [3, 3]))])
case("matmulTalks", "tests/test_talks.nim:21-37",
     [call("c", {"a": T([1, 2, 3, 4], [2, 2]), "b": T(x23, [2, 3])}, T([9, 12, 15, 19, 26, 33], [2, 3]))])
case("relu", "tests/test_model.nim:46-54",
     [call("outp", {"inp": T([0, -1, 10, -20, 0.1, -0.1], [2, 3])}, T([0, 0, 10, 0, 0.1, 0], [2, 3]))])
case("meanSquaredError", "tests/test_model.nim:56-69",
     [call("loss", {"pred": T([1, 2, 3, 4], [2, 2]), "labels": T([1, 2, 3, 4], [2, 2])}, T([0], [1])),
      call("loss", {"pred": T([1, 2, 3, 4], [2, 2]), "labels": T([4, 3, 2, 1], [2, 2])}, T([20], [1]))])
case("transpose", "tests/test_model.nim:71-78", [call("b", {"a": T(x23, [2, 3])}, T([1, 4, 2, 5, 3, 6], [3, 2]))])
case("max", "tests/test_model.nim:80-89",
     [call("z", {"x": T([1, 0, 3, 4, -10, 6], [3, 2]), "y": T([1, 2, -3, 2, 5, 5.5], [3, 2])}, T(x23, [3, 2]))])
case("conv1", "tests/test_model.nim:91-97",
     [call("res", {"image": T([1, 2, 3, 2, 1, 0, -1]), "filter": T([1, 2, 3])}, T([14, 14, 10, 4, -2]))])
case("singleWrite", "tests/test_model.nim:130-134", [call("y", {}, T([10], [1]))])
case("shape", "tests/test_model.nim:136-141", [call("y", {}, T([1] * 6, [3, 2, 1]))])
case("dimensions", "tests/test_model.nim:143-154",
     [call("y", {"x": T(np.zeros(24), [1, 2, 3, 4])}, T([1, 3, 4, 4, 24])),
      call("y", {"x": T(np.zeros(6), [2, 3])}, T([2, 2, 3, 2, 6]))])
case("loopBounds", "tests/test_model.nim:256-262", [call("res", {}, T([-1, 0, 1, 1, 0]))])

x17 = linspace(-8, 8, 17)
case("derive/polynomial", "tests/test_model.nim:264-272",
     [call("x^2+2x+1", {"x": T(x17)}, T(x17 * f32(2.0) + f32(2)))])
x16 = linspace(-8, 8, 16)
case("derive/multiply", "tests/test_model.nim:274-292",
     [call("x^3", {"x": T(x16)}, T(f32(3) * (x16 * x16))),
      call("x/2", {"x": T(x16)}, T(np.full(16, 0.5, dtype=f32))),
      call("1/x", {"x": T(x16)}, T(f32(-1) / (x16 * x16))),
      call("x/x", {"x": T(x16)}, T(np.zeros(16, dtype=f32)), mode="sumsq", eps=0.00001)])
case("derive/trigonometry", "tests/test_model.nim:294-306",
     [call("sin", {"x": T(x17)}, T(m1("cosf", x17))), call("cos", {"x": T(x17)}, T(-m1("sinf", x17)))])
x5 = linspace(1, 8, 5)
case("derive/exp", "tests/test_model.nim:308-334",
     [call("exp(x)", {"x": T(x17)}, T(m1("expf", x17))),
      call("exp(2x)", {"x": T(x17)}, T(m1("expf", f32(2) * x17) * f32(2.0))),
      call("x^3", {"x": T(x17)}, T((x17 * x17) * f32(3.0))),
      call("2^x", {"x": T(x17)}, T(powf(f32(2), x17) * m1("logf", np.array([2], dtype=f32))[0])),
      call("x^x", {"x": T(x5)}, T(powf(x5, x5) * (m1("logf", x5) + f32(1.0))), mode="sumsq", eps=0.01)])
x8 = linspace(1, 8, 8)
ln = lambda v: m1("logf", np.asarray(v, dtype=f32).reshape(-1))  # noqa: E731
case("derive/log", "tests/test_model.nim:336-359",
     [call("ln(x)", {"x": T(x8)}, T(f32(1) / x8)),
      call("log10(x)", {"x": T(x8)}, T(f32(1) / (x8 * ln([10])[0]))),
      call("log2(x)", {"x": T(x8)}, T(f32(1) / (x8 * ln([2])[0]))),
      call("log(x,5)", {"x": T(x8)}, T(f32(1) / (x8 * ln([5])[0]))),
      call("log(2,x)", {"x": T(x8)}, T(-ln([2])[0] / (x8 * ln(x8) * ln(x8))))])

case("increment", "tests/test_talks.nim:49-58",
     [call("increment", {"input": T(x23, [1, 2, 3])}, T(np.array(x23) + 1, [1, 2, 3]))])
case("sumPositive", "tests/test_talks.nim:60-70",
     [call("sumPositive", {"input": T([1, -2, -3, 4, 5, -6], [2, 3])}, T([10], [1]))])
case("multiple", "tests/test_talks.nim:83-96",
     [call("predict", {"input": T([0, 0, 1, 0, 0, 1, 1, 1, 1, 2], [5, 2]), "weights": T([2, 3], [2, 1]),
                       "biases": T([1], [1])}, T([1, 3, 4, 6, 9], [5, 1]))])
case("multiplyAndSquare", "tests/test_talks.nim:98-122",
     [call("multiply", {"a": T([1, 2, 3, 4], [2, 2]), "b": T([1, 2], [2, 1])}, T([5, 11], [2, 1])),
      call("multiplyAndSquare", {"a": T([1, 2, 3, 4], [2, 2]), "b": T([1, 2], [2, 1])}, T([25, 121], [2, 1]))])
xl = np.array([1, 2, -1, -2, 0, 3], dtype=f32)
case("leakyReluGpu", "tests/test_gpu.nim:238-246",
     [call("y", {"x": T(xl)}, T(np.where(xl > 0, xl, f32(0.01) * xl)))])

case("customGrad", "tests/test_model.nim:196-214",
     [call("identity", {"inp": T([1, 2, 3, 4], [2, 2])}, T([1, 2, 3, 4], [2, 2])),
      call("grad", {"inp": T([1, 2, 3, 4], [2, 2])}, T([2, 4, 6, 8], [2, 2]))])

img7 = [1, 2, 3, 2, 1, 0, -1]
blurred = [2, f32(7 / 3), 2, 1, 0]
case("blur", "tests/test_model.nim:99-107", [call("res", {"image": T(img7)}, T(blurred))])
case("blurCenter", "tests/test_model.nim:109-117", [call("res", {"image": T(img7)}, T(blurred))])
case("blurOffset", "tests/test_model.nim:119-128", [call("res", {"image": T(img7)}, T([0] + blurred + [0]))])
x23 = np.array([1, 2, 3, 4, 5, 6], dtype=f32)
for factor in range(-2, 3):
    case(f"extern/{factor}", "tests/test_model.nim:156-167",
         [call("y", {"x": T(x23, [2, 3])}, T(x23 * f32(factor), [2, 3]))])
x32 = np.array([1, 2, 3, 4, 5, 6], dtype=f32)
case("dynamicAst/0", "tests/test_model.nim:215-231", [call("y", {"x": T(x32, [3, 2])}, T(np.ones(6), [3, 2]), "sumsq", 0.001)])
case("dynamicAst/1", "tests/test_model.nim:215-231", [call("y", {"x": T(x32, [3, 2])}, T(x32, [3, 2]), "sumsq", 0.001)])

case("array", "tests/test_model.nim:233-241", [call("y", {}, T([4, 5, 6]))])
case("nestedArray", "tests/test_model.nim:243-255", [call("y", {}, T([1, 2, 3, 4, 5, 6, 7, 8, 9], [3, 3]))])

out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "known_answers.json")
with open(out, "w") as f:
    json.dump(cases, f, indent=1)
print("wrote", out, len(cases), "cases")
