"""ctypes binding of oracle/librefcpu.so (ORACLE — test infrastructure, never the product path).

Every function mirrors one loop nest of the reference's CPU lowering; see oracle/refcpu.c for
the file:line citations.  Arrays are float32, C-contiguous numpy arrays.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "librefcpu.so")

MAP_OPS = {
    "identity": 0, "relu": 1, "leaky_relu": 2, "sigmoid": 3, "tanh": 4,
    "scale": 5, "sin": 6, "xor_leaky": 7, "exp": 8,
}


def build(force=False):
    """Compile the oracle with gcc (seconds)."""
    srcs = [os.path.join(_HERE, f) for f in ("refcpu.c", "refinterp.c", "refinterp_body.h", "Makefile")]
    srcs = [s for s in srcs if os.path.exists(s)]
    if (not force and os.path.exists(_SO)
            and all(os.path.getmtime(_SO) >= os.path.getmtime(s) for s in srcs)):
        return _SO
    subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        i64, f32p, f64p, cint, f32 = (ctypes.c_int64, ctypes.POINTER(ctypes.c_float),
                                      ctypes.POINTER(ctypes.c_double), ctypes.c_int, ctypes.c_float)
        _lib.ref_sgemm.argtypes = [cint, cint, i64, i64, i64, f32p, i64, f32p, i64, f32p, i64, cint]
        _lib.ref_thread_count.argtypes = [i64, i64, cint]
        _lib.ref_thread_count.restype = cint
        _lib.ref_bias_add.argtypes = [i64, i64, f32p, f32p]
        _lib.ref_colsum.argtypes = [i64, i64, f32p, f32p]
        _lib.ref_rowsum.argtypes = [i64, i64, f32p, f32p]
        _lib.ref_sum.argtypes = [i64, f32p, f32p]
        _lib.ref_gradient_descent.argtypes = [i64, f32, f32p, f32p]
        _lib.ref_axpy.argtypes = [i64, f32, f32p, f32p]
        _lib.ref_map.argtypes = [cint, i64, f32p, f32p, f32]
        _lib.ref_map_grad.argtypes = [cint, i64, f32p, f32p, f32p, f32]
        _lib.ref_conv2_nhwc.argtypes = [i64] * 7 + [f32p, f32p, f32p, cint, cint]
        _lib.ref_conv2_nhwc_grad_filter.argtypes = [i64] * 7 + [f32p, f32p, f32p]
        _lib.ref_conv2_nhwc_grad_image.argtypes = [i64] * 7 + [f32p, f32p, f32p]
        _lib.ref_dgemm_from_f32.argtypes = [cint, cint, i64, i64, i64, f32p, i64, f32p, i64, f64p, i64]
        _lib.ref_dgemm.argtypes = [cint, cint, i64, i64, i64, f64p, i64, f64p, i64, f64p, i64]
        _lib.ref_dgemm.restype = None
        for name in ("ref_sgemm", "ref_bias_add", "ref_colsum", "ref_rowsum", "ref_sum",
                     "ref_gradient_descent", "ref_axpy", "ref_map", "ref_map_grad",
                     "ref_conv2_nhwc", "ref_conv2_nhwc_grad_filter", "ref_conv2_nhwc_grad_image", "ref_dgemm_from_f32"):
            getattr(_lib, name).restype = None
    return _lib


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a


def _p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def sgemm(a, b, trans_a=False, trans_b=False, out=None, threads=1):
    """out (+)= op(a) @ op(b); `out` defaults to zeros (a fresh result tensor, model.nim:295-300)."""
    a, b = _f32(a), _f32(b)
    M, K = (a.shape[1], a.shape[0]) if trans_a else a.shape
    N = b.shape[0] if trans_b else b.shape[1]
    assert (b.shape[1] if trans_b else b.shape[0]) == K
    if out is None:
        out = np.zeros((M, N), dtype=np.float32)
    assert out.dtype == np.float32 and out.flags.c_contiguous and out.shape == (M, N)
    lib().ref_sgemm(int(trans_a), int(trans_b), M, N, K, _p(a), a.shape[1], _p(b), b.shape[1],
                    _p(out), N, threads)
    return out


def dgemm64(a, b, trans_a=False, trans_b=False, out=None):
    """compile[float64]: out (+)= op(a) @ op(b) over float64 in the reference's loop order (ref_dgemm)."""
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    M, K = (a.shape[1], a.shape[0]) if trans_a else a.shape
    N = b.shape[0] if trans_b else b.shape[1]
    assert (b.shape[1] if trans_b else b.shape[0]) == K
    if out is None:
        out = np.zeros((M, N), dtype=np.float64)
    assert out.dtype == np.float64 and out.flags.c_contiguous and out.shape == (M, N)
    pd = lambda x: x.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
    lib().ref_dgemm(int(trans_a), int(trans_b), M, N, K, pd(a), a.shape[1], pd(b), b.shape[1], pd(out), N)
    return out


def dgemm(a, b, trans_a=False, trans_b=False):
    a, b = _f32(a), _f32(b)
    M, K = (a.shape[1], a.shape[0]) if trans_a else a.shape
    N = b.shape[0] if trans_b else b.shape[1]
    out = np.zeros((M, N), dtype=np.float64)
    lib().ref_dgemm_from_f32(int(trans_a), int(trans_b), M, N, K, _p(a), a.shape[1], _p(b), b.shape[1],
                             out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), N)
    return out


def thread_count(size, work_per_iter, pool):
    return lib().ref_thread_count(size, work_per_iter, pool)


def bias_add(bias, out):
    bias = _f32(bias)
    rows, cols = out.shape
    lib().ref_bias_add(rows, cols, _p(bias), _p(out))
    return out


def colsum(x, out=None):
    x = _f32(x)
    rows, cols = x.shape
    if out is None:
        out = np.zeros((cols,), dtype=np.float32)
    lib().ref_colsum(rows, cols, _p(x), _p(out))
    return out


def rowsum(x, out=None):
    x = _f32(x)
    rows, cols = x.shape
    if out is None:
        out = np.zeros((rows,), dtype=np.float32)
    lib().ref_rowsum(rows, cols, _p(x), _p(out))
    return out


def total(x, out=None):
    x = _f32(x).reshape(-1)
    if out is None:
        out = np.zeros((1,), dtype=np.float32)
    lib().ref_sum(x.size, _p(x), _p(out))
    return out


def gradient_descent(param, grad, rate):
    grad = _f32(grad)
    assert param.dtype == np.float32 and param.flags.c_contiguous
    lib().ref_gradient_descent(param.size, float(np.float32(rate)), _p(grad), _p(param))
    return param


def axpy(alpha, x, y):
    x = _f32(x)
    lib().ref_axpy(x.size, float(np.float32(alpha)), _p(x), _p(y))
    return y


def map_(op, x, param=0.0, out=None):
    x = _f32(x)
    if out is None:
        out = np.zeros_like(x)
    lib().ref_map(MAP_OPS[op], x.size, _p(x), _p(out), float(np.float32(param)))
    return out


def map_grad(op, x, gout, param=0.0, out=None):
    x, gout = _f32(x), _f32(gout)
    if out is None:
        out = np.zeros_like(x)
    lib().ref_map_grad(MAP_OPS[op], x.size, _p(x), _p(gout), _p(out), float(np.float32(param)))
    return out


def conv2_nhwc(img, flt, out=None, threads_n=1, threads_y=1):
    img, flt = _f32(img), _f32(flt)
    N, H, W, C = img.shape
    F, FH, FW, C2 = flt.shape
    assert C == C2
    if out is None:
        out = np.zeros((N, H - FH + 1, W - FW + 1, F), dtype=np.float32)
    lib().ref_conv2_nhwc(N, H, W, C, F, FH, FW, _p(img), _p(flt), _p(out), threads_n, threads_y)
    return out


def conv2_nhwc_grad_filter(img, gout, flt_shape, out=None):
    img, gout = _f32(img), _f32(gout)
    N, H, W, C = img.shape
    F, FH, FW, _ = flt_shape
    if out is None:
        out = np.zeros(flt_shape, dtype=np.float32)
    lib().ref_conv2_nhwc_grad_filter(N, H, W, C, F, FH, FW, _p(img), _p(gout), _p(out))
    return out


def conv2_nhwc_grad_image(flt, gout, img_shape, out=None):
    flt, gout = _f32(flt), _f32(gout)
    N, H, W, C = img_shape
    F, FH, FW, _ = flt.shape
    if out is None:
        out = np.zeros(img_shape, dtype=np.float32)
    lib().ref_conv2_nhwc_grad_image(N, H, W, C, F, FH, FW, _p(flt), _p(gout), _p(out))
    return out
