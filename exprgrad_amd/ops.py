"""Library fast path (group 2 of the C ABI) on raw device pointers.

Arguments are device addresses (ints): GpuBuffer.ptr, GpuTensor.ptr or torch.Tensor.data_ptr().
All calls enqueue on the context's stream and return immediately.
"""
import ctypes

from ._lib import call

MAP_OPS = {
    "identity": 0, "relu": 1, "leaky_relu": 2, "sigmoid": 3, "tanh": 4,
    "scale": 5, "sin": 6, "xor_leaky": 7, "exp": 8,
}


def _p(x):
    if x is None:
        return ctypes.c_void_p(0)
    if hasattr(x, "data_ptr"):
        x = x.data_ptr()
    elif hasattr(x, "ptr"):
        x = x.ptr
    return ctypes.c_void_p(int(x))


def sgemm(ctx, M, N, K, A, lda, B, ldb, C, ldc, trans_a=False, trans_b=False, accumulate=False, bias=None):
    call("eg_sgemm", ctx.handle, int(trans_a), int(trans_b), M, N, K, _p(A), lda, _p(B), ldb, _p(C), ldc,
         int(accumulate), _p(bias))


def bias_add(ctx, rows, cols, bias, out, accumulate=True):
    call("eg_bias_add", ctx.handle, rows, cols, _p(bias), _p(out), int(accumulate))


def colsum(ctx, rows, cols, x, out, accumulate=False):
    call("eg_colsum", ctx.handle, rows, cols, _p(x), _p(out), int(accumulate))


def rowsum(ctx, rows, cols, x, out, accumulate=False):
    call("eg_rowsum", ctx.handle, rows, cols, _p(x), _p(out), int(accumulate))


def total(ctx, n, x, out, accumulate=False):
    call("eg_sum", ctx.handle, n, _p(x), _p(out), int(accumulate))


def axpy(ctx, n, alpha, x, y):
    call("eg_axpy", ctx.handle, n, float(alpha), _p(x), _p(y))


def fill(ctx, n, value, out):
    call("eg_fill_f32", ctx.handle, n, float(value), _p(out))


def map_(ctx, op, n, x, out, param=0.0, accumulate=False):
    call("eg_map", ctx.handle, MAP_OPS[op], n, _p(x), _p(out), float(param), int(accumulate))


def map_grad(ctx, op, n, x, gout, gin, param=0.0, accumulate=False):
    call("eg_map_grad", ctx.handle, MAP_OPS[op], n, _p(x), _p(gout), _p(gin), float(param), int(accumulate))


def conv2_nhwc(ctx, N, H, W, C, F, FH, FW, img, flt, out, accumulate=False):
    call("eg_conv2_nhwc", ctx.handle, N, H, W, C, F, FH, FW, _p(img), _p(flt), _p(out), int(accumulate))


def conv2_nhwc_grad_filter(ctx, N, H, W, C, F, FH, FW, img, gout, gflt, accumulate=False):
    """gflt[f,dy,dx,c] (+)= sum gout[n,y,x,f] * img[n,y+dy,x+dx,c]  (derived from dnn.nim:45-49)."""
    call("eg_conv2_nhwc_grad_filter", ctx.handle, N, H, W, C, F, FH, FW, _p(img), _p(gout), _p(gflt), int(accumulate))


def conv2_nhwc_grad_image(ctx, N, H, W, C, F, FH, FW, flt, gout, gimg, accumulate=False):
    """gimg[n,y+dy,x+dx,c] (+)= sum gout[n,y,x,f] * flt[f,dy,dx,c]  (derived from dnn.nim:45-49)."""
    call("eg_conv2_nhwc_grad_image", ctx.handle, N, H, W, C, F, FH, FW, _p(flt), _p(gout), _p(gimg), int(accumulate))
