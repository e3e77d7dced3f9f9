"""ctypes loader for libexprgrad_hip.so (the C ABI declared in include/exprgrad_hip.h).

There is no CPU fallback: if the HIP library is missing or a call fails, this raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# EG_LIB_PATH: another build of the library (A/B timing of two builds on one GPU box); the default is the in-tree one
LIB_PATH = os.environ.get("EG_LIB_PATH") or os.path.join(_HERE, "lib", "libexprgrad_hip.so")


class GpuError(RuntimeError):
    """Mirror of exprgrad's GpuError (runtimes/cl.nim:18, raised by `check` at cl.nim:41-43)."""


class RuntimeErrorEG(GpuError):
    """exprgrad's RuntimeError: unknown target / input (model.nim:358-359, 395-396)."""


class ShapeError(GpuError):
    """exprgrad's ShapeError (passes.nim shape inference)."""


EG_OK = 0
_ERR_CLASSES = {6: RuntimeErrorEG, 7: ShapeError}

_lib = None

c_i64 = ctypes.c_int64
c_int = ctypes.c_int
c_f32 = ctypes.c_float
c_f64 = ctypes.c_double
c_void_p = ctypes.c_void_p
c_char_p = ctypes.c_char_p
c_size_t = ctypes.c_size_t
P = ctypes.POINTER

_SIGS = {
    "eg_last_error": (c_char_p, []),
    "eg_version": (c_int, []),
    "eg_device_count": (c_int, [P(c_int)]),
    "eg_device_info": (c_int, [c_int, c_char_p, c_size_t, c_char_p, c_size_t, c_char_p, c_size_t, P(c_int)]),
    "eg_device_props": (c_int, [c_int, P(c_int), P(c_int), P(c_i64), c_char_p, c_size_t]),
    "eg_compiler_info": (c_int, [c_char_p, c_size_t]),
    "eg_kernel_cache_stats": (c_int, [P(c_i64), P(c_i64), P(ctypes.c_double)]),
    "eg_switches_reload": (c_int, []),
    "eg_switch_table": (c_i64, [c_char_p, c_size_t]),
    "eg_ctx_create": (c_int, [c_int, P(c_void_p)]),
    "eg_ctx_create_on_stream": (c_int, [c_int, c_void_p, P(c_void_p)]),
    "eg_ctx_destroy": (c_int, [c_void_p]),
    "eg_ctx_sync": (c_int, [c_void_p]),
    "eg_ctx_stream": (c_void_p, [c_void_p]),
    "eg_ctx_device": (c_int, [c_void_p]),
    "eg_buf_alloc": (c_int, [c_void_p, c_size_t, P(c_void_p)]),
    "eg_buf_wrap": (c_int, [c_void_p, c_void_p, c_size_t, P(c_void_p)]),
    "eg_buf_free": (c_int, [c_void_p]),
    "eg_buf_size": (c_size_t, [c_void_p]),
    "eg_buf_ptr": (c_void_p, [c_void_p]),
    "eg_buf_write": (c_int, [c_void_p, c_void_p, c_size_t]),
    "eg_buf_read": (c_int, [c_void_p, c_void_p, c_size_t]),
    "eg_buf_fill": (c_int, [c_void_p, c_void_p, c_size_t]),
    "eg_host_alloc": (c_int, [c_size_t, P(c_void_p)]),
    "eg_host_free": (c_int, [c_void_p]),
    "eg_kernel_compile": (c_int, [c_void_p, c_char_p, c_char_p, P(c_void_p)]),
    "eg_kernel_free": (c_int, [c_void_p]),
    "eg_kernel_set_arg_buf": (c_int, [c_void_p, c_int, c_void_p]),
    "eg_kernel_set_arg_i64": (c_int, [c_void_p, c_int, c_i64]),
    "eg_kernel_set_arg_f32": (c_int, [c_void_p, c_int, c_f32]),
    "eg_kernel_set_arg_f64": (c_int, [c_void_p, c_int, c_f64]),
    "eg_kernel_launch": (c_int, [c_void_p, c_int, P(c_i64), P(c_i64)]),
    "eg_sgemm": (c_int, [c_void_p, c_int, c_int, c_i64, c_i64, c_i64, c_void_p, c_i64, c_void_p, c_i64,
                         c_void_p, c_i64, c_int, c_void_p]),
    "eg_bias_add": (c_int, [c_void_p, c_i64, c_i64, c_void_p, c_void_p, c_int]),
    "eg_colsum": (c_int, [c_void_p, c_i64, c_i64, c_void_p, c_void_p, c_int]),
    "eg_rowsum": (c_int, [c_void_p, c_i64, c_i64, c_void_p, c_void_p, c_int]),
    "eg_sum": (c_int, [c_void_p, c_i64, c_void_p, c_void_p, c_int]),
    "eg_axpy": (c_int, [c_void_p, c_i64, c_f32, c_void_p, c_void_p]),
    "eg_fill_f32": (c_int, [c_void_p, c_i64, c_f32, c_void_p]),
    "eg_map": (c_int, [c_void_p, c_int, c_i64, c_void_p, c_void_p, c_f32, c_int]),
    "eg_map_grad": (c_int, [c_void_p, c_int, c_i64, c_void_p, c_void_p, c_void_p, c_f32, c_int]),
    "eg_conv2_nhwc": (c_int, [c_void_p] + [c_i64] * 7 + [c_void_p, c_void_p, c_void_p, c_int]),
    "eg_conv2_nhwc_grad_filter": (c_int, [c_void_p] + [c_i64] * 7 + [c_void_p, c_void_p, c_void_p, c_int]),
    "eg_conv2_nhwc_grad_image": (c_int, [c_void_p] + [c_i64] * 7 + [c_void_p, c_void_p, c_void_p, c_int]),
    "eg_model_compile": (c_int, [c_void_p, c_char_p, P(c_void_p)]),
    "eg_model_free": (c_int, [c_void_p]),
    "eg_model_plan_text": (c_char_p, [c_void_p]),
    "eg_model_launch_text": (c_char_p, [c_void_p, c_char_p]),
    "eg_model_set_seed": (c_int, [c_void_p, ctypes.c_uint64]),
    "eg_model_keep_values": (c_int, [c_void_p, c_int]),
    "eg_fill_uniform": (c_int, [c_void_p, c_i64, ctypes.c_float, ctypes.c_float, c_void_p, ctypes.c_uint64, c_void_p]),
    "eg_rng_advance": (c_int, [c_void_p, c_void_p]),
    "eg_model_kernel_count": (c_int, [c_void_p, c_char_p]),
    "eg_model_tensor_count": (c_int, [c_void_p]),
    "eg_model_param_info": (c_int, [c_void_p, c_int, P(c_int), P(c_int), P(c_i64), c_char_p, c_size_t]),
    "eg_model_param_write": (c_int, [c_void_p, c_int, c_void_p, c_i64]),
    "eg_model_param_read": (c_int, [c_void_p, c_int, c_void_p, c_i64]),
    "eg_model_grad_bucket": (c_int, [c_void_p, c_char_p, P(c_void_p), P(c_i64)]),
    "eg_model_param_ptr": (c_int, [c_void_p, c_int, P(c_void_p), P(c_i64)]),
    "eg_model_set_input_host": (c_int, [c_void_p, c_char_p, c_void_p, c_int, P(c_i64)]),
    "eg_model_set_input_device": (c_int, [c_void_p, c_char_p, c_void_p, c_int, P(c_i64)]),
    "eg_model_run": (c_int, [c_void_p, c_char_p]),
    "eg_model_run_backward": (c_int, [c_void_p, c_char_p]),
    "eg_model_run_update": (c_int, [c_void_p, c_char_p]),
    "eg_model_fit": (c_int, [c_void_p, c_char_p, c_int, P(c_char_p), P(c_void_p), P(c_int), P(c_int), P(c_i64), c_i64]),
    "eg_dp_unique_id": (c_int, [c_void_p]),
    "eg_dp_init": (c_int, [c_void_p, c_void_p, c_int, c_int, P(c_void_p)]),
    "eg_dp_free": (c_int, [c_void_p]),
    "eg_dp_rank": (c_int, [c_void_p]),
    "eg_dp_rccl_count": (c_int, [c_void_p]),
    "eg_dp_rccl_rank": (c_int, [c_void_p]),
    "eg_dp_set_split": (c_int, [c_void_p, c_int]),
    "eg_dp_world": (c_int, [c_void_p]),
    "eg_dp_allreduce_sum_f32": (c_int, [c_void_p, c_void_p, c_i64]),
    "eg_model_step_dp": (c_int, [c_void_p, c_char_p, c_void_p, c_int]),
    "eg_dp_last_pieces": (c_int, [c_void_p]),
    "eg_model_set_grad_scale": (c_int, [c_void_p, c_f32]),
    "eg_model_output_shape": (c_int, [c_void_p, c_char_p, P(c_int), P(c_i64)]),
    "eg_model_read_output": (c_int, [c_void_p, c_char_p, c_void_p, c_i64]),
    "eg_model_tensor_shape": (c_int, [c_void_p, c_char_p, c_int, P(c_int), P(c_i64)]),
    "eg_model_read_tensor": (c_int, [c_void_p, c_char_p, c_int, c_void_p, c_i64]),
    "eg_model_tensor_ptr": (c_int, [c_void_p, c_char_p, c_int, P(c_void_p), P(c_i64)]),
    "eg_model_bind_grad_bucket": (c_int, [c_void_p, c_char_p, c_void_p, c_i64]),
    "eg_model_clear_inputs": (c_int, [c_void_p]),
    "eg_model_set_epoch": (c_int, [c_void_p, c_i64]),
    "eg_model_epoch": (c_i64, [c_void_p]),
    "eg_model_state_bytes": (c_int, [c_void_p, P(c_size_t)]),
    "eg_model_store_state": (c_int, [c_void_p, c_void_p, c_size_t, P(c_size_t)]),
    "eg_model_load_state": (c_int, [c_void_p, c_void_p, c_size_t, P(c_size_t)]),
    "eg_model_save": (c_int, [c_void_p, c_char_p]),
    "eg_model_load": (c_int, [c_void_p, c_char_p, P(c_void_p)]),
    "eg_model_source_text": (c_char_p, [c_void_p]),
    "eg_dgemm": (c_int, [c_void_p, c_int, c_int, c_i64, c_i64, c_i64, c_void_p, c_i64, c_void_p, c_i64,
                         c_void_p, c_i64, c_int, c_void_p]),
    "eg_colsum_f64": (c_int, [c_void_p, c_i64, c_i64, c_void_p, c_void_p, c_int]),
    "eg_fill_f64": (c_int, [c_void_p, c_i64, c_f64, c_void_p]),
    "eg_fill_uniform_f64": (c_int, [c_void_p, c_i64, c_f64, c_f64, c_void_p, ctypes.c_uint64, c_void_p]),
    "eg_model_scalar_bytes": (c_int, [c_void_p]),
    "eg_model_param_write_f64": (c_int, [c_void_p, c_int, c_void_p, c_i64]),
    "eg_model_param_read_f64": (c_int, [c_void_p, c_int, c_void_p, c_i64]),
    "eg_model_set_input_host_f64": (c_int, [c_void_p, c_char_p, c_void_p, c_int, P(c_i64)]),
    "eg_model_set_input_device_f64": (c_int, [c_void_p, c_char_p, c_void_p, c_int, P(c_i64)]),
    "eg_model_read_output_f64": (c_int, [c_void_p, c_char_p, c_void_p, c_i64]),
    "eg_model_read_tensor_f64": (c_int, [c_void_p, c_char_p, c_int, c_void_p, c_i64]),
    "eg_model_fit_f64": (c_int, [c_void_p, c_char_p, c_int, P(c_char_p), P(c_void_p), P(c_int), P(c_int), P(c_i64), c_i64]),
    "eg_dp_allreduce_sum_f64": (c_int, [c_void_p, c_void_p, c_i64]),
}

# functions whose int return value is not a status code
_NOT_STATUS = {"eg_version", "eg_model_scalar_bytes", "eg_ctx_device", "eg_model_kernel_count", "eg_model_tensor_count", "eg_dp_rank",
               "eg_dp_world", "eg_dp_last_pieces", "eg_dp_rccl_count", "eg_dp_rccl_rank"}


def declared_symbols():
    """Names this binding expects; tests cross-check them against include/exprgrad_hip.h."""
    return sorted(_SIGS)


def lib():
    """Load the shared library (once).  Raises GpuError if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GpuError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(make -C exprgrad_amd/csrc). There is no CPU fallback.")
        handle = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
        for name, (restype, argtypes) in _SIGS.items():
            fn = getattr(handle, name)  # AttributeError if the .so is stale
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = handle
    return _lib


def last_error():
    msg = lib().eg_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(status):
    """Turn a non-zero status into the exception exprgrad would raise (cl.nim:41-43)."""
    if status != EG_OK:
        raise _ERR_CLASSES.get(status, GpuError)(last_error() or f"status {status}")


def call(name, *args):
    fn = getattr(lib(), name)
    rc = fn(*args)
    if name not in _NOT_STATUS and fn.restype is c_int:
        check(rc)
    return rc


def reload_switches():
    """Re-read the environment switches (csrc/switches.cpp caches them at first use): call after changing an EG_*
    variable in a live process.  A no-op before the library is loaded (the first use reads the environment anyway)."""
    if _lib is not None:
        _lib.eg_switches_reload()


def switch_table():
    """[(name, class, purpose)] of every environment switch the library honours."""
    n = lib().eg_switch_table(None, 0)
    buf = ctypes.create_string_buffer(int(n) + 1)
    lib().eg_switch_table(buf, int(n) + 1)
    return [tuple(line.split("\t")) for line in buf.value.decode().splitlines()]
