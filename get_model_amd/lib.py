"""ctypes binding of libgetmi355.so (include/getmi355.h).

The product path has NO fallback: if the library is missing, or a call returns non-zero, a RuntimeError is
raised.  Tensors are passed as raw device pointers; the current torch HIP stream is passed to every call.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GETMI_LIB", os.path.join(_HERE, "libgetmi355.so"))   # GETMI_LIB: A/B another build

_vp, _i, _i64, _f, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_size_t

# name -> (restype, argtypes); mirrors include/getmi355.h one to one
_SIGNATURES = {
    "getmi_last_error": (ctypes.c_char_p, []),
    "getmi_version": (_i, []),
    "getmi_arch": (ctypes.c_char_p, []),
    "getmi_cast_pad_f32_bf16": (_i, [_vp, _vp, _i64, _i, _i, _vp]),
    "getmi_cast_transpose_f32_bf16": (_i, [_vp, _vp, _i, _i, _vp]),
    "getmi_qkv_bias": (_i, [_vp, _vp, _vp, _i, _vp]),
    "getmi_gemm_tn": (_i, [_i, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _vp, _i, _vp, _i, _vp, _vp, _i, _vp, _vp,
                           _sz, _i, _vp]),
    "getmi_gemm_tn_blocked_ok": (_i, [_i, _i, _i]),
    "getmi_gemm_tn_workspace_bytes": (_sz, [_i, _i]),
    "getmi_gemm_wgrad_workspace_bytes": (_sz, [_i, _i, _i]),
    "getmi_gemm_wgrad": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _i, _vp, _sz, _vp]),
    "getmi_layernorm_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _vp]),
    "getmi_layernorm_bwd_workspace_bytes": (_sz, [_i, _i]),
    "getmi_layernorm_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _sz, _vp]),
    "getmi_layernorm_bwd_stream": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _vp,
                                        _sz, _vp]),
    "getmi_attention_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _f, _vp, _vp]),
    "getmi_attention_bwd_workspace_bytes": (_sz, [_i, _i, _i]),
    "getmi_attention_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _vp, _vp, _sz, _vp]),
    "getmi_fill_cls": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "getmi_embed_bwd_workspace_bytes": (_sz, [_i, _i, _i]),
    "getmi_embed_bwd_prep": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _sz, _vp]),
    "getmi_head_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "getmi_head_bwd_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "getmi_head_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "getmi_loss_workspace_bytes": (_sz, []),
    "getmi_loss_fwd": (_i, [_i, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _i, _f, _f, _vp, _sz, _vp]),
    "getmi_scale_by_scalar": (_i, [_vp, _vp, _f, _vp, _i64, _vp]),
    "getmi_scale_cast_workspace_bytes": (_sz, [_i, _i]),
    "getmi_scale_cast": (_i, [_vp, _vp, _i, _vp, _vp, _i, _i, _i, _vp, _sz, _vp]),
    "getmi_scale_cast_noalloc_max_bytes": (_sz, []),
    "getmi_scale_cast_noalloc": (_i, [_vp, _vp, _i, _vp, _vp, _i, _i, _i, _vp]),
    "getmi_gather_rows_bf16": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    "getmi_mask_compact": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp]),
    "getmi_colsum_workspace_bytes": (_sz, [_i]),
    "getmi_colsum_bf16": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _sz, _vp]),
    "getmi_adamw_step": (_i, [_vp, _i, _i, _vp, _i, _vp]),
    "getmi_gemm_wgrad_pair_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "getmi_gemm_wgrad_pair": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _vp, _i, _i, _vp, _sz, _vp]),
    "getmi_set_cu_limit": (_i, [_i]),
    "getmi_cu_count": (_i, []),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib = None


def load():
    """Load the shared library (once).  Raises RuntimeError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the HIP kernels are not built (run `python -m get_model_amd.build`). "
            "There is no CPU/eager fallback for the GET hot path.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _ptr(x):
    if x is None:
        return None
    if isinstance(x, torch.Tensor):
        return x.data_ptr()
    return x


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream


def call(name, *args):
    """Invoke an int-returning entry point; tensors -> device pointers; appends the current stream."""
    lib = load()
    fn = getattr(lib, name)
    rc = fn(*[_ptr(a) for a in args], stream_ptr())
    if rc != 0:
        raise RuntimeError(f"{name} failed (code {rc}): {lib.getmi_last_error().decode()}")


def query(name, *args):
    """Invoke a size_t/int-returning query (no stream argument)."""
    return getattr(load(), name)(*args)
