"""ctypes binding of libnncf_amd_wc.so (include/nncf_amd_wc.h).  No fallback: if the library is missing
the product path raises."""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_int64, c_size_t, c_void_p

import torch  # noqa: F401  -- must be imported first so that the library binds to torch's HIP runtime

from nncf_amd import errors

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NNCF_AMD_LIB") or os.path.join(_HERE, "libnncf_amd_wc.so")  # override: developer builds

F32, F16, BF16 = 0, 1, 2
ZP_NONE, ZP_I32, ZP_F32 = 0, 1, 2
MAX_BATCH = 92  # tensors per launch of the multi-tensor kernels (csrc/wc_group_kernels.h: kMaxBatch)
MODE_ID = {"int8_asym": 0, "int8_sym": 1, "int4_asym": 2, "int4_sym": 3, "nf4": 4, "mxfp4": 5, "fp4": 6}

OK, ERR_BAD_ARG, ERR_GROUP_SIZE, ERR_DTYPE, ERR_MODE, ERR_LAUNCH, ERR_WORKSPACE, ERR_UNSUPPORTED = (
    0, -1, -2, -3, -4, -5, -6, -7)

# symbol -> (restype, argtypes); mirrors include/nncf_amd_wc.h one to one
SIGNATURES = {
    "nncf_amd_version": (c_int, []),
    "nncf_amd_last_error": (c_char_p, []),
    "nncf_amd_status_string": (c_char_p, [c_int]),
    "nncf_amd_int_quantize": (c_int, [c_void_p, c_int, c_int64, c_int64, c_int64, c_int64, c_int, c_void_p, c_void_p,
                                      c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "nncf_amd_int_quantize_batched": (c_int, [c_int, POINTER(c_void_p), c_int, POINTER(c_int64), POINTER(c_int64),
                                              c_int64, c_int, POINTER(c_void_p), POINTER(c_void_p),
                                              POINTER(c_void_p), POINTER(c_void_p), c_void_p]),
    "nncf_amd_int_dequantize": (c_int, [c_void_p, c_int, c_int64, c_int64, c_int64, c_int64, c_void_p, c_void_p,
                                        c_int, c_void_p, c_void_p]),
    "nncf_amd_int_quant_error_workspace": (c_size_t, [c_int64, c_int64]),
    "nncf_amd_int_quant_error": (c_int, [c_void_p, c_int, c_int64, c_int64, c_int64, c_int64, c_int, c_int,
                                         c_void_p, c_void_p, c_size_t, c_void_p]),
    "nncf_amd_int_quant_error_batched_workspace": (c_size_t, [c_int, POINTER(c_int64)]),
    "nncf_amd_int_quant_error_batched": (c_int, [c_int, POINTER(c_void_p), c_int, POINTER(c_int64), POINTER(c_int64),
                                                 c_int64, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "nncf_amd_int_quant_error_static": (c_int, [c_void_p, c_int, c_int64, c_int64, c_int64, c_int64, c_int, c_int,
                                                c_void_p, c_void_p]),
    "nncf_amd_pack_u4": (c_int, [c_void_p, c_int64, c_void_p, c_void_p]),
    "nncf_amd_pack_i4": (c_int, [c_void_p, c_int64, c_void_p, c_void_p]),
    "nncf_amd_unpack_u4": (c_int, [c_void_p, c_int64, c_void_p, c_void_p]),
    "nncf_amd_unpack_i4": (c_int, [c_void_p, c_int64, c_void_p, c_void_p]),
    "nncf_amd_float_quantize": (c_int, [c_void_p, c_int, c_int64, c_int64, c_int64, c_int64, c_int, c_void_p,
                                        c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "nncf_amd_float_quantize_batched": (c_int, [c_int, POINTER(c_void_p), c_int, POINTER(c_int64), POINTER(c_int64),
                                                c_int64, c_int, POINTER(c_void_p), POINTER(c_void_p),
                                                POINTER(c_void_p), c_void_p]),
    "nncf_amd_reduce_workspace": (c_size_t, [c_int64, c_int64, c_int64]),
    "nncf_amd_reduce_activation_stats": (c_int, [c_void_p, c_int, c_int64, c_int64, c_int64, c_void_p, c_void_p,
                                                 c_void_p, c_size_t, c_void_p]),
    "nncf_amd_se_step_supported": (c_int, [c_int64, c_int64]),
    "nncf_amd_se_step": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                 c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_int64, c_int, c_int, c_float,
                                 c_float, c_float, c_void_p]),
    "nncf_amd_se_search": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_int64, c_int, c_int, c_int,
                                   c_float, c_float, c_void_p]),
    "nncf_amd_se_zero_mask": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64,
                                      c_int, c_void_p, c_void_p, c_void_p]),
    "nncf_amd_awq_sacts_bytes": (c_int64, [c_int64, c_int64, c_int]),
    "nncf_amd_awq_prepare": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float, c_int, c_int, c_int64, c_int64,
                                     c_int, c_void_p, c_void_p, c_void_p]),
    "nncf_amd_awq_search": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int, c_int64, c_void_p, c_int, c_void_p, c_void_p,
                                    c_int64, c_int, c_void_p, c_void_p]),
    "nncf_amd_decompress": (c_int, [c_void_p, c_int, c_int64, c_int64, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "nncf_amd_token_mean_workspace": (c_size_t, [c_int, c_int64, c_int64]),
    "nncf_amd_token_mean_batched": (c_int, [c_int, POINTER(c_void_p), c_int, POINTER(c_int64), c_int64, c_void_p, c_void_p,
                                            c_size_t, c_void_p]),
    "nncf_amd_hessian_accumulate": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int64, c_float, c_float, c_void_p]),
    "nncf_amd_hessian_accumulate_upper": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int64, c_float, c_float,
                                                  c_void_p]),
    "nncf_amd_hessian_accumulate_list": (c_int, [c_void_p, c_int, POINTER(c_void_p), POINTER(c_int64), c_int, c_int64,
                                                 c_float, c_void_p]),
    "nncf_amd_hessian_finalize": (c_int, [c_void_p, c_int64, c_float, c_void_p]),
    "nncf_amd_gptq_block": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64,
                                    c_int64, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "nncf_amd_split3_bf16": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int, c_int, c_int, c_void_p, c_int64, c_void_p]),
    "nncf_amd_gemm_bf16_tn_acc": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int64, c_int64,
                                          c_int, c_void_p]),
    "nncf_amd_cholinv_block": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_void_p]),
    "nncf_amd_gptq_pack8": (c_int, [c_void_p, c_int, c_int64, c_int64, c_void_p, c_void_p]),
    "nncf_amd_gptq_dequant8": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int, c_void_p]),
}

_lib = None


def lib():
    """Loads the shared library once.  Raises if it has not been built (python -m nncf_amd.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            msg = (f"{LIB_PATH} is missing: the HIP extension has not been built. "
                   "Run `python -m nncf_amd.build` (hipcc, gfx950). There is no CPU fallback.")
            raise ImportError(msg)
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name, None)
            if fn is None:
                msg = f"{LIB_PATH} does not export {name}: stale build, run `python -m nncf_amd.build --force`"
                raise ImportError(msg)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def last_error() -> str:
    return lib().nncf_amd_last_error().decode()


def check(status: int) -> None:
    """Maps a C status to the exception type the reference raises for the same condition."""
    if status == OK:
        return
    msg = last_error()
    if status == ERR_GROUP_SIZE:
        raise errors.InvalidGroupSizeError(msg)  # weight_lowering.py:72-74
    if status == ERR_BAD_ARG:
        raise ValueError(msg)  # weight_lowering.py:472-477, 498-500
    if status in (ERR_MODE, ERR_DTYPE, ERR_UNSUPPORTED, ERR_WORKSPACE):
        raise errors.InternalError(msg)
    raise RuntimeError(f"nncf_amd status {status}: {msg}")


def dtype_id(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.float16:
        return F16
    if t.dtype == torch.bfloat16:
        return BF16
    raise errors.InternalError(f"unsupported weight dtype {t.dtype}")


def ptr(t):
    return None if t is None else c_void_p(t.data_ptr())


def stream_of(t: torch.Tensor):
    return c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def require_gpu(t: torch.Tensor, what: str) -> None:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        msg = f"{what}: expected a torch tensor on an AMD GPU ('cuda' device); got {type(t).__name__} on " \
              f"{getattr(t, 'device', '?')}. nncf_amd has no CPU path."
        raise errors.UnsupportedBackendError(msg)
