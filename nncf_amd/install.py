"""Drop-in installation into an importable NNCF (the reference): mirrors the reference's own
`_can_run_optimized` seam (weight_lowering.py:687-702, which diverts to nncf.openvino.optimized_functions) for
torch tensors that live on an AMD GPU.

    import nncf, nncf_amd.install
    nncf_amd.install.install()      # nncf.compress_weights(...) on a model on "cuda" now lowers through HIP

Every patched function keeps the reference's signature, checks `_can_run_hip(...)` and otherwise calls the
original.  Kill switch: environment variable NNCF_AMD_DISABLE (any value), like NNCF_DISABLE_OPTIMIZED_COMPRESSION.
"""
from __future__ import annotations

import functools
import os

import torch

from nncf_amd import errors as _errors
from nncf_amd import gptq as _gptq
from nncf_amd import statistics as _stats
from nncf_amd import weight_lowering as _wl

MIN_INPUT_SIZE = 10000  # constants.py:116 MIN_INPUT_SIZE_FOR_OPTIMIZED_COMPRESSION
_SUPPORTED = {"int8_asym", "int8_sym", "int4_asym", "int4_sym", "nf4", "mxfp4", "fp4"}
_installed = {}


def _raw(t):
    return getattr(t, "data", t) if t is not None and not isinstance(t, torch.Tensor) else t


def _can_run_hip(weight, mode, min_size=MIN_INPUT_SIZE) -> bool:
    w = _raw(weight)
    return (isinstance(w, torch.Tensor) and w.is_cuda and w.numel() >= min_size
            and os.environ.get("NNCF_AMD_DISABLE") is None and getattr(mode, "value", mode) in _SUPPORTED
            and w.dtype in (torch.float32, torch.float16, torch.bfloat16, torch.float64))


def install():
    """Patches the reference in place; returns the list of patched qualified names.  Idempotent."""
    if _installed:
        return sorted(_installed)
    from nncf.common.tensor_statistics import collectors as C
    from nncf.quantization.algorithms.weight_compression import gptq as ref_gptq
    from nncf.quantization.algorithms.weight_compression import weight_lowering as WL
    from nncf.quantization.algorithms.weight_compression.parameters import CompressedWeight as RefCW
    from nncf.tensor import Tensor
    from nncf.torch.quantization import quantize_functions as QF

    def wrap(t):
        return None if t is None else Tensor(t)

    def wrap_cw(cw, scale_given, zp_given):
        # given parameters are returned verbatim (weight_lowering.py:502-505)
        return RefCW(wrap(cw.tensor), scale_given if scale_given is not None else wrap(cw.scale),
                     zp_given if zp_given is not None else wrap(cw.zero_point))

    def patch(module, name, make):
        orig = getattr(module, name)
        new = functools.wraps(orig)(make(orig))
        setattr(module, name, new)
        _installed[f"{module.__name__}.{name}"] = (module, name, orig)

    def mk_do_int(orig):
        def f(weight, config, reduction_axes=None, precomputed_scale=None, precomputed_zero_point=None):
            if not _can_run_hip(weight, config.mode):
                return orig(weight, config, reduction_axes, precomputed_scale, precomputed_zero_point)
            cw = _wl.do_integer_quantization(_raw(weight), config, reduction_axes, _raw(precomputed_scale),
                                             _raw(precomputed_zero_point))
            return wrap_cw(cw, precomputed_scale, precomputed_zero_point)
        return f

    def mk_int_qdq(orig):
        def f(weight, config, reduction_axes=None, precomputed_scale=None, precomputed_zero_point=None,
              return_compressed_weight=False):
            if not _can_run_hip(weight, config.mode):
                return orig(weight, config, reduction_axes, precomputed_scale, precomputed_zero_point,
                            return_compressed_weight)
            res = _wl.integer_quantize_dequantize_weight(_raw(weight), config, reduction_axes, _raw(precomputed_scale),
                                                         _raw(precomputed_zero_point), return_compressed_weight)
            if return_compressed_weight:
                return Tensor(res[0]), wrap_cw(res[1], precomputed_scale, precomputed_zero_point)
            return Tensor(res)
        return f

    def mk_do_float(orig):
        def f(weight, config, reduction_axes=None, precomputed_scale=None):
            if not _can_run_hip(weight, config.mode):
                return orig(weight, config, reduction_axes, precomputed_scale)
            cw = _wl.do_float_quantization(_raw(weight), config, reduction_axes, _raw(precomputed_scale))
            return wrap_cw(cw, precomputed_scale, None)
        return f

    def mk_float_qdq(orig):
        def f(weight, config, reduction_axes=None, precomputed_scale=None, return_compressed_weight=False):
            if not _can_run_hip(weight, config.mode):
                return orig(weight, config, reduction_axes, precomputed_scale, return_compressed_weight)
            res = _wl.float_quantize_dequantize_weight(_raw(weight), config, reduction_axes, _raw(precomputed_scale),
                                                       return_compressed_weight)
            if return_compressed_weight:
                return Tensor(res[0]), wrap_cw(res[1], precomputed_scale, None)
            return Tensor(res)
        return f

    def mk_err(orig):
        def f(weight, reduction_axes, config, reduction):
            if reduction not in ("max_mean", "frobenius") or not _can_run_hip(weight, config.mode):
                return orig(weight, reduction_axes, config, reduction)
            return _wl.get_integer_quantization_error(_raw(weight), reduction_axes, config, reduction)
        return f

    def mk_params(orig):
        def f(weight, reduction_axes, config):
            if not config.is_integer or not _can_run_hip(weight, config.mode, 1):
                return orig(weight, reduction_axes, config)
            s, z = _wl.calculate_integer_quantization_params(_raw(weight), reduction_axes, config)
            return Tensor(s), wrap(z)
        return f

    patch(WL, "do_integer_quantization", mk_do_int)
    patch(WL, "integer_quantize_dequantize_weight", mk_int_qdq)
    patch(WL, "_do_float_quantization_single_scale", mk_do_float)
    patch(WL, "float_quantize_dequantize_weight", mk_float_qdq)
    patch(WL, "get_integer_quantization_error", mk_err)
    patch(WL, "calculate_integer_quantization_params", mk_params)

    def mk_pack(fn):
        def make(orig):
            def f(tensor):
                if isinstance(tensor, torch.Tensor) and tensor.is_cuda and os.environ.get("NNCF_AMD_DISABLE") is None:
                    return fn(tensor)
                return orig(tensor)
            return f
        return make

    patch(QF, "pack_uint4", mk_pack(_wl.pack_uint4))
    patch(QF, "pack_int4", mk_pack(_wl.pack_int4))

    def mk_reducer(cls_amd, needs_keepdims):
        def make(orig):
            def f(self, x):
                t = _raw(x[0])
                if (isinstance(t, torch.Tensor) and t.is_cuda and os.environ.get("NNCF_AMD_DISABLE") is None
                        and (not needs_keepdims or getattr(self, "_keepdims", True))):
                    # the reference resolves its axes through the reducer's axes mode (collectors.py:494-538)
                    axes = C.determine_reduction_axes(t.ndim, self._axes, self._axes_mode)
                    try:
                        return [Tensor(cls_amd(tuple(axes))._reduce_out_of_place([t])[0])]
                    except _errors.UnsupportedModelError:
                        pass  # axes outside the weight-compression statistic forms: reference path
                    # anything else (launch failures, bad status) is an error of this library and propagates
                return orig(self, x)
            return f
        return make

    patch(C.MeanReducer, "_reduce_out_of_place", mk_reducer(_stats.MeanReducer, True))
    patch(C.MeanVarianceReducer, "_reduce_out_of_place", mk_reducer(_stats.MeanVarianceReducer, False))
    patch(C.MaxVarianceReducer, "_reduce_out_of_place", mk_reducer(_stats.MaxVarianceReducer, False))
    patch(C.MeanAbsMaxReducer, "_reduce_out_of_place", mk_reducer(_stats.MeanAbsMaxReducer, False))

    def mk_hessian(orig):
        def f(self, node, inputs, is_3d_weight=False):
            t = _raw(inputs[0])
            if not (isinstance(t, torch.Tensor) and t.is_cuda) or os.environ.get("NNCF_AMD_DISABLE") is not None:
                return orig(self, node, inputs, is_3d_weight)
            return Tensor(_gptq.calculate_hessian([_raw(i) for i in inputs], is_3d_weight))
        return f

    def mk_quantize_weights(orig):
        def f(self, wc_params, hessian, weight_tensor, inputs):
            w = _raw(weight_tensor)
            cfg = wc_params.compression_config
            if (not (isinstance(w, torch.Tensor) and w.is_cuda) or os.environ.get("NNCF_AMD_DISABLE") is not None
                    or cfg.mode.value not in _SUPPORTED or tuple(wc_params.reduction_axes) != (1,)
                    or self._block_size > 128):
                return orig(self, wc_params, hessian, weight_tensor, inputs)
            q, s, z = _gptq.quantize_weights(_raw(hessian), w, cfg, [_raw(i) for i in inputs], self._damp_percent,
                                             self._block_size, self._scale_estimation, wc_params.reduction_axes)
            return Tensor(q), Tensor(s), wrap(z)
        return f

    patch(ref_gptq.GPTQ, "_calculate_hessian", mk_hessian)
    patch(ref_gptq.GPTQ, "_quantize_weights", mk_quantize_weights)
    return sorted(_installed)


def uninstall():
    for module, name, orig in _installed.values():
        setattr(module, name, orig)
    _installed.clear()
