"""Host-side mirror of the reference's lowering seam
(src/nncf/quantization/algorithms/weight_compression/weight_lowering.py) for torch tensors that live on an
AMD GPU.  Same function names, argument meaning and error behaviour; the arithmetic runs in the fused HIP
kernels of libnncf_amd_wc.so.  torch is used for allocation / views / streams only.

Tensors are plain ``torch.Tensor``s (the reference wraps them in ``nncf.tensor.Tensor``; `nncf_amd.install`
does the unwrapping when patching the reference).
"""
from __future__ import annotations

import math
from typing import Optional, Tuple, Union

import torch

from nncf_amd import _lib
from nncf_amd.errors import InternalError, InvalidGroupSizeError, UnsupportedModelError, ValidationError
from nncf_amd.parameters import CompressedWeight, WeightCompressionConfig, as_config

ReductionAxes = Union[int, Tuple[int, ...]]


# ------------------------------------------------------------------ shape plumbing
def get_reduction_channel_size(weight_shape, reduction_axes: ReductionAxes):
    """weight_lowering.py:40-54."""
    if isinstance(reduction_axes, tuple) and len(reduction_axes) == 1:
        reduction_axes = reduction_axes[0]
    if not isinstance(reduction_axes, int):
        msg = f"Group-wise quantization expects a single reduction axis, but given: {reduction_axes}."
        raise UnsupportedModelError(msg)
    return weight_shape[reduction_axes], reduction_axes


def reshape_weight_for_grouped_quantization(weight: torch.Tensor, reduction_axes: ReductionAxes, group_size: int):
    """weight_lowering.py:57-81 (a view; the kernels index groups directly)."""
    assert group_size != -1
    channel_size, reduction_axes = get_reduction_channel_size(weight.shape, reduction_axes)
    if channel_size % group_size != 0:
        msg = f"Channel size {channel_size} should be divisible by size of group {group_size}."
        raise InvalidGroupSizeError(msg)
    shape = list(weight.shape)
    shape[reduction_axes:reduction_axes + 1] = (channel_size // group_size, group_size)
    return weight.reshape(shape), reduction_axes + 1


def ungroup_weights(weights: torch.Tensor, reduction_axis: int) -> torch.Tensor:
    """weight_lowering.py:410-425."""
    shape = list(weights.shape)
    shape[reduction_axis] = shape[reduction_axis] * shape[reduction_axis + 1]
    shape[reduction_axis + 1] = 1
    return weights.reshape(shape).squeeze()


def _prep_weight(weight: torch.Tensor, what: str) -> torch.Tensor:
    _lib.require_gpu(weight, what)
    if weight.dtype == torch.float64:
        weight = weight.to(torch.float32)  # the reference upcasts/downcasts everything to fp32 first (:493-494)
    if weight.dtype not in (torch.float32, torch.float16, torch.bfloat16):
        raise InternalError(f"{what}: unsupported weight dtype {weight.dtype}")
    return weight.contiguous()


def _single_axis(ndim: int, reduction_axes: ReductionAxes) -> int:
    if isinstance(reduction_axes, tuple):
        if len(reduction_axes) != 1:
            msg = f"nncf_amd lowers a single reduction axis, but given: {reduction_axes}."
            raise UnsupportedModelError(msg)
        reduction_axes = reduction_axes[0]
    return reduction_axes % ndim


def _layout(shape, axis: int):
    """[outer, r, inner] view of a contiguous tensor reduced along `axis`."""
    outer = math.prod(shape[:axis])
    inner = math.prod(shape[axis + 1:])
    return outer, shape[axis], inner


def _param_shape(shape, axis: int):
    s = list(shape)
    s[axis] = 1
    return s


def _resolve_groups(weight: torch.Tensor, config: WeightCompressionConfig, reduction_axes, precomputed_scale):
    """Returns (weight_view, axis, gs) where weight_view is the grouped view the reference would work on,
    `axis` the axis that is reduced and gs the C-ABI group size along that axis (-1: whole axis)."""
    if config.group_size != -1 and reduction_axes is not None:
        weight, axis = reshape_weight_for_grouped_quantization(weight, reduction_axes, config.group_size)
        return weight, axis, -1
    if reduction_axes is not None:
        return weight, _single_axis(weight.ndim, reduction_axes), -1
    # already grouped, parameters given: the reduced axis is where the scale has extent 1 (weight_lowering.py:479-482)
    if precomputed_scale is None:
        msg = "Reduction axes are required for computing the scale and (zero point) vectors."
        raise ValueError(msg)
    ps = tuple(precomputed_scale.shape)
    ws = tuple(weight.shape)
    if len(ps) == len(ws):
        diff = [i for i in range(len(ws)) if ps[i] != ws[i]]
        if len(diff) == 1 and ps[diff[0]] == 1:
            return weight, diff[0], -1
        if len(diff) == 0:
            return weight.unsqueeze(-1), weight.ndim, -1  # one parameter per element
    return weight, None, None  # general broadcast: handled by expanding the parameters per element


def _per_group(t: Optional[torch.Tensor], weight_view: torch.Tensor, axis, dtype=None):
    """Precomputed parameter tensor -> contiguous per-group array (or per-element for a general broadcast)."""
    if t is None:
        return None
    if dtype is not None and t.dtype != dtype:
        t = t.to(dtype)
    if axis is None:
        return t.expand(weight_view.shape).contiguous()
    return t.contiguous()


# ------------------------------------------------------------------ integer path
def _int_quantize(weight: torch.Tensor, config: WeightCompressionConfig, reduction_axes, precomputed_scale,
                  precomputed_zero_point, want_codes=True, want_packed=False, want_dq=False):
    if not config.is_integer:
        msg = "The function supports integer quantization only"
        raise InternalError(msg)
    if config.is_asym_mode and (precomputed_scale is None) != (precomputed_zero_point is None):
        msg = ("If precomputed quantization parameters are provided, both scale and zero point are required "
               "for asymmetric quantization.")
        raise ValueError(msg)
    weight = _prep_weight(weight, "do_integer_quantization")
    if precomputed_scale is None and reduction_axes is None:
        msg = "Reduction axes are required for computing the scale and (zero point) vectors."
        raise ValueError(msg)
    wv, axis, _ = _resolve_groups(weight, config, reduction_axes, precomputed_scale)
    dev = weight.device
    if axis is None:  # general broadcast of precomputed parameters: one group per element
        outer, r, inner, gs = wv.numel(), 1, 1, 1
        pshape = list(wv.shape)
    else:
        outer, r, inner = _layout(wv.shape, axis)
        gs = -1
        pshape = _param_shape(wv.shape, axis)
    scale_in = _per_group(precomputed_scale, wv, axis, torch.float32)
    zp_in, zp_dtype = None, _lib.ZP_NONE
    if precomputed_zero_point is not None:
        if precomputed_zero_point.dtype == torch.int32:
            zp_in, zp_dtype = _per_group(precomputed_zero_point, wv, axis), _lib.ZP_I32
        else:
            zp_in, zp_dtype = _per_group(precomputed_zero_point, wv, axis, torch.float32), _lib.ZP_F32
    compute_params = scale_in is None
    if want_packed and (inner != 1 or r % 2):
        want_codes = True  # the C ABI packs such layouts from the byte codes (include/nncf_amd_wc.h)
    codes = torch.empty(wv.shape, dtype=torch.uint8 if config.is_asym_mode else torch.int8, device=dev) \
        if want_codes else None
    packed = torch.empty((wv.numel() // 2, 1), dtype=torch.uint8, device=dev) if want_packed else None
    dq = torch.empty(wv.shape, dtype=torch.float32, device=dev) if want_dq else None
    scale_out = torch.empty(pshape, dtype=torch.float32, device=dev) if compute_params else None
    zp_out = torch.empty(pshape, dtype=torch.int32, device=dev) if compute_params and config.is_asym_mode else None
    if wv.numel() > 0:
        with torch.cuda.device(dev):
            st = _lib.lib().nncf_amd_int_quantize(
                _lib.ptr(wv), _lib.dtype_id(wv), outer, r, inner, gs, _lib.MODE_ID[config.mode.value],
                _lib.ptr(scale_in), _lib.ptr(zp_in), zp_dtype, _lib.ptr(codes), _lib.ptr(packed),
                _lib.ptr(scale_out), _lib.ptr(zp_out), _lib.ptr(dq), _lib.stream_of(wv))
        _lib.check(st)
    # returned parameters are the given ones verbatim (weight_lowering.py:502-505)
    scale = precomputed_scale if precomputed_scale is not None else scale_out
    zero_point = precomputed_zero_point if precomputed_zero_point is not None else zp_out
    return codes, packed, dq, scale, zero_point


def calculate_integer_quantization_params(weight: torch.Tensor, reduction_axes: ReductionAxes,
                                          config: WeightCompressionConfig):
    """weight_lowering.py:292-326 -> (scale fp32 keepdims, zero_point int32 | None)."""
    config = as_config(config)
    cfg = WeightCompressionConfig(config.mode, -1)
    _, _, _, scale, zp = _int_quantize(weight, cfg, reduction_axes, None, None, want_codes=False)
    return scale, zp


def do_integer_quantization(weight: torch.Tensor, config: WeightCompressionConfig,
                            reduction_axes: Optional[ReductionAxes] = None,
                            precomputed_scale: Optional[torch.Tensor] = None,
                            precomputed_zero_point: Optional[torch.Tensor] = None) -> CompressedWeight:
    """weight_lowering.py:451-508."""
    config = as_config(config)
    codes, _, _, scale, zp = _int_quantize(weight, config, reduction_axes, precomputed_scale, precomputed_zero_point)
    return CompressedWeight(codes, scale, zp)


def do_integer_dequantization(compressed_weights: CompressedWeight, reduction_axis: int = -1) -> torch.Tensor:
    """weight_lowering.py:428-448."""
    codes, scale, zp = compressed_weights.tensor, compressed_weights.scale, compressed_weights.zero_point
    _lib.require_gpu(codes, "do_integer_dequantization")
    codes = codes.contiguous()
    ps, ws = tuple(scale.shape), tuple(codes.shape)
    diff = [i for i in range(len(ws)) if len(ps) == len(ws) and ps[i] != ws[i]]
    if len(ps) == len(ws) and len(diff) == 1 and ps[diff[0]] == 1:
        outer, r, inner = _layout(ws, diff[0])
        sc = scale.to(torch.float32).contiguous()
        z = zp.contiguous() if zp is not None else None
    else:
        outer, r, inner = codes.numel(), 1, 1
        sc = scale.to(torch.float32).expand(ws).contiguous()
        z = zp.expand(ws).contiguous() if zp is not None else None
    zp_dtype = _lib.ZP_NONE
    if z is not None:
        if z.dtype == torch.int32:
            zp_dtype = _lib.ZP_I32
        else:
            z, zp_dtype = z.to(torch.float32), _lib.ZP_F32
    out = torch.empty(ws, dtype=torch.float32, device=codes.device)
    if codes.numel():
        with torch.cuda.device(codes.device):
            st = _lib.lib().nncf_amd_int_dequantize(_lib.ptr(codes), int(codes.dtype == torch.int8), outer, r, inner, -1,
                                                    _lib.ptr(sc), _lib.ptr(z), zp_dtype, _lib.ptr(out),
                                                    _lib.stream_of(codes))
        _lib.check(st)
    if reduction_axis > -1:
        out = ungroup_weights(out, reduction_axis)
    return out


def integer_quantize_dequantize_weight(weight: torch.Tensor, config: WeightCompressionConfig,
                                       reduction_axes: Optional[ReductionAxes] = None,
                                       precomputed_scale: Optional[torch.Tensor] = None,
                                       precomputed_zero_point: Optional[torch.Tensor] = None,
                                       return_compressed_weight: Optional[bool] = False):
    """weight_lowering.py:511-554 -- one fused launch instead of quantize + dequantize."""
    config = as_config(config)
    codes, _, dq, scale, zp = _int_quantize(weight, config, reduction_axes, precomputed_scale, precomputed_zero_point,
                                            want_codes=bool(return_compressed_weight), want_dq=True)
    if return_compressed_weight:
        return dq, CompressedWeight(codes, scale, zp)
    return dq


def _queue_quantization_error(weight: torch.Tensor, reduction_axes: ReductionAxes, config: WeightCompressionConfig,
                              reduction: str, res: torch.Tensor):
    """Queues the error metric of one weight into the 1-element device tensor `res` (no synchronisation).  Returns
    what must stay alive until the stream has run."""
    if reduction not in ["max_mean", "frobenius"]:
        exception_str = f"Unsupported aggregation mode: {reduction}."
        raise InternalError(exception_str)
    config = as_config(config)
    weight = _prep_weight(weight, "get_integer_quantization_error")
    axis = _single_axis(weight.ndim, reduction_axes)
    outer, r, inner = _layout(weight.shape, axis)
    if config.group_size != -1 and r % config.group_size != 0:
        msg = f"Channel size {r} should be divisible by size of group {config.group_size}."
        raise InvalidGroupSizeError(msg)
    L = _lib.lib()
    ws_bytes = L.nncf_amd_int_quant_error_workspace(outer, inner)
    ws = torch.empty(max(ws_bytes, 4), dtype=torch.uint8, device=weight.device)
    with torch.cuda.device(weight.device):
        st = L.nncf_amd_int_quant_error(_lib.ptr(weight), _lib.dtype_id(weight), outer, r, inner, config.group_size,
                                        _lib.MODE_ID[config.mode.value], 0 if reduction == "max_mean" else 1,
                                        _lib.ptr(res), _lib.ptr(ws), ws_bytes, _lib.stream_of(weight))
    _lib.check(st)
    return weight, ws


def get_integer_quantization_error(weight: torch.Tensor, reduction_axes: ReductionAxes,
                                   config: WeightCompressionConfig, reduction: str) -> float:
    """weight_lowering.py:329-374."""
    res = torch.empty(1, dtype=torch.float32, device=weight.device)
    keep = _queue_quantization_error(weight, reduction_axes, config, reduction, res)  # noqa: F841
    return res.item()


def get_integer_quantization_errors(weights, reduction_axes_list, config: WeightCompressionConfig,
                                    reduction: str):
    """The error metric of MANY weights (the mixed-precision scoring loop, mixed_precision.py:189-199) with every
    launch queued back to back and ONE device synchronisation at the end, instead of one `.item()` per weight."""
    import ctypes

    weights = list(weights)
    if not weights:
        return []
    if reduction not in ["max_mean", "frobenius"]:
        exception_str = f"Unsupported aggregation mode: {reduction}."
        raise InternalError(exception_str)
    config = as_config(config)
    axes_list = list(reduction_axes_list)
    # 2-D weights of one dtype reduced along their last axis (every nn.Linear): ONE pair of launches per 48 weights
    # (nncf_amd_int_quant_error_batched) instead of a pair per weight
    plain = all(isinstance(w, torch.Tensor) and w.is_cuda and w.ndim == 2 and w.numel() > 0 and w.dtype == weights[0].dtype
                and w.device == weights[0].device and _single_axis(2, ax) == 1 for w, ax in zip(weights, axes_list))
    if plain:
        ws_list = [_prep_weight(w, "get_integer_quantization_errors") for w in weights]
        n, dev = len(ws_list), ws_list[0].device
        for w in ws_list:
            if config.group_size != -1 and w.shape[1] % config.group_size != 0:
                msg = f"Channel size {w.shape[1]} should be divisible by size of group {config.group_size}."
                raise InvalidGroupSizeError(msg)
        L = _lib.lib()
        c_w = (ctypes.c_void_p * n)(*[w.data_ptr() for w in ws_list])
        c_rows = (ctypes.c_int64 * n)(*[w.shape[0] for w in ws_list])
        c_cols = (ctypes.c_int64 * n)(*[w.shape[1] for w in ws_list])
        nbytes = L.nncf_amd_int_quant_error_batched_workspace(n, c_rows)
        work = torch.empty(max(nbytes // 4, 1), dtype=torch.float32, device=dev)
        res = torch.empty(n, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(L.nncf_amd_int_quant_error_batched(n, c_w, _lib.dtype_id(ws_list[0]), c_rows, c_cols,
                                                          config.group_size, _lib.MODE_ID[config.mode.value],
                                                          0 if reduction == "max_mean" else 1, _lib.ptr(res),
                                                          _lib.ptr(work), nbytes, _lib.stream_of(ws_list[0])))
        return res.tolist()
    res = torch.empty(len(weights), dtype=torch.float32, device=weights[0].device)
    keep = [_queue_quantization_error(w, axes, config, reduction, res[i:i + 1])
            for i, (w, axes) in enumerate(zip(weights, axes_list))]
    out = res.tolist()
    del keep
    return out


# ------------------------------------------------------------------ packing (torch/quantization/quantize_functions.py:417-473)
def _pack(tensor: torch.Tensor, expect, fn_name: str) -> torch.Tensor:
    if tensor.dtype != expect:
        msg = f"Invalid tensor dtype {tensor.type}. {expect} type is supported."
        raise ValidationError(msg)
    _lib.require_gpu(tensor, fn_name)
    t = tensor.contiguous()
    out = torch.empty((t.numel() // 2, 1), dtype=torch.uint8, device=t.device)
    with torch.cuda.device(t.device):
        st = getattr(_lib.lib(), fn_name)(_lib.ptr(t), t.numel(), _lib.ptr(out), _lib.stream_of(t))
    _lib.check(st)
    return out


def pack_uint4(tensor: torch.Tensor) -> torch.Tensor:
    return _pack(tensor, torch.uint8, "nncf_amd_pack_u4")


def pack_int4(tensor: torch.Tensor) -> torch.Tensor:
    return _pack(tensor, torch.int8, "nncf_amd_pack_i4")


def _unpack(packed: torch.Tensor, out_dtype, fn_name: str) -> torch.Tensor:
    _lib.require_gpu(packed, fn_name)
    p = packed.contiguous()
    out = torch.empty((*p.shape, 2), dtype=out_dtype, device=p.device)
    with torch.cuda.device(p.device):
        st = getattr(_lib.lib(), fn_name)(_lib.ptr(p), p.numel(), _lib.ptr(out), _lib.stream_of(p))
    _lib.check(st)
    return out


def unpack_uint4(packed_tensor: torch.Tensor) -> torch.Tensor:
    return _unpack(packed_tensor, torch.uint8, "nncf_amd_unpack_u4")


def unpack_int4(packed_tensor: torch.Tensor) -> torch.Tensor:
    return _unpack(packed_tensor, torch.int8, "nncf_amd_unpack_i4")


# ------------------------------------------------------------------ float (NF4 / MXFP4 / FP4) path
def _float_quantize(weight: torch.Tensor, config: WeightCompressionConfig, reduction_axes, precomputed_scale,
                    want_values=True, want_codes=False, want_packed=False, want_dq=False, want_e8m0=False):
    assert not config.is_integer
    weight = _prep_weight(weight, "do_float_quantization")
    if precomputed_scale is None and reduction_axes is None:
        msg = "Reduction axes are required for computing the scale."
        raise ValueError(msg)
    wv, axis, _ = _resolve_groups(weight, config, reduction_axes, precomputed_scale)
    dev = weight.device
    if axis is None:
        outer, r, inner = wv.numel(), 1, 1
        pshape = list(wv.shape)
    else:
        outer, r, inner = _layout(wv.shape, axis)
        pshape = _param_shape(wv.shape, axis)
    scale_in = _per_group(precomputed_scale, wv, axis, torch.float32)
    if want_packed and (inner != 1 or r % 2):
        want_codes = True
    values = torch.empty(wv.shape, dtype=torch.float32, device=dev) if want_values else None
    codes = torch.empty(wv.shape, dtype=torch.uint8, device=dev) if want_codes else None
    packed = torch.empty((wv.numel() // 2, 1), dtype=torch.uint8, device=dev) if want_packed else None
    dq = torch.empty(wv.shape, dtype=torch.float32, device=dev) if want_dq else None
    scale_out = torch.empty(pshape, dtype=torch.float32, device=dev) if scale_in is None else None
    e8m0 = torch.empty(pshape, dtype=torch.uint8, device=dev) if want_e8m0 else None
    if wv.numel() > 0:
        with torch.cuda.device(dev):
            st = _lib.lib().nncf_amd_float_quantize(
                _lib.ptr(wv), _lib.dtype_id(wv), outer, r, inner, -1, _lib.MODE_ID[config.mode.value],
                _lib.ptr(scale_in), _lib.ptr(values), _lib.ptr(codes), _lib.ptr(packed), _lib.ptr(scale_out),
                _lib.ptr(e8m0), _lib.ptr(dq), _lib.stream_of(wv))
        _lib.check(st)
    scale = precomputed_scale if precomputed_scale is not None else scale_out
    return values, codes, packed, dq, scale, e8m0


def calculate_float_quantization_params(weight: torch.Tensor, reduction_axes: ReductionAxes,
                                        config: WeightCompressionConfig, signed: bool = False) -> torch.Tensor:
    """weight_lowering.py:84-128 (unsigned variant, the one compress_weight uses)."""
    if signed:
        raise InternalError("signed float scale is not on the nncf_amd path")
    config = as_config(config)
    cfg = WeightCompressionConfig(config.mode, -1)
    return _float_quantize(weight, cfg, reduction_axes, None, want_values=False)[4]


def do_float_quantization(weight: torch.Tensor, config: WeightCompressionConfig,
                          reduction_axes: Optional[ReductionAxes] = None,
                          precomputed_scale: Optional[torch.Tensor] = None) -> CompressedWeight:
    """weight_lowering.py:151-249 (single-scale modes): tensor = fp32 quantile values like the reference."""
    config = as_config(config)
    values, _, _, _, scale, _ = _float_quantize(weight, config, reduction_axes, precomputed_scale)
    return CompressedWeight(values, scale)


def do_float_dequantization(compressed_weight: CompressedWeight, reduction_axis: int = -1) -> torch.Tensor:
    """weight_lowering.py:131-148: one broadcast multiply (torch), no kernel needed."""
    scale = compressed_weight.scale
    if compressed_weight.global_scale is not None:
        scale = scale * compressed_weight.global_scale
    out = compressed_weight.get_unscaled_tensor() * scale
    if reduction_axis != -1:
        out = ungroup_weights(out, reduction_axis)
    return out


def float_quantize_dequantize_weight(weight: torch.Tensor, config: WeightCompressionConfig,
                                     reduction_axes: Optional[ReductionAxes] = None,
                                     precomputed_scale: Optional[torch.Tensor] = None,
                                     return_compressed_weight: Optional[bool] = False):
    """weight_lowering.py:252-289."""
    config = as_config(config)
    values, _, _, dq, scale, _ = _float_quantize(weight, config, reduction_axes, precomputed_scale,
                                                 want_values=bool(return_compressed_weight), want_dq=True)
    if return_compressed_weight:
        return dq, CompressedWeight(values, scale)
    return dq


# ------------------------------------------------------------------ top level
def compress_weight(weight: torch.Tensor, reduction_axes: ReductionAxes, config: WeightCompressionConfig,
                    precomputed_compressed_weight: Optional[CompressedWeight] = None) -> CompressedWeight:
    """weight_lowering.py:377-407."""
    config = as_config(config)
    scale, zp = ((precomputed_compressed_weight.scale, precomputed_compressed_weight.zero_point)
                 if precomputed_compressed_weight else (None, None))
    if not config.is_integer:
        return do_float_quantization(weight, config, reduction_axes, scale)
    return do_integer_quantization(weight, config, reduction_axes, scale, zp)


def compress_weight_packed(weight: torch.Tensor, reduction_axes: ReductionAxes, config: WeightCompressionConfig,
                           precomputed_compressed_weight: Optional[CompressedWeight] = None) -> CompressedWeight:
    """compress_weight + decompressor.pack_weight in ONE pass (torch_backend.py:463-468 + layers.py:1400-1451):
    tensor = nibble-packed uint8 [N/2, 1] for 4-bit modes (int8/uint8 codes for 8-bit), scale fp32, zero point int32.
    This is the 2.53 B/element path the throughput metric is quoted on."""
    config = as_config(config)
    scale, zp = ((precomputed_compressed_weight.scale, precomputed_compressed_weight.zero_point)
                 if precomputed_compressed_weight else (None, None))
    if config.is_integer:
        four = config.num_bits == 4
        codes, packed, _, scale, zp = _int_quantize(weight, config, reduction_axes, scale, zp, want_codes=not four,
                                                    want_packed=four)
        return CompressedWeight(packed if four else codes, scale, zp)
    _, _, packed, _, scale, _ = _float_quantize(weight, config, reduction_axes, scale, want_values=False,
                                                want_packed=True)
    return CompressedWeight(packed, scale)


def compress_weights_packed_batched(weights, config: WeightCompressionConfig, want_e8m0: bool = False):
    """Data-free compression of MANY 2-D [out, in] weights (reduction along the last axis) in as few launches as
    possible: the lowering part of PTWeightCompressionAlgoBackend.transform_model's per-weight loop
    (torch_backend.py:442-477) + pack_weight.  Returns a list of CompressedWeight with
    tensor = nibble-packed uint8 [N/2, 1] (4-bit modes) or int8/uint8 codes (8-bit), scale fp32 [out, G, 1],
    zero_point int32 (asymmetric) and, for MXFP4 with want_e8m0, codebook = uint8 E8M0 scale bytes."""
    import ctypes

    config = as_config(config)
    if not weights:
        return []
    ws = [_prep_weight(w, "compress_weights_packed_batched") for w in weights]
    dev, dt = ws[0].device, ws[0].dtype
    # 3-D / MoE weights [experts, out, in] are lowered as the [experts * out, in] matrix they are in memory (groups
    # run along the last axis); the parameters get the expert axis back below
    lead = [tuple(w.shape[:-1]) for w in ws]
    ws = [w.reshape(-1, w.shape[-1]) if w.ndim == 3 else w for w in ws]
    for w in ws:
        if w.ndim != 2 or w.device != dev or w.dtype != dt:
            raise InternalError("compress_weights_packed_batched: 2-D / 3-D weights of one dtype on one device expected")
        if config.group_size != -1 and w.shape[1] % config.group_size != 0:
            msg = f"Channel size {w.shape[1]} should be divisible by size of group {config.group_size}."
            raise InvalidGroupSizeError(msg)
    n = len(ws)
    four = config.num_bits == 4
    asym = config.is_integer and config.is_asym_mode
    if four and any((config.group_size if config.group_size != -1 else w.shape[1]) % 2 for w in ws) and not want_e8m0:
        # groups of an odd number of 4-bit codes do not end on a byte: the single-tensor entry packs them from the
        # byte codes (group size 1 is what the reference's mixed-precision tests use)
        return [compress_weight_packed(w, (w.ndim - 1,), config) for w in weights]
    outs, scales, auxs = [], [], []
    for w in ws:
        o, i = w.shape
        g = config.group_size if config.group_size != -1 else i
        pshape = (o, i // g, 1) if config.group_size != -1 else (o, 1)
        gshape = (o, i // g, g) if config.group_size != -1 else (o, i)
        if four:
            outs.append(torch.empty((o * i // 2, 1), dtype=torch.uint8, device=dev))
        else:
            outs.append(torch.empty(gshape, dtype=torch.uint8 if asym else torch.int8, device=dev))
        scales.append(torch.empty(pshape, dtype=torch.float32, device=dev))
        if asym:
            auxs.append(torch.empty(pshape, dtype=torch.int32, device=dev))
        elif want_e8m0 and config.mode.value == "mxfp4":
            auxs.append(torch.empty(pshape, dtype=torch.uint8, device=dev))
        else:
            auxs.append(None)
    VP, I64 = ctypes.c_void_p * n, ctypes.c_int64 * n
    c_w = VP(*[w.data_ptr() for w in ws])
    c_rows, c_cols = I64(*[w.shape[0] for w in ws]), I64(*[w.shape[1] for w in ws])
    c_out, c_scale = VP(*[t.data_ptr() for t in outs]), VP(*[t.data_ptr() for t in scales])
    c_aux = VP(*[t.data_ptr() if t is not None else None for t in auxs]) if any(a is not None for a in auxs) else None
    L = _lib.lib()
    with torch.cuda.device(dev):
        stream = _lib.stream_of(ws[0])
        if config.is_integer:
            st = L.nncf_amd_int_quantize_batched(n, c_w, _lib.dtype_id(ws[0]), c_rows, c_cols, config.group_size,
                                                 _lib.MODE_ID[config.mode.value], None if four else c_out,
                                                 c_out if four else None, c_scale, c_aux, stream)
        else:
            st = L.nncf_amd_float_quantize_batched(n, c_w, _lib.dtype_id(ws[0]), c_rows, c_cols, config.group_size,
                                                   _lib.MODE_ID[config.mode.value], c_out, c_scale, c_aux, stream)
    _lib.check(st)
    res = []
    for o, s, a, ld in zip(outs, scales, auxs, lead):
        if len(ld) == 2:  # [experts, out]: parameters [experts, out, G, 1] (or [experts, out, 1]); 8-bit codes likewise
            s = s.reshape(ld + tuple(s.shape[1:]))
            a = a.reshape(ld + tuple(a.shape[1:])) if a is not None else None
            if not four:
                o = o.reshape(ld + tuple(o.shape[1:]))
        if asym:
            res.append(CompressedWeight(o, s, a))
        else:
            res.append(CompressedWeight(o, s, None, codebook=a))
    return res
