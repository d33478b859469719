"""
oracle/wc_oracle.py -- TEST INFRASTRUCTURE ONLY.

A numpy (CPU) restatement of the tensor-level math on NNCF's weight-compression
lowering path.  It is the checker that the HIP kernels in ``nncf_amd/csrc`` are
compared against; it is NOT a product path and nothing under ``nncf_amd/``
imports it.  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may import this module.

Parity pinning: every function here is checked against (a) the reference's own
known-answer test vectors restated in ``tests/test_oracle_known_answers.py`` and
(b) golden fixtures under ``tests/golden/*.npz`` which ``tools/gen_golden.py``
produced by importing the reference (nncf 3.2.0) in the build container.

Each function cites the reference file:line it follows (paths are relative to
the reference checkout root).  All arithmetic is float32 unless stated.
"""
from __future__ import annotations

import math

import numpy as np

F32 = np.float32
EPS = np.float32(np.finfo(np.float32).eps)  # 2**-23, fns.finfo(float32).eps

# --- tables: src/nncf/quantization/algorithms/weight_compression/constants.py:17-113 (data, bit exact) ---
NF4_QUANTILES = np.array(
    [-1.0, -0.6961928009986877, -0.5250730514526367, -0.39491748809814453, -0.28444138169288635,
     -0.18477343022823334, -0.09105003625154495, 0.0, 0.07958029955625534, 0.16093020141124725,
     0.24611230194568634, 0.33791524171829224, 0.44070982933044434, 0.5626170039176941,
     0.7229568362236023, 1.0], dtype=np.float32)
F4E2M1_QUANTILES = np.array(
    [-6.0, -4.0, -3.0, -2.0, -1.5, -1.0, -0.5, -0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0], dtype=np.float32)
CENTER_OF_NF4_QUANTILES = np.array(
    [-0.84809643, -0.6106329, -0.45999527, -0.33967942, -0.2346074, -0.13791174, -0.045525018,
     0.03979015, 0.120255254, 0.20352125, 0.29201376, 0.38931254, 0.5016634, 0.6427869, 0.8614784],
    dtype=np.float32)
CENTER_OF_F4E2M1_QUANTILES = (F4E2M1_QUANTILES[1:] + F4E2M1_QUANTILES[:-1]) / 2
FP_MAX = {"nf4": 1.0, "mxfp4": 6.0, "fp4": 6.0}

INT_MODES = {
    # mode: (num_bits, is_asym)   config.py:48-75
    "int8_asym": (8, True),
    "int8_sym": (8, False),
    "int4_asym": (4, True),
    "int4_sym": (4, False),
}
FLOAT_MODES = ("nf4", "mxfp4", "fp4")


class InvalidGroupSizeError(ValueError):
    """weight_lowering.py:72-74 raises nncf.errors.InvalidGroupSizeError."""


def is_integer(mode: str) -> bool:
    return mode in INT_MODES


def _f32(x):
    x = np.asarray(x)
    return x if x.dtype == np.float32 else x.astype(np.float32)


def x86_cvtt_int32(x):
    """float32 -> int32 the way numpy/torch-CPU do it on x86 (cvttss2si): truncation,
    and |x| >= 2**31 or NaN gives INT_MIN.  Used by fake_quantize.py:349 via astype(int32)."""
    x = _f32(x)
    bad = ~(np.abs(x) < np.float32(2147483648.0))
    safe = np.where(bad, np.float32(0), x)
    out = np.trunc(safe).astype(np.int64)
    return np.where(bad, np.int64(-2147483648), out).astype(np.int32)


# ------------------------------------------------------------------ grouping
def reshape_weight_for_grouped_quantization(weight, reduction_axes, group_size):
    """weight_lowering.py:57-81 (+ get_reduction_channel_size :40-54)."""
    assert group_size != -1
    if isinstance(reduction_axes, tuple) and len(reduction_axes) == 1:
        reduction_axes = reduction_axes[0]
    if not isinstance(reduction_axes, int):
        raise ValueError(f"Group-wise quantization expects a single reduction axis, but given: {reduction_axes}.")
    channel_size = weight.shape[reduction_axes]
    if channel_size % group_size != 0:
        raise InvalidGroupSizeError(
            f"Channel size {channel_size} should be divisible by size of group {group_size}.")
    shape = list(weight.shape)
    shape[reduction_axes:reduction_axes + 1] = (channel_size // group_size, group_size)
    return weight.reshape(shape), reduction_axes + 1


def ungroup_weights(weights, reduction_axis):
    """weight_lowering.py:410-425 (note the squeeze of ALL unit dims)."""
    shape = list(weights.shape)
    shape[reduction_axis] = shape[reduction_axis] * shape[reduction_axis + 1]
    shape[reduction_axis + 1] = 1
    return np.squeeze(weights.reshape(shape))


# ------------------------------------------------------------------ integer
def calculate_scale_zero_point(input_low, input_high, level_low, level_high):
    """fake_quantize.py:322-350 with narrow_range=False."""
    levels = level_high - level_low + 1
    scale = ((input_high - input_low) / F32(levels - 1)).astype(np.float32)
    scale = np.where(np.abs(scale) < EPS, EPS, scale)
    zero_point = F32(level_low) - np.round(input_low / scale)
    zero_point = np.clip(x86_cvtt_int32(zero_point), level_low, level_high)
    return scale, zero_point.astype(np.int32)


def calculate_signed_scale(weight, reduction_axes, num_bits=4):
    """weight_lowering.py:629-649."""
    factor = F32(2 ** (num_bits - 1))
    w_abs_min = np.abs(np.min(weight, axis=reduction_axes, keepdims=True))
    w_max = np.max(weight, axis=reduction_axes, keepdims=True)
    scale = np.where(w_abs_min >= w_max, w_abs_min, -w_max)
    scale = scale / factor
    scale = np.where(np.abs(scale) < EPS, EPS, scale)
    return scale.astype(np.float32)


def calculate_integer_quantization_params(weight, reduction_axes, mode):
    """weight_lowering.py:292-326."""
    num_bits, asym = INT_MODES[mode]
    weight = _f32(weight)
    if asym:
        lo = np.min(weight, axis=reduction_axes, keepdims=True)
        hi = np.max(weight, axis=reduction_axes, keepdims=True)
        return calculate_scale_zero_point(lo, hi, 0, 2 ** num_bits - 1)
    return calculate_signed_scale(weight, reduction_axes, num_bits), None


def calculate_integer_quantized_weight(weight, mode, scale, zero_point=None):
    """weight_lowering.py:652-684: round(w/scale [+ zp]) -> clip -> int8/uint8."""
    num_bits, asym = INT_MODES[mode]
    weight = _f32(weight)
    scale = _f32(scale)
    dtype = np.uint8 if asym else np.int8
    level_low = 0 if asym else -(2 ** (num_bits - 1))
    level_high = 2 ** num_bits - 1 if asym else 2 ** (num_bits - 1) - 1
    cw = weight / scale
    if zero_point is not None:
        cw = cw + np.asarray(zero_point).astype(np.float32)
    cw = np.round(cw)
    return np.clip(cw, level_low, level_high).astype(dtype)


def do_integer_quantization(weight, mode, group_size=-1, reduction_axes=None,
                            precomputed_scale=None, precomputed_zero_point=None):
    """weight_lowering.py:451-508. Returns (codes, scale, zero_point)."""
    if mode not in INT_MODES:
        raise RuntimeError("The function supports integer quantization only")
    _, asym = INT_MODES[mode]
    if asym and (precomputed_scale is None) != (precomputed_zero_point is None):
        raise ValueError(
            "If precomputed quantization parameters are provided, both scale and zero point are required "
            "for asymmetric quantization.")
    if group_size != -1 and reduction_axes is not None:
        weight, reduction_axes = reshape_weight_for_grouped_quantization(weight, reduction_axes, group_size)
    weight = _f32(weight)
    scale, zero_point = None, None
    if precomputed_scale is None or (asym and precomputed_zero_point is None):
        if reduction_axes is None:
            raise ValueError("Reduction axes are required for computing the scale and (zero point) vectors.")
        scale, zero_point = calculate_integer_quantization_params(weight, reduction_axes, mode)
    if precomputed_scale is not None:
        scale = precomputed_scale
    if precomputed_zero_point is not None:
        zero_point = precomputed_zero_point
    codes = calculate_integer_quantized_weight(weight, mode, scale, zero_point)
    return codes, scale, zero_point


def do_integer_dequantization(codes, scale, zero_point=None, reduction_axis=-1):
    """weight_lowering.py:428-448."""
    dq = codes.astype(np.int32) - zero_point if zero_point is not None else codes
    dq = np.asarray(dq).astype(scale.dtype) * scale
    if reduction_axis > -1:
        dq = ungroup_weights(dq, reduction_axis)
    return dq


def integer_quantize_dequantize_weight(weight, mode, group_size=-1, reduction_axes=None,
                                       precomputed_scale=None, precomputed_zero_point=None,
                                       return_compressed_weight=False):
    """weight_lowering.py:511-554."""
    codes, scale, zp = do_integer_quantization(weight, mode, group_size, reduction_axes,
                                               precomputed_scale, precomputed_zero_point)
    dq = do_integer_dequantization(codes, scale, zp)
    if return_compressed_weight:
        return dq, (codes, scale, zp)
    return dq


def get_integer_quantization_error(weight, reduction_axes, mode, group_size, reduction):
    """weight_lowering.py:329-374."""
    if reduction not in ("max_mean", "frobenius"):
        raise RuntimeError(f"Unsupported aggregation mode: {reduction}.")
    weight = _f32(weight)
    dq = integer_quantize_dequantize_weight(weight, mode, group_size, reduction_axes)
    dq = dq.reshape(weight.shape)
    if reduction == "max_mean":
        diff = (dq - weight) ** 2
        val = np.max(np.mean(diff, axis=reduction_axes))
    else:
        val = np.linalg.norm((dq - weight).astype(np.float32).ravel())
    return float(val)


# ------------------------------------------------------------------ packing
def pack_uint4(t):
    """torch/quantization/quantize_functions.py:417-433: flattened pairs, out [N/2, 1]."""
    if t.dtype != np.uint8:
        raise TypeError(f"Invalid tensor dtype {t.dtype}. torch.uint8 type is supported.")
    p = np.ascontiguousarray(t).reshape(-1, 2)
    return ((p[:, 0::2] & 15) | (p[:, 1::2] << 4)).astype(np.uint8)


def pack_int4(t):
    """quantize_functions.py:447-461: +8 bias then pack_uint4."""
    if t.dtype != np.int8:
        raise TypeError(f"Invalid tensor dtype {t.dtype}. torch.int8 type is supported.")
    return pack_uint4((t + np.int8(8)).astype(np.uint8))


def unpack_uint4(p):
    """quantize_functions.py:436-444."""
    return np.stack((p & 15, p >> 4), axis=-1)


def unpack_int4(p):
    """quantize_functions.py:464-473."""
    return unpack_uint4(p).astype(np.int8) - np.int8(8)


# ------------------------------------------------------------------ float (NF4 / MXFP4 / FP4)
def log2_f32_correctly_rounded(x):
    """fp32 log2 contract: correctly-rounded float32 log2 (via float64)."""
    return np.log2(np.asarray(x, dtype=np.float64)).astype(np.float32)


def calculate_float_quantization_params(weight, reduction_axes, mode, signed=False):
    """weight_lowering.py:84-128."""
    assert mode in FLOAT_MODES
    weight = _f32(weight)
    if signed:
        sneg = np.min(weight, axis=reduction_axes, keepdims=True)
        spos = np.max(weight, axis=reduction_axes, keepdims=True)
        scale = np.where(np.abs(sneg) >= np.abs(spos), sneg, spos)
    else:
        scale = np.max(np.abs(weight), axis=reduction_axes, keepdims=True)
    if mode != "nf4":
        scale = scale / F32(FP_MAX[mode])
    scale = np.where(np.abs(scale) < EPS, EPS, scale).astype(np.float32)
    if mode == "mxfp4":
        e = np.clip(np.ceil(log2_f32_correctly_rounded(scale)), -127, 127)
        scale = np.ldexp(np.float32(1.0), e.astype(np.int32)).astype(np.float32)
    return scale


def calculate_float_quantized_weight(norm_weight, mode):
    """weight_lowering.py:557-586: returns fp32 VALUES of the chosen quantiles (not codes)."""
    idx = float_quantile_indexes(norm_weight, mode)
    q = NF4_QUANTILES if mode == "nf4" else F4E2M1_QUANTILES
    return q[idx]


def float_quantile_indexes(norm_weight, mode):
    """Index into the quantile table chosen by weight_lowering.py:570-583."""
    is_nf4 = mode == "nf4"
    quantiles = NF4_QUANTILES if is_nf4 else F4E2M1_QUANTILES
    centers = CENTER_OF_NF4_QUANTILES if is_nf4 else CENTER_OF_F4E2M1_QUANTILES
    norm_weight = _f32(norm_weight)
    idx = np.searchsorted(centers, norm_weight)  # side='left'
    if not is_nf4:
        sh = np.clip(idx + 1, 0, quantiles.size - 1)
        dl = np.abs(norm_weight - quantiles[idx])
        dr = np.abs(norm_weight - quantiles[sh])
        right = (dr < dl) | ((dl == dr) & ((sh + 1) % 2 == 0))
        idx = np.where(right, sh, idx)
    return idx


def do_float_quantization(weight, mode, group_size=-1, reduction_axes=None, precomputed_scale=None):
    """weight_lowering.py:195-249 (single-scale path). Returns (values_fp32, scale)."""
    assert mode in FLOAT_MODES
    if group_size != -1 and reduction_axes is not None:
        weight, reduction_axes = reshape_weight_for_grouped_quantization(weight, reduction_axes, group_size)
    weight = _f32(weight)
    scale = precomputed_scale
    if scale is None:
        scale = calculate_float_quantization_params(weight, reduction_axes, mode)
    norm = weight / _f32(scale)  # weight_lowering.py:613-626
    return calculate_float_quantized_weight(norm, mode), scale


def do_float_dequantization(values, scale, reduction_axis=-1):
    """weight_lowering.py:131-148 (no global scale)."""
    dq = values * scale
    if reduction_axis != -1:
        dq = ungroup_weights(dq, reduction_axis)
    return dq


def float_quantize_dequantize_weight(weight, mode, group_size=-1, reduction_axes=None, precomputed_scale=None,
                                     return_compressed_weight=False):
    """weight_lowering.py:252-289."""
    values, scale = do_float_quantization(weight, mode, group_size, reduction_axes, precomputed_scale)
    dq = do_float_dequantization(values, scale)
    if return_compressed_weight:
        return dq, (values, scale)
    return dq


def compress_weight(weight, reduction_axes, mode, group_size=-1, scale=None, zero_point=None):
    """weight_lowering.py:377-407. Returns (tensor, scale, zero_point|None)."""
    if not is_integer(mode):
        v, s = do_float_quantization(weight, mode, group_size, reduction_axes, scale)
        return v, s, None
    return do_integer_quantization(weight, mode, group_size, reduction_axes, scale, zero_point)


# The build's definition of 4-bit codes for the float modes (SURVEY §8c "codes caveat"):
def nf4_code_from_index(idx):
    return idx.astype(np.uint8)


E2M1_CODE_OF_INDEX = np.array(
    # index in F4E2M1_QUANTILES -> OCP e2m1 bit pattern (sign<<3 | exp<<1 | man); the single zero -> +0
    [0b1111, 0b1110, 0b1101, 0b1100, 0b1011, 0b1010, 0b1001, 0b0000,
     0b0001, 0b0010, 0b0011, 0b0100, 0b0101, 0b0110, 0b0111], dtype=np.uint8)
E2M1_VALUE_OF_CODE = np.array(
    [0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0, -0.0, -0.5, -1.0, -1.5, -2.0, -3.0, -4.0, -6.0], dtype=np.float32)


# ------------------------------------------------------------------ reducers / statistics
def _bf16_round(x):
    """round-to-nearest-even float32 -> bfloat16 (kept in float32 storage)."""
    u = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)


def _f16_round(x):
    return np.asarray(x, dtype=np.float32).astype(np.float16).astype(np.float32)


def round_to_dtype(x, dtype: str):
    if dtype == "bf16":
        return _bf16_round(x)
    if dtype == "f16":
        return _f16_round(x)
    return np.asarray(x, dtype=np.float32)


def mean_reducer(x, axes, dtype="f32"):
    """collectors.py:512-516 -- mean in the input dtype; torch accumulates bf16/f16 in fp32 and rounds once."""
    return round_to_dtype(np.mean(_f32(x).astype(np.float64), axis=axes, keepdims=True), dtype)


def mean_variance_reducer(x, axes, dtype="f32"):
    """collectors.py:519-524."""
    v = round_to_dtype(np.var(_f32(x).astype(np.float64), axis=axes), dtype)
    return round_to_dtype(np.mean(v.astype(np.float64)), dtype)


def max_variance_reducer(x, axes, dtype="f32"):
    """collectors.py:527-532."""
    v = round_to_dtype(np.var(_f32(x).astype(np.float64), axis=axes), dtype)
    return np.max(v)


def mean_abs_max_reducer(x, axes, dtype="f32"):
    """collectors.py:535-540."""
    m = np.max(np.abs(_f32(x)), axis=axes, keepdims=True)
    return round_to_dtype(np.mean(m.astype(np.float64)), dtype)


def hawq_aggregate(xs):
    """collectors.py:924-936 (the running division is part of the reference behaviour)."""
    acc = 0.0
    for x in xs:
        x = _f32(x)
        acc = (acc + float(np.sum(x.astype(np.float64) * x.astype(np.float64)))) / x.size
    return acc * 2 / len(xs)


def process_stats(mean_values, shape_values, subset_size, act_ch_axis=-1):
    """activation_stats.py:20-57."""
    X = np.stack(mean_values)
    axes = list(range(1, X.ndim)) + [0]
    X_full = np.transpose(X, axes)
    if X_full.shape[-1] > subset_size and subset_size > 0:
        lens = [int(np.prod(tuple(s[:act_ch_axis]) + tuple(s[act_ch_axis % len(s) + 1:]))) for s in shape_values]
        step = X_full.shape[-1] // subset_size
        idxs = [i[0] for i in sorted(enumerate(lens), key=lambda x: -x[1])][::step]
        X = X_full[..., idxs]
    else:
        X = X_full
    s = np.max(np.abs(X_full), axis=-1)
    return s, X


def activations_to_wc_statistics(activations):
    """gptq.py:381-396 -> (mean_values, shapes)."""
    means, shapes = [], []
    for act in activations:
        shapes.append(tuple(act.shape))
        means.append(np.mean(act, axis=tuple(range(act.ndim - 1))))
    return means, shapes


# ------------------------------------------------------------------ scale estimation
def _qdq(weight, mode, group_size, reduction_axes=None, scale=None, zp=None, ret=False):
    if is_integer(mode):
        return integer_quantize_dequantize_weight(weight, mode, group_size, reduction_axes, scale, zp, ret)
    return float_quantize_dequantize_weight(weight, mode, group_size, reduction_axes, scale, ret)


def _quant(weight, mode, group_size, scale, zp):
    if is_integer(mode):
        return do_integer_quantization(weight, mode, group_size, None, scale, zp)[0]
    return do_float_quantization(weight, mode, group_size, None, scale)[0]


def get_target_zero_mask(compressed, zp=None):
    """scale_estimation.py:380-393."""
    target = compressed
    if zp is not None:
        target = target.astype(zp.dtype) - zp
    return target, np.isclose(target, 0)


def estimate_scales(weight, target, zero_mask, importance):
    """scale_estimation.py:396-409."""
    ideal = np.abs(weight) / (np.abs(target) + zero_mask)
    return np.sum(ideal * importance, axis=-1, keepdims=True)


def scale_estimation_params(mean_values, shape_values, weight, reduction_axes, mode, group_size,
                            act_ch_axis=-1, subset_size=32, initial_steps=5, scale_steps=10, weight_penalty=-1.0):
    """scale_estimation.py:160-377 (2-D and 3-D weights)."""
    reduction_axis = reduction_axes[0]
    s, X = process_stats(mean_values, shape_values, subset_size, act_ch_axis)
    X = _f32(X)
    weight = _f32(weight)
    is_3d = weight.ndim == 3
    was_transposed = False
    if reduction_axis == 0 or (reduction_axis == 1 and is_3d):
        weight = np.moveaxis(weight, -1, -2)
        reduction_axis = weight.ndim - 1
        was_transposed = True
    gs = group_size if group_size != -1 else weight.shape[reduction_axis]

    q_weights, (cw, scale, zp) = _qdq_full(weight, mode, gs, reduction_axis)
    if zp is not None:
        zp = zp.astype(scale.dtype)
    s = np.expand_dims(s, -2)
    s, _ = reshape_weight_for_grouped_quantization(s, reduction_axis, gs)
    weight, _ = reshape_weight_for_grouped_quantization(weight, reduction_axis, gs)
    importance = np.ones_like(weight) * s
    target, zero_mask = get_target_zero_mask(cw, zp)
    importance = np.where(zero_mask, F32(0.0), importance)
    denum = np.sum(importance, axis=-1, keepdims=True)
    importance = importance / (denum + EPS)

    X, _ = reshape_weight_for_grouped_quantization(X, -2, gs)
    fp_outs = np.matmul(np.moveaxis(weight, -2, -3), X)
    q_outs = np.matmul(np.moveaxis(q_weights, -2, -3), X)
    mm_diffs = np.moveaxis(np.mean((fp_outs - q_outs) ** 2, axis=-1), -1, -2)
    if weight_penalty > 0.0:
        mm_diffs = mm_diffs + F32(weight_penalty) * np.mean((q_weights - weight) ** 2, axis=-1)
    best_diffs = mm_diffs
    result_scale = scale
    scale_sign = scale / np.abs(scale)
    zero_scale = F32(0.001)
    zero_mask = zero_scale * zero_mask.astype(weight.dtype)

    def _diffs(qw):
        qo = np.matmul(np.moveaxis(qw, -2, -3), X)
        d = np.moveaxis(np.mean((fp_outs - qo) ** 2, axis=-1), -1, -2)
        if weight_penalty > 0.0:
            d = d + F32(weight_penalty) * np.mean((qw - weight) ** 2, axis=-1)
        return d

    for i in range(initial_steps):
        near = estimate_scales(weight, target, zero_mask, importance) * scale_sign
        qw_ = _qdq(weight, mode, group_size, None, near, zp)
        d = _diffs(qw_)
        mask = (d > best_diffs).astype(best_diffs.dtype)
        best_diffs = mask * best_diffs + (F32(1.0) - mask) * d
        mask = np.expand_dims(mask, -1)
        result_scale = mask * result_scale + (F32(1.0) - mask) * near
        if i < initial_steps - 1:
            cw = _quant(weight, mode, group_size, result_scale, zp)
            target, zero_mask = get_target_zero_mask(cw, zp)
            zero_mask = zero_scale * zero_mask.astype(weight.dtype)

    for step in range(scale_steps):
        factor = 1.0 - 0.05 * step
        scaled = (F32(factor) * scale).astype(np.float32)
        cw = _quant(weight, mode, group_size, scaled, zp)
        target, zero_mask = get_target_zero_mask(cw, zp)
        zero_mask = zero_scale * zero_mask.astype(weight.dtype)
        near = estimate_scales(weight, target, zero_mask, importance) * scale_sign
        qw_ = _qdq(weight, mode, group_size, None, near, zp)
        d = _diffs(qw_)
        mask = (d > best_diffs).astype(best_diffs.dtype)
        best_diffs = mask * best_diffs + (F32(1.0) - mask) * d
        mask = np.expand_dims(mask, -1)
        result_scale = mask * result_scale + (F32(1.0) - mask) * near

    if group_size == -1:
        result_scale = np.squeeze(result_scale, axis=-2)
        if zp is not None:
            zp = np.squeeze(zp, axis=-2)
    if was_transposed:
        if group_size == -1:
            result_scale = np.moveaxis(result_scale, -1, -2)
            if zp is not None:
                zp = np.moveaxis(zp, -1, -2)
        else:
            result_scale = np.moveaxis(result_scale, (-1, -2, -3), (-2, -3, -1))
            if zp is not None:
                zp = np.moveaxis(zp, (-1, -2, -3), (-2, -3, -1))
    return result_scale, zp


def _qdq_full(weight, mode, gs, reduction_axis):
    if is_integer(mode):
        dq, (c, s, z) = integer_quantize_dequantize_weight(weight, mode, gs, reduction_axis,
                                                           return_compressed_weight=True)
        return dq, (c, s, z)
    dq, (c, s) = float_quantize_dequantize_weight(weight, mode, gs, reduction_axis, return_compressed_weight=True)
    return dq, (c, s, None)


# ------------------------------------------------------------------ AWQ
def awq_data_free_scale(weight, axis):
    """awq.py:363-366."""
    weight = _f32(weight)
    return F32(1) / np.maximum(np.mean(np.abs(weight), axis=axis), EPS)


FP16_MAX_VALUE = 65504.0
FP16_OVERFLOW_MARGIN = 0.25


def awq_data_aware_scale(weight, reduction_axis, mode, group_size, mean_values, shape_values, act_ch_axis=-1,
                         subset_size=32, percent_to_apply=0.002, alpha_min=0.0, alpha_max=1.0, steps=100,
                         prev_weight=None, prev_mean_values=None, prev_shape_values=None):
    """awq.py:243-357, 2-D and 3-D (MoE) weights, incl. the clamp of the scale when the previous matmul's output
    (it absorbs 1/scale) would leave the fp16 range (awq.py:315-335, _clamp_scale :360-361)."""
    alpha_step = (alpha_max - alpha_min) / steps
    s, X = process_stats(mean_values, shape_values, subset_size, act_ch_axis)
    s = _f32(s)
    X = _f32(X)
    weight = _f32(weight)
    is_2d = weight.ndim == 2
    if is_2d:
        s, X, weight = s[None], X[None], weight[None]
        prev_weight = _f32(prev_weight)[None] if prev_weight is not None else None
        reduction_axis += 1
    prev_s = prev_w = None
    if prev_mean_values is not None and prev_weight is not None:
        ps, _ = process_stats(prev_mean_values, prev_shape_values, subset_size, act_ch_axis)
        prev_s = float(np.max(_f32(ps)))
        prev_w = np.mean(np.abs(_f32(prev_weight)), axis=reduction_axis)
    top_k = max(int(s.shape[-1] * percent_to_apply), 1)
    topk = np.argsort(-s, kind="stable")[:, :top_k]
    gs = group_size if group_size != -1 else s.shape[-1]
    groups = sorted({(b, int(topk[b, k]) // gs) for b in range(topk.shape[0]) for k in range(topk.shape[1])})
    if reduction_axis == 1:
        weight = np.moveaxis(weight, -1, -2)
        reduction_axis = weight.ndim - 1
    scale = np.ones_like(np.mean(X, axis=-1))
    for b, gi in groups:
        off = gi * gs
        gscale = s[b, off:off + gs]
        gweight = weight[b, :, off:off + gs]
        gacts = X[b, off:off + gs, :]
        a_min = np.quantile(gscale, 0.1).astype(np.float32)
        gscale = np.clip(gscale, a_min, F32(1e2))
        fp32_out = np.matmul(gweight, gacts)
        min_diff = np.max(np.abs(fp32_out))
        best = None
        alpha = alpha_min
        for _ in range(steps):
            cur = (gscale ** F32(alpha)).astype(np.float32)
            if prev_s is not None:
                threshold = FP16_OVERFLOW_MARGIN * FP16_MAX_VALUE
                n_prev = prev_weight.shape[reduction_axis]
                magnitudes = (prev_w[b, off:off + gs] / cur) * F32(prev_s) * F32(n_prev)
                if magnitudes.max() >= threshold:
                    clamped = prev_w[b, off:off + gs] * F32(prev_s) * F32(n_prev) / F32(threshold)
                    cur = np.where(magnitudes < threshold, cur, clamped).astype(np.float32)
            wq = gweight * cur
            dq = _qdq(wq, mode, -1, -1)
            sacts = gacts / cur[:, None]
            cur_out = np.matmul(dq, sacts)
            cur_diff = np.mean(np.abs(cur_out - fp32_out))
            if cur_diff < min_diff:
                min_diff = cur_diff
                best = cur
            alpha += alpha_step
        if best is not None:
            scale[b, off:off + gs] = best
    return scale[0] if is_2d else scale


# ------------------------------------------------------------------ GPTQ
def gptq_hessian(inputs, is_3d_weight=False):
    """gptq.py:200-242 (matmul node, non-transposed activation)."""
    nsamples = 0
    k = inputs[0].shape[-1]
    hb = 1 if not is_3d_weight else int(np.prod(inputs[0].shape[:-2]))
    H = np.zeros((hb, k, k), dtype=np.float32)
    for inp in inputs:
        is_3d_act = inp.ndim == 3
        bs = 1 if is_3d_weight or not is_3d_act else inp.shape[0]
        if is_3d_act and not is_3d_weight:
            inp = inp.reshape((-1, inp.shape[-1]))
        inp = np.moveaxis(inp, -1, -2)
        H *= F32(nsamples / (nsamples + bs))
        nsamples += bs
        inp = _f32(inp) * F32(math.sqrt(2 / nsamples))
        H += np.matmul(inp, np.moveaxis(inp, -1, -2))
    return H


def gptq_hinv(hessian, damp_percent=0.1):
    """gptq.py:269-270, 290-296: dead diagonal, damping, cholesky -> cholesky_inverse -> cholesky(upper).
    `hessian` [K,K] is modified in place.  Returns (Hinv upper, dead mask)."""
    H = hessian
    dead = np.diag(H) == 0
    H[dead, dead] = 1
    columns = H.shape[0]
    damp = F32(damp_percent) * np.mean(np.diag(H))
    di = np.arange(columns)
    H[di, di] += damp
    L = np.linalg.cholesky(H)
    Linv = np.linalg.inv(L)                     # numpy_linalg.py:42-48 cholesky_inverse(lower)
    Hi = np.matmul(Linv.T, Linv)
    return np.linalg.cholesky(Hi).T.copy(), dead  # upper


def gptq_quantize_weights(hessian, weight, mode, group_size, inputs=None, damp_percent=0.1, block_size=128,
                          scale_estimation=False, reduction_axes=(1,), hinv=None, dead=None):
    """gptq.py:244-379.  hessian [K,K], weight [N,K] (both are modified like the reference does).
    Returns (quantized fp32 [N,K], scales, zero_points|None).
    `hinv` (+ `dead`): test hook -- the upper factor of gptq.py:296 supplied by the caller (hessian is then unused),
    so that a comparison can hold the LAPACK step fixed and check the column loop alone."""
    W = weight
    if hinv is None:
        Hinv, dead = gptq_hinv(hessian, damp_percent)
    else:
        Hinv = hinv
        dead = np.zeros(W.shape[1], bool) if dead is None else dead
    W[:, dead] = 0
    scales, zps = [], []
    Q = np.zeros_like(W)
    columns = Hinv.shape[0]
    gs = group_size if group_size != -1 else W.shape[1]
    for i1 in range(0, columns, block_size):
        i2 = min(i1 + block_size, columns)
        count = i2 - i1
        Wb = W[:, i1:i2].copy()
        Qb = np.zeros_like(Wb)
        Eb = np.zeros_like(Wb)
        Hb = Hinv[i1:i2, i1:i2]
        for i in range(count):
            wcol = Wb[:, i]
            d = Hb[i, i]
            if (i1 + i) % gs == 0:
                wg = W[:, (i1 + i):(i1 + i + gs)]       # NB: from W, not from the in-block copy (gptq.py:315,324,330)
                if not is_integer(mode):
                    scales.append(calculate_float_quantization_params(wg, reduction_axes, mode))
                else:
                    if scale_estimation and INT_MODES[mode][0] == 4:
                        acts = [x[..., (i1 + i):(i1 + i + gs)] for x in inputs]
                        m, sh = activations_to_wc_statistics(acts)
                        sc, zp = scale_estimation_params(m, sh, wg, reduction_axes, mode, -1)
                    else:
                        sc, zp = calculate_integer_quantization_params(wg, reduction_axes, mode)
                    scales.append(sc)
                    zps.append(zp)
            if not is_integer(mode):
                qcol = float_quantize_dequantize_weight(wcol[:, None], mode, -1, None, scales[-1])
            else:
                qcol = integer_quantize_dequantize_weight(wcol[:, None], mode, -1, None, scales[-1], zps[-1])
            qcol = qcol.reshape(-1)
            Qb[:, i] = qcol
            err = (wcol - qcol) / d
            Wb[:, i:] -= np.matmul(err[:, None], Hb[i, i:][None, :])
            Eb[:, i] = err
        Q[:, i1:i2] = Qb
        W[:, i2:] -= np.matmul(Eb, Hinv[i1:i2, i2:])
    scales = np.stack(scales, axis=1)
    if group_size == -1:
        scales = np.squeeze(scales, axis=-1)
    if mode in ("int8_asym", "int4_asym"):
        zps = np.stack(zps, axis=1)
        if group_size == -1:
            zps = np.squeeze(zps, axis=-1)
    else:
        zps = None
    return Q.astype(np.float32), scales, zps


# ------------------------------------------------------------------------------------------------------------------
# GPTQ v2 tensor layout of a per-channel INT8 weight (test infrastructure, like everything in this file).
# The reference hands this to a third-party package that is absent here (gptqmodel==2.2.0,
# tests/integration/gptq_model/requirements_extra.txt; call site src/nncf/experimental/torch/gptqmodel/
# convertor.py:69-82).  Restated from that package's published `pack`: integer codes re-derived from the decompressed
# weight column by column, then OR-ed into 32-bit words along the input axis.  PARITY UNPINNED for the layout: the
# reference holds no packed tensors; its own test (tests/integration/gptq_model/test_converter.py:25-57) checks that
# the converted layer computes the same linear, which `gptq_dequant8_ref` lets the tests check too.
def gptq_pack8_ref(weight, scale, zero):
    """weight [out, in] decompressed values, scale / zero [out] -> (qweight [in / 4, out] int32, qzeros [1, out / 4]
    int32): intweight[i, o] = round((W[o, i] + zero[o] * scale[o]) / scale[o]), 4 codes per word, low byte first."""
    w = np.asarray(weight, dtype=np.float32)
    s = np.asarray(scale, dtype=np.float16).astype(np.float32).reshape(-1)
    z = np.asarray(zero, dtype=np.float32).reshape(-1)
    out_f, in_f = w.shape
    intweight = np.empty((in_f, out_f), dtype=np.uint32)
    for idx in range(in_f):
        intweight[idx] = np.round((w[:, idx] + z * s) / s).astype(np.int64).astype(np.uint32)
    qweight = np.zeros((in_f // 4, out_f), dtype=np.uint32)
    for row in range(in_f // 4):
        for j in range(4):
            qweight[row] |= intweight[row * 4 + j] << np.uint32(8 * j)
    zi = z.astype(np.uint32)
    qzeros = np.zeros((1, out_f // 4), dtype=np.uint32)
    for col in range(out_f // 4):
        for j in range(4):
            qzeros[0, col] |= zi[col * 4 + j] << np.uint32(8 * j)
    return qweight.view(np.int32), qzeros.view(np.int32)


def gptq_dequant8_ref(qweight, qzeros, scales):
    """The dequantization of that layout in fp16 arithmetic: W[o, 4 i + j] = (byte j of qweight[i, o] - zero[o]) *
    scale[o] -> [out, in] fp16."""
    qw = np.asarray(qweight).view(np.uint32)
    in4, out_f = qw.shape
    z = np.asarray(qzeros).view(np.uint32).reshape(-1)
    zeros = np.stack([(z >> np.uint32(8 * j)) & np.uint32(255) for j in range(4)], axis=1).reshape(-1)[:out_f]
    codes = np.stack([(qw >> np.uint32(8 * j)) & np.uint32(255) for j in range(4)], axis=1).reshape(in4 * 4, out_f)
    d = (codes.astype(np.int32) - zeros.astype(np.int32)[None, :]).astype(np.float16)
    s = np.asarray(scales, dtype=np.float16).reshape(1, -1)
    return (d * s).astype(np.float16).T.copy()
