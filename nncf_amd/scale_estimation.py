"""Host-side mirror of ScaleEstimation.calculate_quantization_params
(src/nncf/quantization/algorithms/weight_compression/scale_estimation.py:160-409) for torch tensors on an AMD GPU.
Control flow stays Python (as in the reference); every quantize / quantize-dequantize inside the search loops is
one fused HIP launch (nncf_amd.weight_lowering), the small batched group matmuls stay torch.matmul."""
from __future__ import annotations

from copy import deepcopy
from typing import Optional, Tuple

import torch

from nncf_amd import _lib
from nncf_amd.parameters import WeightCompressionConfig, as_config
from nncf_amd.statistics import WCTensorStatistic, process_stats
from nncf_amd.weight_lowering import (
    do_float_quantization,
    do_integer_quantization,
    float_quantize_dequantize_weight,
    integer_quantize_dequantize_weight,
    reshape_weight_for_grouped_quantization,
)

EPS = torch.finfo(torch.float32).eps


def get_target_zero_mask(compressed_weights: torch.Tensor, zp: Optional[torch.Tensor] = None):
    """scale_estimation.py:380-393."""
    target = compressed_weights
    if zp is not None:
        target = target.to(zp.dtype) - zp
    zero_mask = torch.isclose(target.float(), torch.zeros((), device=target.device))
    return target, zero_mask


def estimate_scales(weight, target, zero_mask, importance):
    """scale_estimation.py:396-409."""
    ideal_scale = torch.abs(weight) / (torch.abs(target) + zero_mask)
    weighted_scale = ideal_scale * importance
    return torch.sum(weighted_scale, dim=-1, keepdim=True)


def calculate_quantization_params(
    statistics: WCTensorStatistic,
    weight: torch.Tensor,
    reduction_axes: Tuple[int, ...],
    config: WeightCompressionConfig,
    act_ch_axis: int = -1,
    subset_size: int = 32,
    initial_steps: int = 5,
    scale_steps: int = 10,
    weight_penalty: float = -1.0,
):
    """scale_estimation.py:160-377 -> (scale fp32, zero_point fp32 | None)."""
    config = as_config(config)
    reduction_axis = reduction_axes[0]
    s, X = process_stats(statistics, subset_size, act_ch_axis=act_ch_axis)
    X = X.to(torch.float32)
    native = weight if weight.dtype in (torch.bfloat16, torch.float32) else weight.to(torch.float32)
    weight = weight.to(torch.float32) if not FUSED_SEARCH else native  # (the fused search reads bf16 as it is)
    is_3d_weight = weight.ndim == 3
    was_transposed = False
    if reduction_axis == 0 or (reduction_axis == 1 and is_3d_weight):
        weight = torch.movedim(weight, -1, -2).contiguous()
        reduction_axis = weight.ndim - 1
        was_transposed = True
    group_size = config.group_size if config.group_size != -1 else weight.shape[reduction_axis]
    cur_config = deepcopy(config)
    cur_config.group_size = group_size

    fused = (FUSED_SEARCH and weight.ndim in (2, 3) and reduction_axis == weight.ndim - 1 and weight.numel() > 0
             and bool(_lib.lib().nncf_amd_se_step_supported(group_size, X.shape[-1])))
    if fused:
        result_scale, zp = _fused_search(weight.contiguous(), s, X, config, group_size, initial_steps, scale_steps,
                                         weight_penalty)
        return _finish(result_scale, zp, config, was_transposed)
    weight = weight.to(torch.float32)

    if not config.is_integer:
        q_weights, compressed_weight = float_quantize_dequantize_weight(weight, cur_config, reduction_axis,
                                                                         return_compressed_weight=True)
        zp = None
    else:
        q_weights, compressed_weight = integer_quantize_dequantize_weight(weight, cur_config, reduction_axis,
                                                                           return_compressed_weight=True)
        zp = compressed_weight.zero_point
        if zp is not None:
            zp = zp.to(compressed_weight.scale.dtype)
    scale = compressed_weight.scale
    compressed = compressed_weight.get_unscaled_tensor()

    s = s.unsqueeze(-2)
    s, _ = reshape_weight_for_grouped_quantization(s, reduction_axis, group_size)
    weight, _ = reshape_weight_for_grouped_quantization(weight, reduction_axis, group_size)

    importance = torch.ones_like(weight) * s
    target, zero_mask = get_target_zero_mask(compressed, zp)
    importance = torch.where(zero_mask, 0.0, importance)
    denum = torch.sum(importance, dim=-1, keepdim=True)
    importance = importance / (denum + EPS)

    X, _ = reshape_weight_for_grouped_quantization(X, -2, group_size)
    fp_outs = torch.matmul(torch.movedim(weight, -2, -3), X)
    q_outs = torch.matmul(torch.movedim(q_weights, -2, -3), X)
    min_max_scale_diffs = torch.mean((fp_outs - q_outs) ** 2, dim=-1)
    min_max_scale_diffs = torch.movedim(min_max_scale_diffs, -1, -2)
    if weight_penalty > 0.0:
        min_max_scale_diffs = min_max_scale_diffs + weight_penalty * torch.mean((q_weights - weight) ** 2, dim=-1)

    best_diffs = min_max_scale_diffs
    result_scale = scale
    scale_sign = scale / torch.abs(scale)
    zero_scale = 0.001

    def qdq(sc):
        if not config.is_integer:
            return float_quantize_dequantize_weight(weight, config, precomputed_scale=sc)
        return integer_quantize_dequantize_weight(weight, config, precomputed_scale=sc, precomputed_zero_point=zp)

    def diffs_of(q_weights_):
        q_outs_ = torch.matmul(torch.movedim(q_weights_, -2, -3), X)
        d = torch.mean((fp_outs - q_outs_) ** 2, dim=-1)
        d = torch.movedim(d, -1, -2)
        if weight_penalty > 0.0:
            d = d + weight_penalty * torch.mean((q_weights_ - weight) ** 2, dim=-1)
        return d

    # quantize -> target / zero mask -> estimate_scales is one fused pass (nncf_amd_se_estimate) per candidate scale
    weight = weight.contiguous()
    importance = importance.contiguous()
    zp_c = zp.contiguous() if zp is not None else None
    lib = _lib.lib()
    mode_id = _lib.MODE_ID[config.mode.value]
    n_groups = weight.numel() // group_size

    def estimate(sc):
        sc = sc.contiguous()
        out = torch.empty_like(sc)
        with torch.cuda.device(weight.device):
            _lib.check(lib.nncf_amd_se_estimate(_lib.ptr(weight), _lib.ptr(importance), n_groups, group_size, mode_id,
                                                _lib.ptr(sc), _lib.ptr(zp_c), zero_scale, _lib.ptr(out),
                                                _lib.stream_of(weight)))
        return out

    del target, zero_mask
    candidate = scale  # the first estimate uses the min-max quantization (scale_estimation.py:268-276)
    for i in range(initial_steps):
        near_to_ideal_scale = estimate(candidate) * scale_sign
        ideal_scale_diffs = diffs_of(qdq(near_to_ideal_scale))
        mask = (ideal_scale_diffs > best_diffs).to(best_diffs.dtype)
        best_diffs = mask * best_diffs + (1.0 - mask) * ideal_scale_diffs
        mask = mask.unsqueeze(-1)
        result_scale = mask * result_scale + (1.0 - mask) * near_to_ideal_scale
        candidate = result_scale

    for scale_step in range(scale_steps):
        factor = 1.0 - 0.05 * scale_step
        near_to_ideal_scale = estimate(factor * scale) * scale_sign
        ideal_scale_diffs = diffs_of(qdq(near_to_ideal_scale))
        mask = (ideal_scale_diffs > best_diffs).to(best_diffs.dtype)
        best_diffs = mask * best_diffs + (1.0 - mask) * ideal_scale_diffs
        mask = mask.unsqueeze(-1)
        result_scale = mask * result_scale + (1.0 - mask) * near_to_ideal_scale

    return _finish(result_scale, zp, config, was_transposed)


def _finish(result_scale, zp, config, was_transposed):
    """scale_estimation.py:362-377: shapes of the returned parameters."""
    if config.group_size == -1:
        result_scale = result_scale.squeeze(-2)
        if zp is not None:
            zp = zp.squeeze(-2)
    if was_transposed:
        if config.group_size == -1:
            result_scale = torch.movedim(result_scale, -1, -2)
            if zp is not None:
                zp = torch.movedim(zp, -1, -2)
        else:
            result_scale = torch.movedim(result_scale, (-1, -2, -3), (-2, -3, -1))
            if zp is not None:
                zp = torch.movedim(zp, (-1, -2, -3), (-2, -3, -1))
    return result_scale, zp


FUSED_SEARCH = True  # developer switch: False takes the step-by-step path below the fused one for every shape


def _fused_search(weight, s, X, config, group_size, initial_steps, scale_steps, weight_penalty):
    """The search of scale_estimation.py:236-360 with ONE launch per candidate (nncf_amd_se_step): the weight is read
    once per step in its own dtype, the importance is a 1-bit zero mask + 1 / (sum + eps) per group
    (nncf_amd_se_zero_mask), fp_outs comes out of the first step; nothing weight-sized is written.  weight: bf16 or
    fp32, contiguous [(E,) N, K] reduced along the last axis; s: [(E,) K]; X: [(E,) K, n].
    Returns (result_scale [(E,) N, G, 1], zp fp32 | None)."""
    from nncf_amd.weight_lowering import calculate_float_quantization_params, calculate_integer_quantization_params

    lib = _lib.lib()
    dev = weight.device
    batch = weight.shape[0] if weight.ndim == 3 else 1
    N, K = weight.shape[-2], weight.shape[-1]
    G = K // group_size
    wg, axis = reshape_weight_for_grouped_quantization(weight, weight.ndim - 1, group_size)  # [(E,) N, G, gs] view
    unit = WeightCompressionConfig(mode=config.mode, group_size=-1)
    if config.is_integer:
        scale0, zp_i = calculate_integer_quantization_params(wg, (axis,), unit)
        zp = zp_i.to(torch.float32) if zp_i is not None else None  # scale_estimation.py:227-229
    else:
        scale0, zp = calculate_float_quantization_params(wg, (axis,), unit), None
    s = s.to(torch.float32).contiguous()
    Xg, _ = reshape_weight_for_grouped_quantization(X, -2, group_size)  # [(E,) G, gs, n]
    Xg = Xg.contiguous()
    n_s = Xg.shape[-1]
    lead = tuple(weight.shape[:-2])
    fp_outs = torch.empty(lead + (G, N, n_s), dtype=torch.float32, device=dev)  # (:250), written by the first step
    zmask = torch.empty(tuple(scale0.shape[:-1]) + (4,), dtype=torch.int32, device=dev)
    rden = torch.empty(scale0.shape, dtype=torch.float32, device=dev)
    best = torch.empty(scale0.shape, dtype=torch.float32, device=dev)
    result = torch.empty_like(best)
    mode_id = _lib.MODE_ID[config.mode.value]
    dt = _lib.dtype_id(weight)
    with torch.cuda.device(dev):
        stream = _lib.stream_of(weight)
        _lib.check(lib.nncf_amd_se_zero_mask(_lib.ptr(wg), dt, _lib.ptr(s), _lib.ptr(scale0), _lib.ptr(zp), batch, N, G,
                                             group_size, mode_id, _lib.ptr(zmask), _lib.ptr(rden), stream))

        def step(kind, factor=1.0):
            _lib.check(lib.nncf_amd_se_step(_lib.ptr(wg), dt, _lib.ptr(zmask), _lib.ptr(rden), _lib.ptr(s), _lib.ptr(Xg),
                                            _lib.ptr(fp_outs), _lib.ptr(scale0), _lib.ptr(zp), _lib.ptr(best),
                                            _lib.ptr(result), batch, N, G, group_size, n_s, mode_id, kind, factor, 0.001,
                                            weight_penalty, stream))

        step(0)  # min-max baseline: fp_outs, best_diffs, result_scale = scale0
        for _ in range(initial_steps):
            step(1)  # candidate = current result_scale
        for scale_step in range(scale_steps):
            step(2, 1.0 - 0.05 * scale_step)  # candidate = factor * scale0
    return result, zp


class ScaleEstimation:
    """Parameter holder with the reference's defaults (advanced_parameters.py:317-335)."""

    def __init__(self, subset_size: int = 64, initial_steps: int = 5, scale_steps: int = 5, weight_penalty: float = -1.0):
        self._subset_size, self._initial_steps = subset_size, initial_steps
        self._scale_steps, self._weight_penalty = scale_steps, weight_penalty

    calculate_quantization_params = staticmethod(calculate_quantization_params)
