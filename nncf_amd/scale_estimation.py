"""Scale estimation for torch tensors on an AMD GPU: the search of ScaleEstimation.calculate_quantization_params
(src/nncf/quantization/algorithms/weight_compression/scale_estimation.py:160-377) as a sequence of fused launches --
one per candidate scale (nncf_amd_se_search -> nncf_amd_se_step: csrc/wc_se.hip) for every group size and sample count.  What is left on
the host is shape glue: statistics -> (s, X), the grouped views, the order of the candidates and the shapes the
reference returns."""
from __future__ import annotations

from typing import Tuple

import torch

from nncf_amd import _lib
from nncf_amd.parameters import WeightCompressionConfig, as_config
from nncf_amd.statistics import WCTensorStatistic, process_stats
from nncf_amd.weight_lowering import (
    calculate_float_quantization_params,
    calculate_integer_quantization_params,
    reshape_weight_for_grouped_quantization,
)

ZERO_SCALE = 0.001  # scale_estimation.py:266


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
    """scale_estimation.py:160-377 -> (scale fp32, zero_point fp32 | None), shaped as the reference returns them:
    [.., N, G, 1] for grouped modes, [.., N, 1] for group_size -1, moved back when the reduction axis was not last."""
    config = as_config(config)
    _lib.require_gpu(weight, "scale estimation")
    s, X = process_stats(statistics, subset_size, act_ch_axis=act_ch_axis)
    # the step kernel reads bf16 / fp32 weights as they are (the reference's fp32 upcast is exact, :189)
    w = weight if weight.dtype in (torch.bfloat16, torch.float32) else weight.to(torch.float32)
    axis = reduction_axes[0] % w.ndim
    moved = axis != w.ndim - 1  # (:199-205: the reference transposes such weights too)
    if moved:
        w = torch.movedim(w, -1, -2)
    w = w.contiguous()
    group_size = config.group_size if config.group_size != -1 else w.shape[-1]
    scale, zp = _search(w, s.to(torch.float32), X.to(torch.float32), config, group_size, initial_steps, scale_steps,
                        weight_penalty)
    if config.group_size == -1:  # (:362-377)
        scale = scale.squeeze(-2)
        zp = zp.squeeze(-2) if zp is not None else None
        if moved:
            scale = torch.movedim(scale, -1, -2)
            zp = torch.movedim(zp, -1, -2) if zp is not None else None
    elif moved:
        scale = torch.movedim(scale, (-1, -2, -3), (-2, -3, -1))
        zp = torch.movedim(zp, (-1, -2, -3), (-2, -3, -1)) if zp is not None else None
    return scale, zp


def _search(weight, s, X, config, group_size, initial_steps, scale_steps, weight_penalty):
    """weight: bf16 or fp32, contiguous [(E,) N, K] reduced along the last axis; s: fp32 [(E,) K]; X: fp32 [(E,) K, n].
    The min-max parameters (:207-229) come from the lowering kernel; nncf_amd_se_zero_mask turns the importance of
    :236-247 into one bit per element + a normaliser per group; step 0 writes fp_outs and the min-max baseline
    (:249-262), then `initial_steps` rectifications of the current best scale (:268-298) and `scale_steps` grid
    candidates factor * scale0 (:300-360).  Nothing weight-sized is written.
    Returns (result_scale [(E,) N, G, 1], zp fp32 | None)."""
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
    if weight.numel() == 0:
        return scale0, zp
    s = s.contiguous()
    Xg, _ = reshape_weight_for_grouped_quantization(X, -2, group_size)  # [(E,) G, gs, n]
    Xg = Xg.contiguous()
    n_s = Xg.shape[-1]
    fp_outs = torch.empty(tuple(weight.shape[:-2]) + (G, N, n_s), dtype=torch.float32, device=dev)
    zmask = torch.empty(tuple(scale0.shape[:-1]) + (4,), dtype=torch.int32, device=dev)
    rden = torch.empty(scale0.shape, dtype=torch.float32, device=dev)
    best = torch.empty(scale0.shape, dtype=torch.float32, device=dev)
    result = torch.empty_like(best)
    mode_id = _lib.MODE_ID[config.mode.value]
    dt = _lib.dtype_id(weight)
    with torch.cuda.device(dev):
        # zero mask, baseline (writes fp_outs, best, result), rectifications, grid steps: one call queues them all
        _lib.check(lib.nncf_amd_se_search(_lib.ptr(wg), dt, _lib.ptr(zmask), _lib.ptr(rden), _lib.ptr(s), _lib.ptr(Xg),
                                          _lib.ptr(fp_outs), _lib.ptr(scale0), _lib.ptr(zp), _lib.ptr(best),
                                          _lib.ptr(result), batch, N, G, group_size, n_s, mode_id, initial_steps,
                                          scale_steps, ZERO_SCALE, weight_penalty, _lib.stream_of(weight)))
    return result, zp


class ScaleEstimation:
    """Parameter holder with the reference's defaults (advanced_parameters.py:317-335)."""

    def __init__(self, subset_size: int = 64, initial_steps: int = 5, scale_steps: int = 5, weight_penalty: float = -1.0):
        self._subset_size, self._initial_steps = subset_size, initial_steps
        self._scale_steps, self._weight_penalty = scale_steps, weight_penalty

    calculate_quantization_params = staticmethod(calculate_quantization_params)
