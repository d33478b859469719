"""Host-side mirror of the AWQ tensor math (src/nncf/quantization/algorithms/weight_compression/awq.py:243-366)
for torch tensors on an AMD GPU: the alpha grid search per outlier group (fake-quantize through the fused HIP
quantize-dequantize kernel + small matmuls) and the data-free scale.  Graph pattern matching and the insertion of
the activation multiply stay with the caller (the reference's own Python)."""
from __future__ import annotations

from copy import deepcopy
from typing import Optional

import torch

from nncf_amd.parameters import WeightCompressionConfig, as_config
from nncf_amd.statistics import WCTensorStatistic, process_stats
from nncf_amd.weight_lowering import float_quantize_dequantize_weight, integer_quantize_dequantize_weight

FP16_MAX_VALUE = 65504.0
FP16_OVERFLOW_MARGIN = 0.25


def data_free_step(weight: torch.Tensor, axis: int) -> torch.Tensor:
    """awq.py:363-366."""
    eps = torch.finfo(torch.float32).eps
    scale = torch.clamp_min(torch.mean(torch.abs(weight.to(torch.float32)), dim=axis), eps)
    return 1 / scale


def data_aware_step(weight: torch.Tensor, reduction_axes, config: WeightCompressionConfig,
                    statistics: WCTensorStatistic, act_ch_axis: int = -1, subset_size: int = 32,
                    percent_to_apply: float = 0.002, alpha_min: float = 0.0, alpha_max: float = 1.0, steps: int = 100,
                    prev_weight: Optional[torch.Tensor] = None, prev_statistics: Optional[WCTensorStatistic] = None):
    """awq.py:243-357 -> per-input-channel scale [hidden] (or [experts, hidden])."""
    config = as_config(config)
    alpha_step = (alpha_max - alpha_min) / steps
    s, X = process_stats(statistics, subset_size, act_ch_axis)
    s = s.to(torch.float32)
    X = X.to(torch.float32)
    weight = weight.to(torch.float32)
    is_2d_weight = weight.ndim == 2
    reduction_axis = reduction_axes[0]
    if is_2d_weight:
        s, X, weight = s.unsqueeze(0), X.unsqueeze(0), weight.unsqueeze(0)
        prev_weight = prev_weight.unsqueeze(0) if prev_weight is not None else None
        reduction_axis += 1
    prev_s, prev_w = None, None
    if prev_statistics is not None and prev_weight is not None:
        prev_s, _ = process_stats(prev_statistics, subset_size, act_ch_axis)
        prev_s = prev_s.to(torch.float32).max().item()
        prev_w = torch.mean(torch.abs(prev_weight.to(torch.float32)), dim=reduction_axis)
    top_k = max(int(s.shape[-1] * percent_to_apply), 1)
    topk_idxs = torch.argsort(-s, stable=True)[:, :top_k].cpu()
    group_size = config.group_size if config.group_size != -1 else s.shape[-1]
    groups_to_correct = set()
    for batch_idx in range(topk_idxs.shape[0]):
        for k_idx in range(topk_idxs.shape[1]):
            groups_to_correct.add((batch_idx, int(topk_idxs[batch_idx, k_idx]) // group_size))
    groups_to_correct = sorted(groups_to_correct)
    if reduction_axis == 1:
        weight = torch.movedim(weight, -1, -2)
    scale = torch.ones_like(torch.mean(X, dim=-1))
    awq_config = deepcopy(config)
    awq_config.group_size = -1
    qdq = integer_quantize_dequantize_weight if config.is_integer else float_quantize_dequantize_weight
    for batch_idx, gi in groups_to_correct:
        offset = gi * group_size
        gscale = s[batch_idx, offset:offset + group_size]
        gweight = weight[batch_idx, :, offset:offset + group_size]
        gacts = X[batch_idx, offset:offset + group_size, :]
        a_min = torch.quantile(gscale, 0.1).to(torch.float32)
        gscale = torch.clip(gscale, min=a_min, max=1e2)
        fp32_out = torch.matmul(gweight, gacts)
        # The reference walks the 100 alphas one by one (awq.py:300-348: ~10 small ops per step).  Here all steps of
        # a group go through the same arithmetic at once: [steps, N, gs] scaled weights -> ONE quantize-dequantize
        # launch (rows are independent) -> batched matmul -> per-step mean |diff|; the running "strictly better"
        # minimum of the reference is the argmin with the earliest index on ties.
        min_diff = torch.max(torch.abs(fp32_out))
        alphas = []
        alpha = alpha_min
        for _ in range(steps):
            alphas.append(alpha)
            alpha += alpha_step
        alphas_t = torch.tensor(alphas, dtype=torch.float64, device=gscale.device).to(torch.float32)
        all_scales = torch.pow(gscale.unsqueeze(0), alphas_t.unsqueeze(1))  # [steps, gs]
        half = alphas_t == 0.5  # tensor ** python-scalar takes the sqrt path for exactly 0.5
        if bool(half.any()):
            all_scales[half] = torch.sqrt(gscale)
        if prev_s is not None:
            threshold = FP16_OVERFLOW_MARGIN * FP16_MAX_VALUE
            # awq.py:321-335, same operation order: (prev_w / cur_scale) * prev_s * n  vs  prev_w * prev_s * n / threshold
            gprev = prev_w[batch_idx, offset:offset + group_size]
            n_prev = prev_weight.shape[reduction_axis]
            magnitudes = (gprev.unsqueeze(0) / all_scales) * prev_s * n_prev
            clamped = gprev * prev_s * n_prev / threshold
            clip_rows = magnitudes.amax(dim=1, keepdim=True) >= threshold
            all_scales = torch.where(clip_rows & ~(magnitudes < threshold), clamped.unsqueeze(0), all_scales)
        n_rows, n_tok = gweight.shape[0], gacts.shape[1]
        chunk = max(1, min(steps, (1 << 28) // max(1, n_rows * max(group_size, n_tok))))  # <= 1 GiB of fp32 per temporary
        diffs = []
        for c0 in range(0, steps, chunk):
            sc = all_scales[c0:c0 + chunk]
            scaled = (gweight.unsqueeze(0) * sc.unsqueeze(1)).reshape(-1, gweight.shape[1]).contiguous()
            g_dq = qdq(scaled, awq_config, -1).reshape(sc.shape[0], n_rows, -1)
            sacts = gacts.unsqueeze(0) / sc.unsqueeze(2)
            cur_out = torch.bmm(g_dq, sacts)
            diffs.append(torch.mean(torch.abs(cur_out - fp32_out.unsqueeze(0)), dim=(1, 2)))
        diffs = torch.cat(diffs)
        best = int(torch.argmin(diffs))
        best_scale = all_scales[best] if bool(diffs[best] < min_diff) else None
        if best_scale is not None:
            scale[batch_idx, offset:offset + group_size] = best_scale
    if is_2d_weight:
        scale = scale.squeeze(0)
    return scale
