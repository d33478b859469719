"""Host-side mirror of the AWQ tensor math (src/nncf/quantization/algorithms/weight_compression/awq.py:243-366)
for torch tensors on an AMD GPU: the alpha grid search per outlier group (fake-quantize through the fused HIP
quantize-dequantize kernel + small matmuls) and the data-free scale.  Graph pattern matching and the insertion of
the activation multiply stay with the caller (the reference's own Python)."""
from __future__ import annotations

from copy import deepcopy
from typing import Optional

import numpy as np
import torch

from nncf_amd import _lib
from nncf_amd.parameters import WeightCompressionConfig, as_config
from nncf_amd.statistics import WCTensorStatistic, process_stats
from nncf_amd.weight_lowering import float_quantize_dequantize_weight, integer_quantize_dequantize_weight

FP16_MAX_VALUE = 65504.0
FP16_OVERFLOW_MARGIN = 0.25
FUSED_SEARCH = True  # developer switch: False takes the per-group torch path for every shape
GROUPS_PER_LAUNCH = 64


def _quantile_rows(x: torch.Tensor, q: float) -> torch.Tensor:
    """torch.quantile(x, q, dim=1, keepdim=True) ('linear') for a finite 2-D fp32 x, by its own recipe -- sort, the two
    neighbours of rank q*(n-1) (fp32 arithmetic, as the q tensor of torch.quantile is fp32), torch.lerp -- without the
    ~10 other kernels of torch.quantile: their code objects cost 0.35 s to load in the first call of a process.
    (A NaN in a row is not propagated as torch.quantile does; activation magnitudes with NaN are outside the contract.)"""
    n = x.shape[1]
    rank = np.float32(q) * np.float32(n - 1)
    below, above = int(np.floor(rank)), int(np.ceil(rank))
    weight = float(np.float32(rank - np.float32(below)))
    srt = torch.sort(x, dim=1)[0]
    return torch.lerp(srt[:, below:below + 1], srt[:, above:above + 1], weight)


def _candidate_scales(gscale, alpha_min, alpha_step, steps, gprev, prev_s, n_prev):
    """cur_scale of every alpha step, [..., steps, gs] for gscale [..., gs] (awq.py:311-335 incl. the fp16-overflow
    clamp; gprev [..., gs] = mean |previous weight| of the group's channels)."""
    alphas = []
    alpha = alpha_min
    for _ in range(steps):
        alphas.append(alpha)
        alpha += alpha_step
    alphas32 = np.asarray(alphas, dtype=np.float64).astype(np.float32)
    alphas_t = torch.from_numpy(alphas32).to(gscale.device)
    all_scales = torch.pow(gscale.unsqueeze(-2), alphas_t.unsqueeze(-1))
    half = np.nonzero(alphas32 == np.float32(0.5))[0]  # tensor ** python-scalar takes the sqrt path for exactly 0.5
    if half.size:
        root = torch.sqrt(gscale)
        for j in half.tolist():
            all_scales[..., j, :] = root
    if prev_s is not None:
        threshold = FP16_OVERFLOW_MARGIN * FP16_MAX_VALUE
        # awq.py:321-335, same operation order: (prev_w / cur_scale) * prev_s * n  vs  prev_w * prev_s * n / threshold
        magnitudes = (gprev.unsqueeze(-2) / all_scales) * prev_s * n_prev
        clamped = gprev * prev_s * n_prev / threshold
        clip_rows = magnitudes.amax(dim=-1, keepdim=True) >= threshold
        all_scales = torch.where(clip_rows & ~(magnitudes < threshold), clamped.unsqueeze(-2), all_scales)
    return all_scales


def _fused_group_search(weight, s, X, config, group_size, groups, scale, alpha_min, alpha_step, steps, prev_s, prev_w, n_prev):
    """All outlier groups and all alpha steps through nncf_amd_awq_search (one launch per GROUPS_PER_LAUNCH groups):
    the weight slice of a group is read once and stays in registers for its alpha steps; the per-group preparation
    of the reference's loop body (awq.py:298-308) runs batched over the groups.  weight: fp32 contiguous [B, N, K];
    writes the winning scales into `scale` [B, K]."""
    B, N, K = weight.shape
    G = K // group_size
    n = X.shape[-1]
    nt = 1 if n <= 16 else (2 if n <= 32 else 4)
    dev = weight.device
    tiles = (N + 15) // 16
    lib = _lib.lib()
    mode_id = _lib.MODE_ID[config.mode.value]
    w4 = weight.view(B, N, G, group_size)
    for c0 in range(0, len(groups), GROUPS_PER_LAUNCH):
        chunk = groups[c0:c0 + GROUPS_PER_LAUNCH]
        nc = len(chunk)
        bi = torch.tensor([b for b, _ in chunk], dtype=torch.int64, device=dev)
        gi = torch.tensor([g for _, g in chunk], dtype=torch.int64, device=dev)
        gscale = s.reshape(B, G, group_size)[bi, gi]                       # [nc, gs]
        gweight = w4[bi, :, gi, :]                                      # [nc, N, gs]
        gacts = X.reshape(B, G, group_size, n)[bi, gi]                  # [nc, gs, n]
        a_min = _quantile_rows(gscale, 0.1)
        gscale = torch.minimum(torch.maximum(gscale, a_min), torch.full_like(gscale, 1e2))
        fp_out = torch.matmul(gweight, gacts).contiguous()              # [nc, N, n]
        min_diff = fp_out.abs().amax(dim=(1, 2))
        gprev = prev_w.reshape(B, G, group_size)[bi, gi] if prev_s is not None else None
        all_scales = _candidate_scales(gscale, alpha_min, alpha_step, steps, gprev, prev_s, n_prev).contiguous()
        sacts = gacts.unsqueeze(1) / all_scales.unsqueeze(3)            # [nc, steps, gs, n]  (awq.py:342)
        if n < 16 * nt:
            sacts = torch.nn.functional.pad(sacts, (0, 16 * nt - n))
        sacts = sacts.reshape(nc, steps, group_size, nt, 16).transpose(3, 4).contiguous()  # kernel layout [.., 16, NT]
        goff = bi * (N * K) + gi * group_size
        partial = torch.empty((nc, steps, tiles), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(lib.nncf_amd_awq_search(_lib.ptr(weight), N, K, _lib.ptr(goff), nc, group_size,
                                               _lib.ptr(all_scales), steps, _lib.ptr(sacts), _lib.ptr(fp_out), n, mode_id,
                                               _lib.ptr(partial), _lib.stream_of(weight)))
        diffs = partial.sum(dim=-1) / float(N * n)  # mean |cur_out - fp32_out| per step (awq.py:345)
        best = torch.argmin(diffs, dim=1)  # the reference's running "strictly better" minimum = earliest argmin
        best_diff = diffs.gather(1, best.unsqueeze(1)).squeeze(1)
        improved = best_diff < min_diff
        best_scale = all_scales[torch.arange(nc, device=dev), best]     # [nc, gs]
        sv = scale.view(B, G, group_size)
        sv[bi, gi] = torch.where(improved.unsqueeze(1), best_scale, sv[bi, gi])


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
    topk_idxs = torch.sort(-s, dim=-1, stable=True)[1][:, :top_k].cpu()
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
    if (FUSED_SEARCH and weight.is_cuda and groups_to_correct
            and bool(_lib.lib().nncf_amd_se_step_supported(group_size, X.shape[-1]))):
        _fused_group_search(weight.contiguous(), s, X, config, group_size, groups_to_correct, scale, alpha_min, alpha_step,
                            steps, prev_s, prev_w, prev_weight.shape[reduction_axis] if prev_weight is not None else 0)
        return scale.squeeze(0) if is_2d_weight else scale
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
        gprev = prev_w[batch_idx, offset:offset + group_size] if prev_s is not None else None
        all_scales = _candidate_scales(gscale, alpha_min, alpha_step, steps, gprev, prev_s,
                                       prev_weight.shape[reduction_axis] if prev_weight is not None else 0)
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
