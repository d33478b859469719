"""AWQ tensor math (src/nncf/quantization/algorithms/weight_compression/awq.py:243-366) for torch tensors on an AMD
GPU: the alpha grid search of every outlier group in one fused launch (nncf_amd_awq_search, csrc/wc_se.hip: scale,
per-row parameters, quantize-dequantize, contraction with the scaled activations, sum of |diff|) for any group size,
and the data-free scale.  Graph pattern matching and the insertion of the activation multiply stay with the caller."""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from nncf_amd import _lib
from nncf_amd.parameters import WeightCompressionConfig, as_config
from nncf_amd.statistics import WCTensorStatistic, process_stats

FP16_MAX_VALUE = 65504.0
FP16_OVERFLOW_MARGIN = 0.25
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


def _alphas(alpha_min: float, alpha_step: float, steps: int) -> np.ndarray:
    """The alpha grid as the reference's loop produces it (awq.py:309, 347: a running Python float sum), in fp32."""
    alphas, alpha = [], alpha_min
    for _ in range(steps):
        alphas.append(alpha)
        alpha += alpha_step
    return np.asarray(alphas, dtype=np.float64).astype(np.float32)


def _group_search(weight, s, X, config, group_size, groups, scale, alpha_min, alpha_step, steps, prev_s, prev_w, n_prev):
    """All outlier groups and all alpha steps through nncf_amd_awq_search (one launch per GROUPS_PER_LAUNCH groups):
    the weight slice of a group is read once and stays in registers for its alpha steps; the per-group preparation
    of the reference's loop body (awq.py:298-308) runs batched over the groups.  weight: fp32 contiguous [B, N, K];
    writes the winning scales into `scale` [B, K]."""
    B, N, K = weight.shape
    G = K // group_size
    n = X.shape[-1]
    lib = _lib.lib()
    dev = weight.device
    tiles = (N + 15) // 16
    mode_id = _lib.MODE_ID[config.mode.value]
    # bytes of the scaled activations of one (group, step) in the layout the search kernel of this shape and mode reads
    sacts_bytes = int(lib.nncf_amd_awq_sacts_bytes(group_size, n, mode_id))
    w4 = weight.view(B, N, G, group_size)
    alphas_t = torch.from_numpy(_alphas(alpha_min, alpha_step, steps)).to(dev)
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
        gprev = prev_w.reshape(B, G, group_size)[bi, gi].contiguous() if prev_s is not None else None
        # candidate scales of every alpha step incl. the fp16-overflow clamp, and gacts / scale in the search kernel's
        # layout (awq.py:311-342): one launch
        all_scales = torch.empty((nc, steps, group_size), dtype=torch.float32, device=dev)
        sacts = torch.empty((nc, steps, sacts_bytes // 4), dtype=torch.float32, device=dev)
        gscale, gacts = gscale.contiguous(), gacts.contiguous()
        with torch.cuda.device(dev):
            _lib.check(lib.nncf_amd_awq_prepare(_lib.ptr(gscale), _lib.ptr(alphas_t), _lib.ptr(gacts), _lib.ptr(gprev),
                                                float(prev_s) if prev_s is not None else 0.0, float(n_prev), nc, steps,
                                                group_size, n, mode_id, _lib.ptr(all_scales), _lib.ptr(sacts),
                                                _lib.stream_of(weight)))
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
    """awq.py:243-357 -> per-input-channel scale [hidden] (or [experts, hidden]): 1 everywhere except in the groups
    that hold the largest activation channels, where the alpha search of `_group_search` decides."""
    config = as_config(config)
    _lib.require_gpu(weight, "AWQ data-aware step")
    magnitude, samples = process_stats(statistics, subset_size, act_ch_axis)
    experts = weight.ndim == 3
    red = reduction_axes[0] % weight.ndim
    w3 = weight.to(torch.float32)
    if red != weight.ndim - 1:  # hidden axis last: [.., out, hidden]
        w3 = torch.movedim(w3, -1, -2)
    if not experts:
        w3, magnitude, samples = w3.unsqueeze(0), magnitude.unsqueeze(0), samples.unsqueeze(0)
    magnitude, samples = magnitude.to(torch.float32), samples.to(torch.float32)
    hidden = magnitude.shape[-1]
    gs = hidden if config.group_size == -1 else config.group_size
    # the groups of the top 0.2 % channels (at least one; earliest channel first among equals, :279-291)
    n_top = max(int(hidden * percent_to_apply), 1)
    order = torch.sort(-magnitude, dim=-1, stable=True)[1][:, :n_top]
    groups = sorted({(b, c // gs) for b, row in enumerate(order.tolist()) for c in row})
    # merged with the previous matmul: its mean |weight| per channel and the largest activation magnitude in front of
    # it drive the fp16-overflow clamp of the candidates (:263-275, 315-335)
    prev_peak, prev_mean, prev_len = None, None, 0
    if prev_statistics is not None and prev_weight is not None:
        pw = prev_weight.to(torch.float32)
        pred = red if experts else red + 1
        pw = pw if experts else pw.unsqueeze(0)
        prev_peak = process_stats(prev_statistics, subset_size, act_ch_axis)[0].to(torch.float32).max().item()
        prev_mean = pw.abs().mean(dim=pred)
        prev_len = pw.shape[pred]
    scale = torch.ones(magnitude.shape, dtype=torch.float32, device=weight.device)
    if groups:
        _group_search(w3.contiguous(), magnitude, samples, config, gs, groups, scale, alpha_min,
                      (alpha_max - alpha_min) / steps, steps, prev_peak, prev_mean, prev_len)
    return scale if experts else scale.squeeze(0)
