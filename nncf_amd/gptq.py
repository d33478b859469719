"""Host-side mirror of GPTQ's tensor math (src/nncf/quantization/algorithms/weight_compression/gptq.py:200-396) for
torch tensors on an AMD GPU.

  _calculate_hessian : H = (2/n) sum X^T X through the fp32-MFMA SYRK kernel (nncf_amd_hessian_accumulate).
  _quantize_weights  : dead columns, damping, Cholesky trio (torch.linalg on the device: dense LAPACK, not on the
                       custom-kernel path), then per 128-column block one fused HIP column sweep
                       (nncf_amd_gptq_block) + one trailing GEMM (torch.addmm), instead of K Python iterations.
"""
from __future__ import annotations

from functools import reduce
from operator import mul
from typing import List, Optional

import torch

from nncf_amd import _lib
from nncf_amd import scale_estimation as se
from nncf_amd.errors import UnsupportedModelError
from nncf_amd.parameters import WeightCompressionConfig, as_config
from nncf_amd.statistics import WCTensorStatistic, activations_to_wc_statistics


def calculate_hessian(inputs: List[torch.Tensor], is_3d_weight: bool = False) -> torch.Tensor:
    """gptq.py:200-242 for a matmul node with non-transposed activations.  Returns fp32 [batch, K, K]."""
    first = inputs[0]
    _lib.require_gpu(first, "calculate_hessian")
    k = first.shape[-1]
    hb = 1 if not is_3d_weight else reduce(mul, first.shape[:-2], 1)
    H = torch.zeros((hb, k, k), dtype=torch.float32, device=first.device)
    L = _lib.lib()
    nsamples = 0
    for inp in inputs:
        is_3d_act = inp.ndim == 3
        batch_size = 1 if is_3d_weight or not is_3d_act else inp.shape[0]
        if inp.dtype == torch.float64:
            inp = inp.float()
        if is_3d_act and not is_3d_weight:
            xs = [inp.reshape(-1, k)]
        elif is_3d_weight:
            xs = list(inp.reshape(hb, -1, k))
        else:
            xs = [inp.reshape(-1, k)]
        beta = nsamples / (nsamples + batch_size)
        nsamples += batch_size
        alpha = 2.0 / nsamples
        with torch.cuda.device(first.device):
            for b, x in enumerate(xs):
                x = x.contiguous()
                st = L.nncf_amd_hessian_accumulate(_lib.ptr(H[b]), _lib.ptr(x), _lib.dtype_id(x), x.shape[0], k,
                                                   alpha, beta, _lib.stream_of(x))
                _lib.check(st)
    return H


def quantize_weights(hessian: torch.Tensor, weight: torch.Tensor, config: WeightCompressionConfig,
                     inputs: Optional[List[torch.Tensor]] = None, damp_percent: float = 0.1, block_size: int = 128,
                     scale_estimation: bool = False, reduction_axes=(1,),
                     input_stats: Optional[WCTensorStatistic] = None):
    """gptq.py:244-379.  hessian [K, K] and weight [N, K] fp32 are modified in place like the reference does.
    Returns (quantized fp32 [N, K], scales [N, K/gs, 1] (or [N, 1] for group_size -1), zero_points | None).
    `input_stats` (token means over the FULL hidden axis, one per sample) may stand in for `inputs` when scale
    estimation is on: the mean of a column slice is the slice of the mean (gptq.py:320-321, 381-396)."""
    config = as_config(config)
    _lib.require_gpu(weight, "quantize_weights")
    assert hessian.ndim == 2, "Hessian should be 2D"
    if tuple(reduction_axes) != (1,) or weight.ndim != 2:
        raise UnsupportedModelError("nncf_amd GPTQ expects a [out, in] weight reduced along axis 1")
    if block_size > 128:
        raise UnsupportedModelError("nncf_amd GPTQ block_size must be <= 128")
    W = weight
    assert W.dtype == torch.float32 and W.is_contiguous() and hessian.dtype == torch.float32
    N, K = W.shape
    dead = torch.diag(hessian) == 0
    idx = torch.arange(K, device=W.device)
    hessian[idx[dead], idx[dead]] = 1
    W[:, dead] = 0
    gs = config.group_size if config.group_size != -1 else K
    ngroups = K // gs
    damp = damp_percent * torch.mean(torch.diag(hessian))
    hessian[idx, idx] += damp
    hinv = _upper_cholesky_of_inverse(hessian)

    asym = config.is_integer and config.is_asym_mode
    Q = torch.zeros_like(W)
    scales = torch.empty((N, ngroups), dtype=torch.float32, device=W.device)
    zps = torch.empty((N, ngroups), dtype=torch.int32, device=W.device) if asym else None
    use_se = scale_estimation and config.is_integer and config.num_bits == 4
    mode_id = _lib.MODE_ID[config.mode.value]
    L = _lib.lib()
    block_cfg = WeightCompressionConfig(mode=config.mode, group_size=-1)
    with torch.cuda.device(W.device):
        stream = _lib.stream_of(W)
        for i1 in range(0, K, block_size):
            i2 = min(i1 + block_size, K)
            count = i2 - i1
            if use_se:
                # gptq.py:319-327: parameters of every group that STARTS in this block, from the current W
                for c in range(i1, i2):
                    if c % gs:
                        continue
                    if input_stats is not None:
                        stats = WCTensorStatistic([m[..., c:c + gs] for m in input_stats.mean_values],
                                                  [tuple(sh[:-1]) + (gs,) for sh in input_stats.shape_values])
                    else:
                        acts = [inp[..., c:c + gs] for inp in inputs]
                        stats = activations_to_wc_statistics(acts)
                    sc, zp = se.calculate_quantization_params(stats, W[:, c:c + gs].contiguous(), (1,), block_cfg)
                    scales[:, c // gs] = sc.reshape(N)
                    if asym:
                        zps[:, c // gs] = zp.reshape(N).to(torch.int32)
            err = torch.empty((N, count), dtype=torch.float32, device=W.device)
            st = L.nncf_amd_gptq_block(_lib.ptr(W), _lib.ptr(Q), _lib.ptr(err), _lib.ptr(hinv), N, K, i1, count, gs,
                                       mode_id, _lib.ptr(scales), _lib.ptr(zps), int(use_se), stream)
            _lib.check(st)
            if i2 < K:
                W[:, i2:] -= torch.matmul(err, hinv[i1:i2, i2:])  # gptq.py:363, plain library GEMM
    out_scales = scales.reshape(N, ngroups, 1)
    out_zps = zps.reshape(N, ngroups, 1) if asym else None
    if config.group_size == -1:
        out_scales = out_scales.squeeze(-1)
        if out_zps is not None:
            out_zps = out_zps.squeeze(-1)
    return Q, out_scales, out_zps


def _upper_cholesky_of_inverse(hessian: torch.Tensor) -> torch.Tensor:
    """U upper triangular with H^-1 = U^T U: what the reference gets from cholesky -> cholesky_inverse ->
    cholesky(upper=True) (gptq.py:283-285), from ONE factorization instead of two plus an inversion.  With J the
    index reversal, J H J = L L^T (plain Cholesky) gives H = R R^T for the upper triangular R = J L J, hence
    H^-1 = R^-T R^-1 = U^T U with U = R^-1, and a Cholesky factor with positive diagonal is unique.  In fp32 this is
    at least as close to the exact U as the three-step chain (one rounding cascade less)."""
    L = torch.linalg.cholesky(torch.flip(hessian, (0, 1)))
    R = torch.flip(L, (0, 1))
    eye = torch.eye(hessian.shape[0], dtype=hessian.dtype, device=hessian.device)
    return torch.linalg.solve_triangular(R, eye, upper=True).contiguous()


class GPTQ:
    """Parameter holder with the reference's constructor defaults (gptq.py:49-65) and the two tensor-level methods."""

    def __init__(self, damp_percent: float = 0.1, block_size: int = 128, subset_size: int = 128,
                 scale_estimation: bool = False):
        self._damp_percent, self._block_size = damp_percent, block_size
        self._subset_size, self._scale_estimation = subset_size, scale_estimation

    def _calculate_hessian(self, node, inputs: List[torch.Tensor], is_3d_weight: bool = False) -> torch.Tensor:
        return calculate_hessian(inputs, is_3d_weight)

    def _quantize_weights(self, wc_params, hessian: torch.Tensor, weight_tensor: torch.Tensor,
                          inputs: List[torch.Tensor]):
        return quantize_weights(hessian, weight_tensor, wc_params.compression_config, inputs, self._damp_percent,
                                self._block_size, self._scale_estimation, wc_params.reduction_axes)

    activations_to_wc_statistics = staticmethod(activations_to_wc_statistics)
