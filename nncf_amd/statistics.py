"""Host-side mirror of the activation-statistics reducers that feed AWQ, scale estimation and mixed precision
(src/nncf/common/tensor_statistics/collectors.py:468-540, 924-936; activation_stats.py:20-57;
statistics.py:260-307; gptq.py:381-396) for torch tensors on an AMD GPU.  One fused HIP pass
(nncf_amd_reduce_activation_stats) computes every statistic of one activation."""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from functools import reduce
from operator import mul
from typing import List, Optional, Sequence, Tuple

import torch

from nncf_amd import _lib
from nncf_amd.errors import InternalError, UnsupportedModelError


def activation_stats(x: torch.Tensor, want_mean: bool = True, want_stats: bool = True):
    """x viewed as [batch, tokens, hidden] (leading axes folded into batch).
    Returns (mean fp32 [hidden] over batch and tokens | None,
             stats fp32 [4] = mean(var_tokens), max(var_tokens), mean(absmax_tokens), sum(x*x) | None)."""
    _lib.require_gpu(x, "activation_stats")
    if x.dtype == torch.float64:
        x = x.float()
    if x.ndim < 2:
        x = x.reshape(1, 1, -1) if x.ndim == 1 else x.reshape(1, 1, 1)
    x = x.contiguous()
    hidden, tokens = x.shape[-1], x.shape[-2]
    batch = math.prod(x.shape[:-2])
    dev = x.device
    mean = torch.empty(hidden, dtype=torch.float32, device=dev) if want_mean else None
    stats = torch.empty(4, dtype=torch.float32, device=dev) if want_stats else None
    L = _lib.lib()
    need = L.nncf_amd_reduce_workspace(batch, tokens, hidden)
    ws = torch.empty(max(need, 16), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        st = L.nncf_amd_reduce_activation_stats(_lib.ptr(x), _lib.dtype_id(x), batch, tokens, hidden, _lib.ptr(mean),
                                                _lib.ptr(stats), _lib.ptr(ws), need, _lib.stream_of(x))
    _lib.check(st)
    return mean, stats


REDUCER_MAX_BATCH = 65535  # grid limit of the statistics kernel (leading axes folded into the batch)
_REDUCER_DTYPES = (torch.float32, torch.float16, torch.bfloat16, torch.float64)


def _check_reducer_input(t: torch.Tensor, who: str, batch: int) -> None:
    """Inputs the HIP reducer does not take are UNSUPPORTED (the caller, e.g. install()'s patched reducers, takes the
    reference's path for them), not faults of this library: other dtypes, more folded batch entries than the grid holds."""
    if t.dtype not in _REDUCER_DTYPES:
        raise UnsupportedModelError(f"nncf_amd {who}: activation dtype {t.dtype} is not one of fp32 / fp16 / bf16 / fp64")
    if batch > REDUCER_MAX_BATCH:
        raise UnsupportedModelError(f"nncf_amd {who}: {batch} leading entries exceed the kernel's limit of {REDUCER_MAX_BATCH}")


def _norm_axes(ndim: int, axes) -> Tuple[int, ...]:
    if axes is None:
        return tuple(range(ndim))
    return tuple(sorted(a % ndim for a in axes))


class _ReducerBase:
    """collectors.py:83-157 (out-of-place path only)."""

    def __init__(self, axes=None, inplace: bool = False):
        self._axes = axes
        self._keepdims = True
        self.inplace = inplace

    def __call__(self, x: List[torch.Tensor]):
        if any(t.numel() == 0 for t in x):
            return None
        return self._reduce_out_of_place(x)

    def __eq__(self, o):
        return isinstance(o, self.__class__) and self._axes == o._axes

    def __hash__(self):
        return hash((self.__class__.__name__, self._axes))


class RawReducer(_ReducerBase):
    def _reduce_out_of_place(self, x):
        return x


class ShapeReducer(_ReducerBase):
    def _reduce_out_of_place(self, x):
        return [torch.tensor(tuple(x[0].shape), dtype=torch.int32, device=x[0].device)]


class MeanReducer(_ReducerBase):
    """collectors.py:512-516 for axes == all-but-the-last (the weight-compression statistic, algorithm.py:1357-1370)."""

    def _reduce_out_of_place(self, x):
        t = x[0]
        axes = _norm_axes(t.ndim, self._axes)
        if axes != tuple(range(t.ndim - 1)):
            msg = f"nncf_amd MeanReducer reduces all axes but the last; got axes={self._axes} for ndim={t.ndim}"
            raise UnsupportedModelError(msg)
        _check_reducer_input(t, "MeanReducer", 1)
        mean, _ = activation_stats(t.reshape(1, -1, t.shape[-1]), want_stats=False)
        shape = [1] * (t.ndim - 1) + [t.shape[-1]]
        # torch.mean accumulates bf16/f16 in fp32 and rounds once to the input dtype
        return [mean.to(t.dtype if t.dtype.is_floating_point else torch.float32).reshape(shape)]


class _TokenStatReducer(_ReducerBase):
    _index = 0

    def _reduce_out_of_place(self, x):
        t = x[0]
        axes = _norm_axes(t.ndim, self._axes)
        if t.ndim < 2 or axes != (t.ndim - 2,):
            msg = f"nncf_amd {type(self).__name__} reduces axis -2; got axes={self._axes} for ndim={t.ndim}"
            raise UnsupportedModelError(msg)
        _check_reducer_input(t, type(self).__name__, math.prod(t.shape[:-2]))
        _, stats = activation_stats(t, want_mean=False)
        return [stats[self._index].to(t.dtype)]


class MeanVarianceReducer(_TokenStatReducer):
    """collectors.py:519-524."""
    _index = 0


class MaxVarianceReducer(_TokenStatReducer):
    """collectors.py:527-532."""
    _index = 1


class MeanAbsMaxReducer(_TokenStatReducer):
    """collectors.py:535-540."""
    _index = 2


class HAWQAggregator:
    """collectors.py:924-936 (the running division is reference behaviour)."""

    def __init__(self, num_samples: Optional[int] = None):
        self._num_samples = num_samples
        self._collected = 0
        self._container = 0.0

    def register_reduced_input(self, x: torch.Tensor) -> None:
        if self._num_samples is not None and self._collected >= self._num_samples:
            return
        _, stats = activation_stats(x, want_mean=False)
        self._container = (self._container + stats[3]) / x.numel()
        self._collected += 1

    def aggregate(self):
        return self._container * 2 / self._collected


@dataclass
class WCTensorStatistic:
    """statistics.py:260-307."""
    mean_values: List[torch.Tensor] = field(default_factory=list)
    shape_values: List[Tuple[int, ...]] = field(default_factory=list)


def token_means(activations: Sequence[torch.Tensor]) -> List[torch.Tensor]:
    """Per-item token mean [hidden] of many activation samples [..., tokens, hidden] (flattened to [T_i, hidden]) in
    one pair of launches per 64 samples (`nncf_amd_token_mean_batched`); values in the activation dtype, as
    MeanReducer returns them."""
    import ctypes

    acts = [a if a.is_contiguous() else a.contiguous() for a in activations]
    if not acts:
        return []
    _lib.require_gpu(acts[0], "token_means")
    dev, dt, hidden = acts[0].device, acts[0].dtype, acts[0].shape[-1]
    for a in acts:
        if a.device != dev or a.dtype != dt or a.shape[-1] != hidden or a.numel() == 0:
            raise InternalError("token_means: samples of one dtype, device and hidden size expected")
    n = len(acts)
    tokens = [a.numel() // hidden for a in acts]
    lib = _lib.lib()
    means = torch.empty((n, hidden), dtype=torch.float32, device=dev)
    nbytes = lib.nncf_amd_token_mean_workspace(n, max(tokens), hidden)
    ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=dev)
    xs = (ctypes.c_void_p * n)(*[a.data_ptr() for a in acts])
    tk = (ctypes.c_int64 * n)(*tokens)
    with torch.cuda.device(dev):
        _lib.check(lib.nncf_amd_token_mean_batched(n, xs, _lib.dtype_id(acts[0]), tk, hidden, _lib.ptr(means), _lib.ptr(ws),
                                                   nbytes, _lib.stream_of(acts[0])))
    means = means.to(dt)
    return [means[i] for i in range(n)]


def activations_to_wc_statistics(activations: Sequence[torch.Tensor]) -> WCTensorStatistic:
    """gptq.py:381-396."""
    acts = list(activations)
    shapes = [tuple(a.shape) for a in acts]
    uniform = len(acts) > 1 and all(a.dtype == acts[0].dtype and a.shape[-1] == acts[0].shape[-1] and a.device == acts[0].device
                                    for a in acts)
    if uniform:
        return WCTensorStatistic(token_means(acts), shapes)
    means = []
    for act in acts:
        m, _ = activation_stats(act.reshape(1, -1, act.shape[-1]), want_stats=False)
        means.append(m.to(act.dtype))
    return WCTensorStatistic(means, shapes)


def process_stats(stats: WCTensorStatistic, subset_size: int, act_ch_axis: int = -1):
    """activation_stats.py:20-57 (tiny; torch on device).  The Linears that read one activation (q / k / v, gate / up)
    ask for the same (s, X) of the same statistics object: the result is kept on the object per (subset_size, axis)
    while its lists are unchanged (same tensors, same shapes); callers treat s and X as read-only."""
    cached = getattr(stats, "_processed", None)
    means, shapes = stats.mean_values, stats.shape_values
    if (cached is not None and cached[0] == (subset_size, act_ch_axis) and len(cached[1]) == len(means)
            and all(a is b and v == b._version for (a, v), b in zip(cached[1], means)) and cached[2] == list(map(tuple, shapes))):
        return cached[3]
    result = _process_stats(stats, subset_size, act_ch_axis)
    # (the tensors themselves are kept, not their ids: a kept tensor cannot be freed and its address handed to another)
    stats._processed = ((subset_size, act_ch_axis), [(m, m._version) for m in means], list(map(tuple, shapes)), result)
    return result


def _process_stats(stats: WCTensorStatistic, subset_size: int, act_ch_axis: int = -1):
    X = torch.stack(list(stats.mean_values))
    axes = list(range(1, X.ndim)) + [0]
    X_full = X.permute(axes)
    if X_full.shape[-1] > subset_size and subset_size > 0:
        lens = [reduce(mul, tuple(s[:act_ch_axis]) + tuple(s[act_ch_axis % len(s) + 1:]), 1) for s in stats.shape_values]
        step = X_full.shape[-1] // subset_size
        idxs = [i[0] for i in sorted(enumerate(lens), key=lambda v: -v[1])][::step]
        X = X_full[..., idxs]
    else:
        X = X_full
    s = X_full.abs().amax(dim=-1)
    return s, X
