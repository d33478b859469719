"""Model-level weight compression for torch models on an AMD GPU.

`compress_weights` keeps the signature, defaults, validation and per-weight decisions of `nncf.compress_weights`
on the PyTorch backend (src/nncf/quantization/quantize_model.py:396-696, algorithm.py:74-295, 538-574, 961-1232)
and lowers every weight through the HIP path (`nncf_amd.weight_lowering`).  What differs, on purpose:

* no NNCF graph: the model is traced by ONE hooked forward of the first dataset item (execution order of the
  nn.Linear / nn.Embedding modules, tied weights, which Linear inputs come from a multiply or an activation);
  ignored scopes name modules, not graph nodes;
* NF4 / MXFP4 / gptq are not rejected (the reference's Torch gate, quantize_model.py:504-525, torch_backend.py:
  444-452): NF4 / MXFP4 decompressors and a layer-wise GPTQ feeder exist here;
* all statistics of one activation (token mean for AWQ / scale estimation, the mixed-precision statistic) come from
  ONE fused reducer launch per sample, shared by every Linear that reads that activation;
* the compressed weight is stored as the module's own parameter behind a `torch.nn.utils.parametrize`
  parametrization whose module is the decompressor (the reference uses its function-hook storage for that).
"""
from __future__ import annotations

import re
from collections import OrderedDict
from dataclasses import dataclass, replace
from functools import reduce
from operator import mul
from typing import Any, Dict, List, Optional, Tuple

import torch
import torch.nn as nn
from torch.nn.utils import parametrize

from nncf_amd import awq as awq_mod
from nncf_amd import decompressors as dcmp
from nncf_amd import gptq as gptq_mod
from nncf_amd import scale_estimation as se_mod
from nncf_amd import weight_lowering as wl
from nncf_amd import _lib
from nncf_amd.errors import (InternalError, InvalidGroupSizeError, ParameterNotSupportedError,
                             UnsupportedBackendError, ValidationError)
from nncf_amd.parameters import (AdvancedCompressionParameters, BackupMode, CompressedWeight, CompressWeightsMode,
                                 Dataset, GroupSizeFallbackMode, IgnoredScope, SensitivityMetric,
                                 WeightCompressionConfig, _mode_of, enum_of)
from nncf_amd.statistics import WCTensorStatistic, activation_stats

INT8_MODES = (CompressWeightsMode.INT8_SYM, CompressWeightsMode.INT8_ASYM)
SUPPORTED_MODES = INT8_MODES + (CompressWeightsMode.INT4_SYM, CompressWeightsMode.INT4_ASYM, CompressWeightsMode.NF4,
                                CompressWeightsMode.MXFP4)
SUPPORTED_DATA_TYPES = (torch.float16, torch.bfloat16, torch.float32, torch.float64)  # algorithm.py:64-71
THE_LOWEST_SENSITIVITY = 0  # mixed_precision.py:37
_ACTIVATION_GRAD_FNS = ("SiluBackward", "GeluBackward", "ReluBackward", "SigmoidBackward", "TanhBackward",
                        "EluBackward", "LeakyReluBackward", "HardswishBackward", "HardsigmoidBackward",
                        "MishBackward", "SoftplusBackward", "HardtanhBackward", "PreluKernelBackward")
_STAT_INDEX = {SensitivityMetric.MEAN_ACTIVATION_VARIANCE: 0, SensitivityMetric.MAX_ACTIVATION_VARIANCE: 1,
               SensitivityMetric.MEAN_ACTIVATION_MAGNITUDE: 2, SensitivityMetric.HESSIAN_INPUT_ACTIVATION: 3}


# ------------------------------------------------------------------------------------------------ configuration
def get_weight_compression_configuration(mode, dataset, ratio, group_size, all_layers, awq, scale_estimation, gptq,
                                         ignored_scope, sensitivity_metric, backup_mode, advanced_parameters):
    """algorithm.py:74-137."""
    if group_size is None:
        if mode in INT8_MODES:
            group_size = -1
        elif mode == CompressWeightsMode.MXFP4:
            group_size = 32
        else:
            group_size = 128
    if backup_mode is None:
        backup_mode = BackupMode.MXFP8_E4M3 if mode == CompressWeightsMode.MXFP4 else BackupMode.INT8_ASYM
    if sensitivity_metric is None:
        sensitivity_metric = (SensitivityMetric.WEIGHT_QUANTIZATION_ERROR if dataset is None
                              else SensitivityMetric.MAX_ACTIVATION_VARIANCE)
    return dict(mode=mode, ratio=ratio or 1, group_size=group_size, all_layers=all_layers or False, awq=awq or False,
                scale_estimation=scale_estimation or False, gptq=gptq or False,
                ignored_scope=ignored_scope or IgnoredScope(), sensitivity_metric=sensitivity_metric,
                backup_mode=backup_mode, advanced_parameters=advanced_parameters or AdvancedCompressionParameters())


def check_user_compression_configuration(mode, subset_size, dataset, ratio, group_size, all_layers, awq,
                                         scale_estimation, gptq, lora_correction, ignored_scope, sensitivity_metric,
                                         backup_mode, advanced_parameters) -> None:
    """algorithm.py:140-295 (the checks of the modes this build covers), same messages."""
    if mode in INT8_MODES:
        if (ratio and ratio != 1) or (group_size and group_size != -1):
            msg = ("INT8 modes require per-channel quantization of all layers in 8 bit. "
                   "Default values of `ratio` (1) and `group_size` (-1) cannot be overridden.")
            raise ParameterNotSupportedError(msg)
        if advanced_parameters and advanced_parameters.statistics_path:
            msg = "INT8 modes do not support the `statistics_path` option in `AdvancedCompressionParameters`."
            raise ParameterNotSupportedError(msg)
        options = {"all_layers": all_layers, "sensitivity_metric": sensitivity_metric, "dataset": dataset, "awq": awq,
                   "scale_estimation": scale_estimation, "gptq": gptq, "lora_correction": lora_correction,
                   "backup_mode": backup_mode}
        unsupported = [name for name, value in options.items() if value is not None]
        if unsupported:
            msg = f"INT8 modes do not support {', '.join(unsupported)} option(s). Set them to None."
            raise ParameterNotSupportedError(msg)
    if ratio is not None and not (0 <= ratio <= 1):
        msg = f"The ratio should be between 0 and 1, but ratio={ratio} is specified."
        raise ValidationError(msg)
    sizes = [subset_size]
    if advanced_parameters:
        sizes += [advanced_parameters.awq_params.subset_size, advanced_parameters.scale_estimation_params.subset_size,
                  advanced_parameters.gptq_params.subset_size]
    for size in sizes:
        if size <= 0:
            msg = f"The subset_size value should be positive, but subset_size={size} is given."
            raise ValidationError(msg)
    if (ratio and dataset is None and sensitivity_metric is not None
            and sensitivity_metric != SensitivityMetric.WEIGHT_QUANTIZATION_ERROR):
        msg = f"Mixed precision selection with sensitivity metric={sensitivity_metric.value} \
            requires a dataset, but it's not provided."
        raise ValidationError(msg)
    if mode == CompressWeightsMode.MXFP4:
        if group_size not in (None, 32):
            msg = f"MXFP4 and MXFP8_E4M3 types only support group size of 32, group size of {group_size} is given"
            raise ValidationError(msg)
        if advanced_parameters and advanced_parameters.group_size_fallback_mode is GroupSizeFallbackMode.ADJUST:
            msg = ("MXFP4, MXFP8_E4M3 and NVFP4 types do not support the group size"
                   f" fallback mode {advanced_parameters.group_size_fallback_mode.value}."
                   " Please use other group size fallback mode.")
            raise ValidationError(msg)


# ------------------------------------------------------------------------------------------------------ tracing
@dataclass
class WeightCompressionParameters:
    """config.py:138-167 with the graph node replaced by the module."""
    weight_name: str
    module_name: str
    module: nn.Module
    kind: str  # "matmul" | "embedding"
    weight_dtype: torch.dtype
    weight_shape: Tuple[int, ...]
    reduction_axes: Tuple[int, ...]
    compression_config: Optional[WeightCompressionConfig]
    order: int = 0

    @property
    def num_weights(self) -> int:
        return reduce(mul, self.weight_shape, 1)


@dataclass
class _Trace:
    order: List[str]  # module names (Linear / Embedding) in order of first execution
    act_shape: Dict[str, Tuple[int, ...]]  # Linear name -> shape of its input activation
    act_key: Dict[str, int]  # Linear name -> id of its input activation (Linears reading one tensor share it)
    producer: Dict[str, Tuple[str, Optional[str]]]  # Linear name -> ("mul_linear", src) | ("act", None) | ("act_mul", None)


class _EarlyExit(Exception):
    pass


def _to_device(item, device):
    if isinstance(item, torch.Tensor):
        return item.to(device)
    if isinstance(item, dict):
        return {k: _to_device(v, device) for k, v in item.items()}
    if isinstance(item, (list, tuple)):
        return type(item)(_to_device(v, device) for v in item)
    return item


def _forward(model: nn.Module, item):
    """wrap_model's example-input convention (src/nncf/torch/model_creation.py): dict -> kwargs, tuple -> args."""
    if isinstance(item, dict):
        return model(**item)
    if isinstance(item, tuple):
        return model(*item)
    return model(item)


def _model_device(model: nn.Module) -> torch.device:
    for p in model.parameters():
        return p.device
    return torch.device("cpu")


def _weighted_modules(model: nn.Module) -> "OrderedDict[str, nn.Module]":
    res = OrderedDict()
    for name, mod in model.named_modules():
        if isinstance(mod, (nn.Linear, nn.Embedding)) and isinstance(getattr(mod, "weight", None), nn.Parameter):
            res[name] = mod
    return res


def trace_model(model: nn.Module, example_input, with_provenance: bool = False) -> _Trace:
    """One hooked forward: execution order, activation shapes and sharing, and (for AWQ, awq_patterns.py:20-69)
    what produced every Linear input -- read off the autograd graph: `x.grad_fn` is a multiply one of whose inputs
    is the output of another Linear (MatMul -> Multiply -> MatMul), an activation function (Act -> MatMul), or a
    multiply fed by an activation function (Act -> Multiply -> MatMul)."""
    mods = _weighted_modules(model)
    names = {id(m): n for n, m in mods.items()}
    order: List[str] = []
    act_shape: Dict[str, Tuple[int, ...]] = {}
    act_key_raw: Dict[str, Any] = {}
    in_fn: Dict[str, Any] = {}
    out_fn: Dict[Any, str] = {}
    keepalive: List[torch.Tensor] = []
    handles = []

    def pre(mod, args, kwargs):
        name = names[id(mod)]
        if name in act_shape or name in order:
            return
        order.append(name)
        x = args[0] if args else next(iter(kwargs.values()))
        if isinstance(mod, nn.Linear) and isinstance(x, torch.Tensor):
            act_shape[name] = tuple(x.shape)
            act_key_raw[name] = (x.data_ptr(), tuple(x.shape), tuple(x.stride()), x.dtype, x._version)
            keepalive.append(x)  # or the allocator hands the address to a later activation of the same shape
            in_fn[name] = x.grad_fn

    def post(mod, args, output):
        if isinstance(mod, nn.Linear) and isinstance(output, torch.Tensor) and output.grad_fn is not None:
            out_fn.setdefault(output.grad_fn, names[id(mod)])

    for m in mods.values():
        handles.append(m.register_forward_pre_hook(pre, with_kwargs=True))
        if with_provenance:
            handles.append(m.register_forward_hook(post))
    flags = None
    try:
        if with_provenance:
            flags = [(p, p.requires_grad) for p in model.parameters()]
            for p, _ in flags:
                if p.is_floating_point():
                    p.requires_grad_(True)
            with torch.enable_grad():
                keep = _forward(model, example_input)  # noqa: F841  (keeps the autograd graph alive while it is read)
                producer = _read_producers(in_fn, out_fn)
            del keep
        else:
            with torch.no_grad():
                _forward(model, example_input)
            producer = {}
    finally:
        for h in handles:
            h.remove()
        if flags is not None:
            for p, f in flags:
                p.requires_grad_(f)
    keys: Dict[Any, int] = {}
    act_key = {n: keys.setdefault(k, len(keys)) for n, k in act_key_raw.items()}
    return _Trace(order, act_shape, act_key, producer)


def _is_act(fn) -> bool:
    return fn is not None and fn.name().startswith(_ACTIVATION_GRAD_FNS)


def _read_producers(in_fn, out_fn):
    producer = {}
    for name, fn in in_fn.items():
        if fn is None:
            continue
        if fn.name().startswith("MulBackward"):
            srcs = [f for f, _ in fn.next_functions if f is not None]
            lin = [out_fn[f] for f in srcs if f in out_fn]
            if lin:  # MatMul_Mul_MatMul is registered first and wins (awq.py:415-417)
                producer[name] = ("mul_linear", lin[0])
            elif any(_is_act(f) for f in srcs):
                producer[name] = ("act_mul", None)
        elif _is_act(fn):
            producer[name] = ("act", None)
    return producer


# -------------------------------------------------------------------------------------------- weight parameters
def _is_ignored(module_name: str, kind: str, scope: IgnoredScope) -> bool:
    if module_name in scope.names:
        return True
    if any(re.fullmatch(p, module_name) or re.match(p, module_name) for p in scope.patterns):
        return True
    return ("linear" if kind == "matmul" else "embedding") in scope.types


class WeightCompression:
    """algorithm.py:298-1232 for torch modules: parameter resolution, mixed precision, AWQ, scale estimation / GPTQ,
    lowering and decompressor insertion."""

    def __init__(self, mode, ratio, group_size, ignored_scope, all_layers, sensitivity_metric, awq, subset_size,
                 scale_estimation, gptq, backup_mode, advanced_parameters):
        self._mode, self._ratio, self._group_size = mode, ratio, group_size
        self._ignored_scope, self._all_layers = ignored_scope, all_layers
        self._sensitivity_metric, self._awq, self._subset_size = sensitivity_metric, awq, subset_size
        self._scale_estimation, self._gptq, self._backup_mode = scale_estimation, gptq, backup_mode
        self._adv = advanced_parameters
        self._fallback = enum_of(GroupSizeFallbackMode, advanced_parameters.group_size_fallback_mode)
        # algorithm.py:433-442
        self._data_aware_mixed_precision = (sensitivity_metric != SensitivityMetric.WEIGHT_QUANTIZATION_ERROR
                                            and ratio != 1.0)
        self._data_aware_compression = ((awq and advanced_parameters.awq_params.prefer_data_aware_scaling)
                                        or scale_estimation or gptq)
        # The reference collects the per-Linear token means whenever the compression is data-aware; with GPTQ alone
        # (scale estimation included: gptq.py:319-327 takes its statistics from the layer-wise inputs) nothing reads
        # them, so the statistics pass over the calibration set is skipped (one full forward per sample).
        self._needs_token_means = ((awq and advanced_parameters.awq_params.prefer_data_aware_scaling)
                                   or (scale_estimation and not gptq))
        self.bitwidth_report: Dict[str, Optional[WeightCompressionConfig]] = {}

    # ---- algorithm.py:576-620
    def _get_backup_config(self, weight_dtype) -> Optional[WeightCompressionConfig]:
        if self._backup_mode == BackupMode.NONE:
            return None
        if self._backup_mode == BackupMode.INT8_SYM:
            return WeightCompressionConfig(CompressWeightsMode.INT8_SYM, -1)
        if self._backup_mode == BackupMode.INT8_ASYM:
            return WeightCompressionConfig(CompressWeightsMode.INT8_ASYM, -1)
        msg = (f"Backup mode {self._backup_mode.value} (FP8) is outside this build's scope (SURVEY 2.1: "
               "fp8_conversion.py); pass backup_mode=INT8_ASYM, INT8_SYM or NONE.")
        raise ParameterNotSupportedError(msg)

    # ---- algorithm.py:961-1088
    def get_weight_compression_parameters(self, model: nn.Module, trace: _Trace):
        mods = _weighted_modules(model)
        nodes = [n for n in trace.order if n in mods]
        all_params: List[WeightCompressionParameters] = []
        skipped: List[WeightCompressionParameters] = []
        seen = set()
        is_last_layer_skipped = False
        n = len(nodes)
        for i, name in enumerate(nodes):
            mod = mods[name]
            kind = "matmul" if isinstance(mod, nn.Linear) else "embedding"
            is_last_layer = i == n - 1
            w = mod.weight
            if id(w) in seen:  # shared weight: processed once (algorithm.py:1001-1005)
                is_last_layer_skipped = is_last_layer
                continue
            wp = WeightCompressionParameters(name + ".weight", name, mod, kind, w.dtype, tuple(w.shape), (w.ndim - 1,),
                                             None, order=i)
            if not _is_ignored(name, kind, self._ignored_scope) and w.dtype in SUPPORTED_DATA_TYPES:
                wp.compression_config = self._get_backup_config(w.dtype)
                all_params.append(wp)
                seen.add(id(w))
            else:
                is_last_layer_skipped = is_last_layer
                skipped.append(wp)
        ratio_defining = self._get_ratio_defining_params(all_params, is_last_layer_skipped)
        if self._fallback == GroupSizeFallbackMode.IGNORE:
            all_params, ratio_defining, skipped = self._handle_ignore_group_size_fallback(all_params, ratio_defining,
                                                                                          skipped)
        if self._fallback == GroupSizeFallbackMode.ADJUST:
            ratio_defining, group_sizes = self._handle_adjust_group_size_fallback(ratio_defining)
        else:
            group_sizes = {wp.weight_name: self._group_size for wp in ratio_defining}
        for wp in ratio_defining:
            wp.compression_config = WeightCompressionConfig(self._mode, group_sizes[wp.weight_name])
        return all_params, ratio_defining, skipped

    # ---- algorithm.py:538-574
    def _get_ratio_defining_params(self, all_params, is_last_layer_skipped):
        if self._mode in INT8_MODES:
            return all_params
        res = [wp for wp in all_params if wp.kind == "matmul"]
        if not self._all_layers and not is_last_layer_skipped:
            res = res[:-1]
        if self._all_layers:
            res.extend(wp for wp in all_params if wp.kind == "embedding" and len(wp.reduction_axes) == 1)
        return res

    # ---- algorithm.py:679-767
    def _handle_ignore_group_size_fallback(self, all_params, ratio_defining, skipped):
        if self._group_size == -1:
            return all_params, ratio_defining, skipped
        bad = set()
        for wp in ratio_defining:
            if wp.weight_shape[wp.reduction_axes[0]] % self._group_size != 0:
                bad.add(wp.weight_name)
                skipped.append(replace(wp, compression_config=None))
        if bad:
            ratio_defining = [wp for wp in ratio_defining if wp.weight_name not in bad]
            all_params = [wp for wp in all_params if wp.weight_name not in bad]
        return all_params, ratio_defining, skipped

    def _handle_adjust_group_size_fallback(self, weight_params):
        if self._group_size == -1:
            return weight_params, {wp.weight_name: self._group_size for wp in weight_params}
        sizes, valid = {}, []
        for wp in weight_params:
            r = wp.weight_shape[wp.reduction_axes[0]]
            if r % self._group_size == 0:
                valid.append(wp)
                sizes[wp.weight_name] = self._group_size
                continue
            adjusted = r & (~r + 1)  # the largest power of two that divides r
            if adjusted >= self._adv.min_adjusted_group_size:
                valid.append(wp)
                sizes[wp.weight_name] = adjusted
        return valid, sizes

    # ---- algorithm.py:652-677
    def validate_group_size(self, ratio_defining) -> None:
        failed = []
        for wp in ratio_defining:
            cfg = wp.compression_config
            if cfg is None or cfg.group_size == -1:
                continue
            r = wp.weight_shape[wp.reduction_axes[0]]
            if r % cfg.group_size != 0:
                failed.append((wp.module_name, r))
        if failed:
            names = "\n\t".join(f'"{name}" (channel size: {size})' for name, size in failed)
            msg = (f"Failed to apply group-wise quantization with group size value {self._group_size}.\n"
                   "Ensure that the group size is divisible by the channel size, "
                   "or consider setting `group_size_fallback_mode` to IGNORE or ADJUST. Failed nodes:\n\t" + names)
            raise InvalidGroupSizeError(msg)

    # ---- statistics (algorithm.py:903-936, 1319-1423; torch/function_hook/statistics/aggregator.py:34-80)
    def collect_statistics(self, model, dataset, trace, weight_params):
        """One fused reducer launch per (activation, sample): the token mean every data-aware algorithm reads and the
        mixed-precision statistic, shared by all Linears that read the activation (q/k/v, gate/up)."""
        targets = [wp for wp in weight_params if wp.kind == "matmul"]
        if not targets:
            return {}, {}
        by_name = {wp.module_name: wp for wp in targets}
        want_mean = bool(self._needs_token_means)
        want_stat = bool(self._data_aware_mixed_precision)
        if want_stat:
            for wp in targets:
                if len(trace.act_shape.get(wp.module_name, ())) < 2:
                    msg = ("Data-aware mixed precision criteria are not supported for MatMuls with 1D inputs. "
                           f"Node: {wp.module_name}, number of dimensions: {len(trace.act_shape.get(wp.module_name, ()))}.")
                    raise RuntimeError(msg)
        means: Dict[str, WCTensorStatistic] = {n: WCTensorStatistic() for n in by_name}
        scalars: Dict[str, List[torch.Tensor]] = {n: [] for n in by_name}
        hawq: Dict[str, Any] = {n: 0.0 for n in by_name}
        stat_idx = _STAT_INDEX.get(self._sensitivity_metric)
        cache: Dict[Any, Any] = {}
        mean_dtype: Dict[str, torch.dtype] = {}
        names = {id(wp.module): wp.module_name for wp in targets}

        def pre(mod, args, kwargs):
            name = names[id(mod)]
            x = args[0] if args else next(iter(kwargs.values()))
            key = (x.data_ptr(), tuple(x.shape), tuple(x.stride()), x.dtype, x._version)
            hit = cache.get(key)
            if hit is None:
                mean, stats = activation_stats(x, want_mean=want_mean, want_stats=want_stat)
                hit = cache[key] = (mean, stats, x)  # x kept alive so that its address is not reused within the pass
            mean, stats, _ = hit
            if mean is not None:
                # the fp32 mean as the kernel wrote it, shared by the Linears that read this activation; rounded to
                # the activation's dtype after the pass (the hook's host time is not hidden behind the forward)
                means[name].mean_values.append(mean)
                means[name].shape_values.append(tuple(x.shape))
                mean_dtype[name] = x.dtype if x.dtype.is_floating_point else torch.float32
            if stats is not None:
                if self._sensitivity_metric == SensitivityMetric.HESSIAN_INPUT_ACTIVATION:
                    hawq[name] = (hawq[name] + stats[3]) / x.numel()  # collectors.py:929-933
                else:
                    scalars[name].append(stats[stat_idx].to(x.dtype))

        handles = [wp.module.register_forward_pre_hook(pre, with_kwargs=True) for wp in targets]
        n_samples = 0
        device = _model_device(model)
        try:
            with torch.no_grad():
                for item in dataset.get_inference_data():
                    if n_samples >= self._subset_size:
                        break
                    cache.clear()
                    _forward(model, _to_device(item, device))
                    n_samples += 1
        finally:
            cache.clear()
            for h in handles:
                h.remove()
        if n_samples == 0:
            raise ValidationError("Calibration dataset must not be empty.")
        # torch.mean accumulates 16-bit floats in fp32 and rounds once to the dtype: one stack + cast per distinct
        # activation; the Linears that read it get the rows of the same buffer (nothing downstream writes into them)
        rounded: Dict[Any, List[torch.Tensor]] = {}
        for name, st in means.items():
            if not st.mean_values:
                continue
            key = (tuple(id(m) for m in st.mean_values), mean_dtype[name])
            if key not in rounded:
                rounded[key] = list(torch.stack(st.mean_values).to(mean_dtype[name]).unbind(0))
            st.mean_values = list(rounded[key])
        activation_scores: Dict[str, float] = {}
        if want_stat:
            for name in by_name:
                if self._sensitivity_metric == SensitivityMetric.HESSIAN_INPUT_ACTIVATION:
                    activation_scores[name] = float(hawq[name] * 2 / n_samples)  # collectors.py:935-936
                else:
                    activation_scores[name] = float(torch.stack(scalars[name]).mean(dim=0))  # MeanAggregator
        return (means if want_mean else {}), activation_scores

    # ---- mixed_precision.py:71-100, 161-199, 254-271, 341-409
    def apply_mixed_precision(self, ratio_defining, activation_scores) -> None:
        if not (self._ratio < 1 and len(ratio_defining) > 0):
            return
        backup = WeightCompressionConfig()  # INT8_ASYM per-channel
        eps = torch.finfo(torch.float32).eps
        metric = self._sensitivity_metric
        # weight part of every score: all launches queued, one synchronisation
        scored = [wp for wp in ratio_defining
                  if not (metric != SensitivityMetric.WEIGHT_QUANTIZATION_ERROR and wp.kind == "embedding")]
        reduction = "frobenius" if metric == SensitivityMetric.HESSIAN_INPUT_ACTIVATION else "max_mean"
        errs = dict(zip((wp.weight_name for wp in scored), wl.get_integer_quantization_errors(
            [wp.module.weight.data for wp in scored], [wp.reduction_axes for wp in scored], backup, reduction)))
        scores = []
        for wp in ratio_defining:
            if wp.weight_name not in errs:
                scores.append(THE_LOWEST_SENSITIVITY)
                continue
            err = errs[wp.weight_name]
            weight_score = err if metric == SensitivityMetric.HESSIAN_INPUT_ACTIVATION else 1 / (err + eps)
            if metric == SensitivityMetric.WEIGHT_QUANTIZATION_ERROR:
                scores.append(weight_score)
            else:
                scores.append(weight_score * activation_scores[wp.module_name])
        self.sensitivity_scores = {wp.module_name: s for wp, s in zip(ratio_defining, scores)}
        num_all = sum(wp.num_weights for wp in ratio_defining)
        ascending = [i for i, _ in sorted(enumerate(scores), key=lambda v: v[1])]
        primary = set()
        in_4bit = 0
        for index in ascending:
            wp = ratio_defining[index]
            if (in_4bit + wp.num_weights) / num_all >= self._ratio:
                break
            primary.add(wp.weight_name)
            in_4bit += wp.num_weights
        for wp in ratio_defining:
            if wp.weight_name not in primary:
                wp.compression_config = self._get_backup_config(wp.weight_dtype)

    # ---- awq.py:124-241, 368-432
    def apply_awq(self, trace, all_params, statistics):
        adv = self._adv.awq_params
        by_name = {wp.module_name: wp for wp in all_params}
        is_data_free = not statistics or not adv.prefer_data_aware_scaling
        scale_per_target: Dict[str, torch.Tensor] = {}
        self.awq_scale_per_target = {}  # activation scales in the reference's shapes (awq.py:237 _scale_per_target_node)
        # targets in the reference's match order: patterns in registration order (awq_patterns.py:20-69, awq.py:386-421)
        rank = {"mul_linear": 0, "act_mul": 1, "act": 2}
        matched = [wp for wp in all_params if wp.module_name in trace.producer and wp.kind == "matmul"
                   and wp.compression_config.num_bits == 4]
        matched.sort(key=lambda wp: (rank[trace.producer[wp.module_name][0]], wp.order))
        for wp in matched:
            kind, src = trace.producer[wp.module_name]
            weight = wp.module.weight.data
            weight_dtype = weight.dtype
            w32 = weight.to(torch.float32)
            act_shape = trace.act_shape[wp.module_name]
            act_ch_axis = len(act_shape) - 1
            merge_wp = by_name.get(src) if kind == "mul_linear" else None
            is_mergeable = kind == "mul_linear" and src is not None and w32.ndim == 2
            red = wp.reduction_axes[0]
            scale_axis = w32.ndim * 2 - 3 - red  # awq.py:189
            if is_data_free:
                scale = awq_mod.data_free_step(w32, scale_axis)
            else:
                prev_weight = prev_stats = None
                if is_mergeable:
                    src_mod = merge_wp.module if merge_wp is not None else _weighted_modules(self._model)[src]
                    prev_weight = src_mod.weight.data
                    prev_stats = statistics[src] if src in statistics else None
                scale = awq_mod.data_aware_step(w32, wp.reduction_axes, wp.compression_config,
                                                statistics[wp.module_name], act_ch_axis, adv.subset_size,
                                                adv.percent_to_apply, adv.alpha_min, adv.alpha_max, adv.steps,
                                                prev_weight, prev_stats)
            w_scale = scale.unsqueeze(scale_axis)
            a_scale = 1.0 / scale
            wp.module.weight.data = (w32 * w_scale).to(weight_dtype)
            if is_mergeable:  # the activation scale goes into the rows of the producing Linear (awq.py:209-214)
                src_mod = merge_wp.module if merge_wp is not None else _weighted_modules(self._model)[src]
                src_mod.weight.data = (src_mod.weight.data * a_scale.unsqueeze(red)).to(weight_dtype)
                # (a bias of the producing Linear is left as it is: awq.py:210-214 scales the weight only)
                self.awq_scale_per_target[wp.module_name] = a_scale.unsqueeze(red)
            else:  # Act -> [Multiply ->] MatMul: a multiply by 1/scale in front of the Linear (awq.py:215-235)
                shape = tuple(a_scale.shape[-1] if d == act_ch_axis else 1 for d in range(len(act_shape)))
                a_scale = a_scale.reshape(shape)
                insert_sq_multiply(wp.module, a_scale)
                self.awq_scale_per_target[wp.module_name] = a_scale
            scale_per_target[wp.module_name] = a_scale
        # awq.py:424-432 (out of place: Linears that share an activation share the stored mean tensors here)
        if statistics:
            for name, a_scale in scale_per_target.items():
                st = statistics[name]
                st.mean_values = [m * a_scale.reshape(-1).to(m.dtype) for m in st.mean_values]
        return statistics

    # ---- scale_estimation.py:97-158
    def apply_scale_estimation(self, all_params, statistics) -> Dict[str, CompressedWeight]:
        p = self._adv.scale_estimation_params
        res = {}
        for wp in all_params:
            cfg = wp.compression_config
            if cfg.num_bits != 4 or wp.module_name not in statistics:
                res[wp.weight_name] = CompressedWeight()
                continue
            act_ch_axis = len(self._trace.act_shape[wp.module_name]) - 1
            scale, zp = se_mod.calculate_quantization_params(statistics[wp.module_name], wp.module.weight.data,
                                                             wp.reduction_axes, cfg, act_ch_axis, p.subset_size,
                                                             p.initial_steps, p.scale_steps, p.weight_penalty)
            res[wp.weight_name] = CompressedWeight(None, scale, zp, None)
        self.se_scales = {k: v.scale for k, v in res.items() if v.scale is not None}  # fp32, as estimated
        return res

    # ---- gptq.py:82-171 with a torch layer-wise feeder in place of the OpenVINO-only LayerwiseEngine
    def apply_gptq(self, model, dataset, trace, all_params) -> Dict[str, CompressedWeight]:
        from nncf_amd.layerwise import LayerwiseFeeder

        p = self._adv.gptq_params
        targets = [wp for wp in all_params if wp.kind == "matmul"]
        feeder = LayerwiseFeeder(model, dataset, p.subset_size, trace, [wp.module_name for wp in targets])
        by_name = {wp.module_name: wp for wp in targets}
        res: Dict[str, CompressedWeight] = {}
        feed = iter(feeder)
        try:
            self._gptq_groups(feed, by_name, p, res)
        finally:
            feed.close()  # ends the suspended calibration forwards now, also when a group failed
        return res

    def _gptq_groups(self, feed, by_name, p, res) -> None:
        for group, hessian, stats in feed:
            # every Linear of `group` reads the same activation: ONE Hessian, ONE Cholesky chain (and one set of token
            # means) serve all of them
            members = [by_name[name] for name in group if by_name[name].compression_config.mode not in INT8_MODES]
            if not members:
                continue
            hinv, dead = gptq_mod.prepare_hinv(hessian, p.damp_percent)
            del hessian
            for wp in members:
                weight = wp.module.weight.data
                w32 = weight.to(torch.float32).contiguous()
                if w32.data_ptr() == weight.data_ptr():
                    w32 = w32.clone()
                q, scale, zp = gptq_mod.quantize_weights(None, w32, wp.compression_config, None, p.damp_percent,
                                                         p.block_size, self._scale_estimation, wp.reduction_axes,
                                                         input_stats=stats, hinv=hinv, dead=dead)
                wp.module.weight.data = q.to(weight.dtype)
                res[wp.weight_name] = CompressedWeight(None, scale, zp, None)

    # ---- torch_backend.py:352-485
    def transform_model(self, all_params, precomputed: Optional[Dict[str, CompressedWeight]]) -> None:
        precomputed = precomputed or {}
        plain: Dict[Any, List[WeightCompressionParameters]] = {}
        results: Dict[str, CompressedWeight] = {}
        for wp in sorted(all_params, key=lambda wp: wp.num_weights, reverse=True):
            pre = precomputed.get(wp.weight_name)
            w = wp.module.weight.data
            if pre is not None and pre.scale is not None:
                results[wp.weight_name] = _with_e8m0(wl.compress_weight_packed(w, wp.reduction_axes,
                                                                              wp.compression_config, pre),
                                                    wp.compression_config)
            elif w.ndim == 2 and w.dtype != torch.float64:
                plain.setdefault((wp.compression_config.mode, wp.compression_config.group_size, w.dtype, w.device),
                                 []).append(wp)
            else:
                results[wp.weight_name] = _with_e8m0(wl.compress_weight_packed(w, wp.reduction_axes,
                                                                              wp.compression_config),
                                                    wp.compression_config)
        for (_, _, _, _), wps in plain.items():  # data-free weights of one config: multi-tensor launches
            cfg = wps[0].compression_config
            outs = wl.compress_weights_packed_batched([wp.module.weight.data for wp in wps], cfg,
                                                      want_e8m0=cfg.mode == CompressWeightsMode.MXFP4)
            for wp, cw in zip(wps, outs):
                results[wp.weight_name] = cw
        owners: Dict[int, List[nn.Module]] = {}  # weight parameter -> the modules that hold it (tied weights: several)
        for mod in self._model.modules():
            w = getattr(mod, "_parameters", {}).get("weight")
            if w is not None:
                owners.setdefault(id(w), []).append(mod)
        for wp in all_params:
            insert_decompressor(self._model, wp, results[wp.weight_name], owners)

    def _statistics_tag(self) -> str:
        """What a statistics cache file must have been collected for to be reusable."""
        return (f"subset={self._subset_size};mean={int(bool(self._needs_token_means))};"
                f"metric={self._sensitivity_metric.value if self._data_aware_mixed_precision else 'none'}")

    # ---- algorithm.py:1090-1232
    def apply(self, model: nn.Module, dataset: Optional[Dataset], example_input) -> nn.Module:
        self._model = model
        self._trace = trace = trace_model(model, example_input, with_provenance=self._awq)
        all_params, ratio_defining, skipped = self.get_weight_compression_parameters(model, trace)
        statistics, activation_scores = None, {}
        if dataset is not None and (self._data_aware_mixed_precision or self._needs_token_means):
            stat_params = ratio_defining if self._backup_mode == BackupMode.NONE else all_params
            cache = self._adv.statistics_path
            cached = load_statistics(cache, _model_device(model), self._statistics_tag()) if cache else None
            if cached is not None:
                statistics, activation_scores = cached
            else:
                statistics, activation_scores = self.collect_statistics(model, dataset, trace, stat_params)
                if cache:
                    save_statistics(cache, statistics, activation_scores, self._statistics_tag())
            statistics = statistics or None
        self.activation_scores = activation_scores
        self.apply_mixed_precision(ratio_defining, activation_scores)
        self.validate_group_size(ratio_defining)
        self.bitwidth_report = {wp.module_name: wp.compression_config for wp in all_params + skipped}
        all_params = [wp for wp in all_params if wp.compression_config is not None]
        if self._awq:
            statistics = self.apply_awq(trace, all_params, statistics)
        precomputed = None
        if self._gptq:
            precomputed = self.apply_gptq(model, dataset, trace, all_params)
        elif self._scale_estimation:
            precomputed = self.apply_scale_estimation(all_params, statistics or {})
        self.transform_model(all_params, precomputed)
        model.requires_grad_(False)  # torch_backend.py:480
        return model


def _with_e8m0(cw: CompressedWeight, cfg: WeightCompressionConfig) -> CompressedWeight:
    if cfg.mode == CompressWeightsMode.MXFP4 and cw.codebook is None:
        # the MX scale is a power of two: its E8M0 byte is the fp32 exponent field
        cw.codebook = ((cw.scale.contiguous().view(torch.int32) >> 23) & 0xFF).to(torch.uint8)
    return cw


class SQMultiply(nn.Module):
    """src/nncf/torch/quantization/layers.py SQMultiply: the AWQ activation scale in front of a Linear."""

    def __init__(self, scale_value: torch.Tensor):
        super().__init__()
        self.register_buffer("_scale_value", scale_value)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return x * self._scale_value.to(x.dtype)


def _sq_pre_hook(mod, args):
    return (mod.awq_input_scale(args[0]),) + tuple(args[1:])


def insert_sq_multiply(module: nn.Module, a_scale: torch.Tensor) -> None:
    module.add_module("awq_input_scale", SQMultiply(a_scale))
    module.register_forward_pre_hook(_sq_pre_hook)


def make_decompressor(cfg: WeightCompressionConfig, cw: CompressedWeight, weight_shape, weight_dtype):
    """torch_backend.py:379-401 plus the NF4 / MXFP4 decompressors of this build."""
    mode = cfg.mode
    if mode == CompressWeightsMode.INT8_SYM:
        return dcmp.INT8SymmetricWeightsDecompressor(cw.scale, result_dtype=weight_dtype)
    if mode == CompressWeightsMode.INT8_ASYM:
        return dcmp.INT8AsymmetricWeightsDecompressor(cw.scale, cw.zero_point, result_dtype=weight_dtype)
    out = weight_shape[0]
    gs = cfg.group_size
    cshape = tuple(weight_shape) if gs == -1 else (out, weight_shape[1] // gs, gs)
    kw = dict(compressed_weight_shape=cshape, result_shape=tuple(weight_shape), result_dtype=weight_dtype)
    if mode == CompressWeightsMode.INT4_SYM:
        return dcmp.INT4SymmetricWeightsDecompressor(scale=cw.scale, **kw)
    if mode == CompressWeightsMode.INT4_ASYM:
        return dcmp.INT4AsymmetricWeightsDecompressor(scale=cw.scale, zero_point=cw.zero_point, **kw)
    if mode == CompressWeightsMode.NF4:
        return dcmp.NF4WeightsDecompressor(scale=cw.scale, **kw)
    if mode == CompressWeightsMode.MXFP4:
        return dcmp.MXFP4WeightsDecompressor(scale_e8m0=cw.codebook.reshape(cw.scale.shape), **kw)
    raise InternalError(f"no decompressor for {mode}")


def insert_decompressor(model: nn.Module, wp: WeightCompressionParameters, cw: CompressedWeight,
                        owners: Optional[Dict[int, List[nn.Module]]] = None) -> None:
    """torch_backend.py:403-422: the packed tensor becomes the module's parameter, the decompressor runs on every
    read of it.  Modules that share the parameter (tied embeddings) share the packed tensor and the decompressor.
    `owners`: id(weight parameter) -> modules holding it, built once by the caller (one walk over the model
    instead of one per weight)."""
    decompressor = make_decompressor(wp.compression_config, cw, wp.weight_shape, wp.weight_dtype)
    old = wp.module.weight
    packed = nn.Parameter(cw.tensor, requires_grad=False)
    if owners is not None:
        mods = owners.get(id(old), [])
    else:
        mods = [m for m in model.modules() if getattr(m, "_parameters", None) and m._parameters.get("weight") is old]
    for mod in mods:
        mod.weight = packed
        parametrize.register_parametrization(mod, "weight", decompressor, unsafe=True)


# ------------------------------------------------------------------------------------------------------ public
def compress_weights(model: nn.Module, *, mode=CompressWeightsMode.INT8_ASYM, ratio: Optional[float] = None,
                     group_size: Optional[int] = None, ignored_scope: Optional[IgnoredScope] = None,
                     all_layers: Optional[bool] = None, dataset: Optional[Dataset] = None, sensitivity_metric=None,
                     subset_size: int = 128, awq: Optional[bool] = None, scale_estimation: Optional[bool] = None,
                     gptq: Optional[bool] = None, lora_correction: Optional[bool] = None, backup_mode=None,
                     advanced_parameters: Optional[AdvancedCompressionParameters] = None) -> nn.Module:
    """`nncf.compress_weights` for a torch model that lives on an AMD GPU (quantize_model.py:409-696).  Compresses
    the model in place and returns it.  `dataset` (at least one item) is required, as in the reference's Torch
    backend: its first item traces the model."""
    mode = _mode_of(mode)
    if mode not in SUPPORTED_MODES:
        msg = f"nncf_amd supports {[m.value for m in SUPPORTED_MODES]} modes for weight compression."
        raise ParameterNotSupportedError(msg)
    if lora_correction is not None:
        msg = "Torch backend does not support lora_correction option(s). Set them to None."
        raise ParameterNotSupportedError(msg)
    sensitivity_metric = None if sensitivity_metric is None else enum_of(SensitivityMetric, sensitivity_metric)
    backup_mode = None if backup_mode is None else enum_of(BackupMode, backup_mode)
    if dataset is None:
        msg = "Please provide a dataset of at least one element for PyTorch model tracing."
        raise ValidationError(msg)
    if not isinstance(dataset, Dataset):
        dataset = Dataset(dataset)
    device = _model_device(model)
    if device.type != "cuda":
        raise UnsupportedBackendError("nncf_amd.compress_weights needs the model on an AMD GPU ('cuda' device); "
                                      "there is no CPU path in this build.")
    _lib.lib()  # a missing HIP library fails here, loudly
    try:
        example_input = _to_device(next(iter(dataset.get_inference_data())), device)
    except StopIteration:
        raise ValidationError("Please provide a dataset of at least one element for PyTorch model tracing.") from None
    if mode in INT8_MODES:
        dataset = None  # data-aware methods don't support INT8 modes (quantize_model.py:545-546)
    if any((scale_estimation, gptq)) and dataset is None:
        msg = "Scale estimation, GPTQ or Lora Correction algorithm is defined, but dataset is None."
        raise ParameterNotSupportedError(msg)
    if any((awq, scale_estimation, gptq)) and mode == CompressWeightsMode.MXFP4:
        msg = "AWQ, Scale estimation, GPTQ or Lora Correction algorithm is defined, but mode in ['mxfp4']."
        raise ParameterNotSupportedError(msg)
    user_dataset = dataset if mode not in INT8_MODES else None
    check_user_compression_configuration(mode, subset_size, user_dataset, ratio, group_size, all_layers, awq,
                                         scale_estimation, gptq, lora_correction, ignored_scope, sensitivity_metric,
                                         backup_mode, advanced_parameters)
    cfg = get_weight_compression_configuration(mode, dataset, ratio, group_size, all_layers, awq, scale_estimation,
                                               gptq, ignored_scope, sensitivity_metric, backup_mode,
                                               advanced_parameters)
    algo = WeightCompression(cfg["mode"], cfg["ratio"], cfg["group_size"], cfg["ignored_scope"], cfg["all_layers"],
                             cfg["sensitivity_metric"], cfg["awq"], subset_size, cfg["scale_estimation"], cfg["gptq"],
                             cfg["backup_mode"], cfg["advanced_parameters"])
    was_training = model.training
    model.eval()
    try:
        algo.apply(model, dataset, example_input)
    finally:
        model.train(was_training)
    model._nncf_amd_weight_compression = algo.bitwidth_report
    model._nncf_amd_awq_scales = getattr(algo, "awq_scale_per_target", {})
    model._nncf_amd_se_scales = getattr(algo, "se_scales", {})
    model._nncf_amd_mixed_precision = {"activation_scores": dict(getattr(algo, "activation_scores", {})),
                                       "scores": [float(v) for v in getattr(algo, "sensitivity_scores", {}).values()]}
    return model


# ----------------------------------------------------------------------------- statistics cache
def save_statistics(path: str, statistics: Dict[str, WCTensorStatistic], activation_scores: Dict[str, float],
                    tag: str) -> None:
    """The statistics of one calibration run as one safetensors file (the reference caches them per node as
    safetensors too, common/tensor_statistics/statistics_serializer.py:81-148, but rejects `statistics_path` on its
    Torch backend, quantize_model.py:527-529): AWQ / scale estimation / GPTQ / ratio sweeps over one model reuse
    them instead of re-running the calibration forward passes."""
    import json

    from safetensors.torch import save_file

    tensors, shapes = {}, {}
    for name, st in (statistics or {}).items():
        if st.mean_values:
            tensors[name] = torch.stack([m.reshape(-1) for m in st.mean_values]).contiguous()
            shapes[name] = [list(sh) for sh in st.shape_values]
    if not tensors:
        tensors["__empty__"] = torch.zeros(1)
    save_file(tensors, path, metadata={"nncf_amd_statistics": json.dumps(
        {"tag": tag, "shapes": shapes, "activation_scores": activation_scores})})


def load_statistics(path: str, device, tag: str):
    """(statistics, activation_scores) of `save_statistics`, or None if the file is absent or was collected for
    other settings."""
    import json
    import os

    from safetensors import safe_open

    if not os.path.isfile(path):
        return None
    with safe_open(path, framework="pt", device=str(device)) as fh:
        meta = json.loads(fh.metadata()["nncf_amd_statistics"])
        if meta["tag"] != tag:
            return None
        statistics = {}
        for name, shapes in meta["shapes"].items():
            stacked = fh.get_tensor(name)
            statistics[name] = WCTensorStatistic([stacked[i] for i in range(stacked.shape[0])],
                                                 [tuple(sh) for sh in shapes])
    return statistics, {k: float(v) for k, v in meta["activation_scores"].items()}


# ------------------------------------------------------------------------------------ on-disk export
_DTYPES = {"float32": torch.float32, "float16": torch.float16, "bfloat16": torch.bfloat16, "float64": torch.float64}


def _decompressor_classes():
    return {n: c for n, c in vars(dcmp).items()
            if isinstance(c, type) and issubclass(c, dcmp.BaseWeightsDecompressor) and c is not dcmp.BaseWeightsDecompressor
            and not n.startswith("_")}


_DECOMPRESSORS = _decompressor_classes()


def _reference_node_name(model: nn.Module, module_name: str, counters: Dict[str, int]) -> str:
    """Name the reference's tracer gives the call of an nn.Linear: the attribute names on the way to the module
    joined by "/", with nn.ModuleList containers and their indices left out (they are never called, so they do not
    show in the scope), then "/linear/<n>" with n counting the calls of that scope in model order -- e.g.
    blocks.2.mlp.fc2 -> "mlp/fc2/linear/0" when only block 2 has an fc2 [probed: tests/golden/model_tiny.npz].
    Other containers / call patterns are parity unpinned."""
    parts, cur, skip = [], model, False
    for comp in module_name.split("."):
        nxt = getattr(cur, comp) if not comp.isdigit() else cur[int(comp)]
        if isinstance(nxt, nn.ModuleList):
            skip = True  # the container's own name
        elif skip and comp.isdigit():
            skip = False  # the index inside it
        else:
            parts.append(comp)
            skip = False
        cur = nxt
    scope = "/".join(parts)
    n = counters.get(scope, 0)
    counters[scope] = n + 1
    return f"{scope}/linear/{n}"


def nncf_state_dict(model: nn.Module) -> Dict[str, torch.Tensor]:
    """state_dict of a model compressed by this module under the REFERENCE's keys (what `model.state_dict()` holds
    after the reference's nncf.compress_weights on the Torch backend [probed, SURVEY section 5 / 8 f4]; written by
    function_hook/serialization.py + the hook storage, torch_backend.py:352-422, layers.py:1303-1460):
        <module>.weight                                                   packed codes (uint8 / int8)
        __nncf_hooks.post_hooks.<module with ':' for '.'>:weight__0.0._scale          fp16 (E8M0 bytes for MXFP4)
        __nncf_hooks.post_hooks.<...>:weight__0.0._zero_point                       packed / uint8 zero points
        __nncf_hooks.pre_hooks.<node name>__0.0._scale_value                        AWQ activation scale
    Tensors are the ones the modules hold (no copies); weights shared by several modules appear under each module's
    `.weight` key and carry ONE hook, under the first module's name, as the reference stores them."""
    out: Dict[str, torch.Tensor] = {}
    owners: Dict[int, str] = {}
    counters: Dict[str, int] = {}
    mapped = set()
    for name, mod in model.named_modules():
        if parametrize.is_parametrized(mod, "weight"):
            plist = mod.parametrizations.weight
            dec = plist[0]
            prefix = f"{name}.parametrizations.weight."
            mapped.update((prefix + "original", prefix + "0._scale", prefix + "0._zero_point"))
            out[f"{name}.weight"] = plist.original.detach()
            if owners.setdefault(id(plist.original), name) == name:
                hook = "__nncf_hooks.post_hooks." + name.replace(".", ":") + ":weight__0.0."
                out[hook + "_scale"] = dec._scale
                if hasattr(dec, "_zero_point"):
                    out[hook + "_zero_point"] = dec._zero_point
        if isinstance(getattr(mod, "awq_input_scale", None), SQMultiply):
            mapped.add(f"{name}.awq_input_scale._scale_value")
            node = _reference_node_name(model, name, counters)
            out[f"__nncf_hooks.pre_hooks.{node}__0.0._scale_value"] = mod.awq_input_scale._scale_value
    for k, v in model.state_dict().items():
        if k not in mapped:
            out[k] = v
    return out


def save_compressed(model: nn.Module, path: str, layout: str = "native") -> None:
    """Writes the compressed model as ONE safetensors file: the model's state_dict (packed weights, fp16 / E8M0
    scales, packed zero points, AWQ activation scales, everything left uncompressed) plus a JSON header that says
    which module carries which decompressor.  The tensors are the reference's torch DQ storage format
    (layers.py:1303-1460, quantize_functions.py:417-461: nibble pairs low-first, +8 bias for INT4_SYM).
    layout "native": this build's module tree (`<module>.parametrizations.weight.*`); layout "nncf": the
    reference's own keys (`nncf_state_dict`), so that the file's tensors are what a model compressed by the reference
    stores; `load_compressed` reads both."""
    import json

    from safetensors.torch import save_file

    if layout not in ("native", "nncf"):
        raise ValueError(f"save_compressed: layout must be 'native' or 'nncf', got {layout!r}")

    entries, groups = {}, {}
    for name, mod in model.named_modules():
        if parametrize.is_parametrized(mod, "weight"):
            plist = mod.parametrizations.weight
            dec = plist[0]
            e = {"decompressor": type(dec).__name__, "result_dtype": str(dec.result_dtype).replace("torch.", ""),
                 "packed_shape": list(plist.original.shape)}
            for attr in ("compressed_weight_shape", "result_shape", "zero_point_shape"):
                if getattr(dec, attr, None) is not None:
                    e[attr] = list(getattr(dec, attr))
            first = groups.setdefault(id(plist.original), name)
            if first != name:
                e["shares_weight_with"] = first
            entries[name] = e
    sq = [name for name, mod in model.named_modules() if isinstance(getattr(mod, "awq_input_scale", None), SQMultiply)]
    tensors, seen, aliases = {}, {}, {}
    source = model.state_dict() if layout == "native" else nncf_state_dict(model)
    for k, v in source.items():
        key = (v.data_ptr(), tuple(v.shape), tuple(v.stride()), v.storage_offset(), v.dtype)
        if v.numel() and key in seen:  # tied tensors are stored once (safetensors refuses shared storage)
            aliases[k] = seen[key]
            continue
        seen[key] = k
        tensors[k] = v.detach().contiguous()
    header = {"weights": entries, "awq": sq, "aliases": aliases, "layout": layout}
    if layout == "nncf":  # reference key -> this build's key, for load_compressed
        native = {}
        counters: Dict[str, int] = {}
        for name in entries:
            hook = "__nncf_hooks.post_hooks." + name.replace(".", ":") + ":weight__0.0."
            prefix = f"{name}.parametrizations.weight."
            native[f"{name}.weight"] = prefix + "original"
            native[hook + "_scale"] = prefix + "0._scale"
            native[hook + "_zero_point"] = prefix + "0._zero_point"
        for name, mod in model.named_modules():
            if isinstance(getattr(mod, "awq_input_scale", None), SQMultiply):
                node = _reference_node_name(model, name, counters)
                native[f"__nncf_hooks.pre_hooks.{node}__0.0._scale_value"] = f"{name}.awq_input_scale._scale_value"
        header["native_keys"] = {k: v for k, v in native.items() if k in tensors or k in aliases}
    save_file(tensors, path, metadata={"nncf_amd_weight_compression": json.dumps(header)})


def load_compressed(model: nn.Module, path: str) -> nn.Module:
    """Turns a freshly built (uncompressed) `model` on the GPU into the compressed model stored by
    `save_compressed`: installs the decompressors named in the header, then loads every tensor."""
    import json

    from safetensors import safe_open

    device = _model_device(model)
    with safe_open(path, framework="pt", device=str(device)) as fh:
        header = json.loads(fh.metadata()["nncf_amd_weight_compression"])
        tensors = {k: fh.get_tensor(k) for k in fh.keys()}
    if header.get("layout") == "nncf":  # back to this build's keys
        native = header["native_keys"]
        tensors = {native.get(k, k): v for k, v in tensors.items()}
        header["aliases"] = {native.get(k, k): native.get(v, v) for k, v in header.get("aliases", {}).items()}
    mods = dict(model.named_modules())
    made: Dict[str, Tuple[nn.Parameter, nn.Module]] = {}
    for name, e in header["weights"].items():
        mod = mods[name]
        src = e.get("shares_weight_with")
        if src is not None:
            packed, dec = made[src]
        else:
            prefix = f"{name}.parametrizations.weight."
            cls = _DECOMPRESSORS.get(e["decompressor"])
            if cls is None:  # the header is data from a file: only the decompressor classes may be named
                raise InternalError(f"load_compressed: unknown decompressor {e['decompressor']!r} for {name}")
            dec = cls.__new__(cls)
            nn.Module.__init__(dec)
            dec.register_buffer("_scale", tensors[prefix + "0._scale"])
            if prefix + "0._zero_point" in tensors:
                dec.register_buffer("_zero_point", tensors[prefix + "0._zero_point"])
            dec.result_dtype = _DTYPES[e["result_dtype"]]
            for attr in ("compressed_weight_shape", "result_shape", "zero_point_shape"):
                if attr in e:
                    setattr(dec, attr, tuple(e[attr]))
            packed = nn.Parameter(tensors[prefix + "original"], requires_grad=False)
            made[name] = (packed, dec)
        mod.weight = packed
        parametrize.register_parametrization(mod, "weight", dec, unsafe=True)
    for name in header["awq"]:
        insert_sq_multiply(mods[name], tensors[f"{name}.awq_input_scale._scale_value"])
    missing, unexpected = model.load_state_dict(tensors, strict=False)
    if unexpected:
        raise InternalError(f"load_compressed: tensors without a place in the model: {unexpected[:5]}")
    # keys the file does not hold must be the tied / shared ones it recorded (stored once); anything else means a
    # truncated or mismatched file and would leave parameters at their initial values
    aliases = header.get("aliases", {})
    shared = {f"{n}.parametrizations.weight.{k}" for n, e in header["weights"].items() if e.get("shares_weight_with")
              for k in ("original", "0._scale", "0._zero_point")}
    lost = [k for k in missing if k not in aliases and k not in shared]
    if lost:
        raise InternalError(f"load_compressed: the file has no tensor for {lost[:5]} ({len(lost)} keys)")
    model.requires_grad_(False)
    return model
