"""Host logic of nncf_amd.quantize_model that needs no GPU: tracing (execution order, shared activations, AWQ
producers read off the autograd graph), weight-parameter resolution against what the reference compressed
(tests/golden/model_tiny.npz meta), ratio selection, validation messages (algorithm.py:140-295)."""
import json
import os

import numpy as np
import pytest
import torch

from nncf_amd import quantize_model as qm
from nncf_amd.errors import InvalidGroupSizeError, ParameterNotSupportedError, UnsupportedBackendError, ValidationError
from nncf_amd.parameters import (AdvancedCompressionParameters, BackupMode, CompressWeightsMode, Dataset,
                                 GroupSizeFallbackMode, IgnoredScope, SensitivityMetric)
from tests.tiny_models import TINY_CONFIGS, make_tiny_lm, tiny_lm_dataset

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _algo(**kw):
    mode = CompressWeightsMode(kw.pop("mode", "int4_sym"))
    cfg = qm.get_weight_compression_configuration(
        mode, kw.pop("dataset", None), kw.pop("ratio", None), kw.pop("group_size", None), kw.pop("all_layers", None),
        None, None, None, kw.pop("ignored_scope", None), kw.pop("sensitivity_metric", None),
        kw.pop("backup_mode", None), kw.pop("advanced_parameters", None))
    return qm.WeightCompression(cfg["mode"], cfg["ratio"], cfg["group_size"], cfg["ignored_scope"], cfg["all_layers"],
                                cfg["sensitivity_metric"], False, 128, False, False, cfg["backup_mode"],
                                cfg["advanced_parameters"])


def test_trace_order_sharing_and_producers():
    m = make_tiny_lm()
    t = qm.trace_model(m, tiny_lm_dataset()[0], with_provenance=True)
    assert t.order[0] == "embed" and t.order[-1] == "lm_head" and len(t.order) == 22
    assert t.order[1:4] == ["blocks.0.attn.q_proj", "blocks.0.attn.k_proj", "blocks.0.attn.v_proj"]
    k = t.act_key
    assert k["blocks.0.attn.q_proj"] == k["blocks.0.attn.k_proj"] == k["blocks.0.attn.v_proj"]
    assert k["blocks.0.mlp.gate_proj"] == k["blocks.0.mlp.up_proj"] != k["blocks.0.mlp.down_proj"]
    assert len(set(k.values())) == 13  # 3 blocks x (qkv, o, gate/up or fc1, down or fc2) + lm_head
    assert t.producer == {"blocks.0.mlp.down_proj": ("mul_linear", "blocks.0.mlp.up_proj"),
                          "blocks.1.mlp.down_proj": ("mul_linear", "blocks.1.mlp.up_proj"),
                          "blocks.2.mlp.fc2": ("act", None)}
    assert t.act_shape["lm_head"] == (1, 12, 64)
    assert all(not p.requires_grad or p.grad is None for p in m.parameters())


@pytest.mark.parametrize("name", [n for n, (_, kw) in TINY_CONFIGS.items() if not kw.get("ratio")])
def test_parameter_resolution_matches_reference(name):
    """Which weights get compressed and to how many bits, without mixed precision: equal to what the reference did
    (golden meta + stored dtypes/shapes)."""
    z = np.load(os.path.join(GOLDEN, "model_tiny.npz"))
    meta = json.loads(bytes(z["__meta__"]).decode())
    mkw, kw = TINY_CONFIGS[name]
    kw = {k: v for k, v in kw.items() if k not in ("data", "awq", "scale_estimation")}
    fallback, ignored = kw.pop("fallback", None), kw.pop("ignored", None)
    if fallback:
        kw["advanced_parameters"] = AdvancedCompressionParameters(
            group_size_fallback_mode=GroupSizeFallbackMode(fallback), min_adjusted_group_size=16)
    if ignored:
        kw["ignored_scope"] = IgnoredScope(names=tuple(ignored))
    if "backup_mode" in kw:
        kw["backup_mode"] = BackupMode(kw["backup_mode"])
    algo = _algo(**kw)
    m = make_tiny_lm(**mkw)
    trace = qm.trace_model(m, tiny_lm_dataset()[0])
    all_params, ratio_defining, skipped = algo.get_weight_compression_parameters(m, trace)
    algo.validate_group_size(ratio_defining)
    got = {wp.module_name: wp.compression_config for wp in all_params if wp.compression_config is not None}
    assert sorted(got) == meta[name]["modules"]
    for mod, cfg in got.items():
        w = z[f"{name}|{mod}|w"]
        four = w.ndim == 2 and w.shape[1] == 1 and w.shape[0] * 2 == int(np.prod(dict(m.named_modules())[mod].weight.shape))
        assert (cfg.num_bits == 4) == four, (mod, cfg)
        ref_groups = z[f"{name}|{mod}|s"].size
        groups = w.size * (2 if four else 1) // (cfg.group_size if cfg.group_size != -1 else
                                                  dict(m.named_modules())[mod].weight.shape[1])
        assert groups == ref_groups, (mod, cfg)


def test_tied_last_layer_is_skipped_once():
    m = make_tiny_lm(tie=True)
    trace = qm.trace_model(m, tiny_lm_dataset()[0])
    algo = _algo(mode="int4_sym", group_size=32)
    all_params, ratio_defining, _ = algo.get_weight_compression_parameters(m, trace)
    names = [wp.module_name for wp in all_params]
    assert "lm_head" not in names and "embed" in names
    # the shared last layer was skipped, so the last remaining matmul stays ratio-defining (algorithm.py:559-561)
    assert ratio_defining[-1].module_name == "blocks.2.mlp.fc2" and len(ratio_defining) == 20


def test_ratio_selection_walks_scores_in_ascending_order():
    m = make_tiny_lm()
    trace = qm.trace_model(m, tiny_lm_dataset()[0])
    algo = _algo(mode="int4_sym", group_size=32, ratio=0.5)
    _, ratio_defining, _ = algo.get_weight_compression_parameters(m, trace)
    scores = {wp.module_name: float(i) for i, wp in enumerate(ratio_defining)}

    def fake_errs(ws, axes, cfg, reduction="max_mean"):
        return [0.0] * len(ws)

    orig = qm.wl.get_integer_quantization_errors
    qm.wl.get_integer_quantization_errors = fake_errs
    try:
        algo._sensitivity_metric = SensitivityMetric.MAX_ACTIVATION_VARIANCE
        algo.apply_mixed_precision(ratio_defining, scores)
    finally:
        qm.wl.get_integer_quantization_errors = orig
    total = sum(wp.num_weights for wp in ratio_defining)
    acc, want4 = 0, []
    for wp in ratio_defining:  # scores ascend with the index
        if (acc + wp.num_weights) / total >= 0.5:
            break
        want4.append(wp.module_name)
        acc += wp.num_weights
    got4 = [wp.module_name for wp in ratio_defining if wp.compression_config.num_bits == 4]
    assert got4 == want4 and 0 < len(got4) < len(ratio_defining)
    assert all(wp.compression_config.mode == CompressWeightsMode.INT8_ASYM for wp in ratio_defining
               if wp.module_name not in want4)


def test_group_size_error_and_messages():
    m = make_tiny_lm()
    trace = qm.trace_model(m, tiny_lm_dataset()[0])
    algo = _algo(mode="int4_sym", group_size=48)
    _, ratio_defining, _ = algo.get_weight_compression_parameters(m, trace)
    with pytest.raises(InvalidGroupSizeError, match="Failed to apply group-wise quantization with group size value 48"):
        algo.validate_group_size(ratio_defining)
    with pytest.raises(ParameterNotSupportedError, match="INT8 modes require per-channel"):
        qm.check_user_compression_configuration(CompressWeightsMode.INT8_ASYM, 128, None, 0.5, None, None, None, None,
                                                None, None, None, None, None, None)
    with pytest.raises(ParameterNotSupportedError, match="INT8 modes do not support all_layers, awq"):
        qm.check_user_compression_configuration(CompressWeightsMode.INT8_SYM, 128, None, None, None, True, True, None,
                                                None, None, None, None, None, None)
    with pytest.raises(ValidationError, match="The ratio should be between 0 and 1"):
        qm.check_user_compression_configuration(CompressWeightsMode.INT4_SYM, 128, None, 1.5, None, None, None, None,
                                                None, None, None, None, None, None)
    with pytest.raises(ValidationError, match="subset_size value should be positive"):
        qm.check_user_compression_configuration(CompressWeightsMode.INT4_SYM, 0, None, None, None, None, None, None,
                                                None, None, None, None, None, None)
    with pytest.raises(ValidationError, match="requires a dataset"):
        qm.check_user_compression_configuration(CompressWeightsMode.INT4_SYM, 128, None, 0.5, None, None, None, None,
                                                None, None, None, SensitivityMetric.HESSIAN_INPUT_ACTIVATION, None,
                                                None)
    with pytest.raises(ValidationError, match="only support group size of 32"):
        qm.check_user_compression_configuration(CompressWeightsMode.MXFP4, 128, None, None, 64, None, None, None,
                                                None, None, None, None, None, None)


def test_defaults_follow_the_reference():
    c = qm.get_weight_compression_configuration(CompressWeightsMode.INT4_SYM, None, None, None, None, None, None,
                                                None, None, None, None, None)
    assert c["group_size"] == 128 and c["ratio"] == 1 and c["backup_mode"] == BackupMode.INT8_ASYM
    assert c["sensitivity_metric"] == SensitivityMetric.WEIGHT_QUANTIZATION_ERROR
    c = qm.get_weight_compression_configuration(CompressWeightsMode.MXFP4, object(), None, None, None, None, None,
                                                None, None, None, None, None)
    assert c["group_size"] == 32 and c["backup_mode"] == BackupMode.MXFP8_E4M3
    assert c["sensitivity_metric"] == SensitivityMetric.MAX_ACTIVATION_VARIANCE
    c = qm.get_weight_compression_configuration(CompressWeightsMode.INT8_ASYM, None, None, None, None, None, None,
                                                None, None, None, None, None)
    assert c["group_size"] == -1


def test_cpu_model_is_refused_loudly():
    m = make_tiny_lm()
    with pytest.raises(UnsupportedBackendError):
        qm.compress_weights(m, dataset=Dataset(tiny_lm_dataset()[:1]))
    with pytest.raises(ValidationError, match="Please provide a dataset"):
        qm.compress_weights(m)
    with pytest.raises(ParameterNotSupportedError):
        qm.compress_weights(m, mode="fp4", dataset=Dataset(tiny_lm_dataset()[:1]))


def test_dataset_indices_and_transform():
    ds = Dataset(list(range(10)), transform_func=lambda v: v * 2)
    assert list(ds.get_inference_data()) == [v * 2 for v in range(10)]
    assert list(ds.get_inference_data(indices=[1, 3])) == [2, 6]
    assert ds.get_length() == 10


def test_insert_decompressor_shares_tied_weights():
    """Parametrization plumbing with a stand-in decompressor (the real one needs the GPU)."""
    from torch.nn.utils import parametrize

    m = make_tiny_lm(tie=True)

    class Fake(torch.nn.Module):
        def forward(self, x):
            return x.float() * 0.5

    old = m.embed.weight
    packed = torch.nn.Parameter(torch.ones(old.shape, dtype=torch.uint8), requires_grad=False)
    dec = Fake()
    for mod in m.modules():
        if getattr(mod, "_parameters", None) and mod._parameters.get("weight") is old:
            mod.weight = packed
            parametrize.register_parametrization(mod, "weight", dec, unsafe=True)
    assert parametrize.is_parametrized(m.embed, "weight") and parametrize.is_parametrized(m.lm_head, "weight")
    assert m.embed.parametrizations.weight.original is m.lm_head.parametrizations.weight.original
    assert m.lm_head.weight.dtype == torch.float32 and float(m.lm_head.weight[0, 0]) == 0.5
    out = m(tiny_lm_dataset()[0])
    assert out.shape == (1, 12, 160)


def test_statistics_cache_round_trip(tmp_path):
    from nncf_amd.statistics import WCTensorStatistic

    stats = {"a.b": WCTensorStatistic([torch.arange(8.0), torch.arange(8.0) * 2], [(1, 12, 8), (1, 13, 8)]),
             "c": WCTensorStatistic([torch.ones(4, dtype=torch.bfloat16)], [(2, 3, 4)])}
    scores = {"a.b": 1.5, "c": 0.25}
    path = str(tmp_path / "stats.safetensors")
    qm.save_statistics(path, stats, scores, "tag1")
    assert qm.load_statistics(path, "cpu", "other") is None
    assert qm.load_statistics(str(tmp_path / "absent"), "cpu", "tag1") is None
    got, got_scores = qm.load_statistics(path, "cpu", "tag1")
    assert got_scores == scores and sorted(got) == ["a.b", "c"]
    assert got["a.b"].shape_values == [(1, 12, 8), (1, 13, 8)]
    assert torch.equal(got["a.b"].mean_values[1], torch.arange(8.0) * 2)
    assert got["c"].mean_values[0].dtype == torch.bfloat16
