"""The drop-in patch against a real NNCF, when one is importable in this environment (the build container has the
reference checkout; the GPU box does not, there the test is skipped).  CPU tensors must keep flowing through the
reference's own code (the HIP path is for GPU tensors only), signatures must be preserved."""
import inspect
import os
import sys

import numpy as np
import pytest

REF = "/root/reference/src"
if os.path.isdir(REF) and REF not in sys.path:
    sys.path.append(REF)
nncf = pytest.importorskip("nncf")


def test_install_preserves_signatures_and_cpu_behaviour():
    import torch

    import nncf_amd.install as inst
    from nncf.parameters import CompressWeightsMode
    from nncf.quantization.algorithms.weight_compression import weight_lowering as WL
    from nncf.quantization.algorithms.weight_compression.config import WeightCompressionConfig
    from nncf.tensor import Tensor
    before = {n: inspect.signature(getattr(WL, n)) for n in ("do_integer_quantization",
                                                              "integer_quantize_dequantize_weight",
                                                              "float_quantize_dequantize_weight",
                                                              "get_integer_quantization_error")}
    w = torch.randn(128, 128) * 0.05
    cfg = WeightCompressionConfig(CompressWeightsMode.INT4_SYM, 32)
    ref = WL.compress_weight(Tensor(w), (1,), cfg)
    names = inst.install()
    try:
        assert len(names) >= 12 and inst.install() == names
        for n, sig in before.items():
            assert inspect.signature(getattr(WL, n)) == sig
        got = WL.compress_weight(Tensor(w), (1,), cfg)  # 16384 elements >= threshold, but a CPU tensor
        assert np.array_equal(got.tensor.data.numpy(), ref.tensor.data.numpy())
        assert np.array_equal(got.scale.data.numpy(), ref.scale.data.numpy())
    finally:
        inst.uninstall()
    assert not inst._installed


def _oracle_lowering(monkeypatch):
    """Test-only stand-ins for the HIP-backed functions install() diverts to (the oracle does their arithmetic on CPU
    tensors), plus `_can_run_hip -> True`: what is left under test is install()'s own code -- the GPU branch of every
    wrapper: unwrapping of nncf Tensors, wrapping of the results, given parameters returned verbatim."""
    import torch

    import nncf_amd.install as inst
    from nncf_amd.parameters import CompressedWeight
    from oracle import wc_oracle as O

    calls = []

    def t(x):
        return None if x is None else torch.from_numpy(np.ascontiguousarray(x))

    def n(x):
        return None if x is None else x.numpy()

    def do_int(weight, config, reduction_axes=None, scale=None, zp=None):
        assert isinstance(weight, torch.Tensor) and (scale is None or isinstance(scale, torch.Tensor))
        calls.append("do_int")
        c, s, z = O.do_integer_quantization(n(weight), config.mode.value, config.group_size, reduction_axes, n(scale), n(zp))
        return CompressedWeight(t(c), t(s), t(z))

    def int_qdq(weight, config, reduction_axes=None, scale=None, zp=None, return_compressed_weight=False):
        calls.append("int_qdq")
        dq = O.integer_quantize_dequantize_weight(n(weight), config.mode.value, config.group_size, reduction_axes, n(scale), n(zp))
        if return_compressed_weight:
            return t(dq), do_int(weight, config, reduction_axes, scale, zp)
        return t(dq)

    def do_float(weight, config, reduction_axes=None, scale=None):
        calls.append("do_float")
        v, s = O.do_float_quantization(n(weight), config.mode.value, config.group_size, reduction_axes, n(scale))[:2]
        return CompressedWeight(t(v), t(s), None)

    def err(weight, reduction_axes, config, reduction):
        calls.append("err")
        return float(O.get_integer_quantization_error(n(weight), reduction_axes, config.mode.value, config.group_size, reduction))

    monkeypatch.setattr(inst, "_can_run_hip", lambda *a, **k: True)
    monkeypatch.setattr(inst._wl, "do_integer_quantization", do_int)
    monkeypatch.setattr(inst._wl, "integer_quantize_dequantize_weight", int_qdq)
    monkeypatch.setattr(inst._wl, "do_float_quantization", do_float)
    monkeypatch.setattr(inst._wl, "get_integer_quantization_error", err)
    return calls


def test_install_gpu_branch_contract(monkeypatch):
    """The contract of the reference's optimized seam, tests/openvino/optimized_functions/test_compression_functions.py
    :163-281 (:250-254: given scale / zero point come back VERBATIM; results are nncf Tensors; Q_DQ with
    return_compressed_weight gives (Tensor, CompressedWeight)), for install()'s GPU branch."""
    import torch

    import nncf_amd.install as inst
    from nncf.parameters import CompressWeightsMode
    from nncf.quantization.algorithms.weight_compression import weight_lowering as WL
    from nncf.quantization.algorithms.weight_compression.config import WeightCompressionConfig
    from nncf.quantization.algorithms.weight_compression.parameters import CompressedWeight as RefCW
    from nncf.tensor import Tensor

    torch.manual_seed(0)
    w = torch.randn(96, 128) * 0.05
    ref = {}
    for mode, gs in ((CompressWeightsMode.INT4_ASYM, 32), (CompressWeightsMode.INT8_SYM, -1), (CompressWeightsMode.NF4, 32)):
        ref[mode] = WL.compress_weight(Tensor(w), (1,), WeightCompressionConfig(mode, gs))  # the reference itself
    calls = _oracle_lowering(monkeypatch)
    inst.install()
    try:
        for mode, gs in ((CompressWeightsMode.INT4_ASYM, 32), (CompressWeightsMode.INT8_SYM, -1), (CompressWeightsMode.NF4, 32)):
            cfg = WeightCompressionConfig(mode, gs)
            calls.clear()
            got = WL.compress_weight(Tensor(w), (1,), cfg)
            assert calls and isinstance(got, RefCW) and isinstance(got.tensor, Tensor) and isinstance(got.scale, Tensor)
            assert np.array_equal(got.tensor.data.numpy(), ref[mode].tensor.data.numpy())
            assert np.array_equal(got.scale.data.numpy(), ref[mode].scale.data.numpy())
            if ref[mode].zero_point is not None:
                assert isinstance(got.zero_point, Tensor)
                assert np.array_equal(got.zero_point.data.numpy(), ref[mode].zero_point.data.numpy())
            else:
                assert got.zero_point is None
            # precomputed parameters: the SAME objects come back (weight_lowering.py:502-505; the alignment test
            # compares them with atol = rtol = 0)
            sc, zp = ref[mode].scale, ref[mode].zero_point
            if cfg.is_integer:
                cw = WL.do_integer_quantization(Tensor(w), cfg, (1,), sc, zp)
                assert cw.scale is sc and cw.zero_point is zp and isinstance(cw.tensor, Tensor)
                dq, cw2 = WL.integer_quantize_dequantize_weight(Tensor(w), cfg, (1,), sc, zp, return_compressed_weight=True)
                assert isinstance(dq, Tensor) and isinstance(cw2, RefCW) and cw2.scale is sc and cw2.zero_point is zp
                only = WL.integer_quantize_dequantize_weight(Tensor(w), cfg, (1,))
                assert isinstance(only, Tensor) and only.shape == dq.shape
            else:
                cw = WL.compress_weight(Tensor(w), (1,), cfg, RefCW(None, sc, None))
                assert cw.scale is sc
        calls.clear()
        e = WL.get_integer_quantization_error(Tensor(w), (1,), WeightCompressionConfig(CompressWeightsMode.INT8_ASYM, -1), "max_mean")
        assert calls == ["err"] and isinstance(e, float)
    finally:
        inst.uninstall()
    # uninstalled: the reference's own arithmetic again, and it agrees with what went through the wrappers
    again = WL.compress_weight(Tensor(w), (1,), WeightCompressionConfig(CompressWeightsMode.INT4_ASYM, 32))
    assert np.array_equal(again.tensor.data.numpy(), ref[CompressWeightsMode.INT4_ASYM].tensor.data.numpy())
