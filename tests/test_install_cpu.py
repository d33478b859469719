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
