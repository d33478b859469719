"""The reference's "optimized vs reference implementation" alignment matrix
(tests/openvino/optimized_functions/test_compression_functions.py:163-281, checked there with atol = 0, rtol = 0,
:521-559) with the HIP path in the place of the OpenVINO-compiled functions and the oracle in the place of the numpy
path: weight (10000, 32) ~ N(0, 1) seed 0 x every mode on the path (per-channel, group size 2, MXFP4 g32) x
fp32 / fp16 / bf16 x {quantize, quantize-dequantize, quantize-dequantize returning the compressed weight} x
{computed, precomputed scale / zero point}.  Everything bit-exact."""
import numpy as np
import pytest
import torch

from oracle import wc_oracle as O
from tests.gpu_utils import host, representable, to_dev
from tests.helpers import assert_equal_nan

pytestmark = pytest.mark.gpu

CONFIGS = [("int8_asym", -1), ("int8_sym", -1), ("int4_asym", -1), ("int4_sym", -1), ("int4_asym", 2), ("int4_sym", 2),
           ("nf4", -1), ("fp4", -1), ("nf4", 2), ("fp4", 2), ("mxfp4", 32)]
SHAPE = (10000, 32)


def _cfg(mode, gs):
    from nncf_amd.parameters import WeightCompressionConfig
    return WeightCompressionConfig(mode=mode, group_size=gs)


def _weight(dtype):
    np.random.seed(0)
    data = np.random.normal(size=SHAPE)
    data = data.astype(np.float16 if dtype == "f16" else np.float32)
    return representable(data.astype(np.float32), dtype)


@pytest.mark.parametrize("precompute", [False, True], ids=["no-precompute", "precompute"])
@pytest.mark.parametrize("dtype", ["f32", "f16", "bf16"])
@pytest.mark.parametrize("task", ["Q", "Q_DQ", "Q_DQ_RQ"])
@pytest.mark.parametrize("mode,gs", CONFIGS, ids=[f"{m}_g{g}" for m, g in CONFIGS])
def test_quantization_alignment(mode, gs, task, dtype, precompute):
    from nncf_amd import weight_lowering as wl
    cfg = _cfg(mode, gs)
    is_int = mode.startswith("int")
    asym = mode.endswith("asym")
    w32 = _weight(dtype)
    d1, d2 = SHAPE
    pshape = (d1, 1) if gs == -1 else (d1, d2 // gs, 1)
    sc = zp = None
    axes = (1,)
    if precompute:  # inputs are reshaped beforehand and reduction axes are not passed (:205-217)
        if gs != -1:
            w32 = w32.reshape(d1, d2 // gs, gs)
        np.random.seed(0)
        sc = np.random.normal(size=pshape).astype(np.float32)
        if asym:
            np.random.seed(0)
            zp = np.random.randint(0, 2 ** (8 if mode.startswith("int8") else 4) - 1, size=pshape).astype(np.int32)
        axes = None
    w = to_dev(w32, dtype)
    sc_d = torch.from_numpy(sc).cuda() if sc is not None else None
    zp_d = torch.from_numpy(zp).cuda() if zp is not None else None
    # ---- HIP
    if is_int:
        if task == "Q":
            cw, dq = wl.do_integer_quantization(w, cfg, axes, sc_d, zp_d), None
        elif task == "Q_DQ":
            cw, dq = None, wl.integer_quantize_dequantize_weight(w, cfg, axes, sc_d, zp_d)
        else:
            dq, cw = wl.integer_quantize_dequantize_weight(w, cfg, axes, sc_d, zp_d, return_compressed_weight=True)
    else:
        if task == "Q":
            cw, dq = wl.do_float_quantization(w, cfg, axes, sc_d), None
        elif task == "Q_DQ":
            cw, dq = None, wl.float_quantize_dequantize_weight(w, cfg, axes, sc_d)
        else:
            dq, cw = wl.float_quantize_dequantize_weight(w, cfg, axes, sc_d, return_compressed_weight=True)
    # ---- oracle
    with np.errstate(all="ignore"):
        if is_int:
            r_dq, (r_t, r_s, r_z) = O.integer_quantize_dequantize_weight(w32, mode, gs, axes, sc, zp, True)
        else:
            r_dq, (r_t, r_s) = O.float_quantize_dequantize_weight(w32, mode, gs, axes, sc, True)
            r_z = None
    if cw is not None:
        if precompute:  # the returned parameters are the given ones (:250-254)
            assert np.array_equal(host(cw.scale), sc)
            if asym:
                assert np.array_equal(host(cw.zero_point), zp)
        assert_equal_nan(host(cw.tensor), r_t.reshape(tuple(cw.tensor.shape)), "compressed weight")
        assert host(cw.tensor).dtype == r_t.dtype
        assert_equal_nan(host(cw.scale), np.asarray(r_s).reshape(tuple(cw.scale.shape)), "scale")
        if asym:
            assert_equal_nan(host(cw.zero_point), np.asarray(r_z).reshape(tuple(cw.zero_point.shape)), "zero point")
    if dq is not None:
        assert dq.dtype == torch.float32
        assert_equal_nan(host(dq), r_dq.reshape(tuple(dq.shape)), "decompressed weight")


@pytest.mark.parametrize("dtype", ["f32", "f16", "bf16"])
@pytest.mark.parametrize("reduction", ["max_mean", "frobenius"])
@pytest.mark.parametrize("mode,gs", [c for c in CONFIGS if c[0].startswith("int")])
def test_integer_quantization_error_alignment(mode, gs, reduction, dtype):
    """:284-324, the reference's own tolerance (atol 1e-6, rtol 1e-4: the reductions sum in another order)."""
    from nncf_amd import weight_lowering as wl
    w32 = _weight(dtype)
    got = wl.get_integer_quantization_error(to_dev(w32, dtype), (1,), _cfg(mode, gs), reduction)
    ref = O.get_integer_quantization_error(w32, (1,), mode, gs, reduction)
    np.testing.assert_allclose(got, ref, atol=1e-6, rtol=1e-4)
