"""Pins oracle/wc_oracle.py against the golden fixtures that tools/gen_golden.py produced by running
the reference (nncf 3.2.0).  CPU only."""
import numpy as np
import pytest

from oracle import wc_oracle as O
from tests.helpers import assert_equal_nan, golden_weight

INT_MODES = ("int8_asym", "int8_sym", "int4_asym", "int4_sym")


@pytest.mark.parametrize("dtype", ["f32", "f16", "bf16"])
@pytest.mark.parametrize("gs", [-1, 32, 64, 128])
@pytest.mark.parametrize("mode", INT_MODES)
def test_int_lowering(golden, mode, gs, dtype):
    g = golden("int_lowering")
    w = golden_weight(g, dtype)
    codes, scale, zp = O.compress_weight(w, (1,), mode, gs)
    key = f"{mode}_g{gs}_{dtype}"
    assert_equal_nan(codes, g[key + "_codes"], key)
    assert codes.dtype == g[key + "_codes"].dtype
    assert_equal_nan(scale, g[key + "_scale"], key)
    if zp is not None:
        assert zp.dtype == np.int32
        assert_equal_nan(zp, g[key + "_zp"], key)
    if dtype == "f32":
        assert_equal_nan(O.do_integer_dequantization(codes, scale, zp), g[key + "_dq"], key)
        if mode.startswith("int4"):
            packed = O.pack_uint4(codes) if mode == "int4_asym" else O.pack_int4(codes)
            assert_equal_nan(packed, g[key + "_packed"], key)
            un = O.unpack_uint4(packed) if mode == "int4_asym" else O.unpack_int4(packed)
            assert_equal_nan(un.reshape(codes.shape), codes, key)


@pytest.mark.parametrize("mode", INT_MODES)
def test_int_precomputed(golden, mode):
    g = golden("int_lowering")
    w, sc, zi = g["pre_w"], g["pre_scale"], g["pre_zp_i32"]
    if mode.endswith("asym"):
        for zname, z in (("i32", zi), ("f32", zi.astype(np.float32))):
            dq, (codes, s, zz) = O.integer_quantize_dequantize_weight(w, mode, 64, None, sc, z, True)
            assert_equal_nan(codes, g[f"pre_{mode}_{zname}_codes"])
            assert_equal_nan(dq, g[f"pre_{mode}_{zname}_dq"])
            assert s is sc and zz is z
    else:
        dq, (codes, s, _) = O.integer_quantize_dequantize_weight(w, mode, 64, None, sc, None, True)
        assert_equal_nan(codes, g[f"pre_{mode}_codes"])
        assert_equal_nan(dq, g[f"pre_{mode}_dq"])


@pytest.mark.parametrize("red", ["max_mean", "frobenius"])
@pytest.mark.parametrize("gs", [-1, 64])
@pytest.mark.parametrize("mode", INT_MODES)
def test_int_error(golden, mode, gs, red):
    g = golden("int_lowering")
    got = O.get_integer_quantization_error(g["err_w"], (1,), mode, gs, red)
    np.testing.assert_allclose(got, float(g[f"err_{mode}_g{gs}_{red}"]), rtol=1e-6)


@pytest.mark.parametrize("dtype", ["f32", "f16", "bf16"])
@pytest.mark.parametrize("gs", [-1, 32, 64])
@pytest.mark.parametrize("mode", ["nf4", "mxfp4"])
def test_float_lowering(golden, mode, gs, dtype):
    g = golden("float_lowering")
    w = golden_weight(g, dtype)
    v, s, _ = O.compress_weight(w, (1,), mode, gs)
    key = f"{mode}_g{gs}_{dtype}"
    assert_equal_nan(s, g[key + "_scale"], key)
    assert_equal_nan(v, g[key + "_values"], key)
    if dtype == "f32":
        assert_equal_nan(O.do_float_dequantization(v, s), g[key + "_dq"], key)


@pytest.mark.parametrize("mode", ["nf4", "mxfp4"])
def test_float_boundaries(golden, mode):
    g = golden("float_lowering")
    v, s, _ = O.compress_weight(g["w_boundary"], (1,), mode, 32)
    assert_equal_nan(s, g[f"boundary_{mode}_scale"])
    assert_equal_nan(v, g[f"boundary_{mode}_values"])
    mid = "e2m1" if mode == "mxfp4" else "nf4"
    v, s, _ = O.compress_weight(g[f"w_mid_{mid}"], (1,), mode, -1)
    assert_equal_nan(s, g[f"mid_{mid}_scale"])
    assert_equal_nan(v, g[f"mid_{mid}_values"])
    dq, (v, s) = O.float_quantize_dequantize_weight(g["pre_w"], mode, 32, None, g["pre_scale"], True)
    assert_equal_nan(v, g[f"pre_{mode}_values"])
    assert_equal_nan(dq, g[f"pre_{mode}_dq"])


@pytest.mark.parametrize("dtype,tol", [("f32", 2e-6), ("bf16", 0), ("f16", 0)])
def test_reducers(golden, dtype, tol):
    g = golden("reducers")
    x = O.round_to_dtype(g["x_f32"], dtype)
    # bf16/f16: torch accumulates in fp32 and rounds once -> the rounded results must agree to within 1 ulp of dtype
    rt = {"f32": tol, "bf16": 2 ** -7, "f16": 2 ** -10}[dtype]
    np.testing.assert_allclose(O.mean_reducer(x, (0, 1), dtype), g[f"mean_{dtype}"], rtol=rt, atol=1e-6)
    np.testing.assert_allclose(O.mean_variance_reducer(x, (1,), dtype), g[f"meanvar_{dtype}"], rtol=rt)
    np.testing.assert_allclose(O.max_variance_reducer(x, (1,), dtype), g[f"maxvar_{dtype}"], rtol=rt)
    np.testing.assert_allclose(O.mean_abs_max_reducer(x, (1,), dtype), g[f"meanabsmax_{dtype}"], rtol=rt)


def test_hawq_and_process_stats(golden):
    g = golden("reducers")
    np.testing.assert_allclose(O.hawq_aggregate([g[f"hawq_x{i}"] for i in range(4)]), float(g["hawq_result"]),
                               rtol=1e-6)
    means = list(g["ps_means"])
    shapes = [tuple(int(v) for v in s) for s in g["ps_shapes"]]
    for subset in (4, 3, 32):
        s, X = O.process_stats(means, shapes, subset, -1)
        assert_equal_nan(s, g[f"ps_s_{subset}"])
        assert_equal_nan(X, g[f"ps_X_{subset}"])


def _se_inputs(g):
    return list(g["means"]), [tuple(int(v) for v in s) for s in g["shapes"]]


@pytest.mark.parametrize("gs", [32, -1])
@pytest.mark.parametrize("mode", ["int4_sym", "int4_asym", "nf4"])
def test_scale_estimation(golden, mode, gs):
    g = golden("se_awq")
    means, shapes = _se_inputs(g)
    s, z = O.scale_estimation_params(means, shapes, g["w"], (1,), mode, gs, -1, 8, 3, 4)
    np.testing.assert_allclose(s, g[f"se_{mode}_g{gs}_scale"], rtol=1e-5, atol=1e-9)
    if z is not None:
        assert_equal_nan(z, g[f"se_{mode}_g{gs}_zp"])


@pytest.mark.parametrize("mode", ["int4_sym", "int4_asym"])
def test_awq(golden, mode):
    g = golden("se_awq")
    means, shapes = _se_inputs(g)
    sc = O.awq_data_aware_scale(g["w"], 1, mode, 32, means, shapes, -1, 8, 0.02, 0.0, 1.0, 20)
    np.testing.assert_allclose(sc, g[f"awq_{mode}_scale"], rtol=1e-5)
    np.testing.assert_allclose(O.awq_data_free_scale(g["w"], 0), g["awq_data_free_scale"], rtol=1e-6)


def test_gptq_hessian(golden):
    g = golden("gptq")
    H = O.gptq_hessian(list(g["xs"]))
    np.testing.assert_allclose(H, g["H_numpy"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(H, g["H_torch"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("se", [0, 1])
@pytest.mark.parametrize("gs", [16, 64, -1])
@pytest.mark.parametrize("mode", ["int4_sym", "int4_asym", "nf4"])
def test_gptq_quantize(golden, mode, gs, se):
    if se and mode == "nf4":
        pytest.skip("reference runs scale estimation inside GPTQ for integer modes only")
    g = golden("gptq")
    H = g["H_numpy"][0].copy()
    q, s, z = O.gptq_quantize_weights(H, g["w"].copy(), mode, gs, list(g["xs"]), 0.1, 32, bool(se))
    key = f"numpy_{mode}_g{gs}_se{se}"
    assert s.shape == g[key + "_scale"].shape
    # GPTQ is chaotic under rounding: the same reference differs between its numpy and torch backends,
    # so require agreement on (nearly) all entries rather than bit equality.
    close = np.isclose(q, g[key + "_q"], rtol=1e-4, atol=1e-6)
    assert close.mean() > 0.99, close.mean()
    np.testing.assert_allclose(s, g[key + "_scale"], rtol=1e-3, atol=1e-6)
    if z is not None:
        assert (z == g[key + "_zp"]).mean() > 0.98


# ------------------------------------------------------------------ round-2 fixtures (tools/gen_golden_extra.py)
def _shapes(a):
    return [tuple(int(v) for v in r) for r in a]


@pytest.mark.parametrize("mode", ["int4_sym", "int4_asym"])
def test_awq_clamp_branch(golden, mode):
    """AWQ with the previous matmul merged: the fp16-overflow clamp of awq.py:315-335 is taken (80 times while the
    fixture was generated)."""
    g = golden("se_awq_extra")
    assert int(g["clamp_calls"]) > 0
    sc = O.awq_data_aware_scale(g["clamp_w"], 1, mode, 32, list(g["clamp_means"]), _shapes(g["clamp_shapes"]), -1, 8, 0.02,
                                0.0, 1.0, 20, g["clamp_prev_w"], list(g["clamp_prev_means"]), _shapes(g["clamp_prev_shapes"]))
    np.testing.assert_allclose(sc, g[f"clamp_awq_{mode}_scale"], rtol=1e-6)


@pytest.mark.parametrize("mode", ["int4_sym", "int4_asym"])
def test_awq_3d_weights(golden, mode):
    g = golden("se_awq_extra")
    w_ekn = np.ascontiguousarray(np.moveaxis(g["moe_w_enk"], -1, -2))
    sc = O.awq_data_aware_scale(w_ekn, 1, mode, 32, list(g["moe_means"]), _shapes(g["moe_shapes"]), -1, 8, 0.05, 0.0, 1.0, 20)
    assert sc.shape == g[f"moe_awq_{mode}_scale"].shape
    np.testing.assert_allclose(sc, g[f"moe_awq_{mode}_scale"], rtol=1e-6)


@pytest.mark.parametrize("axis", [2, 1])
@pytest.mark.parametrize("gs", [32, -1])
@pytest.mark.parametrize("mode", ["int4_sym", "int4_asym", "nf4"])
def test_scale_estimation_3d_weights(golden, mode, gs, axis):
    g = golden("se_awq_extra")
    w = g["moe_w_enk"] if axis == 2 else np.ascontiguousarray(np.moveaxis(g["moe_w_enk"], -1, -2))
    s, z = O.scale_estimation_params(list(g["moe_means"]), _shapes(g["moe_shapes"]), w, (axis,), mode, gs, -1, 8, 3, 4)
    key = f"moe_se_{mode}_g{gs}_ax{axis}"
    assert s.shape == g[key + "_scale"].shape
    np.testing.assert_allclose(s, g[key + "_scale"], rtol=1e-6)
    if z is not None:
        assert np.array_equal(z, g[key + "_zp"])


def autogptq_inputs():
    """The seeded inputs of the reference's own GPTQ test (tests/openvino/native/quantization/test_gptq.py:355-371)."""
    np.random.seed(0)
    inputs = [np.random.rand(4, 128, 32).astype(np.float32) for _ in range(10)]
    weights = np.random.rand(4, 20, 32).astype(np.float32)
    return inputs, weights


@pytest.mark.parametrize("expert", [0, 1, 2, 3])
def test_gptq_against_the_independent_autogptq_reference(golden, expert):
    """The oracle's GPTQ (Hessian + column loop) against the outputs of the AutoGPTQ-derived torch implementation the
    reference's own test compares with (test_gptq.py:168-325), INT4_SYM, group 16, damp 0.1, same bar as that test
    (np.isclose on the Hessian, the scales and the quantized weights, :421-452)."""
    g = golden("gptq_autogptq")
    inputs, weights = autogptq_inputs()
    assert np.array_equal(inputs[0][0, 0], g["check_first_input_row"]) and np.array_equal(weights[0, 0], g["check_weight_row"])
    H = O.gptq_hessian([x[expert][None] for x in inputs])[0]
    assert np.all(np.isclose(g[f"H_{expert}"], H))
    q, s, _ = O.gptq_quantize_weights(H.copy(), weights[expert].copy(), "int4_sym", 16)
    assert np.all(np.isclose(g[f"scale_{expert}"].reshape(s.shape), s))
    assert np.all(np.isclose(g[f"weight_{expert}"], q))
