"""Known-answer vectors restated from the reference's own tests (data only; citations give file:line
under the reference checkout).  They pin oracle/wc_oracle.py independently of tools/gen_golden.py."""
import numpy as np
import pytest

from oracle import wc_oracle as O

EPS = np.float32(2 ** -23)

# tests/openvino/native/quantization/test_weights_compression.py:2226-2243
E2M1_EDGE = [(-7.0, -6.0), (7.0, 6.0), (-5.0, -4.0), (-3.5, -4.0), (1.75, 2.0), (2.5, 2.0), (-4.0, -4.0),
             (0.0, 0.0), (-0.0, 0.0), (-0.25, 0.0), (0.25, 0.0), (-0.49, -0.5), (-0.51, -0.5)]


@pytest.mark.parametrize("x,expected", E2M1_EDGE)
def test_mxfp4_edge_cases(x, expected):
    got = O.calculate_float_quantized_weight(np.array([x], np.float32), "mxfp4")
    assert got[0] == expected


def test_e2m1_rule_is_rne_on_dense_grid():
    # the reference tie rule equals IEEE round-to-nearest-even on the e2m1 grid, saturating at +-6
    x = np.linspace(-8, 8, 64001).astype(np.float32)
    got = O.calculate_float_quantized_weight(x, "mxfp4")
    grid = np.array([0, .5, 1, 1.5, 2, 3, 4, 6], np.float32)
    even = np.array([1, 0, 1, 0, 1, 0, 1, 0], bool)
    a = np.abs(x)
    d = np.abs(a[:, None] - grid[None])
    best = d.min(1, keepdims=True)
    cand = d == best
    pick = np.where(cand.sum(1) > 1, np.argmax(cand & even[None], 1), np.argmax(cand, 1))
    exp = np.sign(x) * grid[pick]
    assert np.array_equal(np.abs(got), np.abs(exp)) and np.all((got == exp))


def test_nf4_mid_quant():
    # test_weights_compression.py:2210-2223: norm weight one ulp above the first NF4 center -> 2nd quantile
    w = np.array([-0.07263] * 2, np.float16).astype(np.float32)
    s = np.array([0.08564102] * 2, np.float32)
    got = O.calculate_float_quantized_weight(w / s, "nf4")
    assert np.all(got == O.NF4_QUANTILES[1])


# SURVEY Appendix A11 (probed on the reference; numpy == torch), INT4, group size 8
DEGENERATE = [
    (np.zeros(8), [0] * 8, EPS, 0, [0] * 8, EPS),
    (np.full(8, .5), [15] * 8, EPS, 0, [7] * 8, 0.0625),
    (np.full(8, -.5), [0] * 8, EPS, 15, [-8] * 8, 0.0625),
    (np.full(8, -300.), [0] * 8, EPS, 0, [-8] * 8, 37.5),
    ([-7.5, -6.5, -.5, .5, 1.5, 2.5, 6.5, 7.5], [0, 2, 8, 8, 10, 10, 14, 15], 1.0, 8, [-8, -7, -1, 1, 2, 3, 7, 7],
     0.9375),
    (np.arange(1, 9), [2, 4, 6, 9, 11, 13, 15, 15], 0.46666667, 0, [-1, -2, -3, -4, -5, -6, -7, -8], -1.0),
    (-np.arange(1, 9), [13, 11, 9, 6, 4, 2, 0, 0], 0.46666667, 15, [-1, -2, -3, -4, -5, -6, -7, -8], 1.0),
    (np.full(8, 1e-40), [0] * 8, EPS, 0, [0] * 8, EPS),
    ([1e-3] * 7 + [10], [0] * 7 + [15], 0.66659999, 0, [0] * 7 + [-8], -1.25),
]


@pytest.mark.parametrize("row,qa,sa,za,qs,ss", DEGENERATE)
def test_degenerate_rows(row, qa, sa, za, qs, ss):
    w = np.asarray(row, np.float32)[None]
    c, s, z = O.compress_weight(w, (1,), "int4_asym", 8)
    assert c.ravel().tolist() == list(qa) and int(z.ravel()[0]) == za
    assert np.float32(s.ravel()[0]) == np.float32(sa)
    c, s, _ = O.compress_weight(w, (1,), "int4_sym", 8)
    assert c.ravel().tolist() == list(qs) and np.float32(s.ravel()[0]) == np.float32(ss)


# test_weights_compression.py:700-775 (test_quantization_error_calculation)
L = np.arange(0, 256, 17, dtype=np.float32)
TWO = np.vstack((L * 1.2, L * 3.4))
LA = np.arange(0, 16, dtype=np.float32)
TWOA = np.vstack((LA * 1.2, LA * 3.4))
LS = np.arange(-8, 7, dtype=np.float32)
TWOS = np.vstack((LS * 1.2, LS * 3.4))
ERR = [
    (TWO, "int8_asym", -1, 0, 1e-8), (TWOS, "int4_sym", -1, 0, 1e-8), (TWOA, "int4_asym", -1, 0, 1e-8),
    (np.hstack((TWOS, np.vstack((LS * 5.6, LS * 7.8)))), "int4_sym", 15, 0, 1e-8),
    (np.hstack((TWOA, np.vstack((LA * 5.6, LA * 7.8)))), "int4_asym", 16, 0, 1e-8),
    (TWO[:, 1:-1], "int8_asym", -1, 239, 1), (TWO.T, "int8_asym", -1, 46818, 1),
    (TWOA, "int4_sym", -1, 4.12, 1), (TWOA.T, "int4_sym", -1, 0.63, 1),
    (TWOA[:, 1:-1], "int4_asym", -1, 1.49, 1), (TWOA.T, "int4_asym", -1, 162, 1),
]


@pytest.mark.parametrize("w,mode,gs,ref,atol", ERR)
def test_quantization_error_calculation(w, mode, gs, ref, atol):
    assert np.allclose(O.get_integer_quantization_error(w, 1, mode, gs, "max_mean"), ref, atol=atol)


def test_calculate_scale_per_group():
    # test_weights_compression.py:788-830
    W = np.array([[-4, -3, -2, -1], [0, 11, 2, 3]])
    am = np.array([4, 2, 11, 3])
    cases = [(W, am.reshape(2, 2, 1), 1, 2), (W, np.abs(W).reshape(2, 1, 4), 0, 1),
             (W.reshape(1, 2, 4, 1), am.reshape(1, 2, 2, 1, 1), 2, 2),
             (W.reshape(2, 2, 2), np.array([4, 3, 2, 11]).reshape(2, 1, 1, 2), 1, 2),
             (W.reshape(2, 2, 2), np.array([4, 11, 2, 3]).reshape(1, 1, 2, 2), 0, 2)]
    for w, ref, axis, gs in cases:
        r, ax = O.reshape_weight_for_grouped_quantization(w, axis, gs)
        assert np.allclose(np.max(np.abs(r), axis=ax, keepdims=True), ref)
    with pytest.raises(O.InvalidGroupSizeError):
        O.reshape_weight_for_grouped_quantization(W, 1, 3)


def test_pack_roundtrip():
    # tests/torch/function_hook/quantization/test_weights_compression.py:557-572
    rng = np.random.default_rng(0)
    u = rng.integers(0, 16, (4, 4)).astype(np.uint8)
    p = O.pack_uint4(u)
    assert p.dtype == np.uint8 and p.size * 2 == u.size and p.shape == (8, 1)
    assert np.array_equal(O.unpack_uint4(p).reshape(u.shape), u)
    i = rng.integers(-8, 8, (4, 4)).astype(np.int8)
    assert np.array_equal(O.unpack_int4(O.pack_int4(i)).reshape(i.shape), i)
    with pytest.raises(TypeError):
        O.pack_uint4(i)


def test_reducer_literals():
    # tests/common/test_reducers_and_aggregators.py:62, 264-295 (axes None == all axes)
    x = np.array([[[1, 2, 0], [1, -3, 10]], [[-1, 2, -3], [4, 5, -6]]], np.float32)
    mean_var = {(0, 1, 2): 16.1666, (0,): 14.25, (0, 1): 15.875}
    mean_abs_max = {(0, 1, 2): 10.0, (0,): 4.16666, (0, 1): 6.33333}
    max_var = {(0, 1, 2): 16.1666, (0,): 64.0, (0, 1): 36.1875}
    for axes, ref in mean_var.items():
        assert np.isclose(O.mean_variance_reducer(x, axes), ref, atol=1e-4)
    for axes, ref in mean_abs_max.items():
        assert np.isclose(O.mean_abs_max_reducer(x, axes), ref, atol=1e-4)
    for axes, ref in max_var.items():
        assert np.isclose(O.max_variance_reducer(x, axes), ref, atol=1e-4)


def test_hawq_literals():
    # tests/common/test_reducers_and_aggregators.py:713-726
    assert np.isclose(O.hawq_aggregate([np.arange(10, dtype=np.float32)]), 57.0)
    assert np.isclose(O.hawq_aggregate([np.arange(12, dtype=np.float32).reshape(2, 6),
                                        np.arange(24, dtype=np.float32).reshape(4, 6)]), 181.92361111)
    assert np.isclose(O.hawq_aggregate([np.arange(8 * i, dtype=np.float32).reshape(1, 8, i) for i in range(1, 5)]),
                      165.61627, rtol=1e-6)


def test_mixed_precision_literals():
    # tests/cross_fw/test_templates/template_test_weights_compression.py:52-58
    x = np.array([[[-4, 1, 2], [0, 0, 0], [0, 0, 0]]], np.float32)
    assert np.isclose(O.max_variance_reducer(x, (1,)), 3.555555, atol=1e-5)
    assert np.isclose(O.mean_variance_reducer(x, (1,)), 1.555555, atol=1e-5)
    assert np.isclose(O.mean_abs_max_reducer(x, (1,)), 2.333333, atol=1e-5)
    assert np.isclose(O.hawq_aggregate([x]), (16 + 1 + 4) * 2 / 9)


def test_x86_int32_overflow_rule():
    v = np.array([2.5e9, -2.5e9, np.nan, 3.7, -3.7, 2147483520.0], np.float32)
    assert O.x86_cvtt_int32(v).tolist() == [-2 ** 31, -2 ** 31, -2 ** 31, 3, -3, 2147483520]


def test_nf4_centers_stay_clear_of_bucket_boundaries():
    """The HIP fast path (wc_float_policy.h, kNF4Margin) looks only at the center of t's own 1/16-wide bucket:
    sound as long as no center sits within the rounding margin of a bucket boundary."""
    c = np.asarray(O.CENTER_OF_NF4_QUANTILES, dtype=np.float64) * 16.0
    assert np.min(np.abs(c - np.round(c))) > 0.02
    buckets = np.floor(c).astype(int)
    assert len(set(buckets.tolist())) == len(buckets)  # at most one center per bucket


# ---- scale estimation literals of the reference's template test
from tests.se_literals import se_literal_inputs, se_literal_params  # noqa: E402


@pytest.mark.parametrize("mode,init,steps,transpose_a,n_samples,ref_scale,ref_zp", se_literal_params())
def test_scale_estimation_literals(mode, init, steps, transpose_a, n_samples, ref_scale, ref_zp):
    """template_test_weights_compression.py:566-623: transpose_a -> reduction axis 0 / act_ch_axis 1, else 1 / 2;
    subset_size 2, group size 8, atol 1e-3."""
    w, acts = se_literal_inputs(n_samples)
    means, shapes = O.activations_to_wc_statistics(acts)
    axes, ch = ((0,), 1) if transpose_a else ((1,), 2)
    sc, zp = O.scale_estimation_params(means, shapes, w, axes, mode, 8, ch, 2, init, steps)
    np.testing.assert_allclose(sc, np.array(ref_scale, np.float32).reshape(sc.shape), atol=1e-3, rtol=0)
    assert sc.dtype == np.float32
    if ref_zp is None:
        assert zp is None
    else:
        np.testing.assert_allclose(zp, np.array(ref_zp, np.float32).reshape(zp.shape), atol=1e-3, rtol=0)


_PROCESS_STATS_CASES = [
    # (reduced_shape, activation_shapes, subset_size, act_ch_axis | None, ref_s, ref_X)
    ((2,), [(1, 2), (3, 2), (5, 2), (10, 2)], 2, None, [6, 7], np.array([6, 2, 7, 3]).reshape(2, 2)),
    ((2,), [(2, 1), (2, 3), (2, 5), (2, 10)], 2, 0, [6, 7], np.array([6, 2, 7, 3]).reshape(2, 2)),
    ((2,), [(5, 2), (5, 2)], 2, None, [2, 3], np.array([0, 2, 1, 3]).reshape(2, 2)),
    ((2, 4), [(1, 2, 4), (3, 2, 4), (5, 2, 4), (10, 2, 4)], 2, None, np.arange(24, 32).reshape(2, 4),
     np.array([24, 8, 25, 9, 26, 10, 27, 11, 28, 12, 29, 13, 30, 14, 31, 15]).reshape(2, 4, 2)),
    ((2, 4), [(1, 100000, 2, 4), (3, 10000, 2, 4), (5, 1000, 2, 4), (10, 5, 2, 4)], 2, 1, np.arange(24, 32).reshape(2, 4),
     np.array([24, 8, 25, 9, 26, 10, 27, 11, 28, 12, 29, 13, 30, 14, 31, 15]).reshape(2, 4, 2)),
    ((2, 4), [(1, 2, 4), (1, 2, 4)], 2, None, np.arange(8, 16).reshape(2, 4),
     np.array([0, 8, 1, 9, 2, 10, 3, 11, 4, 12, 5, 13, 6, 14, 7, 15]).reshape(2, 4, 2)),
]


@pytest.mark.parametrize("case", range(len(_PROCESS_STATS_CASES)))
def test_process_stats_literals(case):
    """The reference's known answers for process_stats (template_test_weights_compression.py:980-1054): which
    calibration samples survive the subset sampling and the per-channel magnitudes; the oracle and the host mirror
    (plain torch, runs on the CPU)."""
    import torch

    from nncf_amd.statistics import WCTensorStatistic, process_stats
    reduced_shape, act_shapes, subset, axis, ref_s, ref_X = _PROCESS_STATS_CASES[case]
    total = int(np.prod(reduced_shape))
    means = [np.arange(i * total, (i + 1) * total, dtype=np.float32).reshape(reduced_shape) for i in range(len(act_shapes))]
    kw = {} if axis is None else {"act_ch_axis": axis}
    s, X = O.process_stats(means, act_shapes, subset, **kw)
    assert s.shape == np.shape(ref_s) and X.shape == ref_X.shape
    assert np.allclose(s, ref_s) and np.allclose(X, ref_X)
    s_t, X_t = process_stats(WCTensorStatistic([torch.from_numpy(m) for m in means], list(act_shapes)), subset, **kw)
    assert tuple(s_t.shape) == np.shape(ref_s) and tuple(X_t.shape) == ref_X.shape
    assert np.allclose(s_t.numpy(), ref_s) and np.allclose(X_t.numpy(), ref_X)


_RANGE_DATA = {"neg": [-8.0, -7.0, -6.0, -5.0, -4.0, -3.0, -2.0, -1.0, 0.0], "pos": [0.0, 1.0, 2.0, 3.0, 4.0, 5.0, 6.0, 7.0, 8.0],
               "neg-pos": [-8.0, -7.0, -6.0, -5.0, -4.0, -3.0, -2.0, -1.0, 0.0, 1.0, 2.0, 3.0, 4.0, 5.0, 6.0, 7.0]}
_FP4_RANGE_REF = {
    "neg": [-8.0, -8.0, -5.333333492279053, -5.333333492279053, -4.0, -2.6666667461395264, -2.0, -1.3333333730697632, -0.0],
    "pos": [-0.0, 1.3333333730697632, 2.0, 2.6666667461395264, 4.0, 5.333333492279053, 5.333333492279053, 8.0, 8.0],
    "neg-pos": [-8.0, -8.0, -5.333333492279053, -5.333333492279053, -4.0, -2.6666667461395264, -2.0, -1.3333333730697632, -0.0,
                1.3333333730697632, 2.0, 2.6666667461395264, 4.0, 5.333333492279053, 5.333333492279053, 8.0]}
_MXFP4_RANGE_REF = {"neg": [-8.0, -8.0, -6.0, -4.0, -4.0, -3.0, -2.0, -1.0, -0.0], "pos": [-0.0, 1.0, 2.0, 3.0, 4.0, 4.0, 6.0, 8.0, 8.0],
                    "neg-pos": [-8.0, -8.0, -6.0, -4.0, -4.0, -3.0, -2.0, -1.0, -0.0, 1.0, 2.0, 3.0, 4.0, 4.0, 6.0, 8.0]}
RANGE_REFS = {"mxfp4": _MXFP4_RANGE_REF, "fp4": _FP4_RANGE_REF}


@pytest.mark.parametrize("ident", ["neg", "pos", "neg-pos"])
def test_int_compressed_weights_range(ident):
    """tests/openvino/native/quantization/test_weights_compression.py:1489-1507: INT4_SYM codes of an integer ramp
    equal the ramp in magnitude."""
    w = np.array(_RANGE_DATA[ident], np.float32)
    codes, _, _ = O.do_integer_quantization(w, "int4_sym", -1, (-1,))
    assert np.allclose(np.abs(codes), np.abs(w))


@pytest.mark.parametrize("mode", ["mxfp4", "fp4"])
@pytest.mark.parametrize("ident", ["neg", "pos", "neg-pos"])
def test_float_compressed_weights_range(mode, ident):
    """:1510-1626 (TEST_FLOAT_COMPRESSED_REFS, the modes in scope): quantize + dequantize of the ramps."""
    w = np.array(_RANGE_DATA[ident], np.float32)
    values, scale = O.do_float_quantization(w, mode, -1, (-1,))
    assert np.allclose(O.do_float_dequantization(values, scale), np.array(RANGE_REFS[mode][ident], np.float32))


PRECOMPUTED_CASES = [("int8_asym", False, False, False), ("int8_asym", True, True, False), ("int8_asym", True, False, True),
                     ("int8_asym", False, True, True), ("int4_asym", False, False, False), ("int4_asym", True, True, False),
                     ("int4_asym", True, False, True), ("int4_asym", False, True, True), ("int8_sym", True, False, False),
                     ("int8_sym", False, False, False), ("int4_sym", True, False, False), ("int4_sym", False, False, False)]


def precomputed_inputs(pre_scale, pre_zp):
    weight = ((np.arange(11) - 5) / 10).astype(np.float32)[:, None]
    scale = -((np.arange(11) - 5) / 100).astype(np.float32)[:, None] if pre_scale else None
    zp = np.arange(11).astype(np.int32)[:, None] if pre_zp else None
    return weight, scale, zp


@pytest.mark.parametrize("mode,pre_scale,pre_zp,raises", PRECOMPUTED_CASES)
def test_int_quantization_with_precomputed_parameters(mode, pre_scale, pre_zp, raises):
    """:1646-1688: given parameters come back verbatim (a zero scale among them included), half a pair raises."""
    weight, scale, zp = precomputed_inputs(pre_scale, pre_zp)
    if raises:
        with pytest.raises(ValueError):
            O.do_integer_quantization(weight, mode, -1, (-1,), scale, zp)
        return
    _, s, z = O.do_integer_quantization(weight, mode, -1, (-1,), scale, zp)
    if pre_scale:
        assert np.allclose(s, scale)
    if mode.endswith("asym"):
        if pre_zp:
            assert np.allclose(z, zp)
    else:
        assert z is None


def test_process_stats_memo_follows_the_statistics():
    """nncf_amd.statistics.process_stats keeps its result on the statistics object (the Linears reading one activation
    ask for the same (s, X)): same object again while nothing changed; recomputed after another subset size, an
    in-place edit of a mean, or a replaced list."""
    import torch

    from nncf_amd.statistics import WCTensorStatistic, process_stats

    st = WCTensorStatistic([torch.randn(8) for _ in range(5)], [(1, 3 + i, 8) for i in range(5)])
    a = process_stats(st, 4)
    assert process_stats(st, 4) is a
    c = process_stats(st, 2)
    assert c is not a and c[1].shape[-1] == 3  # every (5 // 2)-th of the 5 samples (activation_stats.py:44-50)
    st.mean_values[0].mul_(2)
    d = process_stats(st, 2)
    assert d is not c and not torch.equal(d[0], c[0])
    st.mean_values = [m.clone() for m in st.mean_values]
    e = process_stats(st, 2)
    assert e is not d and torch.equal(e[0], d[0]) and torch.equal(e[1], d[1])
    st.shape_values = [(1, 9 - i, 8) for i in range(5)]
    f = process_stats(st, 2)
    assert f is not e and not torch.equal(f[1], e[1])
