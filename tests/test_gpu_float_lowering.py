"""GPU parity: HIP NF4 / MXFP4 lowering vs the oracle and the golden fixtures (values bit exact, scales bit exact)."""
import numpy as np
import pytest
import torch

from oracle import wc_oracle as O
from tests.gpu_utils import host, representable, to_dev
from tests.helpers import assert_equal_nan, golden_weight

pytestmark = pytest.mark.gpu


def _cfg(mode, gs):
    from nncf_amd.parameters import WeightCompressionConfig
    return WeightCompressionConfig(mode=mode, group_size=gs)


@pytest.mark.parametrize("dtype", ["f32", "f16", "bf16"])
@pytest.mark.parametrize("gs", [-1, 32, 64])
@pytest.mark.parametrize("mode", ["nf4", "mxfp4"])
def test_golden_float_lowering(golden, mode, gs, dtype):
    from nncf_amd import weight_lowering as wl
    g = golden("float_lowering")
    w = to_dev(golden_weight(g, dtype), dtype)
    key = f"{mode}_g{gs}_{dtype}"
    cw = wl.compress_weight(w, (1,), _cfg(mode, gs))
    assert_equal_nan(host(cw.scale), g[key + "_scale"], key + " scale")
    assert_equal_nan(host(cw.tensor), g[key + "_values"], key + " values")
    if dtype == "f32":
        assert_equal_nan(host(wl.do_float_dequantization(cw)), g[key + "_dq"], key + " dq")
        assert_equal_nan(host(wl.float_quantize_dequantize_weight(w, _cfg(mode, gs), (1,))), g[key + "_dq"])


@pytest.mark.parametrize("mode", ["nf4", "mxfp4"])
def test_golden_boundaries(golden, mode):
    from nncf_amd import weight_lowering as wl
    g = golden("float_lowering")
    cw = wl.compress_weight(to_dev(g["w_boundary"], "f32"), (1,), _cfg(mode, 32))
    assert_equal_nan(host(cw.scale), g[f"boundary_{mode}_scale"], "boundary scale")
    assert_equal_nan(host(cw.tensor), g[f"boundary_{mode}_values"], "boundary values")
    mid = "e2m1" if mode == "mxfp4" else "nf4"
    cw = wl.compress_weight(to_dev(g[f"w_mid_{mid}"], "f32"), (1,), _cfg(mode, -1))
    assert_equal_nan(host(cw.scale), g[f"mid_{mid}_scale"])
    assert_equal_nan(host(cw.tensor), g[f"mid_{mid}_values"])
    sc = torch.from_numpy(g["pre_scale"]).cuda()
    dq, cw = wl.float_quantize_dequantize_weight(to_dev(g["pre_w"], "f32"), _cfg(mode, 32), None, sc, True)
    assert_equal_nan(host(cw.tensor), g[f"pre_{mode}_values"])
    assert_equal_nan(host(dq), g[f"pre_{mode}_dq"])
    assert cw.scale is sc


SHAPES = [((64, 512), 1, 32), ((64, 512), 1, 64), ((9, 512), 1, 128), ((5, 768), 1, -1), ((3, 4096), 1, -1),
          ((2, 40000), 1, -1), ((10, 30), 1, 15), ((48, 20), 0, 16), ((4, 32, 6), 1, 8), ((17, 64), 1, 8)]


@pytest.mark.parametrize("dtype", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("shape,axis,gs", SHAPES)
@pytest.mark.parametrize("mode", ["nf4", "mxfp4", "fp4"])
def test_oracle_float_shapes(mode, shape, axis, gs, dtype):
    from nncf_amd import weight_lowering as wl
    from nncf_amd.weight_lowering import _float_quantize
    rng = np.random.default_rng(hash((shape, axis, gs)) % 2**31)
    w32 = representable((rng.standard_normal(shape) * 0.03).astype(np.float32), dtype)
    w = to_dev(w32, dtype)
    dq, cw = wl.float_quantize_dequantize_weight(w, _cfg(mode, gs), (axis,), return_compressed_weight=True)
    rdq, (rv, rs) = O.float_quantize_dequantize_weight(w32, mode, gs, (axis,), return_compressed_weight=True)
    assert_equal_nan(host(cw.scale), rs, "scale")
    assert_equal_nan(host(cw.tensor), rv, "values")
    assert_equal_nan(host(dq), rdq, "dq")
    # the build's 4-bit codes: LUT[code] == reference value
    _, codes, _, _, _, e8 = _float_quantize(w, _cfg(mode, gs), (axis,), None, want_values=False, want_codes=True,
                                            want_e8m0=(mode == "mxfp4"))
    lut = O.NF4_QUANTILES if mode == "nf4" else O.E2M1_VALUE_OF_CODE
    assert np.array_equal(lut[host(codes)], rv)
    if mode == "mxfp4":
        assert np.array_equal(np.ldexp(np.float32(1), host(e8).astype(np.int32) - 127), rs)
    if w32.size % 2 == 0 and axis == w32.ndim - 1:
        fused = wl.compress_weight_packed(w, (axis,), _cfg(mode, gs))
        c = host(codes).reshape(-1, 2)
        assert np.array_equal(host(fused.tensor).ravel(), (c[:, 0] | (c[:, 1] << 4)).astype(np.uint8))


def test_dense_normalized_grid():
    """every decision boundary +-2 ulp, saturation, zeros: unit scale via precomputed scale."""
    from nncf_amd import weight_lowering as wl
    for mode, centers, hi in (("nf4", O.CENTER_OF_NF4_QUANTILES, 1.5), ("mxfp4", O.CENTER_OF_F4E2M1_QUANTILES, 8.0)):
        pts = [np.linspace(-hi, hi, 20001).astype(np.float32)]
        for c in np.concatenate([centers, O.NF4_QUANTILES if mode == "nf4" else O.F4E2M1_QUANTILES]):
            x = np.float32(c)
            lo = x
            up = x
            for _ in range(3):
                lo = np.nextafter(lo, np.float32(-np.inf))
                up = np.nextafter(up, np.float32(np.inf))
                pts.append(np.array([lo, up], np.float32))
            pts.append(np.array([x], np.float32))
        n = np.concatenate(pts + [np.array([0.0, -0.0, 1e-30, -1e-30, 100.0, -100.0], np.float32)])
        n = n[: n.size // 8 * 8].reshape(-1, 8)
        sc = np.ones((n.shape[0], 1), np.float32)
        cw = wl.do_float_quantization(to_dev(n, "f32"), _cfg(mode, 8), None, torch.from_numpy(sc).cuda())
        rv, _ = O.do_float_quantization(n, mode, 8, None, sc)
        assert_equal_nan(host(cw.tensor), rv, mode)


def test_mx_scale_boundaries_dense():
    """group maxima 6*2^k*(1+m*2^-23) for every k in [-24, 100], m in [0, 24]: E8M0 exponent parity."""
    from nncf_amd import weight_lowering as wl
    rows = []
    for k in range(-24, 101):
        for m in range(0, 25):
            s = np.float32(2.0 ** k) * (np.float32(1) + np.float32(m) * np.float32(2.0 ** -23))
            r = np.zeros(32, np.float32)
            r[3] = -s * np.float32(6.0)
            r[4] = s
            rows.append(r)
    w32 = np.stack(rows)
    cw = wl.compress_weight(to_dev(w32, "f32"), (1,), _cfg("mxfp4", 32))
    rv, rs, _ = O.compress_weight(w32, (1,), "mxfp4", 32)
    assert_equal_nan(host(cw.scale), rs, "mx scale")
    assert_equal_nan(host(cw.tensor), rv, "mx values")


def _unpack_nibbles(packed):
    p = host(packed).ravel()
    return np.stack([p & 15, p >> 4], axis=1).ravel()


def _adversarial_groups(mode, gs, rng, big=2.0 ** 20):
    """Groups whose normalised weights sit on / within a few ulp of every decision boundary, for scales that
    are and are not powers of two; plus zeros, negative underflow (the reference has a single zero) and saturation."""
    centers = O.CENTER_OF_NF4_QUANTILES if mode == "nf4" else O.CENTER_OF_F4E2M1_QUANTILES
    top = np.float32(1.0 if mode == "nf4" else 6.0)
    anchors = [1.0, 0.75, 3.1415927, 1e-3, 1.9999999, 2.0 ** -12 * 1.2345, 37.5, big, 5e-8]
    rows = []
    for a in anchors:
        amax = np.float32(a) * top
        scale = np.float32(a) if mode == "nf4" else np.float32(2.0) ** np.ceil(np.log2(np.float64(np.float32(amax / np.float32(6)))))
        cand = []
        for c in np.concatenate([centers, [0.0]]):
            x = np.float32(np.float32(c) * np.float32(scale))
            lo = up = x
            cand.append(x)
            for _ in range(3):
                lo = np.nextafter(lo, np.float32(-np.inf))
                up = np.nextafter(up, np.float32(np.inf))
                cand += [lo, up]
        cand += [np.float32(-1e-9) * scale, np.float32(1e-9) * scale, np.float32(-0.0), -amax, amax * np.float32(0.999999)]
        cand = np.array(cand, np.float32)
        cand = cand[np.abs(cand) <= amax]
        n_rows = -(-cand.size // (gs - 1))
        buf = np.zeros((n_rows, gs), np.float32)
        buf[:, 0] = amax  # pins the group scale
        body = np.resize(cand, n_rows * (gs - 1)).reshape(n_rows, gs - 1)
        buf[:, 1:] = body
        rows.append(buf)
    w = np.concatenate(rows)
    extra = (rng.standard_normal((7, gs)) * 0.05).astype(np.float32)
    return np.concatenate([w, extra])


@pytest.mark.parametrize("dtype", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("mode", ["nf4", "mxfp4"])
def test_packed_fast_paths_adversarial(mode, dtype):
    """The multi-tensor kernel's NF4 reciprocal fast path (margin + exact redo) and MXFP4 hardware e2m1 convert
    against the oracle, on weights built to sit on the decision boundaries."""
    from nncf_amd import weight_lowering as wl
    rng = np.random.default_rng(7)
    gs = 32
    w32 = representable(_adversarial_groups(mode, gs, rng, big=2.0 ** 12 if dtype == "f16" else 2.0 ** 20), dtype)  # no inf: NaN is outside the contract
    w32 = w32.reshape(-1, 4 * gs) if w32.shape[0] % 4 == 0 else np.concatenate([w32, w32[: 4 - w32.shape[0] % 4]]).reshape(-1, 4 * gs)
    lut = O.NF4_QUANTILES if mode == "nf4" else O.E2M1_VALUE_OF_CODE
    ws = [w32, w32[:5], np.ascontiguousarray(w32[::-1])]
    got = wl.compress_weights_packed_batched([to_dev(x, dtype) for x in ws], _cfg(mode, gs), want_e8m0=(mode == "mxfp4"))
    for x, cw in zip(ws, got):
        rv, rs, _ = O.compress_weight(x, (1,), mode, gs)
        assert_equal_nan(host(cw.scale).reshape(rs.shape), rs, "scale")
        assert np.array_equal(lut[_unpack_nibbles(cw.tensor)], rv.ravel()), f"{mode} {dtype} values"
        if mode == "mxfp4":
            assert np.array_equal(np.ldexp(np.float32(1), host(cw.codebook).astype(np.int32) - 127).reshape(rs.shape), rs)
    single = wl.compress_weight_packed(to_dev(w32, dtype), (1,), _cfg(mode, gs))
    assert np.array_equal(host(single.tensor), host(got[0].tensor))


@pytest.mark.parametrize("mode", ["nf4", "mxfp4"])
def test_c5_batched_full_size_vs_oracle(mode):
    """BASELINE configs[4] shapes (Mixtral expert w1 14336 x 4096 and w2 4096 x 14336, g32) + a ragged tensor whose
    last tile is partial, through the multi-tensor kernel in one launch, against the oracle on the WHOLE tensors:
    every 4-bit code decodes to the oracle's value, scales (and E8M0 bytes) bit-exact."""
    from nncf_amd import weight_lowering as wl
    torch.manual_seed(9)
    ws = [(torch.randn(r, c, device="cuda") * 0.02).to(torch.bfloat16) for r, c in [(14336, 4096), (4096, 14336), (333, 4096)]]
    ws[2][5, 64:96] = 0.0
    got = wl.compress_weights_packed_batched(ws, _cfg(mode, 32), want_e8m0=True)
    lut = O.NF4_QUANTILES if mode == "nf4" else O.E2M1_VALUE_OF_CODE
    for w, cw in zip(ws, got):
        rc, rs, _ = O.compress_weight(host(w.float()), (1,), mode, 32)
        assert_equal_nan(host(cw.scale).reshape(rs.shape), rs, "scale")
        p = host(cw.tensor).ravel()
        codes = np.stack([p & 15, p >> 4], -1).ravel()
        assert np.array_equal(lut[codes], rc.ravel())
        if mode == "mxfp4":
            assert np.array_equal(np.ldexp(np.float32(1), host(cw.codebook).astype(np.int32).ravel() - 127), rs.ravel())


@pytest.mark.parametrize("mode", ["mxfp4", "fp4"])
@pytest.mark.parametrize("ident", ["neg", "pos", "neg-pos"])
def test_reference_float_range_literals(mode, ident):
    """tests/openvino/native/quantization/test_weights_compression.py:1510-1626 (the modes in scope) through the HIP
    path: quantize + dequantize of the integer ramps, 1-D weight."""
    from nncf_amd import weight_lowering as WL
    from tests.test_oracle_known_answers import _RANGE_DATA, RANGE_REFS
    w = np.array(_RANGE_DATA[ident], np.float32)
    cw = WL.do_float_quantization(to_dev(w, "f32"), _cfg(mode, -1), -1)
    assert np.allclose(host(WL.do_float_dequantization(cw)), np.array(RANGE_REFS[mode][ident], np.float32))
