"""GPU parity: HIP integer lowering (through the C ABI) vs the oracle and the golden fixtures.
Codes, zero points and packed bytes must be bit exact; scales are compared bit exact as well (the
arithmetic is IEEE fp32 with the same operation order; north_star allows <=1e-5 relative)."""
import numpy as np
import pytest
import torch

from oracle import wc_oracle as O
from tests.gpu_utils import host, representable, to_dev
from tests.helpers import assert_equal_nan, golden_weight

pytestmark = pytest.mark.gpu

INT_MODES = ("int8_asym", "int8_sym", "int4_asym", "int4_sym")


def _cfg(mode, gs):
    from nncf_amd.parameters import WeightCompressionConfig
    return WeightCompressionConfig(mode=mode, group_size=gs)


@pytest.mark.parametrize("dtype", ["f32", "f16", "bf16"])
@pytest.mark.parametrize("gs", [-1, 32, 64, 128])
@pytest.mark.parametrize("mode", INT_MODES)
def test_golden_int_lowering(golden, mode, gs, dtype):
    from nncf_amd import weight_lowering as wl
    g = golden("int_lowering")
    w = to_dev(golden_weight(g, dtype), dtype)
    cw = wl.compress_weight(w, (1,), _cfg(mode, gs))
    key = f"{mode}_g{gs}_{dtype}"
    assert tuple(cw.tensor.shape) == g[key + "_codes"].shape
    assert_equal_nan(host(cw.tensor), g[key + "_codes"], key + " codes")
    assert_equal_nan(host(cw.scale), g[key + "_scale"], key + " scale")
    if mode.endswith("asym"):
        assert cw.zero_point.dtype == torch.int32
        assert_equal_nan(host(cw.zero_point), g[key + "_zp"], key + " zp")
    else:
        assert cw.zero_point is None
    if dtype == "f32":
        assert_equal_nan(host(wl.do_integer_dequantization(cw)), g[key + "_dq"], key + " dq")
        dq, cw2 = wl.integer_quantize_dequantize_weight(w, _cfg(mode, gs), (1,), return_compressed_weight=True)
        assert_equal_nan(host(dq), g[key + "_dq"], key + " fused dq")
        assert_equal_nan(host(cw2.tensor), g[key + "_codes"], key + " fused codes")
        if mode.startswith("int4"):
            packed = wl.pack_uint4(cw.tensor) if mode == "int4_asym" else wl.pack_int4(cw.tensor)
            assert_equal_nan(host(packed), g[key + "_packed"], key + " packed")
            fused = wl.compress_weight_packed(w, (1,), _cfg(mode, gs))
            assert_equal_nan(host(fused.tensor), g[key + "_packed"], key + " fused packed")
            assert_equal_nan(host(fused.scale), g[key + "_scale"], key + " fused scale")
            un = wl.unpack_uint4(packed) if mode == "int4_asym" else wl.unpack_int4(packed)
            assert_equal_nan(host(un).reshape(g[key + "_codes"].shape), g[key + "_codes"], key + " unpack")


@pytest.mark.parametrize("mode", INT_MODES)
def test_golden_precomputed(golden, mode):
    from nncf_amd import weight_lowering as wl
    g = golden("int_lowering")
    w = to_dev(g["pre_w"], "f32")
    sc = torch.from_numpy(g["pre_scale"]).cuda()
    zi = torch.from_numpy(g["pre_zp_i32"]).cuda()
    if mode.endswith("asym"):
        for zname, z in (("i32", zi), ("f32", zi.float())):
            dq, cw = wl.integer_quantize_dequantize_weight(w, _cfg(mode, 64), None, sc, z, True)
            assert_equal_nan(host(cw.tensor), g[f"pre_{mode}_{zname}_codes"])
            assert_equal_nan(host(dq), g[f"pre_{mode}_{zname}_dq"])
            assert cw.scale is sc and cw.zero_point is z  # returned verbatim
        with pytest.raises(ValueError):
            wl.do_integer_quantization(w, _cfg(mode, 64), None, sc, None)
    else:
        dq, cw = wl.integer_quantize_dequantize_weight(w, _cfg(mode, 64), None, sc, None, True)
        assert_equal_nan(host(cw.tensor), g[f"pre_{mode}_codes"])
        assert_equal_nan(host(dq), g[f"pre_{mode}_dq"])


@pytest.mark.parametrize("red", ["max_mean", "frobenius"])
@pytest.mark.parametrize("gs", [-1, 64])
@pytest.mark.parametrize("mode", INT_MODES)
def test_golden_error_metric(golden, mode, gs, red):
    # tolerance of the reference's own alignment test (test_compression_functions.py:284-324): reductions reorder
    from nncf_amd import weight_lowering as wl
    g = golden("int_lowering")
    got = wl.get_integer_quantization_error(to_dev(g["err_w"], "f32"), (1,), _cfg(mode, gs), red)
    np.testing.assert_allclose(got, float(g[f"err_{mode}_g{gs}_{red}"]), rtol=1e-4, atol=1e-6)


SHAPES = [
    # (shape, reduction axis, group size): every kernel path -- grouped (LPG 1..64), row (wave / block teams),
    # generic (odd sizes, inner > 1)
    ((64, 512), 1, 128), ((64, 512), 1, 64), ((64, 512), 1, 32), ((33, 512), 1, 16), ((17, 64), 1, 8),
    ((7, 1024), 1, 512), ((5, 768), 1, -1), ((3, 4096), 1, -1), ((2, 14336), 1, -1), ((2, 40000), 1, -1),
    ((10, 30), 1, 15), ((10, 30), 1, -1), ((6, 7), 1, -1), ((48, 20), 0, 16), ((48, 20), 0, -1),
    ((4, 32, 6), 1, 8), ((2, 3, 64), 2, 32), ((1, 128), 1, 128), ((1, 8), 1, 4), ((513, 256), 1, 128),
]


@pytest.mark.parametrize("dtype", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("shape,axis,gs", SHAPES)
@pytest.mark.parametrize("mode", INT_MODES)
def test_oracle_int_shapes(mode, shape, axis, gs, dtype):
    from nncf_amd import weight_lowering as wl
    rng = np.random.default_rng(hash((shape, axis, gs)) % 2**31)
    w32 = representable((rng.standard_normal(shape) * 0.03).astype(np.float32), dtype)
    w = to_dev(w32, dtype)
    dq, cw = wl.integer_quantize_dequantize_weight(w, _cfg(mode, gs), (axis,), return_compressed_weight=True)
    rdq, (rc, rs, rz) = O.integer_quantize_dequantize_weight(w32, mode, gs, (axis,), return_compressed_weight=True)
    assert_equal_nan(host(cw.tensor), rc, "codes")
    assert_equal_nan(host(cw.scale), rs, "scale")
    if rz is not None:
        assert_equal_nan(host(cw.zero_point), rz, "zp")
    assert_equal_nan(host(dq), rdq, "dq")
    s2, z2 = wl.calculate_integer_quantization_params(w, (axis,), _cfg(mode, -1))
    rs2, rz2 = O.calculate_integer_quantization_params(w32, (axis,), mode)
    assert_equal_nan(host(s2), rs2, "params scale")
    if rz2 is not None:
        assert_equal_nan(host(z2), rz2, "params zp")
    if mode.startswith("int4") and w32.size % 2 == 0 and axis == w32.ndim - 1:
        fused = wl.compress_weight_packed(w, (axis,), _cfg(mode, gs))
        ref = O.pack_uint4(rc) if mode == "int4_asym" else O.pack_int4(rc)
        assert_equal_nan(host(fused.tensor), ref, "fused packed")


def test_adversarial_values_match_oracle():
    """rows built to hit ties (x.5), near-ties, saturation, denormals, huge values, constant groups."""
    from nncf_amd import weight_lowering as wl
    rng = np.random.default_rng(11)
    rows = []
    for _ in range(64):
        s = np.float32(rng.uniform(0.001, 3.0))
        k = rng.integers(-8, 8, 128).astype(np.float32) + 0.5
        r = k * s
        r[0], r[1] = -8 * s, 7 * s  # pin min/max so that the scale is s (sym) or near it
        r[2::7] = np.nextafter(r[2::7], np.float32(np.inf))
        r[3::7] = np.nextafter(r[3::7], np.float32(-np.inf))
        rows.append(r.astype(np.float32))
    rows.append(np.full(128, 1e-40, np.float32))
    rows.append(np.full(128, -300.0, np.float32))
    rows.append(np.concatenate([np.full(64, 3e38, np.float32), np.full(64, -3e38, np.float32)]))
    rows.append(np.concatenate([np.zeros(127, np.float32), np.float32([1e-30])]))
    w32 = np.stack(rows)
    w = to_dev(w32, "f32")
    for mode in INT_MODES:
        for gs in (128, 32, -1):
            cw = wl.compress_weight(w, (1,), _cfg(mode, gs))
            rc, rs, rz = O.compress_weight(w32, (1,), mode, gs)
            assert_equal_nan(host(cw.tensor), rc, f"{mode} g{gs} codes")
            assert_equal_nan(host(cw.scale), rs, f"{mode} g{gs} scale")
            if rz is not None:
                assert_equal_nan(host(cw.zero_point), rz, f"{mode} g{gs} zp")


def test_error_behaviour():
    from nncf_amd import errors
    from nncf_amd import weight_lowering as wl
    w = torch.randn(4, 30, device="cuda")
    with pytest.raises(errors.InvalidGroupSizeError):
        wl.compress_weight(w, (1,), _cfg("int4_sym", 16))
    with pytest.raises(errors.InternalError):
        wl.do_integer_quantization(w, _cfg("nf4", -1), (1,))
    with pytest.raises(ValueError):
        wl.do_integer_quantization(w, _cfg("int4_sym", -1), None)
    with pytest.raises(errors.UnsupportedBackendError):
        wl.compress_weight(w.cpu(), (1,), _cfg("int4_sym", -1))
    with pytest.raises(errors.ValidationError):
        wl.pack_uint4(torch.zeros(4, dtype=torch.int8, device="cuda"))
    with pytest.raises(errors.InternalError):
        wl.get_integer_quantization_error(w, (1,), _cfg("int8_asym", -1), "l1")
    empty = wl.compress_weight(torch.empty(0, 16, device="cuda"), (1,), _cfg("int8_asym", -1))
    assert empty.tensor.numel() == 0


def test_large_tensor_properties():
    """Llama-3-8B gate_proj shape: size-independent properties at full size + oracle on a row slice."""
    from nncf_amd import weight_lowering as wl
    torch.manual_seed(0)
    w = (torch.randn(14336, 4096, device="cuda") * 0.02).to(torch.bfloat16)
    cfg = _cfg("int4_sym", 128)
    cw = wl.compress_weight(w, (1,), cfg)
    fused = wl.compress_weight_packed(w, (1,), cfg)
    assert torch.equal(wl.pack_int4(cw.tensor), fused.tensor)
    assert torch.equal(wl.unpack_int4(fused.tensor).reshape(cw.tensor.shape), cw.tensor)
    assert torch.equal(cw.scale, fused.scale)
    assert int(cw.tensor.min()) == -8 and int(cw.tensor.max()) <= 7
    # every group attains -8 (the signed-scale convention maps the extreme element to -8)
    assert bool((cw.tensor.amin(dim=-1) == -8).all())
    # idempotence: re-quantizing the dequantized weight with the same scale gives the same codes
    dq = wl.do_integer_dequantization(cw)
    again = wl.do_integer_quantization(dq, cfg, None, cw.scale)
    assert torch.equal(again.tensor, cw.tensor)
    sl = slice(5000, 5064)
    rc, rs, _ = O.compress_weight(w[sl].float().cpu().numpy(), (1,), "int4_sym", 128)
    assert_equal_nan(host(cw.tensor[sl]), rc)
    assert_equal_nan(host(cw.scale[sl]), rs)


@pytest.mark.parametrize("dtype", ["bf16", "f32"])
@pytest.mark.parametrize("mode,gs", [("int4_sym", 128), ("int4_asym", 64), ("int4_sym", 32), ("int4_sym", -1),
                                     ("int8_asym", -1), ("nf4", 32), ("mxfp4", 32), ("int4_asym", 8)])
def test_batched_matches_single_and_oracle(mode, gs, dtype):
    """multi-tensor launches: ragged tensor sizes (partial tiles, tiny tensors), more tensors than one launch takes."""
    from nncf_amd import weight_lowering as wl
    rng = np.random.default_rng(3)
    shapes = [(1, 256), (3, 256), (8, 1024), (64, 512), (5, 768) if gs in (-1, 32, 64, 8) else (5, 1024), (129, 256),
              (16, 2048), (2, 4096)] * 13  # 104 tensors (92 per launch)
    ws32 = [representable((rng.standard_normal(s) * 0.05).astype(np.float32), dtype) for s in shapes]
    ws = [to_dev(w, dtype) for w in ws32]
    cfg = _cfg(mode, gs)
    got = wl.compress_weights_packed_batched(ws, cfg, want_e8m0=True)
    assert len(got) == len(ws)
    for w, w32, cw in zip(ws, ws32, got):
        single = wl.compress_weight_packed(w, (1,), cfg)
        assert torch.equal(cw.tensor.reshape(-1), single.tensor.reshape(-1))
        assert torch.equal(cw.scale.reshape(-1), single.scale.reshape(-1))
        rc, rs, rz = O.compress_weight(w32, (1,), mode, gs)
        assert_equal_nan(host(cw.scale).reshape(rs.shape), rs, "scale")
        if mode.startswith("int"):
            if mode.startswith("int4"):
                ref = O.pack_uint4(rc) if mode == "int4_asym" else O.pack_int4(rc)
            else:
                ref = rc
            assert_equal_nan(host(cw.tensor).reshape(ref.shape), ref, "codes")
            if rz is not None:
                assert_equal_nan(host(cw.zero_point).reshape(rz.shape), rz, "zp")
        else:
            p = host(cw.tensor).ravel()
            codes = np.stack([p & 15, p >> 4], -1).ravel()
            lut = O.NF4_QUANTILES if mode == "nf4" else O.E2M1_VALUE_OF_CODE
            assert np.array_equal(lut[codes], rc.ravel())
            if mode == "mxfp4":
                assert np.array_equal(np.ldexp(np.float32(1), host(cw.codebook).astype(np.int32).ravel() - 127), rs.ravel())


def _oracle_packed(w, mode, gs):
    """(packed codes or int8 codes, scale, zp) of the oracle for a device tensor."""
    rc, rs, rz = O.compress_weight(host(w.float()), (1,), mode, gs)
    if mode == "int4_sym":
        rc = O.pack_int4(rc)
    elif mode == "int4_asym":
        rc = O.pack_uint4(rc)
    return rc, rs, rz


def _assert_cw_equals_oracle(cw, ref, what=""):
    rc, rs, rz = ref
    assert_equal_nan(host(cw.tensor).reshape(rc.shape), rc, what + " codes")
    assert_equal_nan(host(cw.scale).reshape(rs.shape), rs, what + " scale")
    if rz is not None:
        assert_equal_nan(host(cw.zero_point).reshape(rz.shape), rz, what + " zp")


def test_many_launches_of_changing_sizes():
    """Many multi-tensor launches of changing sizes and both kernel families (tile / row) on one stream, back to
    back without synchronisation: every result is compared with the ORACLE's packed codes, scales, zero points."""
    from nncf_amd import weight_lowering as wl
    torch.manual_seed(1)
    sets = [("int4_sym", 128, [(4096, 4096), (1024, 4096), (14336, 4096)]), ("int4_sym", 128, [(8, 128)]),
            ("int4_asym", 128, [(3, 256), (515, 1024), (64, 2048)]), ("int8_asym", -1, [(2048, 8192)] * 3),
            ("int4_sym", 128, [(2048, 8192)] * 5), ("int8_sym", -1, [(77, 512), (300, 14336)])]
    data = [(m, gs, [(torch.randn(r, c, device="cuda") * 0.02).to(torch.bfloat16) for r, c in shapes])
            for m, gs, shapes in sets]
    want = [[_oracle_packed(w, m, gs) for w in ws] for m, gs, ws in data]
    for it in range(90):
        k = it % len(data)
        m, gs, ws = data[k]
        got = wl.compress_weights_packed_batched(ws, _cfg(m, gs))
        if it < 12 or it % 7 == 0 or it > 80:
            for ref, cw in zip(want[k], got):
                _assert_cw_equals_oracle(cw, ref, f"iteration {it} set {k}")


def test_streams_and_graph_replay():
    """(1) multi-tensor launches on several streams at once; (2) a recorded multi-tensor launch is replayed (the
    library keeps no device-side state between launches).  All results against the oracle."""
    from nncf_amd import weight_lowering as wl
    torch.manual_seed(2)
    cfg = _cfg("int4_sym", 128)
    ws = [(torch.randn(r, c, device="cuda") * 0.02).to(torch.bfloat16) for r, c in [(2048, 4096), (512, 1024), (4096, 4096)]]
    want = [_oracle_packed(w, "int4_sym", 128) for w in ws]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(6)]
    results = []
    for rep in range(3):
        for st in streams:
            with torch.cuda.stream(st):
                results.append(wl.compress_weights_packed_batched(ws, cfg))
    torch.cuda.synchronize()
    for got in results:
        for ref, cw in zip(want, got):
            _assert_cw_equals_oracle(cw, ref, "streams")
    # graph: record one batched call, replay it five times into the same output buffers
    gstream = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(gstream):
        wl.compress_weights_packed_batched(ws, cfg)  # warm-up on the capture stream
    torch.cuda.synchronize()
    with torch.cuda.graph(graph, stream=gstream):
        captured = wl.compress_weights_packed_batched(ws, cfg)
    for rep in range(5):
        for cw in captured:
            cw.tensor.zero_()
            cw.scale.zero_()
        graph.replay()
        torch.cuda.synchronize()
        for ref, cw in zip(want, captured):
            _assert_cw_equals_oracle(cw, ref, f"graph replay {rep}")


@pytest.mark.parametrize("mode,gs", [("int4_sym", 128), ("int4_asym", 128), ("int4_sym", 64), ("int8_asym", -1),
                                     ("int8_sym", -1), ("int4_sym", -1)])
def test_batched_kernel_full_size_vs_oracle(mode, gs):
    """The multi-tensor kernels -- the ones bench.py times -- against the oracle on WHOLE BASELINE-sized tensors
    (Llama-3-8B gate_proj 14336 x 4096, q_proj 4096 x 4096, down_proj 4096 x 14336 and a ragged 1000 x 4096 whose
    last tile is partial) in one launch: every code byte, scale and zero point bit-exact."""
    from nncf_amd import weight_lowering as wl
    torch.manual_seed(4)
    ws = [(torch.randn(r, c, device="cuda") * 0.02).to(torch.bfloat16)
          for r, c in [(14336, 4096), (4096, 4096), (4096, 14336), (1000, 4096)]]
    ws[1][17, 256:384] = 0.0          # an all-zero group
    ws[1][18, :] = 0.5                # constant row
    got = wl.compress_weights_packed_batched(ws, _cfg(mode, gs))
    for w, cw in zip(ws, got):
        _assert_cw_equals_oracle(cw, _oracle_packed(w, mode, gs), f"{mode} g{gs} {tuple(w.shape)}")


@pytest.mark.parametrize("mode,gs", [("int4_sym", 128), ("int4_asym", 128), ("int4_asym", 32), ("int8_asym", -1),
                                     ("int8_sym", -1), ("int4_sym", -1)])
def test_exact_division_path_high_volume(mode, gs):
    """The lowering kernels take their codes from a correctly rounded quotient built with fma corrections
    (IntPolicyT::quant_n_div) instead of an IEEE divide.  8 M full-mantissa fp32 weights whose rows span magnitudes
    2^-50 ... 2^50 (every quotient then lands near a rounding boundary of its own), rows beyond the fast range
    (2^-70, 2^70: the IEEE-divide branch), negative-scale groups (largest element positive), denormal and constant
    rows: codes, scales and zero points bit-exact against the oracle."""
    from nncf_amd import weight_lowering as wl
    rng = np.random.default_rng(2024)
    rows, cols = 2048, 4096
    w = rng.standard_normal((rows, cols)).astype(np.float32)
    # exact ties and near-ties on a quarter of the rows: multiples of half the scale, nudged by one ulp
    half = (rng.integers(-16, 16, (rows // 4, cols)).astype(np.float32) + 0.5) * np.float32(0.37)
    half[:, 0] = np.float32(-8 * 0.37)
    half[:, 1::9] = np.nextafter(half[:, 1::9], np.float32(np.inf))
    half[:, 2::9] = np.nextafter(half[:, 2::9], np.float32(-np.inf))
    w[: rows // 4] = half
    mag = np.exp2(rng.integers(-50, 51, (rows, 1))).astype(np.float32)
    mag[5], mag[6], mag[7], mag[8] = 2.0 ** -70, 2.0 ** 70, 2.0 ** -100, 2.0 ** 100
    w = (w * mag).astype(np.float32)
    w[9] = np.abs(w[9])                    # all positive: negative scale in the symmetric modes
    w[10] = 1e-42                          # denormal constant row
    w[11] = -3.25                          # constant row
    w[12, ::2] = 0.0
    ws = [to_dev(w[:1024], "f32"), to_dev(w[1024:], "f32")]
    got = wl.compress_weights_packed_batched(ws, _cfg(mode, gs))
    for part, cw in zip((w[:1024], w[1024:]), got):
        rc, rs, rz = O.compress_weight(part, (1,), mode, gs)
        ref = O.pack_uint4(rc) if mode == "int4_asym" else O.pack_int4(rc) if mode == "int4_sym" else rc
        assert_equal_nan(host(cw.scale).reshape(rs.shape), rs, f"{mode} g{gs} scale")
        if rz is not None:
            assert_equal_nan(host(cw.zero_point).reshape(rz.shape), rz, f"{mode} g{gs} zp")
        assert_equal_nan(host(cw.tensor).reshape(ref.shape), ref, f"{mode} g{gs} codes")


@pytest.mark.parametrize("dtype", ["bf16", "f32", "f16"])
@pytest.mark.parametrize("mode", ["int8_asym", "int8_sym", "int4_sym", "int4_asym"])
def test_per_channel_row_kernel_all_classes(mode, dtype):
    """Per-channel (group_size = -1) batched lowering: every team/keep class of the row kernel (rows of 8 ... 3584
    16-byte vectors, wave- and block-owned), ragged row lengths inside a class, several classes in one call."""
    from nncf_amd import weight_lowering as wl
    rng = np.random.default_rng(11)
    cols = [64, 200, 512, 1000, 2048, 4096, 5000, 8192, 14336, 28672] if dtype != "f32" else [32, 104, 520, 2048, 4096, 7168, 14336]
    shapes = [(3 if c > 8000 else 9, c) for c in cols] + [(1, cols[2]), (70, cols[1])]
    ws32 = [representable((rng.standard_normal(s) * 0.05).astype(np.float32), dtype) for s in shapes]
    ws32[0][1, :] = 0.0                      # all-zero row
    ws32[3][2, :] = ws32[3][2, 0]            # constant row
    ws = [to_dev(w, dtype) for w in ws32]
    cfg = _cfg(mode, -1)
    got = wl.compress_weights_packed_batched(ws, cfg)
    for w32, cw in zip(ws32, got):
        rc, rs, rz = O.compress_weight(w32, (1,), mode, -1)
        assert_equal_nan(host(cw.scale).reshape(rs.shape), rs, f"scale {w32.shape}")
        if mode.startswith("int4"):
            ref = O.pack_uint4(rc) if mode == "int4_asym" else O.pack_int4(rc)
        else:
            ref = rc
        assert_equal_nan(host(cw.tensor).reshape(ref.shape), ref, f"codes {w32.shape}")
        if rz is not None:
            assert_equal_nan(host(cw.zero_point).reshape(rz.shape), rz, f"zp {w32.shape}")


@pytest.mark.parametrize("dtype", ["bf16", "f32"])
@pytest.mark.parametrize("reduction", ["max_mean", "frobenius"])
@pytest.mark.parametrize("mode,gs", [("int8_asym", -1), ("int4_sym", 64), ("int4_asym", -1), ("int4_asym", 32),
                                     ("int8_sym", 32), ("int4_sym", 16)])
def test_batched_error_metric_vs_oracle_and_single(mode, gs, reduction, dtype):
    """nncf_amd_int_quant_error_batched (one pair of launches for many weights, what the mixed-precision scoring
    uses) against the oracle's get_integer_quantization_error (rtol 1e-4: the reference's own tolerance for this
    reduction) and equal to the per-weight entry point up to the order of the fp32 sums (the batched path takes the
    tile kernel for grouped modes on 64-byte rows, the per-weight path a block per channel); > 48 tensors = several
    launches."""
    from nncf_amd import weight_lowering as wl
    rng = np.random.default_rng(5)
    shapes = [(7, 256), (64, 1024), (300, 512), (1, 64), (33, 4096)] * 11  # 55 tensors
    ws32 = [representable((rng.standard_normal(s) * (0.01 + 0.01 * (i % 5))).astype(np.float32), dtype)
            for i, s in enumerate(shapes)]
    ws = [to_dev(w, dtype) for w in ws32]
    cfg = _cfg(mode, gs)
    got = wl.get_integer_quantization_errors(ws, [(1,)] * len(ws), cfg, reduction)
    assert len(got) == len(ws)
    for i in range(0, len(ws), 4):
        np.testing.assert_allclose(got[i], wl.get_integer_quantization_error(ws[i], (1,), cfg, reduction), rtol=2e-6, atol=1e-12)
        ref = O.get_integer_quantization_error(ws32[i], (1,), mode, gs, reduction)
        np.testing.assert_allclose(got[i], ref, rtol=1e-4, atol=1e-12)
    from nncf_amd import errors
    with pytest.raises(errors.InvalidGroupSizeError):
        wl.get_integer_quantization_errors([torch.zeros(4, 30, device="cuda")], [(1,)], _cfg("int4_sym", 16), "max_mean")


@pytest.mark.parametrize("dtype", ["bf16", "f32"])
@pytest.mark.parametrize("mode", ["int8_asym", "int8_sym", "int4_asym"])
def test_per_channel_error_metric_long_and_odd_rows(mode, dtype):
    """Whole-row groups (the backup configuration the mixed-precision scoring measures): every team / keep class of
    the row kernel up to its longest row, a ragged last vector pair, and a launch that falls back to the generic
    kernel (a row that is not a multiple of 16 bytes) -- batched and per weight, against the oracle."""
    from nncf_amd import weight_lowering as wl
    rng = np.random.default_rng(11)
    for shapes in ([(5, 14336), (2, 8192), (3, 16384), (9, 2056), (4, 136)], [(6, 1000), (3, 30), (2, 4096)]):
        ws32 = [representable((rng.standard_normal(s) * 0.02 + (0.01 if i % 2 else 0.0)).astype(np.float32), dtype)
                for i, s in enumerate(shapes)]
        ws = [to_dev(w, dtype) for w in ws32]
        cfg = _cfg(mode, -1)
        for reduction in ("max_mean", "frobenius"):
            got = wl.get_integer_quantization_errors(ws, [(1,)] * len(ws), cfg, reduction)
            for i, w32 in enumerate(ws32):
                ref = O.get_integer_quantization_error(w32, (1,), mode, -1, reduction)
                np.testing.assert_allclose(got[i], ref, rtol=1e-4, atol=1e-12)
                one = wl.get_integer_quantization_error(ws[i], (1,), cfg, reduction)
                np.testing.assert_allclose(got[i], one, rtol=2e-6, atol=1e-12)


@pytest.mark.parametrize("mode,gs", [("int4_sym", 64), ("int4_asym", 32), ("int8_asym", -1), ("nf4", 32), ("mxfp4", 32)])
def test_batched_lowering_of_3d_moe_weights(mode, gs):
    """compress_weights_packed_batched with 3-D / MoE weights [experts, out, in] mixed with 2-D ones: codes, scales and
    zero points equal the oracle's compress_weight on the 3-D tensor reduced along its last axis."""
    from nncf_amd import weight_lowering as wl
    rng = np.random.default_rng(8)
    ws32 = [representable((rng.standard_normal(s) * 0.05).astype(np.float32), "bf16") for s in [(4, 24, 256), (16, 512), (2, 7, 128)]]
    got = wl.compress_weights_packed_batched([to_dev(w, "bf16") for w in ws32], _cfg(mode, gs))
    for w32, cw in zip(ws32, got):
        rc, rs, rz = O.compress_weight(w32, (w32.ndim - 1,), mode, gs)
        assert tuple(cw.scale.shape) == rs.shape
        assert_equal_nan(host(cw.scale), rs, "scale")
        if rz is not None:
            assert_equal_nan(host(cw.zero_point), rz, "zp")
        if mode.startswith("int8"):
            assert_equal_nan(host(cw.tensor), rc.reshape(tuple(cw.tensor.shape)), "codes")
        elif mode.startswith("int4"):
            ref = O.pack_uint4(rc) if mode == "int4_asym" else O.pack_int4(rc)
            assert_equal_nan(host(cw.tensor).ravel(), ref.ravel(), "packed")
        else:
            p = host(cw.tensor).ravel()
            codes = np.stack([p & 15, p >> 4], -1).ravel()
            lut = O.NF4_QUANTILES if mode == "nf4" else O.E2M1_VALUE_OF_CODE
            assert np.array_equal(lut[codes], rc.ravel())


@pytest.mark.parametrize("ident", ["neg", "pos", "neg-pos"])
def test_reference_int_range_literals(ident):
    """tests/openvino/native/quantization/test_weights_compression.py:1489-1507 through the HIP path (1-D weight)."""
    from tests.test_oracle_known_answers import _RANGE_DATA
    w = np.array(_RANGE_DATA[ident], np.float32)
    from nncf_amd import weight_lowering as WL
    cw = WL.do_integer_quantization(to_dev(w, "f32"), _cfg("int4_sym", -1), -1)
    assert np.allclose(np.abs(host(cw.tensor).astype(np.float32)), np.abs(w))


@pytest.mark.parametrize("mode,pre_scale,pre_zp,raises", __import__("tests.test_oracle_known_answers", fromlist=["x"]).PRECOMPUTED_CASES)
def test_reference_precomputed_parameter_cases(mode, pre_scale, pre_zp, raises):
    """:1646-1688 through the HIP path: parameters come back verbatim, half a pair raises ValueError, and the codes
    equal the oracle's wherever the quotient is a number (one precomputed scale is 0 under a zero weight: 0 / 0, whose
    integer cast is undefined in the reference too, numpy warns about it)."""
    from tests.test_oracle_known_answers import precomputed_inputs
    from nncf_amd import weight_lowering as WL
    weight, scale, zp = precomputed_inputs(pre_scale, pre_zp)
    d = lambda a, k: None if a is None else torch.from_numpy(a).cuda()  # noqa: E731
    if raises:
        with pytest.raises(ValueError):
            WL.do_integer_quantization(to_dev(weight, "f32"), _cfg(mode, -1), -1, d(scale, "f32"), d(zp, "i32"))
        return
    cw = WL.do_integer_quantization(to_dev(weight, "f32"), _cfg(mode, -1), -1, d(scale, "f32"), d(zp, "i32"))
    ref_q, ref_s, ref_z = O.do_integer_quantization(weight, mode, -1, (-1,), scale, zp)
    if pre_scale:
        assert np.allclose(host(cw.scale), scale)
    if mode.endswith("asym"):
        if pre_zp:
            assert np.allclose(host(cw.zero_point), zp)
    else:
        assert cw.zero_point is None
    defined = np.ones(weight.shape, bool) if scale is None else scale != 0
    assert np.array_equal(host(cw.tensor)[defined], ref_q[defined])
