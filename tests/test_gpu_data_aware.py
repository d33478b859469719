"""GPU parity of the data-aware rows: scale estimation, AWQ, GPTQ Hessian and GPTQ column sweep vs the golden
fixtures (reference outputs) and the oracle.  Tolerances: scales / adjusted weights <= 1e-5 relative where the
arithmetic is order-stable; GPTQ after the Cholesky trio is compared like the reference's own test does
(np.isclose on most entries, tests/openvino/native/quantization/test_gptq.py:421-452) because it is chaotic
under last-bit differences of LAPACK."""
import numpy as np
import pytest
import torch

from oracle import wc_oracle as O
from tests.gpu_utils import host, to_dev

pytestmark = pytest.mark.gpu


def _cfg(mode, gs):
    from nncf_amd.parameters import WeightCompressionConfig
    return WeightCompressionConfig(mode=mode, group_size=gs)


def _stats(g):
    from nncf_amd.statistics import WCTensorStatistic
    means = [torch.from_numpy(m).cuda() for m in g["means"]]
    shapes = [tuple(int(v) for v in s) for s in g["shapes"]]
    return WCTensorStatistic(means, shapes)


@pytest.mark.parametrize("gs", [32, -1])
@pytest.mark.parametrize("mode", ["int4_sym", "int4_asym", "nf4"])
def test_golden_scale_estimation(golden, mode, gs):
    from nncf_amd import scale_estimation as SE
    g = golden("se_awq")
    s, z = SE.calculate_quantization_params(_stats(g), to_dev(g["w"], "f32"), (1,), _cfg(mode, gs), -1, 8, 3, 4)
    ref = g[f"se_{mode}_g{gs}_scale"]
    assert tuple(s.shape) == ref.shape
    np.testing.assert_allclose(host(s), ref, rtol=1e-5, atol=1e-9)
    if z is not None:
        assert np.array_equal(host(z), g[f"se_{mode}_g{gs}_zp"])


@pytest.mark.parametrize("mode", ["int4_sym", "int4_asym"])
def test_golden_awq(golden, mode):
    from nncf_amd import awq
    g = golden("se_awq")
    w = to_dev(g["w"], "f32")
    sc = awq.data_aware_step(w, (1,), _cfg(mode, 32), _stats(g), -1, 8, 0.02, 0.0, 1.0, 20)
    np.testing.assert_allclose(host(sc), g[f"awq_{mode}_scale"], rtol=1e-5)
    np.testing.assert_allclose(host(awq.data_free_step(w, 0)), g["awq_data_free_scale"], rtol=1e-6)


def test_golden_hessian(golden):
    from nncf_amd import gptq
    g = golden("gptq")
    xs = [torch.from_numpy(x).cuda() for x in g["xs"]]
    H = gptq.calculate_hessian(xs)
    assert tuple(H.shape) == g["H_torch"].shape
    # entries are sums with cancellation: the error scales with the diagonal (fp32 eps x |H|max), not with the entry
    np.testing.assert_allclose(host(H), g["H_torch"], rtol=1e-5, atol=1e-6 * np.abs(g["H_torch"]).max())
    assert np.array_equal(host(H)[0], host(H)[0].T)  # written symmetric by construction


@pytest.mark.parametrize("dtype", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("T,K", [(300, 256), (37, 100), (1000, 384), (64, 4096), (700, 2312), (33, 2048), (4500, 2560)])
def test_hessian_vs_float64(T, K, dtype):
    from nncf_amd import gptq
    rng = np.random.default_rng(T + K)
    xs32 = [rng.standard_normal((1, T, K)).astype(np.float32) for _ in range(3)]
    xs = [to_dev(x, dtype) for x in xs32]
    xr = [host(x).astype(np.float64) for x in xs]
    H = host(gptq.calculate_hessian(xs))[0]
    ref = sum(x[0].T @ x[0] for x in xr) * (2.0 / 3.0)
    np.testing.assert_allclose(H, ref, rtol=1e-5, atol=1e-5 * np.abs(ref).max())
    # moe / 3-D weight form keeps the expert axis (gptq.py:216-223)
    x3 = [to_dev(rng.standard_normal((2, 16, 64)).astype(np.float32), "f32") for _ in range(2)]
    H3 = host(gptq.calculate_hessian(x3, is_3d_weight=True))
    ref3 = O.gptq_hessian([host(x) for x in x3], is_3d_weight=True)
    # off-diagonal entries are sums with cancellation: the error scales with the diagonal, not with the entry
    np.testing.assert_allclose(H3, ref3, rtol=1e-5, atol=1e-6 * np.abs(ref3).max())


@pytest.mark.parametrize("se", [0, 1])
@pytest.mark.parametrize("gs", [16, 64, -1])
@pytest.mark.parametrize("mode", ["int4_sym", "int4_asym", "nf4"])
def test_golden_gptq_quantize(golden, mode, gs, se):
    if se and mode == "nf4":
        pytest.skip("reference runs scale estimation inside GPTQ for integer modes only")
    from nncf_amd import gptq
    g = golden("gptq")
    H = torch.from_numpy(g["H_torch"][0].copy()).cuda()
    W = torch.from_numpy(g["w"].copy()).cuda()
    xs = [torch.from_numpy(x).cuda() for x in g["xs"]]
    q, s, z = gptq.quantize_weights(H, W, _cfg(mode, gs), xs, 0.1, 32, bool(se))
    key = f"torch_{mode}_g{gs}_se{se}"
    assert tuple(s.shape) == g[key + "_scale"].shape
    close = np.isclose(host(q), g[key + "_q"], rtol=1e-4, atol=1e-6)
    assert close.mean() > 0.97, close.mean()
    sclose = np.isclose(host(s), g[key + "_scale"], rtol=1e-3, atol=1e-6)
    assert sclose.mean() > 0.97, sclose.mean()
    if z is not None:
        assert (host(z) == g[key + "_zp"]).mean() > 0.97


@pytest.mark.parametrize("mode,gs", [("int4_sym", 128), ("int4_asym", 64), ("nf4", 32), ("mxfp4", 32), ("int4_sym", -1)])
def test_gptq_sweep_matches_oracle_given_same_hinv(mode, gs):
    """The HIP column sweep against the oracle's Python loop with the SAME inverse-Hessian factor (computed once, in
    float64 on the host), so that the only difference left is the sweep itself: codes must match exactly."""
    from nncf_amd import _lib
    rng = np.random.default_rng(5)
    N, K, bs = 48, 256, 128
    W = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
    X = rng.standard_normal((512, K))
    Hd = 2.0 / 512 * X.T @ X
    Hd[np.diag_indices(K)] += 0.1 * np.mean(np.diag(Hd))
    hinv = np.linalg.cholesky(np.linalg.inv(Hd)).T.astype(np.float32).copy()
    # oracle sweep
    Wo, Q_ref = W.copy(), np.zeros_like(W)
    g = gs if gs != -1 else K
    scales_ref = []
    for i1 in range(0, K, bs):
        i2 = min(i1 + bs, K)
        Wb, Eb = Wo[:, i1:i2].copy(), np.zeros((N, i2 - i1), np.float32)
        for i in range(i2 - i1):
            c = i1 + i
            if c % g == 0:
                wg = Wo[:, c:c + g]
                if O.is_integer(mode):
                    sc, zp = O.calculate_integer_quantization_params(wg, (1,), mode)
                else:
                    sc, zp = O.calculate_float_quantization_params(wg, (1,), mode), None
                scales_ref.append(sc)
            col = Wb[:, i][:, None]
            qc = (O.integer_quantize_dequantize_weight(col, mode, -1, None, sc, zp) if O.is_integer(mode)
                  else O.float_quantize_dequantize_weight(col, mode, -1, None, sc)).reshape(-1)
            Q_ref[:, c] = qc
            err = (Wb[:, i] - qc) / hinv[c, c]
            Wb[:, i:] -= err[:, None] * hinv[c, c:i2][None, :]
            Eb[:, i] = err
        Wo[:, i2:] -= Eb @ hinv[i1:i2, i2:]
    # HIP sweep through the C ABI, trailing GEMM in float32 numpy-equivalent (torch.matmul) -> compare per block
    Wd = torch.from_numpy(W.copy()).cuda()
    hd = torch.from_numpy(hinv).cuda()
    Qd = torch.zeros_like(Wd)
    ng = K // g
    sc_d = torch.empty((N, ng), dtype=torch.float32, device="cuda")
    zp_d = torch.empty((N, ng), dtype=torch.int32, device="cuda")
    L = _lib.lib()
    for i1 in range(0, K, bs):
        i2 = min(i1 + bs, K)
        err = torch.empty((N, i2 - i1), dtype=torch.float32, device="cuda")
        st = L.nncf_amd_gptq_block(_lib.ptr(Wd), _lib.ptr(Qd), _lib.ptr(err), _lib.ptr(hd), N, K, i1, i2 - i1, g,
                                   _lib.MODE_ID[mode], _lib.ptr(sc_d), _lib.ptr(zp_d), 0, _lib.stream_of(Wd))
        _lib.check(st)
        if i2 < K:
            Wd[:, i2:] -= err @ hd[i1:i2, i2:]
    ref_scales = np.concatenate(scales_ref, axis=1)
    np.testing.assert_allclose(host(sc_d), ref_scales, rtol=2e-5)
    close = np.isclose(host(Qd), Q_ref, rtol=1e-4, atol=1e-7)
    assert close.mean() > 0.995, close.mean()
    # first block: no GEMM involved yet -> bit exact
    assert np.array_equal(host(Qd)[:, :bs], Q_ref[:, :bs])


def _same_hinv_case(N, K, seed):
    """Weight + the upper factor U of gptq.py:296 computed ONCE in fp64 on the device and rounded to fp32: handed
    to both the oracle's column loop and gptq.quantize_weights, the LAPACK step drops out of the comparison."""
    g = torch.Generator(device="cuda").manual_seed(seed)
    W = torch.randn(N, K, device="cuda", generator=g) * 0.05
    X = torch.randn(2 * K, K, device="cuda", generator=g, dtype=torch.float64)
    X[:, ::97] *= 6.0  # outlier channels
    H = X.T @ X * (2.0 / (2 * K))
    H += 0.1 * torch.mean(torch.diag(H)) * torch.eye(K, device="cuda", dtype=torch.float64)
    U = torch.linalg.cholesky(torch.cholesky_inverse(torch.linalg.cholesky(H)), upper=True).float().contiguous()
    return W, U


# (mode, group size, K, block size): >= 2 super-blocks of 8 blocks with a ragged tail (the far `errs` GEMM of
# nncf_amd/gptq.py, taken by every BASELINE shape), group sizes that straddle a super-block end (192, 384, 768 with
# block 128; 128 with block 100), a group size beyond the super-block (1536), per-channel (-1)
_TWO_LEVEL = [("int4_sym", 64, 1152, 128), ("int4_asym", 128, 1152, 128), ("nf4", 32, 1152, 128),
              ("int4_sym", 64, 2304, 128), ("int4_asym", 128, 2304, 128), ("nf4", 32, 2304, 128),
              ("int4_sym", 64, 4096, 128), ("int4_asym", 128, 4096, 128), ("nf4", 32, 4096, 128),
              ("mxfp4", 32, 2304, 128), ("int8_asym", 128, 1152, 128),
              ("int4_sym", 192, 2304, 128), ("int4_asym", 384, 2304, 128), ("nf4", 768, 2304, 128),
              ("int4_sym", 1536, 4608, 128), ("int4_asym", 128, 2304, 100), ("int4_sym", 64, 1600, 50),
              ("int4_sym", -1, 1152, 128)]


@pytest.mark.parametrize("mode,gs,K,bs", _TWO_LEVEL)
def test_gptq_quantize_weights_two_level_same_hinv(mode, gs, K, bs):
    """gptq.quantize_weights (two-level trailing update) against the oracle's single-level column loop
    (gptq.py:298-363) given the SAME fp32 Hinv factor.  Bar: first block bit-exact (no GEMM before it), all blocks
    >= 99.5 % isclose (the far GEMM sums the same products in another order), scales <= 2e-5, zero points equal on
    >= 99.5 %."""
    from nncf_amd import gptq
    N = 64
    W, U = _same_hinv_case(N, K, seed=K + bs)
    q_ref, s_ref, z_ref = O.gptq_quantize_weights(None, host(W).copy(), mode, gs, block_size=bs, hinv=host(U))
    q, s, z = gptq.quantize_weights(None, W.clone(), _cfg(mode, gs), None, 0.1, bs, False, hinv=U,
                                    dead=torch.zeros(K, dtype=torch.bool, device="cuda"))
    assert tuple(s.shape) == s_ref.shape
    assert np.array_equal(host(q)[:, :bs], q_ref[:, :bs])
    close = np.isclose(host(q), q_ref, rtol=1e-4, atol=1e-7)
    assert close.mean() >= 0.995, close.mean()
    sclose = np.isclose(host(s), s_ref, rtol=2e-5, atol=0)
    assert sclose.mean() >= 0.995, sclose.mean()
    if z_ref is None:
        assert z is None
    else:
        assert tuple(z.shape) == z_ref.shape and (host(z) == z_ref).mean() >= 0.995


def test_gptq_straddling_group_sees_pending_errors():
    """Regression for the group-straddles-super-block case: with gs = 192 the group starting at column 960 runs to
    1152, past the super-block end at 1024; its scale must come from columns that already carry the errors of blocks
    0..6 (gptq.py:312-333).  The scales of exactly those groups are compared with the oracle at 2e-5."""
    from nncf_amd import gptq
    N, K, gs = 32, 2304, 192
    W, U = _same_hinv_case(N, K, seed=11)
    _, s_ref, _ = O.gptq_quantize_weights(None, host(W).copy(), "int4_sym", gs, hinv=host(U))
    _, s, _ = gptq.quantize_weights(None, W.clone(), _cfg("int4_sym", gs), None, hinv=U,
                                    dead=torch.zeros(K, dtype=torch.bool, device="cuda"))
    straddlers = [c // gs for c in range(0, K, gs) if (c % 1024) and (c // 1024 != (c + gs - 1) // 1024)]
    assert straddlers == [5, 10]
    np.testing.assert_allclose(host(s)[:, straddlers], s_ref[:, straddlers], rtol=2e-5)
    np.testing.assert_allclose(host(s), s_ref, rtol=2e-5)


@pytest.mark.parametrize("mode,gs", [("int4_sym", 64), ("int4_asym", 32), ("nf4", 64), ("mxfp4", 32)])
def test_gptq_sweep_many_rows_four_lanes_per_row(mode, gs):
    """From 16384 rows on the launcher takes the 4-lanes-per-row form of the sweep kernel (32 columns per lane); same
    check as for the 16-lane form: gptq.quantize_weights against the oracle's column loop with the SAME factor U on a
    slice of the rows (rows are independent), first block bit-exact, the ragged last workgroup included."""
    from nncf_amd import gptq
    N, K = 16384 + 37, 320
    W, U = _same_hinv_case(N, K, seed=11)
    Q, sc, zp = gptq.quantize_weights(None, W.clone(), _cfg(mode, gs), hinv=U,
                                      dead=torch.zeros(K, dtype=torch.bool, device="cuda"))
    for rows in (slice(0, 24), slice(N - 50, N)):
        q_ref, s_ref, z_ref = O.gptq_quantize_weights(None, host(W[rows]).copy(), mode, gs, hinv=host(U))
        assert np.array_equal(host(Q[rows])[:, :128], q_ref[:, :128])
        assert np.isclose(host(Q[rows]), q_ref, rtol=1e-4, atol=1e-7).mean() >= 0.995
        assert np.isclose(host(sc[rows]), s_ref, rtol=2e-5, atol=0).mean() >= 0.995
        if z_ref is not None:
            assert (host(zp[rows]) == z_ref).mean() >= 0.995


def test_quantize_units_stream_modes_agree():
    """gptq.quantize_units: the units of a layer on one stream, with every unit's Hinv + sweeps on a stream of its own
    (all Hessians first), and with the next Hessian on a second stream -- the same kernels in the same order within a
    unit; a Hessian that is not positive definite raises the reference's error
    in the asynchronous mode too (its flag is read once, at the end)."""
    from nncf_amd import gptq
    g = torch.Generator(device="cuda").manual_seed(4)
    cfg = _cfg("int4_asym", 32)

    def make_units():
        units = []
        for u, (k, outs) in enumerate(((256, (40, 24)), (384, (64,)), (1280, (48, 16, 16)))):
            xs = [torch.randn(1, 300 + 20 * i, k, device="cuda", generator=g).to(torch.bfloat16) for i in range(4)]
            ws = {f"u{u}.w{j}": (torch.randn(n, k, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
                  for j, n in enumerate(outs)}
            units.append((xs, ws))
        return units

    units = make_units()
    ref = gptq.quantize_units(units, cfg, overlap="none")
    assert sorted(ref) == sorted(n for _, ws in units for n in ws)
    for mode in ("units", "hessian"):
        got = gptq.quantize_units(units, cfg, overlap=mode)
        torch.cuda.synchronize()
        assert sorted(got) == sorted(ref)
        # (the SYRK adds its token splits with fp32 atomics, so a Hessian is reproducible to rounding only and GPTQ
        # turns a last-bit difference into a few different codes: the reference's own bar for two GPTQ runs,
        # tests/openvino/native/quantization/test_gptq.py:421-452)
        for name in ref:
            for a, b in zip(got[name], ref[name]):
                assert a.shape == b.shape and a.dtype == b.dtype
                assert float(torch.isclose(a.float(), b.float(), rtol=1e-4, atol=1e-7).float().mean()) >= 0.97, (mode, name)
    packed = gptq.quantize_units(units, cfg, overlap="units", on_weight=lambda name, w, q, sc, zp: (q != 0).sum())
    assert all(int(v) > 0 for v in packed.values())
    # weights are not modified, and the single-unit API gives the same result as the pipeline
    xs, ws = units[1]
    before = ws["u1.w0"].clone()
    q, sc, zp = gptq.quantize_node_weight(ws["u1.w0"], xs, cfg)
    assert torch.equal(ws["u1.w0"], before)
    assert float(torch.isclose(q, ref["u1.w0"][0], rtol=1e-4, atol=1e-7).float().mean()) >= 0.97
    assert tuple(sc.shape) == tuple(ref["u1.w0"][1].shape) and tuple(zp.shape) == tuple(ref["u1.w0"][2].shape)
    # dead columns (all-zero channels) go through the sync-free dead-column fix
    xd = [torch.randn(1, 200, 128, device="cuda", generator=g) for _ in range(2)]
    for x in xd:
        x[..., 5] = 0.0
    wd = torch.randn(8, 128, device="cuda", generator=g) * 0.05
    qd = gptq.quantize_units([(xd, {"d": wd})], cfg, overlap="units")["d"][0]
    assert bool((qd[:, 5] == 0).all()) and bool(torch.isfinite(qd).all())
    # a Hessian the factorization cannot take (NaN activations): the reference's error, also from the asynchronous mode
    xn = [torch.full((1, 64, 128), float("nan"), device="cuda")]
    for mode in ("none", "units"):
        with pytest.raises(torch.linalg.LinAlgError):
            gptq.quantize_units([(units[0][0], units[0][1]), (xn, {"n": wd})], cfg, overlap=mode)
    with pytest.raises(ValueError):
        gptq.quantize_units(units, cfg, overlap="all")


def _se_case(shape, n_samples, seed, device="cuda"):
    """Weight + per-sample token means with outlier channels (what WCTensorStatistic carries)."""
    from nncf_amd.statistics import WCTensorStatistic
    g = torch.Generator(device=device).manual_seed(seed)
    w = torch.randn(*shape, device=device, generator=g) * 0.05
    K = shape[-1]
    lead = tuple(shape[:-2])  # expert axis of a 3-D weight
    means = []
    for _ in range(n_samples):
        m = torch.randn(*lead, K, device=device, generator=g)
        m[..., ::61] *= 12.0
        means.append(m)
    shapes = [lead + (7 + 3 * i, K) for i in range(n_samples)]
    return w, WCTensorStatistic(means, shapes)


@pytest.mark.parametrize("shape,gs,n_samples,subset", [((96, 256), 32, 8, 8), ((200, 512), 64, 40, 32), ((130, 384), 128, 70, 64),
                                                        ((64, 128), -1, 5, 32), ((3, 70, 256), 64, 33, 64)])
@pytest.mark.parametrize("mode", ["int4_sym", "int4_asym", "nf4", "mxfp4", "int8_asym"])
def test_fused_se_search_matches_oracle(mode, shape, gs, n_samples, subset):
    """The fused search (nncf_amd_se_step: one launch per candidate scale) against the oracle's restatement of
    scale_estimation.py:160-377 on the same statistics: scales <= 1e-5 on >= 99.5 % of the groups (a candidate whose
    output error ties with the best one within fp32 summation noise may go either way; the reference's numpy and
    torch backends differ from each other in the same way), zero points equal.  Row counts that are not multiples of
    the 32-row MFMA tile, 1 and 2 sample tiles, a 3-D (MoE) weight."""
    from nncf_amd import scale_estimation as SE
    w, stats = _se_case(shape, n_samples, seed=len(shape) * 100 + n_samples)
    axes = (len(shape) - 1,)
    sc, zp = SE.calculate_quantization_params(stats, w, axes, _cfg(mode, gs), -1, subset, 4, 5)
    ref_s, ref_z = O.scale_estimation_params([host(m) for m in stats.mean_values], stats.shape_values, host(w), axes,
                                             mode, gs, -1, subset, 4, 5)
    assert tuple(sc.shape) == ref_s.shape
    close = np.isclose(host(sc), ref_s, rtol=1e-5, atol=0)
    assert close.mean() >= 0.995, close.mean()
    if ref_z is None:
        assert zp is None
    else:
        assert np.array_equal(host(zp), ref_z)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("mode,gs", [("int4_asym", 128), ("int4_sym", 32), ("nf4", 64)])
def test_fused_se_search_16bit_weights(mode, gs, dtype):
    """The fused search reads a bf16 weight as it is (no fp32 copy; fp16 is upcast by the host mirror): same result
    as the oracle on the upcast weight (scale_estimation.py:197 casts to fp32 first)."""
    from nncf_amd import scale_estimation as SE
    w, stats = _se_case((260, 512), 48, seed=17)
    w = w.to(dtype)
    sc, zp = SE.calculate_quantization_params(stats, w, (1,), _cfg(mode, gs), -1, 64, 5, 5)
    ref_s, ref_z = O.scale_estimation_params([host(m) for m in stats.mean_values], stats.shape_values, host(w.float()), (1,),
                                             mode, gs, -1, 64, 5, 5)
    close = np.isclose(host(sc), ref_s, rtol=1e-5, atol=0)
    assert close.mean() >= 0.995, close.mean()
    if ref_z is not None:
        assert np.array_equal(host(zp), ref_z)


@pytest.mark.parametrize("shape,gs,n_samples,subset", [((70, 240), 30, 12, 32), ((33, 192), 96, 70, 128), ((50, 200), -1, 9, 32),
                                                        ((19, 1000), 250, 130, 256), ((40, 96), 2, 5, 32), ((2, 21, 150), 50, 300, 512),
                                                        ((45, 77), 7, 3, 32)])
@pytest.mark.parametrize("mode", ["int4_sym", "int4_asym", "nf4", "mxfp4", "int8_asym"])
def test_se_search_any_group_size_matches_oracle(mode, shape, gs, n_samples, subset):
    """Group sizes other than 32 / 64 / 128 (odd ones and whole rows included) and more than 64 samples go through the
    same entry point (nncf_amd_se_step -> k_se_step_any); same bar against the oracle as the tiled kernel."""
    from nncf_amd import scale_estimation as SE
    w, stats = _se_case(shape, n_samples, seed=len(shape) * 1000 + n_samples + abs(gs))
    axes = (len(shape) - 1,)
    sc, zp = SE.calculate_quantization_params(stats, w, axes, _cfg(mode, gs), -1, subset, 3, 4)
    ref_s, ref_z = O.scale_estimation_params([host(m) for m in stats.mean_values], stats.shape_values, host(w), axes,
                                             mode, gs, -1, subset, 3, 4)
    assert tuple(sc.shape) == ref_s.shape
    close = np.isclose(host(sc), ref_s, rtol=1e-5, atol=0)
    # groups of 2 or 7 elements: candidates often quantize to the SAME codes, their output errors then differ only by
    # the order of summation and the reference's `diff > best` goes either way
    assert close.mean() >= (0.95 if 0 < gs < 16 else 0.99), close.mean()
    if ref_z is None:
        assert zp is None
    else:
        assert np.array_equal(host(zp), ref_z)


@pytest.mark.parametrize("penalty", [-1.0, 0.3])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("mode,gs", [("int4_asym", 128), ("int4_sym", 64), ("nf4", 32), ("mxfp4", 32)])
def test_se_tiled_kernel_equals_any_kernel(mode, gs, dtype, penalty):
    """The two kernel families behind nncf_amd_se_step on the SAME problem, including the weight-penalty term: a
    weight whose storage is not 16-byte aligned takes k_se_step_any, its aligned copy the tiled MFMA kernel.  They
    sum the output error in different orders, hence 1e-5 / 99.8 %."""
    from nncf_amd import scale_estimation as SE
    w, stats = _se_case((1000, 1024), 64, seed=3)
    w = w.to(dtype)
    buf = torch.empty(w.numel() + 8, dtype=dtype, device="cuda")
    shifted = buf[1:1 + w.numel()].view(w.shape)
    shifted.copy_(w)
    assert shifted.is_contiguous() and shifted.data_ptr() % 16 != 0 and w.data_ptr() % 16 == 0
    cfg = _cfg(mode, gs)
    a_s, a_z = SE.calculate_quantization_params(stats, w, (1,), cfg, -1, 64, 5, 5, penalty)
    b_s, b_z = SE.calculate_quantization_params(stats, shifted, (1,), cfg, -1, 64, 5, 5, penalty)
    assert a_s.shape == b_s.shape
    close = torch.isclose(a_s, b_s, rtol=1e-5, atol=0).float().mean()
    assert float(close) >= 0.998, float(close)
    if b_z is not None:
        assert torch.equal(a_z, b_z)


def _stat_from(means, shapes):
    from nncf_amd.statistics import WCTensorStatistic
    return WCTensorStatistic([torch.from_numpy(np.ascontiguousarray(m)).cuda() for m in means],
                             [tuple(int(v) for v in s) for s in shapes])


@pytest.mark.parametrize("mode", ["int4_sym", "int4_asym"])
def test_golden_awq_clamp_branch(golden, mode):
    """AWQ data-aware step with the previous matmul merged: the fp16-overflow clamp (awq.py:315-335, _clamp_scale
    :360) against what the reference returned (tools/gen_golden_extra.py)."""
    from nncf_amd import awq
    g = golden("se_awq_extra")
    sc = awq.data_aware_step(to_dev(g["clamp_w"], "f32"), (1,), _cfg(mode, 32), _stat_from(g["clamp_means"], g["clamp_shapes"]),
                             -1, 8, 0.02, 0.0, 1.0, 20, prev_weight=to_dev(g["clamp_prev_w"], "f32"),
                             prev_statistics=_stat_from(g["clamp_prev_means"], g["clamp_prev_shapes"]))
    np.testing.assert_allclose(host(sc), g[f"clamp_awq_{mode}_scale"], rtol=1e-5)


@pytest.mark.parametrize("mode", ["int4_sym", "int4_asym"])
def test_golden_awq_3d_weights(golden, mode):
    """MoE weight [experts, hidden, out] reduced along axis 1 (awq.py:255-289)."""
    from nncf_amd import awq
    g = golden("se_awq_extra")
    w_ekn = np.ascontiguousarray(np.moveaxis(g["moe_w_enk"], -1, -2))
    sc = awq.data_aware_step(to_dev(w_ekn, "f32"), (1,), _cfg(mode, 32), _stat_from(g["moe_means"], g["moe_shapes"]),
                             -1, 8, 0.05, 0.0, 1.0, 20)
    assert tuple(sc.shape) == g[f"moe_awq_{mode}_scale"].shape
    np.testing.assert_allclose(host(sc), g[f"moe_awq_{mode}_scale"], rtol=1e-5)


@pytest.mark.parametrize("axis", [2, 1])
@pytest.mark.parametrize("gs", [32, -1])
@pytest.mark.parametrize("mode", ["int4_sym", "int4_asym", "nf4"])
def test_golden_scale_estimation_3d_weights(golden, mode, gs, axis):
    """MoE weights through scale estimation, reduction along the last axis ([E, N, K]) and along axis 1 ([E, K, N],
    the transposed branch, scale_estimation.py:206-212) against the reference's outputs."""
    from nncf_amd import scale_estimation as SE
    g = golden("se_awq_extra")
    w = g["moe_w_enk"] if axis == 2 else np.ascontiguousarray(np.moveaxis(g["moe_w_enk"], -1, -2))
    s, z = SE.calculate_quantization_params(_stat_from(g["moe_means"], g["moe_shapes"]), to_dev(w, "f32"), (axis,),
                                            _cfg(mode, gs), -1, 8, 3, 4)
    key = f"moe_se_{mode}_g{gs}_ax{axis}"
    assert tuple(s.shape) == g[key + "_scale"].shape
    np.testing.assert_allclose(host(s), g[key + "_scale"], rtol=1e-5, atol=1e-9)
    if z is not None:
        assert np.array_equal(host(z), g[key + "_zp"])


from tests.se_literals import se_literal_inputs, se_literal_params  # noqa: E402


@pytest.mark.parametrize("mode,init,steps,transpose_a,n_samples,ref_scale,ref_zp", se_literal_params())
def test_scale_estimation_reference_literals(mode, init, steps, transpose_a, n_samples, ref_scale, ref_zp):
    """The literal scales / zero points of the reference's own template test
    (tests/cross_fw/test_templates/template_test_weights_compression.py:409-623), its tolerance (atol 1e-3)."""
    from nncf_amd import scale_estimation as SE
    from nncf_amd.statistics import activations_to_wc_statistics
    w, acts = se_literal_inputs(n_samples)
    stats = activations_to_wc_statistics([torch.from_numpy(a).cuda() for a in acts])
    axes, ch = ((0,), 1) if transpose_a else ((1,), 2)
    sc, zp = SE.calculate_quantization_params(stats, torch.from_numpy(w).cuda(), axes, _cfg(mode, 8), ch, 2, init, steps)
    assert sc.dtype == torch.float32
    np.testing.assert_allclose(host(sc), np.array(ref_scale, np.float32).reshape(tuple(sc.shape)), atol=1e-3, rtol=0)
    if ref_zp is None:
        assert zp is None
    else:
        np.testing.assert_allclose(host(zp), np.array(ref_zp, np.float32).reshape(tuple(zp.shape)), atol=1e-3, rtol=0)


def test_gptq_against_the_independent_autogptq_reference(golden):
    """HIP Hessian (2-D per expert and the 3-D / MoE form) + gptq.quantize_weights against the AutoGPTQ-derived torch
    reference the reference's own GPTQ test compares with (tests/openvino/native/quantization/test_gptq.py:168-325,
    :355-452): INT4_SYM, group 16, damp 0.1, seeded inputs of that test.  Its bar is np.isclose on everything; after
    the Cholesky chain single entries may flip under last-bit differences, so the weights take >= 99 %."""
    from nncf_amd import gptq
    from tests.test_oracle_golden import autogptq_inputs
    g = golden("gptq_autogptq")
    inputs, weights = autogptq_inputs()
    H3 = gptq.calculate_hessian([torch.from_numpy(x).cuda() for x in inputs], is_3d_weight=True)
    for e in range(4):
        H = gptq.calculate_hessian([torch.from_numpy(x[e][None]).cuda() for x in inputs])[0]
        assert np.all(np.isclose(g[f"H_{e}"], host(H), rtol=1e-5, atol=1e-5))
        assert np.all(np.isclose(g[f"H_{e}"], host(H3[e]), rtol=1e-5, atol=1e-5))
        q, s, _ = gptq.quantize_weights(H.clone(), torch.from_numpy(weights[e].copy()).cuda(), _cfg("int4_sym", 16))
        assert np.all(np.isclose(g[f"scale_{e}"].reshape(tuple(s.shape)), host(s)))
        assert np.isclose(g[f"weight_{e}"], host(q)).mean() >= 0.99


@pytest.mark.parametrize("merged", [False, True])
@pytest.mark.parametrize("mode,gs,n_samples,subset", [("int4_sym", 128, 32, 32), ("int4_asym", 64, 40, 64), ("nf4", 32, 9, 32),
                                                       ("int8_asym", 128, 64, 64), ("int4_asym", 48, 20, 32), ("int4_sym", 100, 70, 128),
                                                       ("nf4", -1, 12, 32), ("int4_sym", 64, 90, 128), ("mxfp4", 10, 5, 32),
                                                       ("mxfp4", 32, 24, 32), ("int4_sym", 32, 12, 32)])
def test_awq_search_matches_oracle(mode, gs, n_samples, subset, merged):
    """nncf_amd_awq_search (all groups x alpha steps in one launch; the tiled MFMA kernel for group sizes 32 / 64 / 128
    with <= 64 samples, k_awq_search_any for every other shape) against the oracle's restatement of awq.py:243-357,
    with and without the merged previous matmul (fp16-overflow clamp): the same winning scale on (nearly) every
    channel -- the sums of |cur_out - fp32_out| run in another order, so a step whose error ties with the best one
    within fp32 noise may win on one side only."""
    from nncf_amd import awq
    N, K, Kp = 150, {48: 576, 100: 600, 10: 600, -1: 600}.get(gs, 512), 256
    w, stats = _se_case((N, K), n_samples, seed=21 + n_samples)
    g = torch.Generator(device="cuda").manual_seed(1)
    prev = torch.randn(K, Kp, device="cuda", generator=g)
    _, pstats = _se_case((4, Kp), n_samples, seed=22)
    for m in pstats.mean_values:
        m *= 30.0
    kw = dict(prev_weight=prev, prev_statistics=pstats) if merged else {}
    a = awq.data_aware_step(w, (1,), _cfg(mode, gs), stats, -1, subset, 0.01, 0.0, 1.0, 25, **kw)
    okw = (host(prev), [host(m) for m in pstats.mean_values], pstats.shape_values) if merged else ()
    ref = O.awq_data_aware_scale(host(w), 1, mode, gs, [host(m) for m in stats.mean_values], stats.shape_values, -1, subset,
                                 0.01, 0.0, 1.0, 25, *okw)
    assert tuple(a.shape) == ref.shape and bool((a != 1).any())
    same = np.isclose(host(a), ref, rtol=1e-5, atol=0).mean()
    assert same >= 0.9, same


@pytest.mark.parametrize("n", [1, 2, 15, 16, 17, 63, 64, 65, 100, 127, 128])
def test_cholinv_block_kernel(n):
    """nncf_amd_cholinv_block (one workgroup, n <= 128) vs fp64 cholesky + triangular inverse: the block is a
    strided view inside a larger matrix, its strict upper triangle holds NaN (must not be read), the strict upper
    triangle and everything outside the block of the output must stay untouched."""
    from nncf_amd import _lib

    torch.manual_seed(n)
    x = torch.randn(3 * n + 8, n, device="cuda", dtype=torch.float64)
    A64 = x.T @ x / n + 0.05 * torch.eye(n, device="cuda", dtype=torch.float64)
    big = torch.full((n + 5, n + 9), float("nan"), device="cuda")
    blk = big[3:3 + n, 2:2 + n]
    blk.copy_(torch.tril(A64.float()) + torch.triu(torch.full((n, n), float("nan"), device="cuda"), 1))
    out = torch.full((n + 2, n + 3), 7.0, device="cuda")
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    _lib.check(_lib.lib().nncf_amd_cholinv_block(_lib.ptr(blk), big.stride(0), _lib.ptr(out[1:, 2:]), out.stride(0), n,
                                                 _lib.ptr(info), _lib.stream_of(big)))
    assert int(info.item()) == 0
    A_seen = torch.tril(A64.float()).double()
    A_seen = A_seen + torch.tril(A_seen, -1).T
    ref = torch.linalg.inv(torch.linalg.cholesky(A_seen))
    got = out[1:1 + n, 2:2 + n]
    upper = torch.triu(torch.ones(n, n, dtype=torch.bool, device="cuda"), 1)
    assert torch.all(got[upper] == 7.0) and torch.all(out[0] == 7.0) and torch.all(out[:, :2] == 7.0)
    err = (torch.tril(got).double() - ref).abs().max() / ref.abs().max()
    assert float(err) < 2e-5, float(err)


def test_cholinv_block_reports_non_positive_definite():
    from nncf_amd import _lib
    from nncf_amd import gptq as G

    n = 40
    A = torch.eye(n, device="cuda")
    A[17, 17] = -1.0
    out = torch.zeros(n, n, device="cuda")
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    _lib.check(_lib.lib().nncf_amd_cholinv_block(_lib.ptr(A), n, _lib.ptr(out), n, n, _lib.ptr(info), _lib.stream_of(A)))
    assert int(info.item()) == 18  # LAPACK convention: 1 + index of the failing pivot
    assert _lib.lib().nncf_amd_cholinv_block(_lib.ptr(A), n, _lib.ptr(out), n, 129, _lib.ptr(info), None) == _lib.ERR_BAD_ARG
    H = torch.eye(300, device="cuda")
    H[5, 5] = -2.0
    with pytest.raises(torch.linalg.LinAlgError):
        G._upper_cholesky_of_inverse(H)


@pytest.mark.parametrize("K", [1, 7, 64, 100, 128, 129, 200, 512, 1152, 2304])
def test_hinv_recursion_matches_reference_chain(K):
    """`_upper_cholesky_of_inverse` (GEMM-based recursive Cholesky-with-inverse) vs the reference's three-step chain
    cholesky -> cholesky_inverse -> cholesky(upper) (gptq.py:293-295) in fp64 on the same damped Hessian."""
    from nncf_amd import gptq as G

    torch.manual_seed(K)
    x = torch.randn(2 * K, K, device="cuda")
    H = x.T @ x / K
    H += 0.1 * torch.mean(torch.diag(H)) * torch.eye(K, device="cuda")
    U = G._upper_cholesky_of_inverse(H.clone())
    H64 = H.double()
    ref = torch.linalg.cholesky(torch.cholesky_inverse(torch.linalg.cholesky(H64)), upper=True)
    assert torch.equal(U, torch.triu(U))
    err = (U.double() - ref).abs().max() / ref.abs().max()
    assert float(err) < 1e-5, float(err)


def test_hinv_full_size_property():
    """At a BASELINE-sized K (Llama-3-8B hidden = 4096) the fp64 chain is too slow for a test; the size-independent
    property is used instead: U is upper triangular with a positive diagonal and U^T U H = I."""
    from nncf_amd import gptq as G

    K = 4096
    torch.manual_seed(3)
    x = torch.randn(2 * K, K, device="cuda")
    x[:, ::97] *= 8.0  # outlier channels, as in calibration activations
    H = x.T @ x * (2.0 / (2 * K))
    Hd = H.clone()
    U, dead = G.prepare_hinv(Hd, 0.1)
    assert not bool(dead.any()) and torch.equal(U, torch.triu(U)) and bool((torch.diag(U) > 0).all())
    R = (U.double().T @ U.double()) @ Hd.double() - torch.eye(K, device="cuda", dtype=torch.float64)
    assert float(R.abs().max()) < 2e-4, float(R.abs().max())


@pytest.mark.parametrize("K", [2304, 12296])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_hessian_item_list_kernel(dtype, K):
    """nncf_amd_hessian_accumulate_list: the MFMA kernels walk a list of separately allocated, ragged calibration
    items (token counts that are not multiples of their 32-token stage, a 1-token item, > 128 items = two launches)
    against the fp64 sum, through the C ABI and through HessianAccumulator (which no longer concatenates).
    K = 2304: k_syrk16b; K = 12296 (49 tiles across, the last one ragged): the list form of k_gemm16t."""
    import ctypes
    from nncf_amd import _lib, gptq as G
    torch.manual_seed(4)
    toks = [37, 1, 64, 100, 33] * 27  # 135 items
    xs = [torch.randn(t, K, device="cuda").to(dtype) for t in toks]
    ref = sum(x.double().T @ x.double() for x in xs)
    H = torch.zeros(K, K, device="cuda")
    n = len(xs)
    c_x = (ctypes.c_void_p * n)(*[x.data_ptr() for x in xs])
    c_t = (ctypes.c_int64 * n)(*toks)
    _lib.check(_lib.lib().nncf_amd_hessian_accumulate_list(_lib.ptr(H), n, c_x, c_t, _lib.dtype_id(xs[0]), K, 1.0,
                                                           _lib.stream_of(H)))
    _lib.check(_lib.lib().nncf_amd_hessian_finalize(_lib.ptr(H), K, 1.0, _lib.stream_of(H)))
    scale = float(ref.abs().max())
    torch.testing.assert_close(H.double(), ref, rtol=1e-5, atol=1e-6 * scale)
    assert torch.equal(H, H.T)
    acc = G.HessianAccumulator(K, xs[0].device)
    for x in xs:
        acc.add(x, 1)
    torch.testing.assert_close(acc.result().double(), ref * (2.0 / n), rtol=1e-5, atol=1e-6 * scale * 2 / n)


def test_hessian_chunking_equals_item_by_item():
    """HessianAccumulator: one SYRK launch per ~32k buffered tokens == the item-by-item running average."""
    from nncf_amd import gptq as G

    torch.manual_seed(0)
    xs = [torch.randn(1, 100 + 10 * i, 256, device="cuda").to(torch.bfloat16) for i in range(7)]
    H = G.calculate_hessian(xs)[0]
    acc = G.HessianAccumulator(256, xs[0].device, chunk_tokens=1)  # flushes after every item
    for x in xs:
        acc.add(x.reshape(-1, 256), 1)
    scale = float(H.abs().max())  # entries are sums with cancellation: errors scale with the diagonal
    torch.testing.assert_close(H, acc.result(), rtol=1e-5, atol=1e-6 * scale)
    ref = sum(x[0].double().T @ x[0].double() for x in xs) * 2 / len(xs)
    torch.testing.assert_close(H.double(), ref, rtol=1e-5, atol=1e-6 * scale)


@pytest.mark.parametrize("mode,gs", [("int4_sym", 16), ("int4_asym", 32), ("nf4", -1)])
def test_gptq_node_3d_weight_vs_oracle(mode, gs):
    """GPTQ on a 3-D / MoE weight [experts, out, in] (gptq.py:136-169: one Hessian per expert from `inp[expert]`, results
    stacked): Hessians against the oracle, and per expert the column sweep against the oracle's given the same Hinv."""
    from nncf_amd import gptq
    E, N, K, T = 3, 24, 128, 96
    torch.manual_seed(6)
    w = torch.randn(E, N, K, device="cuda") * 0.05
    xs = [torch.randn(E, T, K, device="cuda") for _ in range(4)]
    q, s, z = gptq.quantize_node_weight(w, xs, _cfg(mode, gs), (2,))
    G = 1 if gs == -1 else K // gs
    assert tuple(q.shape) == (E, N, K) and tuple(s.shape) == ((E, N, G, 1) if gs != -1 else (E, N, 1))
    H = gptq.calculate_hessian(xs, is_3d_weight=True)
    ref_H = O.gptq_hessian([host(x) for x in xs], is_3d_weight=True)
    np.testing.assert_allclose(host(H), ref_H, rtol=1e-5, atol=1e-6 * np.abs(ref_H).max())
    for e in range(E):
        U, dead = gptq.prepare_hinv(H[e].clone(), 0.1)
        q_ref, s_ref, z_ref = O.gptq_quantize_weights(None, host(w[e]).copy(), mode, gs, hinv=host(U), dead=host(dead))
        close = np.isclose(host(q[e]), q_ref, rtol=1e-4, atol=1e-7)
        assert close.mean() >= 0.99, close.mean()
        assert np.isclose(host(s[e]), s_ref, rtol=2e-5, atol=0).mean() >= 0.99
        if z_ref is not None:
            assert (host(z[e]) == z_ref).mean() >= 0.99
    assert (z is None) == (mode != "int4_asym")


@pytest.mark.gpu
def test_awq_preparation_helpers_equal_the_torch_ops():
    """awq._quantile_rows replaces torch.quantile in the search preparation (its code objects are the first AWQ call's
    largest one-time cost): same bits on finite inputs, ties and constant rows included."""
    from nncf_amd import awq
    g = torch.Generator(device="cuda").manual_seed(3)
    for n in (8, 16, 32, 64, 100, 128, 256):
        x = torch.rand((40, n), device="cuda", generator=g) * 3 + 1e-3
        x[1] = 0.25                      # constant row
        x[2, : n // 2] = x[2, 0]         # ties around the rank
        x[3] = x[3].round()              # few distinct values
        for q in (0.1, 0.5, 0.9):
            want = torch.quantile(x, q, dim=1, keepdim=True)
            got = awq._quantile_rows(x, q)
            assert got.shape == want.shape and torch.equal(got, want), (n, q)
    s = torch.rand((2, 500), device="cuda", generator=g).round(decimals=2)  # many equal keys: stable order matters
    assert torch.equal(torch.sort(-s, dim=-1, stable=True)[1], torch.argsort(-s, stable=True))


# ---- the bf16 split GEMM behind the large products of GPTQ's dense algebra (csrc/wc_hessian.hip: k_split3_*, k_gemm16t)
def _split3_call(x, contract_cols, order, negate=False):
    from nncf_amd import gptq
    return gptq._split3(x, contract_cols, order, negate)


@pytest.mark.parametrize("contract_cols", [False, True])
@pytest.mark.parametrize("order", [0, 1])
def test_split3_is_the_exact_three_way_split(contract_cols, order):
    """x = h + m + l exactly with 8-bit pieces; row 6t + s of the result holds slot s (order 0: h h h m m l, order 1:
    h m l h m h) of contraction index t; views with a row pitch, ragged sizes, negation."""
    g = torch.Generator(device="cuda").manual_seed(5)
    base = torch.randn(301, 530, device="cuda", generator=g) * torch.exp(4 * torch.randn(301, 530, device="cuda", generator=g))
    base[3, 7] = 0.0
    base[5, 9] = 1e-41  # subnormal
    x = base[:, 3:523]  # pitch 530, first column not 16-byte aligned
    d = _split3_call(x, contract_cols, order, negate=bool(order)).float()
    src = (x.T if contract_cols else x) * (-1.0 if order else 1.0)
    k = src.shape[0]
    h = (src.contiguous().view(torch.int32) & -65536).view(torch.float32)
    r1 = src - h
    m = (r1.view(torch.int32) & -65536).view(torch.float32)
    l = ((r1 - m).view(torch.int32) & -65536).view(torch.float32)  # (a cut only below 2^-133: subnormal inputs)
    normal = src.abs() > 1e-30
    assert torch.equal((h + m + l)[normal], src[normal])
    assert float((h + m + l - src).abs().max()) < 1e-39
    want = [h, h, h, m, m, l] if order == 0 else [h, m, l, h, m, h]
    pieces = d.reshape(k, 6, -1)
    for s in range(6):
        assert torch.equal(pieces[:, s], want[s]), s


@pytest.mark.parametrize("m,k,n", [(256, 32, 256), (264, 40, 520), (1000, 136, 776), (512, 2048, 512), (8, 3, 8),
                                   (2048, 1024, 4096)])
def test_gemm_bf16_tn_acc_matches_float64(m, k, n):
    """C -= A B through the split + k_gemm16t against float64 (as close as the fp32 GEMM: 2e-6 of max|C|), on a strided
    view of C, with ragged tiles and token splits."""
    from nncf_amd import _lib
    g = torch.Generator(device="cuda").manual_seed(m + k + n)
    A = torch.randn(m, k, device="cuda", generator=g)
    B = torch.randn(k, n + 8, device="cuda", generator=g)[:, 8:]
    Cfull = torch.randn(m, n + 16, device="cuda", generator=g)
    C = Cfull[:, 16:]
    ref = C.double() - A.double() @ B.double()
    guard = Cfull[:, :16].clone()
    P = _split3_call(A, True, 0, negate=True)
    Q = _split3_call(B, False, 1)
    _lib.check(_lib.lib().nncf_amd_gemm_bf16_tn_acc(_lib.ptr(C), C.stride(0), _lib.ptr(P), m, _lib.ptr(Q), n, m, n, 6 * k, 0,
                                                    _lib.stream_of(C)))
    err = float((C.double() - ref).abs().max() / ref.abs().max())
    assert err < 2e-6, err
    assert torch.equal(Cfull[:, :16], guard)  # nothing outside the view is written
    # B given transposed ([n, k]: contraction over its columns) gives the same sums
    C2 = torch.zeros(m, n, device="cuda")
    Q2 = _split3_call(B.T.contiguous(), True, 1)
    assert torch.equal(Q2, Q)
    _lib.check(_lib.lib().nncf_amd_gemm_bf16_tn_acc(_lib.ptr(C2), n, _lib.ptr(P), m, _lib.ptr(Q2), n, m, n, 6 * k, 0,
                                                    _lib.stream_of(C2)))
    ref2 = -(A.double() @ B.double())
    assert float((C2.double() - ref2).abs().max() / ref2.abs().max()) < 2e-6


def test_gemm_bf16_tn_acc_rejects_what_it_cannot_address():
    from nncf_amd import _lib
    L = _lib.lib()
    c = torch.zeros(8, 8, device="cuda")
    p = torch.zeros(6, 8, dtype=torch.bfloat16, device="cuda")
    assert L.nncf_amd_gemm_bf16_tn_acc(_lib.ptr(c), 8, _lib.ptr(p), 4, _lib.ptr(p), 8, 8, 8, 6, 0, None) == _lib.ERR_BAD_ARG  # ldp < m
    assert L.nncf_amd_gemm_bf16_tn_acc(_lib.ptr(c), 8, _lib.ptr(p), 12, _lib.ptr(p), 8, 8, 8, 6, 0, None) == _lib.ERR_UNSUPPORTED  # pitch % 8
    assert L.nncf_amd_gemm_bf16_tn_acc(_lib.ptr(c), 8, _lib.ptr(p), 8, _lib.ptr(p), 8, 8, 8, 1 << 29, 0, None) == _lib.ERR_UNSUPPORTED  # >= 4 GiB


def test_mm_takes_the_same_decision_for_both_routes():
    """gptq._mm: large aligned products on the bf16 route, the rest on the fp32 GEMM; both within fp32-GEMM accuracy of
    float64, for every combination the recursion uses (B transposed or not, accumulate into out or a copy, negated)."""
    from nncf_amd import gptq
    g = torch.Generator(device="cuda").manual_seed(11)
    for (m, k, n) in ((3072, 2048, 3072), (200, 64, 120)):  # 3.9e10 flops: bf16 route; tiny: fp32 GEMM
        A = torch.randn(m, k, device="cuda", generator=g)
        for b_t in (False, True):
            B = torch.randn((n, k) if b_t else (k, n), device="cuda", generator=g)
            prod = A.double() @ (B.double().T if b_t else B.double())
            for negate in (False, True):
                sgn = -1.0 if negate else 1.0
                out = torch.empty(m, n, device="cuda")
                gptq._mm(out, A, B, b_t=b_t, negate=negate)
                assert float((out.double() - sgn * prod).abs().max() / prod.abs().max()) < 4e-6
                C = torch.randn(m, n, device="cuda", generator=g)
                ref = C.double() + sgn * prod
                out2 = C.clone()
                gptq._mm(out2, A, B, b_t=b_t, acc=out2, negate=negate)
                assert float((out2.double() - ref).abs().max() / ref.abs().max()) < 4e-6
                out3 = torch.empty(m, n, device="cuda")
                gptq._mm(out3, A, B, b_t=b_t, acc=C, negate=negate)
                assert float((out3.double() - ref).abs().max() / ref.abs().max()) < 4e-6


@pytest.mark.parametrize("n", [3328, 3336])
def test_triangular_products_in_one_launch(n):
    """The four products of a level of the Hinv recursion through k_gemm16t with the zero / unwanted tiles skipped in
    the kernel (tri = 1..4) against float64: X T^T, X T, T X for a lower triangular T, and the lower triangle of
    C - A A^T.  n = 3336: a ragged last tile."""
    from nncf_amd import gptq
    g = torch.Generator(device="cuda").manual_seed(n)
    X = torch.randn(n, n, device="cuda", generator=g)
    T = torch.tril(torch.randn(n, n, device="cuda", generator=g))
    assert gptq._bf16x3_route(n, n, n, X, T, X, tri=1)

    def close(got, ref, mask=None):
        d = (got.double() - ref).abs()
        if mask is not None:
            d = d * mask
        return float(d.max() / ref.abs().max())

    assert close(gptq._mm_dense_lowerT(X, T), X.double() @ T.double().T) < 4e-6
    assert close(gptq._mm_dense_lower(X, T), X.double() @ T.double()) < 4e-6
    out = torch.empty(n, n, device="cuda")
    assert close(gptq._mm_lower_dense(T, X, out), T.double() @ X.double()) < 4e-6
    C = torch.randn(n, n, device="cuda", generator=g)
    A = torch.randn(n, 2048, device="cuda", generator=g)
    lower = torch.tril(torch.ones(n, n, device="cuda", dtype=torch.float64))
    ref = C.double() - A.double() @ A.double().T
    got = gptq._syrk_lower_sub(C, A)  # in place
    assert got.data_ptr() == C.data_ptr()
    assert close(got, ref, lower) < 4e-6
