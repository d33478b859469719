#!/usr/bin/env python3
"""Randomized parity campaign: the HIP lowering (single-tensor and multi-tensor entry points, every mode) against the
oracle on random shapes, group sizes, dtypes and value distributions; bit-exact codes / scales / zero points.

Not collected by pytest (no test_ prefix); run on a GPU box as

    python -m tests.fuzz_parity --cases 3000 --seed 0

It prints a summary line and every mismatch (seed + case description reproduce it).  The fixed-seed parity tests of
tests/test_gpu_*.py stay the gate; this widens the net (results of the last run: DESIGN.md section 3)."""
import argparse
import json
import sys
import time

import numpy as np
import torch

from oracle import wc_oracle as O
from tests.gpu_utils import host, representable, to_dev

WIDE = False  # --wide: magnitudes over the whole fp32 range (subnormal groups, sums that overflow)
INT_MODES = ["int8_asym", "int8_sym", "int4_asym", "int4_sym"]
FLOAT_MODES = ["nf4", "mxfp4"]


def _cfg(mode, gs):
    from nncf_amd.parameters import WeightCompressionConfig
    return WeightCompressionConfig(mode, gs)


def _same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    if a.shape != b.shape:
        return False
    if a.dtype.kind == "f":
        return bool(((a == b) | (np.isnan(a) & np.isnan(b))).all())
    return bool((a == b).all())


def _diagnose(w32, codes, rc, scale, rs, zp, rz):
    out = f" | max|w| {np.abs(w32).max():.3e} min|w|>0 {np.abs(w32[w32 != 0]).min() if (w32 != 0).any() else 0:.3e}"
    bad_s = ~((scale == rs) | (np.isnan(scale) & np.isnan(rs)))
    out += f" | scale mismatches {int(bad_s.sum())}"
    if bad_s.any():
        i = tuple(np.argwhere(bad_s)[0])
        out += f" first got {scale[i]!r} ref {rs[i]!r}"
    if zp is not None:
        out += f" | zp mismatches {int((zp != rz).sum())}"
    if codes.shape == rc.shape:
        bad_c = ~((codes == rc) | ((codes != codes) & (rc != rc)))
        out += f" | code mismatches {int(bad_c.sum())} of {rc.size}"
        if bad_c.any():
            i = tuple(np.argwhere(bad_c)[0])
            wv = w32.reshape(rc.shape)[i]
            si = tuple(list(i[:-1]) + [0])
            out += f" first w {wv!r} got {codes[i]!r} ref {rc[i]!r} scale {rs[si]!r}"
    return out


def random_values(rng, shape):
    """One of several distributions: plain normal at a random magnitude, outlier channels, constant / zero groups,
    values on and next to rounding ties, a wide dynamic range."""
    kind = rng.integers(0, 7)
    mag = np.float32(10.0 ** (rng.uniform(-42, 37) if WIDE else rng.uniform(-6, 3)))
    w = rng.standard_normal(shape).astype(np.float32) * mag
    if kind == 1:  # outlier columns
        idx = rng.integers(0, shape[-1], max(1, shape[-1] // 50))
        w[..., idx] *= np.float32(rng.uniform(5, 50))
    elif kind == 2:  # some constant or zero runs
        flat = w.reshape(-1)
        for _ in range(3):
            a = int(rng.integers(0, flat.size))
            b = min(flat.size, a + int(rng.integers(1, 300)))
            flat[a:b] = 0.0 if rng.random() < 0.5 else flat[a]
    elif kind == 3:  # integers + 0.5 times a step: exact ties for the symmetric modes' scales
        step = np.float32(2.0 ** rng.integers(-12, 4))
        w = (rng.integers(-8, 8, shape).astype(np.float32) + np.float32(0.5)) * step
        flat = w.reshape(-1)
        sel = rng.random(flat.size) < 0.3
        flat[sel] = np.nextafter(flat[sel], np.float32(np.inf if rng.random() < 0.5 else -np.inf))
    elif kind == 4:  # wide dynamic range inside a row
        w = w * np.float32(10.0) ** rng.uniform(-3, 3, shape).astype(np.float32)
    elif kind == 5:  # all positive / all negative rows (asymmetric zero points at the range ends)
        w = np.abs(w) * (np.float32(1.0) if rng.random() < 0.5 else np.float32(-1.0))
    with np.errstate(all="ignore"):  # finite weights only: inf / NaN weights are outside the contract
        return np.clip(np.nan_to_num(w.astype(np.float32), nan=0.0), -3.0e38, 3.0e38).astype(np.float32)


def random_case(rng):
    mode = str(rng.choice(INT_MODES + FLOAT_MODES))
    dtype = str(rng.choice(["f32", "bf16", "f16"]))
    if mode == "mxfp4":
        gs = 32
    elif mode.startswith("int8") and rng.random() < 0.6:
        gs = -1
    else:
        gs = int(rng.choice([-1, 4, 8, 16, 32, 64, 128, 256, 24, 96]))
    rows = int(rng.integers(1, 200))
    if gs == -1:
        cols = int(rng.choice([int(rng.integers(1, 300)), int(rng.integers(1, 40)) * 8, int(rng.choice([512, 1024, 4096, 5120]))]))
    else:
        cols = gs * int(rng.integers(1, max(2, 2048 // gs)))
    three_d = rng.random() < 0.1
    shape = (int(rng.integers(1, 4)), rows, cols) if three_d else (rows, cols)
    return mode, dtype, gs, shape


def check_case(rng, wl, stats, failures, batched_pool):
    mode, dtype, gs, shape = random_case(rng)
    desc = f"{mode} g{gs} {dtype} {shape}"
    w32 = representable(random_values(rng, shape), dtype)
    if dtype == "f16":
        w32 = np.clip(w32, -60000.0, 60000.0)
    w = to_dev(w32, dtype)
    axis = len(shape) - 1
    rc, rs, rz = O.compress_weight(w32, (axis,), mode, gs)
    cw = wl.compress_weight(w, (axis,), _cfg(mode, gs))
    ok = _same(host(cw.tensor), rc) and _same(host(cw.scale), rs) and (rz is None or _same(host(cw.zero_point), rz))
    stats["single"] += 1
    if not ok and not np.isfinite(w32).all():
        stats["nonfinite_skipped"] = stats.get("nonfinite_skipped", 0) + 1  # inf / NaN weights: outside the contract
    elif not ok:
        failures.append("single " + desc + _diagnose(w32, host(cw.tensor), rc, host(cw.scale), rs,
                                                     None if rz is None else host(cw.zero_point), rz))
    # the same weight against GIVEN parameters (another tensor's scale, perturbed): the path scale estimation, AWQ
    # and GPTQ results take into the lowering; and the quantize-dequantize form
    from nncf_amd.parameters import CompressedWeight
    if mode in INT_MODES and rng.random() < 0.5:
        factor = np.float32(rng.uniform(0.3, 3.0))
        gscale = (rs * factor).astype(np.float32)
        gzp = None if rz is None else np.clip(rz + rng.integers(-2, 3, rz.shape), 0, 255 if "8" in mode else 15).astype(rz.dtype)
        rc2, rs2, rz2 = O.compress_weight(w32, (axis,), mode, gs, gscale, gzp)
        pre = CompressedWeight(None, torch.from_numpy(gscale).cuda(), None if gzp is None else torch.from_numpy(gzp).cuda())
        cw2 = wl.compress_weight(w, (axis,), _cfg(mode, gs), pre)
        stats["given"] += 1
        if not _same(host(cw2.tensor), rc2):
            failures.append("given-parameters " + desc)
        if shape[-1] % 2 == 0 and mode.startswith("int4"):
            packed = wl.compress_weight_packed(w, (axis,), _cfg(mode, gs), pre)
            ref = O.pack_uint4(rc2) if mode == "int4_asym" else O.pack_int4(rc2)
            if not _same(host(packed.tensor).reshape(-1), np.asarray(ref).reshape(-1)):
                failures.append("given-parameters packed " + desc)
    if mode in FLOAT_MODES and rng.random() < 0.5:
        # a given scale (what scale estimation hands to the NF4 lowering); a power-of-two factor for the MX format
        factor = np.float32(2.0 ** rng.integers(-2, 3)) if mode == "mxfp4" else np.float32(rng.uniform(0.3, 3.0))
        gscale = (rs * factor).astype(np.float32)
        if np.isfinite(gscale).all() and (gscale != 0).all():
            rv2, _, _ = O.compress_weight(w32, (axis,), mode, gs, gscale)
            cw2 = wl.compress_weight(w, (axis,), _cfg(mode, gs), CompressedWeight(None, torch.from_numpy(gscale).cuda()))
            stats["given"] += 1
            if not _same(host(cw2.tensor), rv2):
                failures.append("given-scale " + desc)
    if rng.random() < 0.3:
        if mode in INT_MODES:
            dq = wl.integer_quantize_dequantize_weight(w, _cfg(mode, gs), (axis,))
            rdq = O.integer_quantize_dequantize_weight(w32, mode, gs, (axis,))
        else:
            dq = wl.float_quantize_dequantize_weight(w, _cfg(mode, gs), (axis,))
            rdq = O.float_quantize_dequantize_weight(w32, mode, gs, (axis,))
        stats["qdq"] += 1
        if not _same(host(dq), rdq):
            failures.append("quantize-dequantize " + desc)
    if len(shape) == 2 and (shape[1] % 2 == 0 or mode.startswith("int8")):
        batched_pool.setdefault((mode, gs, dtype), []).append((w, w32, rc, rs, rz, desc))


def check_batched(wl, stats, failures, batched_pool):
    for (mode, gs, dtype), items in batched_pool.items():
        outs = wl.compress_weights_packed_batched([it[0] for it in items], _cfg(mode, gs), want_e8m0=mode == "mxfp4")
        for (w, w32, rc, rs, rz, desc), cw in zip(items, outs):
            got = host(cw.tensor).reshape(-1)
            if mode in FLOAT_MODES:  # the build's 4-bit codes: LUT[code] == the reference's value
                lut = O.NF4_QUANTILES if mode == "nf4" else O.E2M1_VALUE_OF_CODE
                got = lut[np.stack([got & 15, got >> 4], axis=1).reshape(-1)]
                ref = rc
            elif mode == "int4_asym":
                ref = O.pack_uint4(rc)
            elif mode == "int4_sym":
                ref = O.pack_int4(rc)
            else:
                ref = rc
            ok = _same(got, np.asarray(ref).reshape(-1)) and _same(host(cw.scale).reshape(-1), rs.reshape(-1))
            if rz is not None:
                ok = ok and _same(host(cw.zero_point).reshape(-1), rz.reshape(-1))
            stats["batched"] += 1
            if not ok:
                extra = ""
                if np.isfinite(w32).all():
                    refv = np.asarray(ref).reshape(-1)
                    bad = ~((got == refv) | ((got != got) & (refv != refv))) if got.shape == refv.shape else None
                    if bad is not None and bad.any() and mode in FLOAT_MODES:
                        i = int(np.argwhere(bad)[0][0])
                        extra = (f" | {int(bad.sum())} of {bad.size} values differ; first w {w32.reshape(-1)[i]!r} got {got[i]!r} "
                                 f"ref {refv[i]!r} scale {rs.reshape(-1)[i // gs]!r} group max {np.abs(w32.reshape(-1, gs)[i // gs]).max()!r}")
                    sb = host(cw.scale).reshape(-1) != rs.reshape(-1)
                    extra += f" | scale mismatches {int(sb.sum())}"
                    failures.append("batched " + desc + extra)
                else:
                    stats["nonfinite_skipped"] = stats.get("nonfinite_skipped", 0) + 1


def check_reducers(rng, n_cases, failures):
    """activation_stats (token mean + the four mixed-precision statistics) on random [batch, tokens, hidden] against
    float64 / the oracle's reducers, at the tolerances of tests/test_gpu_reducers.py; mean-only call = same bits."""
    from nncf_amd import statistics as S
    for _ in range(n_cases):
        dtype = str(rng.choice(["f32", "bf16", "f16"]))
        hidden = int(rng.choice([int(rng.integers(1, 700)), 8 * int(rng.integers(1, 300)), 4096, 5120]))
        tokens = int(rng.choice([int(rng.integers(1, 40)), int(rng.integers(200, 3000)), 256, 2048]))
        batch = int(rng.choice([1, 1, 1, 2, 5]))
        x32 = rng.standard_normal((batch, tokens, hidden)).astype(np.float32) * np.float32(10.0 ** rng.uniform(-2, 1))
        x32[..., int(rng.integers(0, hidden))] += np.float32(rng.uniform(-200, 200))  # a channel with a large mean
        x = to_dev(x32, dtype)
        xr = host(x).astype(np.float32)
        mean, stats = S.activation_stats(x)
        only, _ = S.activation_stats(x, want_stats=False)
        rt = 3e-6 if dtype == "f32" else 2 ** -7
        x64 = xr.astype(np.float64)
        try:
            assert torch.equal(only, mean), "mean-only bits"
            np.testing.assert_allclose(host(mean), x64.mean(axis=(0, 1)), rtol=3e-6, atol=5e-6 * max(1.0, np.abs(x64).max()))
            np.testing.assert_allclose(float(stats[0]), O.mean_variance_reducer(xr, (1,), dtype), rtol=max(rt, 1e-4))
            np.testing.assert_allclose(float(stats[1]), O.max_variance_reducer(xr, (1,), dtype), rtol=max(rt, 1e-4))
            np.testing.assert_allclose(float(stats[2]), O.mean_abs_max_reducer(xr, (1,)), rtol=3e-6)
            np.testing.assert_allclose(float(stats[3]), np.sum(x64 ** 2), rtol=1e-5)
        except AssertionError as e:
            failures.append(f"reducers {dtype} {(batch, tokens, hidden)}: {str(e).strip().splitlines()[-1][:120]}")


def check_error_metric(rng, n_cases, failures):
    """get_integer_quantization_error[s] (per weight and batched; grouped and whole-row groups) against the oracle at
    the reference's rtol 1e-4."""
    from nncf_amd import weight_lowering as wl
    pool = {}
    for _ in range(n_cases):
        mode = str(rng.choice(INT_MODES))
        dtype = str(rng.choice(["f32", "bf16", "f16"]))
        gs = int(rng.choice([-1, -1, 8, 16, 32, 64, 128, 24]))
        rows = int(rng.integers(1, 120))
        cols = (int(rng.choice([int(rng.integers(1, 300)), 8 * int(rng.integers(1, 600)), 4096, 14336])) if gs == -1
                else gs * int(rng.integers(1, max(2, 4096 // gs))))
        w32 = representable(rng.standard_normal((rows, cols)).astype(np.float32) * np.float32(10.0 ** rng.uniform(-3, 1)), dtype)
        pool.setdefault((mode, gs, dtype), []).append(w32)
    for (mode, gs, dtype), ws32 in pool.items():
        ws = [to_dev(w, dtype) for w in ws32]
        for reduction in ("max_mean", "frobenius"):
            got = wl.get_integer_quantization_errors(ws, [(1,)] * len(ws), _cfg(mode, gs), reduction)
            for i, w32 in enumerate(ws32):
                ref = O.get_integer_quantization_error(w32, (1,), mode, gs, reduction)
                one = wl.get_integer_quantization_error(ws[i], (1,), _cfg(mode, gs), reduction) if i % 3 == 0 else got[i]
                if not (np.isclose(got[i], ref, rtol=1e-4, atol=1e-12) and np.isclose(one, ref, rtol=1e-4, atol=1e-12)):
                    failures.append(f"error metric {mode} g{gs} {dtype} {w32.shape} {reduction}: {got[i]} / {one} vs {ref}")


def check_decompress(rng, n_cases, failures):
    """The decompressor forward on random codes / scales / zero points against the reference's op chain in torch
    (layers.py decompress_symmetric / _asymmetric, fp16 arithmetic; NF4 / e2m1 look-up times scale): bit-exact, fp16
    scales over their whole range (subnormals, products that overflow), any even group size."""
    from nncf_amd import decompressors as D
    from tests.test_gpu_decompress import _unpack_uint4
    g = torch.Generator(device="cuda").manual_seed(int(rng.integers(0, 2 ** 31)))
    e2m1 = torch.tensor([0, 0.5, 1, 1.5, 2, 3, 4, 6, -0.0, -0.5, -1, -1.5, -2, -3, -4, -6], device="cuda")
    nf4 = torch.tensor(O.NF4_QUANTILES, device="cuda", dtype=torch.float32)
    for _ in range(n_cases):
        mode = str(rng.choice(INT_MODES + FLOAT_MODES))
        out_dtype = [torch.float16, torch.bfloat16, torch.float32][int(rng.integers(0, 3))]
        gs = 32 if mode == "mxfp4" else int(rng.choice([2 * int(rng.integers(1, 80)), 8 * int(rng.integers(1, 600)), 32, 64, 128, 4096]))
        n_groups = int(rng.choice([int(rng.integers(1, 50)), int(rng.integers(50, 20000))]))
        if n_groups * gs > 1 << 25:
            n_groups = max(1, (1 << 25) // gs)
        four = not mode.startswith("int8")
        n = n_groups * gs
        codes = torch.randint(0, 256, (n // 2 if four else n,), dtype=torch.uint8, device="cuda", generator=g)
        if mode == "mxfp4":
            scale = torch.randint(1, 255, (n_groups,), dtype=torch.uint8, device="cuda", generator=g)
        else:  # any finite fp16 bit pattern, both signs
            bits = torch.randint(0, 0x7c00, (n_groups,), dtype=torch.int32, device="cuda", generator=g)
            sign = torch.randint(0, 2, (n_groups,), dtype=torch.int32, device="cuda", generator=g) << 15
            scale = (bits | sign).to(torch.int16).view(torch.float16)
        zp = None
        if mode == "int4_asym":
            zp = torch.randint(0, 256, ((n_groups + 1) // 2,), dtype=torch.uint8, device="cuda", generator=g)
        if mode == "int8_asym":
            zp = torch.randint(0, 256, (n_groups,), dtype=torch.uint8, device="cuda", generator=g)
        got = D.decompress(codes, mode, n_groups, gs, scale, zp, out_dtype)
        c = _unpack_uint4(codes).reshape(n_groups, gs) if four else codes.reshape(n_groups, gs)
        if mode == "mxfp4":
            ref = e2m1[c.long()] * torch.exp2(scale.float() - 127).unsqueeze(1)
        elif mode == "nf4":
            ref = nf4[c.long()].to(torch.float16) * scale.unsqueeze(1)
        elif mode == "int4_sym":
            ref = (c.to(torch.float16) - 8) * scale.unsqueeze(1)
        elif mode == "int4_asym":
            z = _unpack_uint4(zp).reshape(-1)[:n_groups]
            ref = (c.to(torch.float16) - z.to(torch.float16).unsqueeze(1)) * scale.unsqueeze(1)
        elif mode == "int8_asym":
            ref = (c.to(torch.float16) - zp.to(torch.float16).unsqueeze(1)) * scale.unsqueeze(1)
        else:
            ref = c.view(torch.int8).to(torch.float16) * scale.unsqueeze(1)
        ref = ref.reshape(-1).to(out_dtype)
        same = (got == ref) | (got.isnan() & ref.isnan())
        if not bool(same.all()) or not torch.equal(torch.signbit(got), torch.signbit(ref)):
            failures.append(f"decompress {mode} g{gs} x {n_groups} -> {out_dtype}: {int((~same).sum())} values differ")


def check_misc(rng, n_cases, failures):
    """Nibble packing / unpacking, integer dequantization and the token means of a list of samples."""
    from nncf_amd import statistics as S, weight_lowering as wl
    from nncf_amd.parameters import CompressedWeight
    for _ in range(n_cases):
        n = 2 * int(rng.choice([int(rng.integers(1, 40)), int(rng.integers(40, 100000))]))
        u = rng.integers(0, 16, n).astype(np.uint8)
        i4 = rng.integers(-8, 8, n).astype(np.int8)
        pu, pi = wl.pack_uint4(torch.from_numpy(u).cuda()), wl.pack_int4(torch.from_numpy(i4).cuda())
        if not (_same(host(pu).reshape(-1), O.pack_uint4(u).reshape(-1)) and _same(host(pi).reshape(-1), O.pack_int4(i4).reshape(-1))):
            failures.append(f"pack n={n}")
        if not (_same(host(wl.unpack_uint4(pu)).reshape(-1), u) and _same(host(wl.unpack_int4(pi)).reshape(-1), i4)):
            failures.append(f"unpack n={n}")
        # dequantization of grouped codes
        mode = str(rng.choice(INT_MODES))
        gs = int(rng.choice([4, 16, 24, 128]))
        rows, groups = int(rng.integers(1, 60)), int(rng.integers(1, 20))
        w32 = (rng.standard_normal((rows, groups * gs)) * 10.0 ** rng.uniform(-4, 2)).astype(np.float32)
        rc, rs, rz = O.compress_weight(w32, (1,), mode, gs)
        cw = CompressedWeight(torch.from_numpy(rc).cuda(), torch.from_numpy(rs).cuda(), None if rz is None else torch.from_numpy(rz).cuda())
        if not _same(host(wl.do_integer_dequantization(cw, 1)), O.do_integer_dequantization(rc, rs, rz, 1)):
            failures.append(f"dequantize {mode} g{gs} {(rows, groups * gs)}")
        # token means of a ragged list of samples
        dtype = str(rng.choice(["f32", "bf16", "f16"]))
        hidden = int(rng.choice([int(rng.integers(1, 300)), 8 * int(rng.integers(1, 200)), 4096]))
        counts = [int(rng.choice([1, int(rng.integers(1, 70)), int(rng.integers(70, 1500))])) for _ in range(int(rng.integers(2, 80)))]
        xs32 = [(rng.standard_normal((1, t, hidden)) + rng.standard_normal(hidden) * 5).astype(np.float32) for t in counts]
        xs = [to_dev(x, dtype) for x in xs32]
        got = S.activations_to_wc_statistics(xs)
        tol = 1e-6 if dtype == "f32" else (8e-3 if dtype == "bf16" else 1e-3)
        for x, m in zip(xs, got.mean_values):
            ref = host(x).astype(np.float64).reshape(-1, hidden).mean(0)
            if not np.allclose(host(m).astype(np.float64), ref, rtol=tol, atol=tol * max(1e-30, np.abs(ref).max())):
                failures.append(f"token means {dtype} hidden {hidden} tokens {x.shape[1]} of {len(xs)} samples")
                break


def check_data_aware(rng, n_cases, failures):
    """Scale estimation and the AWQ search on random shapes / sample counts against the oracle on the same statistics,
    at the bars of tests/test_gpu_data_aware.py (scales 1e-5 on >= 99.5 % of the groups, zero points equal; AWQ
    scales 1e-5 on >= 97 % of the channels)."""
    from nncf_amd import awq, scale_estimation as SE
    from nncf_amd.statistics import WCTensorStatistic
    for _ in range(n_cases):
        mode = str(rng.choice(["int4_sym", "int4_asym", "nf4", "int8_asym", "mxfp4"]))
        # two thirds on the tiled MFMA kernels' shapes (32 / 64 / 128, <= 64 samples), one third on everything else
        # (k_se_step_any / k_awq_search_any: other group sizes, whole rows, more samples)
        tiled = rng.random() < 0.67
        gs = int(rng.choice([32, 64, 128])) if tiled else int(rng.choice([8, 16, 24, 40, 48, 96, 100, 160, 256]))
        rows = int(rng.integers(1, 300))
        K = gs * int(rng.integers(1, 9))
        n_samples = int(rng.integers(1, 80)) if tiled else int(rng.integers(1, 200))
        w32 = (rng.standard_normal((rows, K)) * 0.05).astype(np.float32)
        means = []
        for _i in range(n_samples):
            m = rng.standard_normal(K).astype(np.float32)
            m[:: int(rng.integers(20, 70))] *= np.float32(12.0)
            means.append(m)
        shapes = [(int(rng.integers(1, 30)), K) for _i in range(n_samples)]
        stats = WCTensorStatistic([torch.from_numpy(m).cuda() for m in means], shapes)
        w = torch.from_numpy(w32).cuda()
        subset = int(rng.choice([8, 32, 64])) if tiled else int(rng.choice([8, 64, 128, 256]))
        init, steps = int(rng.integers(1, 6)), int(rng.integers(1, 6))
        whole_row = (not tiled) and rng.random() < 0.2
        cfg_gs = -1 if whole_row else gs
        desc = f"{mode} g{cfg_gs} {(rows, K)} samples {n_samples} subset {subset} steps {init}/{steps}"
        sc, zp = SE.calculate_quantization_params(stats, w, (1,), _cfg(mode, cfg_gs), -1, subset, init, steps)
        ref_s, ref_z = O.scale_estimation_params(means, shapes, w32, (1,), mode, cfg_gs, -1, subset, init, steps)
        close = np.isclose(host(sc), ref_s, rtol=1e-5, atol=0)
        # (a candidate whose output error ties with the best one within fp32 summation noise may go either way: one
        # group is allowed to differ whatever the group count)
        if (~close).sum() > max(1, 0.005 * close.size) or (ref_z is not None and (host(zp) != ref_z).sum() > max(1, 0.005 * close.size)):
            failures.append(f"scale estimation {desc}: {close.mean():.4f} of the scales within 1e-5")
        if mode != "int8_asym" and rng.random() < 0.5:
            a = awq.data_aware_step(w, (1,), _cfg(mode, cfg_gs), stats, -1, subset, 0.02, 0.0, 1.0, 20)
            ref = O.awq_data_aware_scale(w32, 1, mode, cfg_gs, means, shapes, -1, subset, 0.02, 0.0, 1.0, 20)
            ok = np.isclose(host(a), ref, rtol=1e-5, atol=0).mean()
            if ok < 0.97 and (1 - ok) * ref.size > 1:
                failures.append(f"awq {desc}: {ok:.4f} of the channel scales within 1e-5")


def check_gptq(rng, n_cases, failures):
    """gptq.quantize_weights against the oracle's column loop given the SAME fp32 inverse-Hessian factor, random K /
    rows / group size / block size (groups that straddle blocks and super-blocks included), at the bar of
    test_gptq_quantize_weights_two_level_same_hinv: first block bit-exact, >= 99.5 % of the values close, scales 2e-5."""
    from nncf_amd import gptq
    from tests.test_gpu_data_aware import _same_hinv_case
    for _ in range(n_cases):
        mode = str(rng.choice(["int4_sym", "int4_asym", "nf4", "mxfp4"]))  # (the reference runs GPTQ on 4-bit weights only)
        gs = 32 if mode == "mxfp4" else int(rng.choice([16, 32, 48, 64, 96, 128, 192, -1]))
        bs = int(rng.choice([128, 128, 100, 64, 50, 32]))
        unit = 32 if gs == -1 else gs
        K = unit * int(rng.integers(1, max(2, 1400 // unit)))
        N = int(rng.integers(1, 80))
        W, U = _same_hinv_case(N, K, seed=int(rng.integers(0, 2 ** 31)))
        desc = f"gptq {mode} g{gs} block {bs} {(N, K)}"
        q_ref, s_ref, z_ref = O.gptq_quantize_weights(None, host(W).copy(), mode, gs, block_size=bs, hinv=host(U))
        q, sc, z = gptq.quantize_weights(None, W.clone(), _cfg(mode, gs), None, 0.1, bs, False, hinv=U,
                                         dead=torch.zeros(K, dtype=torch.bool, device="cuda"))
        b0 = min(bs, K)
        close = np.isclose(host(q), q_ref, rtol=1e-4, atol=1e-7)
        sclose = np.isclose(host(sc), s_ref, rtol=2e-5, atol=0)
        slack = lambda a: (~a).sum() > max(2, 0.005 * a.size)  # noqa: E731
        if not np.array_equal(host(q)[:, :b0], q_ref[:, :b0]):
            failures.append(desc + ": first block differs")
        elif (slack(close) or slack(sclose)) and (~close).any(axis=1).sum() > max(1, 0.02 * close.shape[0]):
            # (one row whose code flips at a rounding tie -- the trailing GEMMs sum in another order than the oracle's
            # column loop -- differs from that column on: the algorithm is chaotic per row; more rows than that is a defect)
            rows_bad = np.argwhere((~close).any(axis=1)).reshape(-1)
            firsts = sorted(int(np.argmax(~close[r])) for r in rows_bad)
            k0 = firsts[0]
            r0 = int(rows_bad[[int(np.argmax(~close[r])) for r in rows_bad].index(k0)])
            failures.append(desc + f": values close {close.mean():.4f}, scales close {sclose.mean():.4f}; rows with a "
                            f"difference {len(rows_bad)} of {close.shape[0]}, first differing column per such row {firsts[:12]}; "
                            f"row {r0} col {k0}: got {host(q)[r0, k0]!r} ref {q_ref[r0, k0]!r}, the row's W there {host(W)[r0, k0]!r}")
        elif z_ref is not None and slack(host(z) == z_ref):
            failures.append(desc + ": zero points differ")


def check_hessian(rng, n_cases, failures):
    """gptq.calculate_hessian on random lists of calibration items (ragged token counts, every kernel: fp32 MFMA,
    16-bit 128- and 256-tile kernels, the item-list form, odd K) against float64 at the bar of test_hessian_vs_float64."""
    from nncf_amd import gptq
    for _ in range(n_cases):
        dtype = str(rng.choice(["f32", "bf16", "f16"]))
        K = int(rng.choice([int(rng.integers(1, 400)), 8 * int(rng.integers(1, 200)), 2048 + 8 * int(rng.integers(0, 150)),
                            2048 + int(rng.integers(0, 600))]))
        n_items = int(rng.integers(1, 12))
        xs32 = [rng.standard_normal((1, int(rng.choice([1, int(rng.integers(1, 100)), int(rng.integers(100, 1200))])), K)).astype(np.float32)
                for _i in range(n_items)]
        xs = [to_dev(x, dtype) for x in xs32]
        H = host(gptq.calculate_hessian(xs))[0]
        ref = sum(host(x).astype(np.float64)[0].T @ host(x).astype(np.float64)[0] for x in xs) * (2.0 / n_items)
        if not np.allclose(H, ref, rtol=1e-5, atol=1e-5 * np.abs(ref).max()):
            err = np.abs(H - ref).max() / np.abs(ref).max()
            failures.append(f"hessian {dtype} K={K} tokens {[x.shape[1] for x in xs32]}: max error {err:.2e} of max|H|")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=1000)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--reducers", type=int, default=0, help="cases of the activation-statistics reducer instead")
    ap.add_argument("--errors", type=int, default=0, help="cases of the quantization-error metric instead")
    ap.add_argument("--decompress", type=int, default=0, help="cases of the decompressor forward instead")
    ap.add_argument("--wide", action="store_true", help="weight magnitudes over the whole fp32 range")
    ap.add_argument("--misc", type=int, default=0, help="cases of pack / unpack / dequantize / token means instead")
    ap.add_argument("--data-aware", type=int, default=0, help="cases of scale estimation / AWQ instead")
    ap.add_argument("--gptq", type=int, default=0, help="cases of the GPTQ sweep with a given inverse-Hessian factor instead")
    ap.add_argument("--hessian", type=int, default=0, help="cases of the Hessian accumulation instead")
    args = ap.parse_args()
    if args.hessian:
        failures = []
        check_hessian(np.random.default_rng(args.seed), args.hessian, failures)
        for f in failures[:50]:
            print("MISMATCH", f)
        print(json.dumps({"seed": args.seed, "hessian_cases": args.hessian, "mismatches": len(failures)}))
        return 1 if failures else 0
    if args.gptq:
        failures = []
        check_gptq(np.random.default_rng(args.seed), args.gptq, failures)
        for f in failures[:50]:
            print("MISMATCH", f)
        print(json.dumps({"seed": args.seed, "gptq_cases": args.gptq, "mismatches": len(failures)}))
        return 1 if failures else 0
    if args.data_aware:
        failures = []
        check_data_aware(np.random.default_rng(args.seed), args.data_aware, failures)
        for f in failures[:50]:
            print("MISMATCH", f)
        print(json.dumps({"seed": args.seed, "data_aware_cases": args.data_aware, "mismatches": len(failures)}))
        return 1 if failures else 0
    if args.misc:
        failures = []
        check_misc(np.random.default_rng(args.seed), args.misc, failures)
        for f in failures[:50]:
            print("MISMATCH", f)
        print(json.dumps({"seed": args.seed, "misc_cases": args.misc, "mismatches": len(failures)}))
        return 1 if failures else 0
    global WIDE
    WIDE = args.wide
    if args.decompress:
        failures = []
        check_decompress(np.random.default_rng(args.seed), args.decompress, failures)
        for f in failures[:50]:
            print("MISMATCH", f)
        print(json.dumps({"seed": args.seed, "decompress_cases": args.decompress, "mismatches": len(failures)}))
        return 1 if failures else 0
    if args.errors:
        failures = []
        check_error_metric(np.random.default_rng(args.seed), args.errors, failures)
        for f in failures[:50]:
            print("MISMATCH", f)
        print(json.dumps({"seed": args.seed, "error_metric_cases": args.errors, "mismatches": len(failures)}))
        return 1 if failures else 0
    if args.reducers:
        failures = []
        check_reducers(np.random.default_rng(args.seed), args.reducers, failures)
        for f in failures[:50]:
            print("MISMATCH", f)
        print(json.dumps({"seed": args.seed, "reducer_cases": args.reducers, "mismatches": len(failures)}))
        return 1 if failures else 0
    from nncf_amd import weight_lowering as wl
    rng = np.random.default_rng(args.seed)
    stats = {"single": 0, "batched": 0, "given": 0, "qdq": 0}
    failures, pool = [], {}
    t0 = time.time()
    for i in range(args.cases):
        check_case(rng, wl, stats, failures, pool)
        if (i + 1) % 200 == 0:
            check_batched(wl, stats, failures, pool)
            pool = {}
            print(f"[{i + 1}/{args.cases}] {stats} failures {len(failures)}  {time.time() - t0:.0f}s", flush=True)
    check_batched(wl, stats, failures, pool)
    torch.cuda.synchronize()
    for f in failures[:50]:
        print("MISMATCH", f)
    print(json.dumps({"seed": args.seed, "cases": args.cases, **stats, "mismatches": len(failures)}))
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
