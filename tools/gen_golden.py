#!/usr/bin/env python3
"""
Generates the golden fixtures under tests/golden/ by IMPORTING THE REFERENCE (nncf 3.2.0 from
/root/reference/src).  Runs only in the build container; the reference never travels to the GPU box.

    PYTHONPATH=/root/reference/src PYTHONDONTWRITEBYTECODE=1 python3 tools/gen_golden.py

Fixtures are data only: seeded inputs and the outputs the reference produced for them.
"""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

REF = os.environ.get("NNCF_REFERENCE_SRC", "/root/reference/src")
sys.path.insert(0, REF)
sys.dont_write_bytecode = True

import nncf  # noqa: E402
from nncf.common.tensor_statistics import collectors as col  # noqa: E402
from nncf.common.tensor_statistics.statistics import WCTensorStatistic  # noqa: E402
from nncf.parameters import CompressWeightsMode as M  # noqa: E402
from nncf.quantization.algorithms.weight_compression import weight_lowering as wl  # noqa: E402
from nncf.quantization.algorithms.weight_compression.activation_stats import process_stats  # noqa: E402
from nncf.quantization.algorithms.weight_compression.awq import AWQ  # noqa: E402
from nncf.quantization.algorithms.weight_compression.config import WeightCompressionConfig as Cfg  # noqa: E402
from nncf.quantization.algorithms.weight_compression.config import WeightCompressionParameters  # noqa: E402
from nncf.quantization.algorithms.weight_compression.gptq import GPTQ  # noqa: E402
from nncf.quantization.algorithms.weight_compression.parameters import CompressedWeight  # noqa: E402
from nncf.quantization.algorithms.weight_compression.scale_estimation import ScaleEstimation  # noqa: E402
from nncf.tensor import Tensor  # noqa: E402
from nncf.tensor.definitions import TensorDataType  # noqa: E402
from nncf.torch.quantization.quantize_functions import pack_int4, pack_uint4  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
MODES = {"int8_asym": M.INT8_ASYM, "int8_sym": M.INT8_SYM, "int4_asym": M.INT4_ASYM, "int4_sym": M.INT4_SYM,
         "nf4": M.NF4, "mxfp4": M.MXFP4}


def d(t):
    """nncf Tensor / torch / numpy -> numpy."""
    if t is None:
        return None
    if isinstance(t, Tensor):
        t = t.data
    if isinstance(t, torch.Tensor):
        if t.dtype == torch.bfloat16:
            return t.float().numpy()
        return t.numpy()
    return np.asarray(t)


def adversarial_rows(cols):
    rows = []
    tie = np.array([-7.5, -6.5, -0.5, 0.5, 1.5, 2.5, 6.5, 7.5], dtype=np.float32)
    rows.append(np.zeros(cols, np.float32))
    rows.append(np.full(cols, 0.5, np.float32))
    rows.append(np.full(cols, -0.5, np.float32))
    rows.append(np.full(cols, -300.0, np.float32))
    rows.append(np.full(cols, 1e-40, np.float32))
    rows.append(np.tile(tie, cols // 8))
    rows.append(np.tile(np.arange(1, 9, dtype=np.float32), cols // 8))
    rows.append(-np.tile(np.arange(1, 9, dtype=np.float32), cols // 8))
    r = np.full(cols, 1e-3, np.float32)
    r[7::8] = 10.0
    rows.append(r)
    r = np.linspace(-1, 1, cols).astype(np.float32)
    rows.append(r)
    r = np.linspace(0, 15, cols).astype(np.float32)  # exact grid for asym
    rows.append(r)
    r = np.zeros(cols, np.float32)
    r[::2] = 3e4
    r[1::2] = -6e4
    rows.append(r)
    return np.stack(rows)


def make_weight(rows, cols, seed):
    rng = np.random.default_rng(seed)
    w = (rng.standard_normal((rows, cols)) * 0.02).astype(np.float32)
    return np.concatenate([w, adversarial_rows(cols)], 0)


def to_backend(w32, dtype):
    if dtype == "f32":
        return w32
    if dtype == "f16":
        with np.errstate(over="ignore"):
            return w32.astype(np.float16)
    return torch.from_numpy(w32).to(torch.bfloat16)


def gen_int_lowering():
    out = {}
    w32 = make_weight(20, 256, 0)
    for dtype in ("f32", "f16", "bf16"):
        wb = to_backend(w32, dtype)
        if dtype == "bf16":
            out["w_bf16_bits"] = wb.view(torch.int16).numpy().view(np.uint16)
        else:
            out[f"w_{dtype}"] = wb
        for mname in ("int8_asym", "int8_sym", "int4_asym", "int4_sym"):
            for gs in (-1, 32, 64, 128):
                cfg = Cfg(mode=MODES[mname], group_size=gs)
                cw = wl.compress_weight(Tensor(wb), (1,), cfg)
                key = f"{mname}_g{gs}_{dtype}"
                out[key + "_codes"] = d(cw.tensor)
                out[key + "_scale"] = d(cw.scale)
                if cw.zero_point is not None:
                    out[key + "_zp"] = d(cw.zero_point)
                if dtype == "f32":
                    dq = wl.do_integer_dequantization(cw)
                    out[key + "_dq"] = d(dq)
                    if cfg.num_bits == 4:
                        tt = torch.from_numpy(d(cw.tensor))
                        out[key + "_packed"] = (pack_uint4(tt) if cfg.is_asym_mode else pack_int4(tt)).numpy()
    # precomputed scale / zero point (int32 and fp32 zero points), already-grouped weight, reduction_axes=None
    rng = np.random.default_rng(5)
    wg = (rng.standard_normal((16, 4, 64)) * 0.05).astype(np.float32)
    sc = (np.abs(rng.standard_normal((16, 4, 1))) * 0.01 + 1e-3).astype(np.float32)
    sc[::3] *= -1
    zp_i = rng.integers(0, 16, (16, 4, 1)).astype(np.int32)
    out["pre_w"] = wg
    out["pre_scale"] = sc
    out["pre_zp_i32"] = zp_i
    for mname in ("int8_asym", "int8_sym", "int4_asym", "int4_sym"):
        cfg = Cfg(mode=MODES[mname], group_size=64)
        if cfg.is_asym_mode:
            for zname, z in (("i32", zp_i), ("f32", zp_i.astype(np.float32))):
                dq, cw = wl.integer_quantize_dequantize_weight(Tensor(wg), cfg, None, Tensor(sc), Tensor(z),
                                                               return_compressed_weight=True)
                out[f"pre_{mname}_{zname}_codes"] = d(cw.tensor)
                out[f"pre_{mname}_{zname}_dq"] = d(dq)
        else:
            dq, cw = wl.integer_quantize_dequantize_weight(Tensor(wg), cfg, None, Tensor(sc), None,
                                                           return_compressed_weight=True)
            out[f"pre_{mname}_codes"] = d(cw.tensor)
            out[f"pre_{mname}_dq"] = d(dq)
    # error metric
    w_err = make_weight(20, 256, 1)
    out["err_w"] = w_err
    for mname in ("int8_asym", "int4_sym", "int4_asym", "int8_sym"):
        for gs in (-1, 64):
            for red in ("max_mean", "frobenius"):
                cfg = Cfg(mode=MODES[mname], group_size=gs)
                out[f"err_{mname}_g{gs}_{red}"] = np.float64(wl.get_integer_quantization_error(
                    Tensor(w_err), (1,), cfg, red))
    np.savez_compressed(os.path.join(OUT, "int_lowering.npz"), **out)
    print("int_lowering", len(out))


def gen_float_lowering():
    out = {}
    w32 = make_weight(20, 256, 2)
    # mxfp4 boundary scales: group max = 6 * 2^k (1 + m 2^-23)
    rows = []
    for k in range(-20, 8):
        for m in (0, 1, 2, 5, 11):
            s = np.float32(2.0 ** k) * (np.float32(1) + np.float32(m) * np.float32(2.0 ** -23))
            r = np.linspace(-1, 1, 32).astype(np.float32) * np.float32(0.3) * s
            r[5] = s * np.float32(6.0)
            rows.append(r)
    wb32 = np.stack(rows).reshape(-1, 64)
    out["w_boundary"] = wb32
    # e2m1/nf4 midpoints +- ulps with unit scale: group max fixed to 6 (mxfp4) / 1 (nf4)
    mids_e = np.array([-5, -3.5, -2.5, -1.75, -1.25, -0.75, -0.25, 0.25, 0.75, 1.25, 1.75, 2.5, 3.5, 5], np.float32)
    ne = []
    for mval in mids_e:
        ne += [np.nextafter(mval, np.float32(-np.inf)), mval, np.nextafter(mval, np.float32(np.inf))]
    ne = np.array(ne + [6.0, -6.0, 0.0, -0.0, 5.9, 7.0][:0], np.float32)
    row_e = np.zeros(64, np.float32)
    row_e[:ne.size] = ne
    row_e[-1] = 6.0
    from nncf.quantization.algorithms.weight_compression.constants import CENTER_OF_NF4_QUANTILES as CN
    nn_ = []
    for c in CN:
        nn_ += [np.nextafter(c, np.float32(-np.inf)), c, np.nextafter(c, np.float32(np.inf))]
    row_n = np.zeros(64, np.float32)
    row_n[:len(nn_)] = np.array(nn_, np.float32)
    row_n[-1] = 1.0
    out["w_mid_e2m1"] = row_e[None]
    out["w_mid_nf4"] = row_n[None]
    for dtype in ("f32", "f16", "bf16"):
        wb = to_backend(w32, dtype)
        if dtype == "bf16":
            out["w_bf16_bits"] = wb.view(torch.int16).numpy().view(np.uint16)
        else:
            out[f"w_{dtype}"] = wb
        for mname in ("nf4", "mxfp4"):
            for gs in (-1, 32, 64):
                cfg = Cfg(mode=MODES[mname], group_size=gs)
                cw = wl.compress_weight(Tensor(wb), (1,), cfg)
                key = f"{mname}_g{gs}_{dtype}"
                out[key + "_values"] = d(cw.tensor)
                out[key + "_scale"] = d(cw.scale)
                if dtype == "f32":
                    out[key + "_dq"] = d(wl.do_float_dequantization(cw))
    for mname in ("nf4", "mxfp4"):
        cfg = Cfg(mode=MODES[mname], group_size=32)
        cw = wl.compress_weight(Tensor(wb32), (1,), cfg)
        out[f"boundary_{mname}_values"] = d(cw.tensor)
        out[f"boundary_{mname}_scale"] = d(cw.scale)
        cwt = wl.compress_weight(Tensor(torch.from_numpy(wb32)), (1,), cfg)
        assert np.array_equal(d(cwt.scale), d(cw.scale)) and np.array_equal(d(cwt.tensor), d(cw.tensor)), mname
    cw = wl.compress_weight(Tensor(row_e[None]), (1,), Cfg(mode=M.MXFP4, group_size=-1))
    out["mid_e2m1_values"] = d(cw.tensor)
    out["mid_e2m1_scale"] = d(cw.scale)
    cw = wl.compress_weight(Tensor(row_n[None]), (1,), Cfg(mode=M.NF4, group_size=-1))
    out["mid_nf4_values"] = d(cw.tensor)
    out["mid_nf4_scale"] = d(cw.scale)
    # precomputed scale
    rng = np.random.default_rng(7)
    wg = (rng.standard_normal((16, 4, 32)) * 0.05).astype(np.float32)
    sc = (np.abs(rng.standard_normal((16, 4, 1))) * 0.02 + 1e-3).astype(np.float32)
    sc[::3] *= -1
    out["pre_w"] = wg
    out["pre_scale"] = sc
    for mname in ("nf4", "mxfp4"):
        cfg = Cfg(mode=MODES[mname], group_size=32)
        dq, cw = wl.float_quantize_dequantize_weight(Tensor(wg), cfg, None, Tensor(sc), return_compressed_weight=True)
        out[f"pre_{mname}_values"] = d(cw.tensor)
        out[f"pre_{mname}_dq"] = d(dq)
    np.savez_compressed(os.path.join(OUT, "float_lowering.npz"), **out)
    print("float_lowering", len(out))


def gen_reducers():
    out = {}
    rng = np.random.default_rng(3)
    x32 = rng.standard_normal((2, 48, 96)).astype(np.float32)
    x32[:, :, 5] *= 20
    out["x_f32"] = x32
    for dtype in ("f32", "bf16", "f16"):
        xt = torch.from_numpy(x32)
        xt = {"f32": xt, "bf16": xt.to(torch.bfloat16), "f16": xt.to(torch.float16)}[dtype]
        T = Tensor(xt)
        out[f"mean_{dtype}"] = d(col.MeanReducer(axes=(0, 1))._reduce_out_of_place([T])[0]).astype(np.float32)
        out[f"meanvar_{dtype}"] = d(col.MeanVarianceReducer(axes=(1,))._reduce_out_of_place([T])[0]).astype(np.float32)
        out[f"maxvar_{dtype}"] = d(col.MaxVarianceReducer(axes=(1,))._reduce_out_of_place([T])[0]).astype(np.float32)
        out[f"meanabsmax_{dtype}"] = d(col.MeanAbsMaxReducer(axes=(1,))._reduce_out_of_place([T])[0]).astype(np.float32)
    # HAWQ
    agg = col.HAWQAggregator(num_samples=4)
    xs = [rng.standard_normal((1, 8, 16)).astype(np.float32) for _ in range(4)]
    for i, x in enumerate(xs):
        out[f"hawq_x{i}"] = x
        agg.register_reduced_input(Tensor(x))
    out["hawq_result"] = np.float64(d(agg.aggregate()))
    # process_stats
    means = [rng.standard_normal(24).astype(np.float32) for _ in range(10)]
    shapes = [(1, int(rng.integers(3, 40)), 24) for _ in range(10)]
    out["ps_means"] = np.stack(means)
    out["ps_shapes"] = np.array(shapes, np.int64)
    for subset in (4, 3, 32):
        s, X = process_stats(WCTensorStatistic([Tensor(m) for m in means], shapes), subset, act_ch_axis=-1)
        out[f"ps_s_{subset}"] = d(s)
        out[f"ps_X_{subset}"] = d(X)
    np.savez_compressed(os.path.join(OUT, "reducers.npz"), **out)
    print("reducers", len(out))


def gen_se_awq():
    out = {}
    rng = np.random.default_rng(42)
    N, K, n = 24, 128, 12
    w = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
    acts = [rng.standard_normal((1, int(rng.integers(4, 12)), K)).astype(np.float32) for _ in range(n)]
    for i in (3, 77):
        for a in acts:
            a[..., i] *= 15
    stats = GPTQ.activations_to_wc_statistics([Tensor(a) for a in acts])
    out["w"] = w
    out["means"] = np.stack([d(m) for m in stats.mean_values])
    out["shapes"] = np.array([tuple(s) for s in stats.shape_values], np.int64)
    for mname in ("int4_sym", "int4_asym", "nf4"):
        for gs in (32, -1):
            cfg = Cfg(mode=MODES[mname], group_size=gs)
            s, z = ScaleEstimation.calculate_quantization_params(stats, Tensor(w), (1,), cfg, act_ch_axis=-1,
                                                                 subset_size=8, initial_steps=3, scale_steps=4)
            out[f"se_{mname}_g{gs}_scale"] = d(s)
            if z is not None:
                out[f"se_{mname}_g{gs}_zp"] = d(z)
    # AWQ data-aware on a (weight, stats) pair, no merge node
    for mname in ("int4_sym", "int4_asym"):
        cfg = Cfg(mode=MODES[mname], group_size=32)
        awq = AWQ(subset_size=8, percent_to_apply=0.02, alpha_min=0.0, alpha_max=1.0, steps=20)
        wp = SimpleNamespace(compression_config=cfg, reduction_axes=(1,))
        sc = awq._data_aware_step(wp, Tensor(w.copy()), stats, -1)
        out[f"awq_{mname}_scale"] = d(sc)
    awq = AWQ()
    out["awq_data_free_scale"] = d(awq._data_free_step(Tensor(w), axis=0))
    np.savez_compressed(os.path.join(OUT, "se_awq.npz"), **out)
    print("se_awq", len(out))


def gen_gptq():
    out = {}
    rng = np.random.default_rng(0)
    N, K = 20, 64
    w = (rng.standard_normal((N, K)) * 0.1).astype(np.float32)
    xs = [rng.standard_normal((1, 16, K)).astype(np.float32) for _ in range(6)]
    for x in xs:
        x[..., 9] = 0.0  # a dead column
    out["w"] = w
    out["xs"] = np.stack(xs)
    g = GPTQ(damp_percent=0.1, block_size=32, subset_size=6)
    g._backend_entity = SimpleNamespace(convolution_metatypes=[], matmul_metatypes=["MM"])
    node = SimpleNamespace(metatype="MM", layer_attributes=SimpleNamespace(
        input_attributes={"transpose": False}, constant_attributes={1: {"transpose": True}}))
    for backend in ("numpy", "torch"):
        conv = (lambda a: a) if backend == "numpy" else torch.from_numpy
        H = g._calculate_hessian(node, [Tensor(conv(x)) for x in xs])
        out[f"H_{backend}"] = d(H)
        for mname in ("int4_sym", "int4_asym", "nf4"):
            for gs in (16, 64, -1):
                for se in (False, True):
                    if se and mname == "nf4":
                        continue
                    g._scale_estimation = se
                    cfg = Cfg(mode=MODES[mname], group_size=gs)
                    wp = WeightCompressionParameters("w", node, 1, TensorDataType.float32, (N, K), (1,), cfg)
                    Hc = conv(d(H)[0].copy())
                    q, s, z = g._quantize_weights(wp, Tensor(Hc), Tensor(conv(w.copy())), [Tensor(conv(x)) for x in xs])
                    key = f"{backend}_{mname}_g{gs}_se{int(se)}"
                    out[key + "_q"] = d(q)
                    out[key + "_scale"] = d(s)
                    if z is not None:
                        out[key + "_zp"] = d(z)
    np.savez_compressed(os.path.join(OUT, "gptq.npz"), **out)
    print("gptq", len(out))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    print("reference nncf", nncf.__version__, "torch", torch.__version__, "numpy", np.__version__)
    gen_int_lowering()
    gen_float_lowering()
    gen_reducers()
    gen_se_awq()
    gen_gptq()
