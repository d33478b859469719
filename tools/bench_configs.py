#!/usr/bin/env python3
"""Per-layer wall time of the data-aware BASELINE configurations on ONE MI355X (synthetic activations and weights of
the named shapes, resident in HBM; phases timed with a device synchronize on both sides):

  c3  Llama-3-8B layer, INT4_ASYM g128, AWQ + scale estimation, 128 calibration sequences x 2048 tokens (bf16)
  c4  Llama-3-70B layer, INT4_SYM g64, GPTQ (damp 0.1, block 128), 128 x 2048 tokens (bf16)

    python tools/bench_configs.py c3|c4 [--samples 128] [--tokens 2048]

Prints one JSON object; whole-model figures are the per-layer time x the layer count (stated as such).
"""
import argparse
import json
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from nncf_amd import awq, gptq, scale_estimation as se, statistics as S, weight_lowering as wl  # noqa: E402
from nncf_amd.parameters import CompressedWeight, WeightCompressionConfig  # noqa: E402


class Phases:
    def __init__(self):
        self.ms = {}

    def run(self, name, fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        self.ms[name] = self.ms.get(name, 0.0) + (time.perf_counter() - t0) * 1e3
        return r


def make_acts(n, tokens, k, outliers=True):
    g = torch.Generator(device="cuda").manual_seed(k)
    acts = []
    scale = torch.ones(k, device="cuda")
    if outliers:  # 0.2 % of the channels x20
        idx = torch.randperm(k, device="cuda", generator=g)[: max(1, k // 500)]
        scale[idx] = 20.0
    for _ in range(n):
        acts.append((torch.randn((1, tokens, k), device="cuda", generator=g) * scale).to(torch.bfloat16))
    return acts


def c3(args):
    H, I = 4096, 14336
    shapes = {"q": (H, H), "k": (1024, H), "v": (1024, H), "o": (H, H), "gate": (I, H), "up": (I, H), "down": (H, I)}
    reads = {"q": "a", "k": "a", "v": "a", "o": "b", "gate": "c", "up": "c", "down": "d"}
    cfg = WeightCompressionConfig("int4_asym", 128)
    g = torch.Generator(device="cuda").manual_seed(0)
    W = {n: (torch.randn(s, device="cuda", generator=g) * 0.02).to(torch.bfloat16) for n, s in shapes.items()}
    acts_h = make_acts(args.samples, args.tokens, H)
    acts_i = make_acts(args.samples, args.tokens, I)
    ph = Phases()
    passes = []
    for it in range(4):  # one warm-up pass, then three timed ones: the median per phase is reported
        ph.ms = {}
        stats = {}
        stats = ph.run("statistics (4 activations x %d samples)" % args.samples, lambda: {
            key: S.activations_to_wc_statistics(acts) for key, acts in (("a", acts_h), ("b", acts_h), ("c", acts_h), ("d", acts_i))})
        # AWQ: the MatMul -> Multiply -> MatMul pattern of the gated MLP (target down_proj, merged into up_proj)
        scale = ph.run("awq search (down_proj)", lambda: awq.data_aware_step(
            W["down"].float(), (1,), cfg, stats["d"], prev_weight=W["up"], prev_statistics=stats["c"]))
        wd = ph.run("awq apply", lambda: ((W["down"].float() * scale.unsqueeze(0)).to(torch.bfloat16),
                                          (W["up"].float() / scale.unsqueeze(1)).to(torch.bfloat16)))
        Wl = dict(W)
        Wl["down"], Wl["up"] = wd
        stats["d"] = S.WCTensorStatistic([m / scale.to(m.dtype) for m in stats["d"].mean_values], stats["d"].shape_values)
        # the seven weights of a phase are queued back to back, as the model-level driver does (one device
        # synchronisation per phase, not per weight)
        pre = ph.run("scale estimation (7 weights)", lambda: {n: se.calculate_quantization_params(
            stats[reads[n]], Wl[n], (1,), cfg, subset_size=64, initial_steps=5, scale_steps=5) for n in shapes})
        ph.run("lowering + pack with the estimated scales", lambda: [wl.compress_weight_packed(
            Wl[n], (1,), cfg, CompressedWeight(None, pre[n][0], pre[n][1])) for n in shapes])
        if it > 0:
            passes.append(dict(ph.ms))
    ph.ms = {k: sorted(p[k] for p in passes)[len(passes) // 2] for k in passes[0]}
    total = sum(ph.ms.values())
    act_bytes = args.samples * args.tokens * (3 * H + I) * 2
    return {"config": "c3: Llama-3-8B layer, int4_asym g128, awq + scale_estimation, %d x %d tokens bf16" % (
        args.samples, args.tokens), "phases_ms": {k: round(v, 3) for k, v in ph.ms.items()},
        "timed_passes": len(passes), "layer_ms_per_pass": [round(sum(p.values()), 3) for p in passes],
        "layer_ms": round(total, 3), "model_s_32_layers_x_layer_time": round(total * 32 / 1e3, 3),
        "statistics_GBps": round(act_bytes / ph.ms["statistics (4 activations x %d samples)" % args.samples] / 1e6, 1),
        "weight_GBps_bf16_in": round(sum(a * b for a, b in shapes.values()) * 2 / total / 1e6, 2)}


def c4(args):
    H, I = 8192, 28672
    shapes = {"q": (H, H), "k": (1024, H), "v": (1024, H), "o": (H, H), "gate": (I, H), "up": (I, H), "down": (H, I)}
    groups = [("a", H, ["q", "k", "v"]), ("b", H, ["o"]), ("c", H, ["gate", "up"]), ("d", I, ["down"])]
    cfg = WeightCompressionConfig(args.mode, 64)
    g = torch.Generator(device="cuda").manual_seed(0)
    W = {n: (torch.randn(s, device="cuda", generator=g) * 0.02).to(torch.bfloat16) for n, s in shapes.items()}
    ph = Phases()
    flops = 0.0
    warm = torch.randn(2048, 2048, device="cuda")
    gptq.prepare_hinv(warm.T @ warm / 2048, 0.1)  # library initialisation (rocSOLVER / hipBLASLt) stays untimed
    gptq.quantize_weights(None, torch.randn(256, 2048, device="cuda"), cfg,
                          hinv=gptq.prepare_hinv(warm.T @ warm / 2048, 0.1)[0], dead=torch.zeros(2048, dtype=torch.bool, device="cuda"))
    del warm
    for key, k, names in groups:
        acts = make_acts(args.samples, args.tokens, k, outliers=False)
        gptq.calculate_hessian(acts[:2])  # warm-up (kernel load)
        Hm = ph.run("hessian accumulation (MFMA SYRK)", lambda: gptq.calculate_hessian(acts)[0])
        flops += args.samples * args.tokens * k * k
        del acts
        hinv, dead = ph.run("Hinv: damping + Cholesky-with-inverse", lambda: gptq.prepare_hinv(Hm, 0.1))
        del Hm
        for n in names:
            w32 = W[n].float()
            q, sc, zp = ph.run("column sweeps + trailing GEMMs", lambda: gptq.quantize_weights(
                None, w32, cfg, hinv=hinv, dead=dead))
            ph.run("lowering + pack with the GPTQ scales", lambda: wl.compress_weight_packed(
                q.to(torch.bfloat16), (1,), cfg, CompressedWeight(None, sc, zp)))
        del hinv
        print("c4: unit %s done, phases so far %s" % (key, {k: round(v, 1) for k, v in ph.ms.items()}), file=sys.stderr, flush=True)
    total = sum(ph.ms.values())
    # the same layer through the product's unit pipeline (gptq.quantize_units): Hessian of unit i + 1 on a second
    # stream under unit i's Hinv and sweeps; activations of all four units resident before the clock starts
    layer_wall = {}
    if not args.no_pipeline:
        all_acts = {k: make_acts(args.samples, args.tokens, k, outliers=False) for k in (H, I)}

        def units():
            for key, k, names in groups:
                yield all_acts[k], {n: W[n] for n in names}

        def pack(name, w, q, sc, zp):
            return wl.compress_weight_packed(q.to(torch.bfloat16), (1,), cfg, CompressedWeight(None, sc, zp)).tensor

        for mode, ov, ms in (("sequential", "none", 1), ("unit_streams", "units", 4), ("next_hessian_stream", "hessian", 1),
                             ("sequential_again", "none", 1), ("unit_streams_again", "units", 4)):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            res = gptq.quantize_units(units(), cfg, overlap=ov, on_weight=pack, max_streams=ms)
            torch.cuda.synchronize()
            layer_wall[mode] = round((time.perf_counter() - t0) * 1e3, 2)
            print("c4: %s %.1f ms" % (mode, layer_wall[mode]), file=sys.stderr, flush=True)
            del res
    return {"config": "c4: Llama-3-70B layer, %s g64, gptq, %d x %d tokens bf16" % (args.mode, args.samples, args.tokens),
            "phases_ms": {k: round(v, 3) for k, v in ph.ms.items()}, "layer_ms": round(total, 3),
            "layer_wall_ms_quantize_units": layer_wall,
            "hessian_TFLOPs_real": round(flops / ph.ms["hessian accumulation (MFMA SYRK)"] / 1e9, 1),
            "model_s_80_layers_x_layer_time_1gpu": round(total * 80 / 1e3, 2),
            "model_s_8gpus_layer_sharded": round(total * 10 / 1e3, 2),
            "weight_GBps_bf16_in": round(sum(a * b for a, b in shapes.values()) * 2 / total / 1e6, 2)}


def cpu_baseline(which):
    """CPU leg of the data-aware configurations: the oracle (oracle/wc_oracle.py, a numpy port of the reference's
    chain; numpy's BLAS threads) timed on a REDUCED sample of the layer and extrapolated with the stated operation
    counts -- a reported baseline next to the GPU figure, not a measurement of the full layer."""
    import os

    import numpy as np

    from oracle import wc_oracle as O

    rng = np.random.default_rng(0)
    threads = int(os.environ.get("OMP_NUM_THREADS", "16"))
    if which == "c3":
        H, I, n = 4096, 14336, 128
        rows = 64  # rows of every weight that are timed; scale estimation and the AWQ search are linear in the rows
        means_h = [rng.standard_normal(H).astype(np.float32) for _ in range(n)]
        means_i = [rng.standard_normal(I).astype(np.float32) for _ in range(n)]
        sh_h, sh_i = [(1, 2048, H)] * n, [(1, 2048, I)] * n
        t0 = time.perf_counter()
        O.scale_estimation_params(means_h, sh_h, (rng.standard_normal((rows, H)) * 0.02).astype(np.float32), (1,), "int4_asym", 128,
                                  -1, 64, 5, 5)
        t_se_h = time.perf_counter() - t0
        t0 = time.perf_counter()
        O.scale_estimation_params(means_i, sh_i, (rng.standard_normal((rows, I)) * 0.02).astype(np.float32), (1,), "int4_asym", 128,
                                  -1, 64, 5, 5)
        t_se_i = time.perf_counter() - t0
        t0 = time.perf_counter()
        O.awq_data_aware_scale((rng.standard_normal((rows, I)) * 0.02).astype(np.float32), 1, "int4_asym", 128, means_i, sh_i)
        t_awq = time.perf_counter() - t0
        out_rows_h = 4096 + 1024 + 1024 + 4096 + 14336 + 14336  # rows of the six weights with K = 4096
        se_s = t_se_h * out_rows_h / rows + t_se_i * 4096 / rows
        awq_s = t_awq * 4096 / rows
        return {"kind": "port", "cores": threads, "sample": f"{rows} rows of every weight of the layer, extrapolated linearly in the rows",
                "scale_estimation_s_per_layer": round(se_s, 2), "awq_search_s_per_layer": round(awq_s, 2),
                "layer_s": round(se_s + awq_s, 2)}
    K, N, T = 1024, 256, 4096
    xs = [rng.standard_normal((1, T // 2, K)).astype(np.float32) for _ in range(2)]
    O.gptq_hessian([x[:, :64] for x in xs])  # BLAS thread start-up stays untimed
    t0 = time.perf_counter()
    Hm = O.gptq_hessian(xs)[0]
    t_h = time.perf_counter() - t0
    t0 = time.perf_counter()
    hinv, dead = O.gptq_hinv(Hm.copy(), 0.1)
    t_inv = time.perf_counter() - t0
    t0 = time.perf_counter()
    O.gptq_quantize_weights(None, (rng.standard_normal((N, K)) * 0.02).astype(np.float32), "int4_sym", 64, hinv=hinv, dead=dead)
    t_sw = time.perf_counter() - t0
    # one Llama-3-70B layer: Hessians and Hinv of K = 8192 (x3) and 28672 (x1) over 262144 tokens; sweeps of the 7 weights
    tok = 128 * 2048
    hes = t_h * sum(k * k * tok for k in (8192, 8192, 8192, 28672)) / (K * K * T)
    inv = t_inv * sum(k ** 3 for k in (8192, 8192, 8192, 28672)) / K ** 3
    shapes = [(8192, 8192), (1024, 8192), (1024, 8192), (8192, 8192), (28672, 8192), (28672, 8192), (8192, 28672)]
    sw = t_sw * sum(n_ * k * k for n_, k in shapes) / (N * K * K)
    return {"kind": "port", "cores": threads, "sample": f"K = {K}, N = {N}, T = {T}; Hessian ~ T K^2, Hinv ~ K^3, sweeps ~ N K^2",
            "hessian_s_per_layer": round(hes, 1), "hinv_s_per_layer": round(inv, 1), "sweeps_s_per_layer": round(sw, 1),
            "layer_s": round(hes + inv + sw, 1)}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("which", choices=["c3", "c4"])
    ap.add_argument("--samples", type=int, default=128)
    ap.add_argument("--tokens", type=int, default=2048)
    ap.add_argument("--mode", default="int4_sym")
    ap.add_argument("--no-pipeline", action="store_true", help="c4: skip the quantize_units wall-time legs")
    ap.add_argument("--cpu-baseline", action="store_true", help="add the oracle's CPU time on a reduced sample, extrapolated")
    a = ap.parse_args()
    res = {"c3": c3, "c4": c4}[a.which](a)
    if a.cpu_baseline:
        res["cpu_baseline"] = cpu_baseline(a.which)
    print(json.dumps(res), flush=True)
