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
    for warm in (True, False):
        ph.ms = {}
        stats = {}
        for key, acts in (("a", acts_h), ("b", acts_h), ("c", acts_h), ("d", acts_i)):
            stats[key] = ph.run("statistics (4 activations x %d samples)" % args.samples,
                                lambda: S.activations_to_wc_statistics(acts))
        # AWQ: the MatMul -> Multiply -> MatMul pattern of the gated MLP (target down_proj, merged into up_proj)
        scale = ph.run("awq search (down_proj)", lambda: awq.data_aware_step(
            W["down"].float(), (1,), cfg, stats["d"], prev_weight=W["up"], prev_statistics=stats["c"]))
        wd = ph.run("awq apply", lambda: ((W["down"].float() * scale.unsqueeze(0)).to(torch.bfloat16),
                                          (W["up"].float() / scale.unsqueeze(1)).to(torch.bfloat16)))
        Wl = dict(W)
        Wl["down"], Wl["up"] = wd
        stats["d"] = S.WCTensorStatistic([m / scale.to(m.dtype) for m in stats["d"].mean_values], stats["d"].shape_values)
        pre = {}
        for n in shapes:
            pre[n] = ph.run("scale estimation (7 weights)", lambda: se.calculate_quantization_params(
                stats[reads[n]], Wl[n], (1,), cfg, subset_size=64, initial_steps=5, scale_steps=5))
        for n in shapes:
            ph.run("lowering + pack with the estimated scales", lambda: wl.compress_weight_packed(
                Wl[n], (1,), cfg, CompressedWeight(None, pre[n][0], pre[n][1])))
    total = sum(ph.ms.values())
    act_bytes = args.samples * args.tokens * (3 * H + I) * 2
    return {"config": "c3: Llama-3-8B layer, int4_asym g128, awq + scale_estimation, %d x %d tokens bf16" % (
        args.samples, args.tokens), "phases_ms": {k: round(v, 3) for k, v in ph.ms.items()},
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
    total = sum(ph.ms.values())
    return {"config": "c4: Llama-3-70B layer, %s g64, gptq, %d x %d tokens bf16" % (args.mode, args.samples, args.tokens),
            "phases_ms": {k: round(v, 3) for k, v in ph.ms.items()}, "layer_ms": round(total, 3),
            "hessian_TFLOPs_real": round(flops / ph.ms["hessian accumulation (MFMA SYRK)"] / 1e9, 1),
            "model_s_80_layers_x_layer_time_1gpu": round(total * 80 / 1e3, 2),
            "model_s_8gpus_layer_sharded": round(total * 10 / 1e3, 2),
            "weight_GBps_bf16_in": round(sum(a * b for a, b in shapes.values()) * 2 / total / 1e6, 2)}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("which", choices=["c3", "c4"])
    ap.add_argument("--samples", type=int, default=128)
    ap.add_argument("--tokens", type=int, default=2048)
    ap.add_argument("--mode", default="int4_sym")
    a = ap.parse_args()
    print(json.dumps({"c3": c3, "c4": c4}[a.which](a)), flush=True)
