#!/usr/bin/env python3
"""Wall time of nncf_amd.compress_weights on a random-init transformers Llama-style model on one MI355X (bf16).

    python tools/bench_model_level.py [--hidden 2048 --inter 5632 --layers 8 --samples 32 --tokens 1024]
"""
import argparse
import json
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import nncf_amd  # noqa: E402
from nncf_amd.parameters import AdvancedCompressionParameters, AdvancedGPTQParameters  # noqa: E402


def build(args):
    import transformers

    cfg = transformers.LlamaConfig(hidden_size=args.hidden, intermediate_size=args.inter, num_hidden_layers=args.layers,
                                   num_attention_heads=args.hidden // 128, num_key_value_heads=max(1, args.hidden // 512),
                                   vocab_size=32000, max_position_embeddings=args.tokens, tie_word_embeddings=False)
    torch.manual_seed(0)
    with torch.device("cuda"):
        model = transformers.LlamaForCausalLM(cfg).to(torch.bfloat16).eval()
    return model


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hidden", type=int, default=2048)
    ap.add_argument("--inter", type=int, default=5632)
    ap.add_argument("--layers", type=int, default=8)
    ap.add_argument("--samples", type=int, default=32)
    ap.add_argument("--tokens", type=int, default=1024)
    ap.add_argument("--only", default="", help="comma-separated substrings of the run names to keep")
    ap.add_argument("--rerun-feeder", action="store_true", help="GPTQ with the re-run feeder (layerwise.py) for comparison")
    args = ap.parse_args()
    if args.rerun_feeder:
        from nncf_amd import layerwise
        layerwise.DEFAULT_SUSPEND = False
    g = torch.Generator().manual_seed(1)
    data = [{"input_ids": torch.randint(0, 32000, (1, args.tokens), generator=g).cuda()} for _ in range(args.samples)]
    out = {"model": vars(args)}
    runs = {
        "int4_sym_g128_data_free": dict(mode="int4_sym", group_size=128),
        "int4_asym_g128_ratio0.8_maxvar": dict(mode="int4_asym", group_size=128, ratio=0.8),
        "int4_asym_g128_awq_se": dict(mode="int4_asym", group_size=128, awq=True, scale_estimation=True),
        "int4_sym_g128_gptq": dict(mode="int4_sym", group_size=128, gptq=True, advanced_parameters=AdvancedCompressionParameters(
            gptq_params=AdvancedGPTQParameters(subset_size=args.samples))),
    }
    keep = [k for k in args.only.split(",") if k]
    for name, kw in runs.items():
        if keep and not any(k in name for k in keep):
            continue
        model = build(args)
        params = sum(p.numel() for p in model.parameters())
        ds = nncf_amd.Dataset(data if len(kw) > 2 else data[:1])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        nncf_amd.compress_weights(model, dataset=ds, subset_size=args.samples, **kw)
        torch.cuda.synchronize()
        out[name] = {"seconds": round(time.perf_counter() - t0, 3)}
        out["model"]["parameters"] = params
        print(name, out[name], flush=True)
        del model
        torch.cuda.empty_cache()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
