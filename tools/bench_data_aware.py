#!/usr/bin/env python3
"""Developer bench of the data-aware rows on one MI355X: reducers -> scale estimation / AWQ search on one
Llama-3-8B-shaped linear (4096 x 4096 and 14336 x 4096), INT4_ASYM g128 (BASELINE config C3)."""
import json
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from nncf_amd import awq, scale_estimation as se, statistics as S  # noqa: E402
from nncf_amd.parameters import WeightCompressionConfig  # noqa: E402


def timeit(fn, reps=3, warm=1):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    out = {}
    cfg = WeightCompressionConfig("int4_asym", 128)
    K = 4096
    acts = [torch.randn((1, 2048, K), device="cuda", dtype=torch.bfloat16) for _ in range(16)]
    t = timeit(lambda: S.activations_to_wc_statistics(acts))
    out["wc_statistics_16x2048x4096_bf16"] = {"ms": t * 1e3, "GBps": 16 * 2048 * K * 2 / t / 1e9}
    stats = S.activations_to_wc_statistics(acts)
    for N in (4096, 14336):
        W = torch.randn((N, K), device="cuda") * 0.02
        t = timeit(lambda: se.calculate_quantization_params(stats, W, (1,), cfg))
        out[f"scale_estimation_{N}x{K}_g128"] = {"ms": t * 1e3}
        t = timeit(lambda: awq.data_aware_step(W, (1,), cfg, stats))
        out[f"awq_data_aware_{N}x{K}_g128"] = {"ms": t * 1e3}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
