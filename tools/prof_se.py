#!/usr/bin/env python3
"""rocprofv3 target: scale estimation on one 14336 x 4096 layer, INT4_ASYM g128."""
import sys, time, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from nncf_amd import scale_estimation as se, statistics as S
from nncf_amd.parameters import WeightCompressionConfig
K, N = 4096, 14336
acts = [torch.randn((1, 2048, K), device="cuda", dtype=torch.bfloat16) for _ in range(16)]
stats = S.activations_to_wc_statistics(acts)
W = torch.randn((N, K), device="cuda") * 0.02
cfg = WeightCompressionConfig("int4_asym", 128)
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    se.calculate_quantization_params(stats, W, (1,), cfg)
    torch.cuda.synchronize(); print("SE %.2f ms" % ((time.perf_counter() - t0) * 1e3))
