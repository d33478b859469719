#!/usr/bin/env python3
"""rocprofv3 target: scale estimation of the 7 weights of one Llama-3-8B layer, INT4_ASYM g128, 64-sample subset
(the C3 search phase).  `python tools/prof_se.py [reps]`."""
import sys, time, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from nncf_amd import scale_estimation as se, statistics as S
from nncf_amd.parameters import WeightCompressionConfig
H, I = 4096, 14336
shapes = [(H, H), (1024, H), (1024, H), (H, H), (I, H), (I, H), (H, I)]
g = torch.Generator(device="cuda").manual_seed(0)
stats = {k: S.WCTensorStatistic([torch.randn(k, device="cuda", generator=g).to(torch.bfloat16) for _ in range(128)],
                                [(1, 2048, k)] * 128) for k in (H, I)}
W = [(torch.randn(s, device="cuda", generator=g) * 0.02).to(torch.bfloat16) for s in shapes]
cfg = WeightCompressionConfig("int4_asym", 128)
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for w in W:
        se.calculate_quantization_params(stats[w.shape[1]], w, (1,), cfg, subset_size=64, initial_steps=5, scale_steps=5)
    torch.cuda.synchronize(); print("SE of one layer (7 weights): %.2f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
