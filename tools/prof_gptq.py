#!/usr/bin/env python3
"""rocprofv3 target: one GPTQ layer (4096x4096, int4_sym g128) so that the kernel stats show where the time goes."""
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from nncf_amd import gptq  # noqa: E402
from nncf_amd.parameters import WeightCompressionConfig  # noqa: E402

K = N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
x = torch.randn((1, 8192, K), device="cuda")
H = gptq.calculate_hessian([x])[0]
W = torch.randn((N, K), device="cuda") * 0.02
cfg = WeightCompressionConfig("int4_sym", 128)
for it in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    gptq.quantize_weights(H.clone(), W.clone(), cfg)
    torch.cuda.synchronize()
    print("gptq quantize %dx%d: %.2f ms" % (N, K, (time.perf_counter() - t0) * 1e3))
