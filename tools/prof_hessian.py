#!/usr/bin/env python3
"""rocprofv3 target: bf16 Hessian accumulation, K x K from T tokens (argv: K T)."""
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from nncf_amd import gptq  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
T = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
x = torch.randn((1, T, K), device="cuda", dtype=torch.bfloat16)
for it in range(int(sys.argv[3]) if len(sys.argv) > 3 else 4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    gptq.calculate_hessian([x])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("hessian K=%d T=%d: %.3f ms  %.1f TFLOP/s (symmetric half), %.1f full-equivalent" % (K, T, dt * 1e3, T * K * K / dt / 1e12, 2 * T * K * K / dt / 1e12))
