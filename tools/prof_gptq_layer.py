#!/usr/bin/env python3
"""Phase timing of one GPTQ layer on the GPU (BASELINE config[3] shapes): Hessian accumulation, Hinv (Cholesky chain),
block sweeps + trailing GEMMs.   argv: N K [group_size] [tokens] [mode]"""
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from nncf_amd import _lib, gptq  # noqa: E402
from nncf_amd.parameters import WeightCompressionConfig  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
K = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
GS = int(sys.argv[3]) if len(sys.argv) > 3 else 64
T = int(sys.argv[4]) if len(sys.argv) > 4 else 16384
MODE = sys.argv[5] if len(sys.argv) > 5 else "int4_sym"


def timed(fn, n=1):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, r


xs = [torch.randn((1, 2048, K), device="cuda", dtype=torch.bfloat16) for _ in range(T // 2048)]
ms, H = timed(lambda: gptq.calculate_hessian(xs))
ms, H = timed(lambda: gptq.calculate_hessian(xs))
print(f"hessian K={K} T={T} bf16: {ms:.2f} ms ({T * K * K / ms / 1e9:.0f} TFLOP/s real)", flush=True)
H = H[0]
W = torch.randn((N, K), device="cuda") * 0.02
cfg = WeightCompressionConfig(MODE, GS)
Hd = H.clone()
idx = torch.arange(K, device="cuda")
Hd[idx, idx] += 0.1 * torch.mean(torch.diag(Hd))
for _ in range(2):
    ms, hinv = timed(lambda: gptq._upper_cholesky_of_inverse(Hd))
    print(f"Hinv (flipped cholesky + triangular solve) K={K}: {ms:.2f} ms", flush=True)
for _ in range(2):
    ms, _ = timed(lambda: gptq.quantize_weights(H.clone(), W.clone(), cfg))
    print(f"quantize_weights {N}x{K} g{GS} {MODE}: {ms:.2f} ms total", flush=True)
