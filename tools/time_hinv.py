#!/usr/bin/env python3
"""Developer timing of nncf_amd.gptq._upper_cholesky_of_inverse vs the library chain.  argv: K"""
import sys, time, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from nncf_amd import gptq as G
K = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
x = torch.randn(2 * K, K, device="cuda")
H = x.T @ x / K
H += 0.1 * torch.mean(torch.diag(H)) * torch.eye(K, device="cuda")
del x
def t(f, n=2):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): r = f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3, r
ms, L = t(lambda: torch.linalg.cholesky(H[:1024, :1024])); print("potrf 1024: %.2f ms" % ms)
ms, _ = t(lambda: torch.linalg.solve_triangular(L, torch.eye(1024, device="cuda"), upper=False)); print("trsm(I) 1024: %.2f ms" % ms)
ms, L = t(lambda: torch.linalg.cholesky(H[:512, :512])); print("potrf 512: %.2f ms" % ms)
ms, L = t(lambda: torch.linalg.cholesky(H[:256, :256])); print("potrf 256: %.2f ms" % ms)
for base, split in ((1024, 2048), (512, 2048), (2048, 4096), (1024, 1 << 30), (1024, 1024)):
    G.CHOLINV_BASE, G.TRI_SPLIT_MIN = base, split
    ms, U = t(lambda: G._upper_cholesky_of_inverse(H))
    print("K=%d base=%d tri_split=%d: %.2f ms" % (K, base, split, ms), flush=True)
if K <= 8192:
    ref = torch.linalg.cholesky(torch.cholesky_inverse(torch.linalg.cholesky(H.double())), upper=True)
    print("max rel err vs fp64 chain: %.3e" % float((U.double() - ref).abs().max() / ref.abs().max()))
