#!/usr/bin/env python3
"""Developer timing of nncf_amd.gptq._upper_cholesky_of_inverse (GEMM recursion over nncf_amd_cholinv_block) vs the
library chain.  argv: K"""
import sys, time, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from nncf_amd import _lib, gptq as G
K = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
x = torch.randn(2 * K, K, device="cuda")
H = x.T @ x / K
H += 0.1 * torch.mean(torch.diag(H)) * torch.eye(K, device="cuda")
del x
def t(f, n=3):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): r = f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3, r
ms, L = t(lambda: torch.linalg.cholesky(H[:128, :128])); print("library potrf 128: %.3f ms" % ms)
ms, _ = t(lambda: torch.linalg.solve_triangular(L, torch.eye(128, device="cuda"), upper=False)); print("library trsm(I) 128: %.3f ms" % ms)
out = torch.zeros(128, 128, device="cuda"); info = torch.zeros(1, dtype=torch.int32, device="cuda")
def blk():
    for _ in range(100):
        G._cholinv_base(H[:128, :128], out, info)
ms, _ = t(blk); print("nncf_amd_cholinv_block 128: %.4f ms per block" % (ms / 100), flush=True)
for split in (1024, 512, 2048):
    G.TRI_SPLIT_MIN = split
    ms, U = t(lambda: G._upper_cholesky_of_inverse(H))
    print("K=%d tri_split=%d: %.2f ms" % (K, split, ms), flush=True)
if K <= 8192:
    ref = torch.linalg.cholesky(torch.cholesky_inverse(torch.linalg.cholesky(H.double())), upper=True)
    print("max rel err vs fp64 chain: %.3e" % float((U.double() - ref).abs().max() / ref.abs().max()))
    U3 = torch.linalg.cholesky(torch.cholesky_inverse(torch.linalg.cholesky(H)), upper=True)
    print("library fp32 chain, same measure: %.3e" % float((U3.double() - ref).abs().max() / ref.abs().max()))
