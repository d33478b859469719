#!/usr/bin/env python3
"""Developer timing of the dense LAPACK-style steps of GPTQ on the GPU (argv: K)."""
import sys, time, torch
K = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
x = torch.randn(2 * K, K, device="cuda")
H = x.T @ x / K + 0.1 * torch.eye(K, device="cuda")
def t(f, n=3):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): r = f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3, r
ms, L = t(lambda: torch.linalg.cholesky(H)); print("cholesky %.2f ms" % ms)
ms, _ = t(lambda: torch.cholesky_inverse(L)); print("cholesky_inverse %.2f ms" % ms)
eye = torch.eye(K, device="cuda")
ms, _ = t(lambda: torch.linalg.solve_triangular(L, eye, upper=False)); print("solve_triangular(L, I) %.2f ms" % ms)
ms, _ = t(lambda: torch.linalg.solve_triangular(L[:128, :128], H[:, :128].T.contiguous(), upper=False)); print("panel trsm 128 x K: %.3f ms" % ms)
ms, _ = t(lambda: H @ H); print("gemm KxKxK %.2f ms (%.1f TF)" % (ms, 2 * K**3 / ms / 1e9))
ms, _ = t(lambda: torch.linalg.inv(L)); print("inv(L) %.2f ms" % ms)
