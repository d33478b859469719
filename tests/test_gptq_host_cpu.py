"""Host-side dense algebra of the GPTQ path that is plain torch (runs on CPU tensors too): the GEMM-based recursive
Cholesky-with-inverse must equal the reference's chain cholesky -> cholesky_inverse -> cholesky(upper)
(src/nncf/quantization/algorithms/weight_compression/gptq.py:293-295).  The base case of the recursion is a HIP
kernel in the product (no CPU path); here it is replaced by a torch stand-in so that the recursion's host logic
(splits, triangular-aware products, Schur complements, flips) is what gets tested.  The kernel itself is checked on
the GPU in tests/test_gpu_data_aware.py."""
import pytest
import torch

from nncf_amd import gptq as G


def _torch_base(A, Linv, info):
    L = torch.linalg.cholesky(torch.tril(A) + torch.tril(A, -1).T)
    Linv.copy_(torch.linalg.solve_triangular(L, torch.eye(A.shape[0], dtype=A.dtype), upper=False))


@pytest.fixture(autouse=True)
def small_bases(monkeypatch):
    monkeypatch.setattr(G, "_cholinv_base", _torch_base)
    monkeypatch.setattr(G, "TRI_SPLIT_MIN", 256)
    yield


@pytest.mark.parametrize("K", [64, 129, 700, 1536])
@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-11), (torch.float32, 2e-4)])
def test_upper_cholesky_of_inverse(K, dtype, tol, small_bases):
    torch.manual_seed(K)
    x = torch.randn(2 * K, K, dtype=torch.float64)
    H = x.T @ x / K + 0.1 * torch.eye(K, dtype=torch.float64)
    ref = torch.linalg.cholesky(torch.cholesky_inverse(torch.linalg.cholesky(H)), upper=True)
    U = G._upper_cholesky_of_inverse(H.to(dtype).clone())
    assert torch.equal(U, torch.triu(U))
    assert float((U.double() - ref).abs().max() / ref.abs().max()) < tol
    # U^T U is the inverse
    err = (U.double().T @ U.double()) @ H - torch.eye(K, dtype=torch.float64)
    assert float(err.abs().max()) < tol * 50


def test_triangular_products(small_bases):
    torch.manual_seed(0)
    n = 640
    T = torch.tril(torch.randn(n, n, dtype=torch.float64))
    X = torch.randn(300, n, dtype=torch.float64)
    Y = torch.randn(n, 200, dtype=torch.float64)
    torch.testing.assert_close(G._mm_dense_lowerT(X, T), X @ T.T)
    torch.testing.assert_close(G._mm_dense_lower(X, T), X @ T)
    out = torch.empty(n, 200, dtype=torch.float64)
    torch.testing.assert_close(G._mm_lower_dense(T, Y, out), T @ Y)
    C = torch.randn(n, n, dtype=torch.float64)
    A = torch.randn(n, 96, dtype=torch.float64)
    want = torch.tril(C - A @ A.T)
    S = G._syrk_lower_sub(C, A)  # in place: C is the recursion's own copy
    assert S.data_ptr() == C.data_ptr()
    torch.testing.assert_close(torch.tril(S), want)
    out2 = torch.empty(n, 200, dtype=torch.float64)
    torch.testing.assert_close(G._mm_lower_dense(T, Y, out2, negate=True), -(T @ Y))


def test_prepare_hinv_dead_columns_and_damping():
    torch.manual_seed(1)
    K = 96
    x = torch.randn(400, K)
    x[:, 7] = 0
    H = (x.T @ x) * (2 / 400)
    Hc = H.clone()
    U, dead = G.prepare_hinv(Hc, 0.1)
    assert dead.nonzero().flatten().tolist() == [7]
    Hd = H.clone()
    Hd[7, 7] = 1
    Hd += 0.1 * torch.mean(torch.diag(Hd)) * torch.eye(K)
    torch.testing.assert_close(Hc, Hd)
    ref = torch.linalg.cholesky(torch.cholesky_inverse(torch.linalg.cholesky(Hd.double())), upper=True)
    assert float((U.double() - ref).abs().max() / ref.abs().max()) < 1e-4
