// wc_cholinv.hip -- inverse Cholesky factor of one small SPD block (n <= 128) in ONE workgroup, for gfx950.
//
// Base case of the GEMM-based Cholesky-with-inverse recursion that produces GPTQ's `hessian_inv`
// (gptq.py:283-285: cholesky -> cholesky_inverse -> cholesky(upper); host side: nncf_amd/gptq.py _cholinv_lower).
// The library chain for a block (rocSOLVER potrf at ~2 us per column + a triangular solve against the identity)
// is latency-bound; here the whole block lives in registers:
//
//   256 threads = 16 x 16 grid (ti, tk); thread (ti, tk) owns A[i][k] and X[i][k] for i = 16 ii + ti,
//   k = 16 kk + tk (8 x 8 each).  Step s (right-looking, one barrier):
//     * the 16 lanes that own column s of A (tk == s % 16: one 16-lane group, so A[s][s] reaches them by one
//       in-group shuffle) write L[:, s] = A[:, s] / sqrt(A[s][s]) into LDS,
//     * the 16 threads that own row s of X (ti == s % 16) write the raw row into LDS,
//     * barrier; everyone reads its 8 + 8 + 8 operands (broadcast reads) and applies the two rank-1 updates
//       A[i][k] -= L[i][s] L[k][s]   and   X[i][k] -= L[i][s] (X[s][k] / L[s][s]);  the scaled row is final.
//   X starts as the identity, so after step n-1 the finished rows are L^-1.  Entries above the diagonal and
//   rows/columns that are already finished carry junk in registers; nothing reads them again.
#include <hip/hip_runtime.h>

#include "../../include/nncf_amd_wc.h"
#include "wc_common.h"

namespace wc {

constexpr int CB = 128;       // largest block
constexpr int CLD = CB + 1;   // LDS row pitch (floats)
constexpr int CR = CB / 16;   // register blocks per dimension

__global__ __launch_bounds__(256) void k_cholinv_block(const float* __restrict__ a_in, int64_t lda,
                                                       float* __restrict__ linv_out, int64_t ldo, int n,
                                                       int* __restrict__ info) {
    extern __shared__ float sm[];
    float* colT = sm;                  // [CB][CLD]: colT[s][i] = L[i][s]; used as the staging image of A first
    float* rows = sm + CB * CLD;       // [CB][CLD]: finished rows of X = L^-1
    float* xraw = rows + CB * CLD;     // [2][CB]: raw row s of X, double-buffered by the parity of s
    float* dinv = xraw + 2 * CB;       // [2]: 1 / L[s][s], same double buffering
    const int t = threadIdx.x, ti = t & 15, tk = t >> 4;

    // A -> LDS with coalesced reads (lower triangle only: the rest of the input is unspecified), identity padding
    for (int idx = t; idx < CB * CB; idx += 256) {
        const int i = idx / CB, k = idx % CB;
        float v = (i == k) ? 1.0f : 0.0f;
        if (i < n && k <= i) v = a_in[(int64_t)i * lda + k];
        colT[i * CLD + k] = v;
    }
    __syncthreads();
    float a[CR][CR], x[CR][CR];
#pragma unroll
    for (int ii = 0; ii < CR; ++ii)
#pragma unroll
        for (int kk = 0; kk < CR; ++kk) {
            const int i = ii * 16 + ti, k = kk * 16 + tk;
            a[ii][kk] = colT[i * CLD + k];
            x[ii][kk] = (i == k) ? 1.0f : 0.0f;
        }
    __syncthreads();

    int bad = 0;
#pragma unroll
    for (int jb = 0; jb < CR; ++jb) {
        // Register blocks that still matter while s is in [16 jb, 16 jb + 16): rows ii >= jb; for A the columns
        // jb <= kk <= ii (lower triangle), for X the columns kk <= jb (row s of X is zero beyond column s).
        for (int jl = 0; jl < 16; ++jl) {
            const int s = jb * 16 + jl;
            if (s >= n) break;  // uniform
            // A[s][s] sits in lane (ti == jl, tk == jl) of wave jl / 4: one readlane with a uniform index in that
            // wave; the other waves read something they do not use
            const float ass = __builtin_bit_cast(
                float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, a[jb][jb]), ((jl & 3) << 4) + jl));
            if (tk == jl) {
                if (!(ass > 0.0f) && ti == jl && bad == 0) bad = s + 1;
                const float d = sqrtf(ass);
                const float rinv = 1.0f / d;
                if (ti == jl) dinv[s & 1] = rinv;
#pragma unroll
                for (int ii = jb; ii < CR; ++ii) {
                    const int i = ii * 16 + ti;
                    colT[s * CLD + i] = i > s ? a[ii][jb] * rinv : (i == s ? d : 0.0f);
                }
            }
            float* xr = xraw + (s & 1) * CB;
            if (ti == jl) {
#pragma unroll
                for (int kk = 0; kk <= jb; ++kk) xr[kk * 16 + tk] = x[jb][kk];
            }
            __syncthreads();
            const float rinv = dinv[s & 1];
            float li[CR], lk[CR], xs[CR];
#pragma unroll
            for (int r = jb; r < CR; ++r) {
                li[r] = colT[s * CLD + r * 16 + ti];
                lk[r] = colT[s * CLD + r * 16 + tk];
            }
#pragma unroll
            for (int kk = 0; kk <= jb; ++kk) xs[kk] = xr[kk * 16 + tk] * rinv;
            if (ti == jl) {
#pragma unroll
                for (int kk = 0; kk <= jb; ++kk) rows[s * CLD + kk * 16 + tk] = xs[kk];
            }
#pragma unroll
            for (int ii = jb; ii < CR; ++ii) {
#pragma unroll
                for (int kk = jb; kk <= ii; ++kk) a[ii][kk] = __builtin_fmaf(-li[ii], lk[kk], a[ii][kk]);
#pragma unroll
                for (int kk = 0; kk <= jb; ++kk) x[ii][kk] = __builtin_fmaf(-li[ii], xs[kk], x[ii][kk]);
            }
        }
    }
    // the first failing pivot of the first failing block wins (later pivots only see the NaNs it left behind)
    __shared__ int first_bad;
    if (t == 0) first_bad = 0x7fffffff;
    __syncthreads();
    if (bad) atomicMin(&first_bad, bad);
    __syncthreads();
    if (t == 0 && first_bad != 0x7fffffff) atomicCAS(info, 0, first_bad);
    for (int idx = t; idx < n * CB; idx += 256) {
        const int i = idx / CB, k = idx % CB;
        if (k <= i) linv_out[(int64_t)i * ldo + k] = rows[i * CLD + k];
    }
}

}  // namespace wc

extern "C" int nncf_amd_cholinv_block(const float* a, int64_t lda, float* linv, int64_t ldo, int64_t n, int32_t* info,
                                      void* stream) {
    using namespace wc;
    if (!a || !linv || !info || n <= 0 || n > CB || lda < n || ldo < n) {
        set_error("cholinv_block: bad argument (n=%lld lda=%lld ldo=%lld; n must be in 1..%d)", (long long)n,
                  (long long)lda, (long long)ldo, CB);
        return NNCF_AMD_ERR_BAD_ARG;
    }
    constexpr size_t lds_bytes = (size_t)(2 * CB * CLD + 2 * CB + 2) * sizeof(float);  // ~130 KiB of dynamic LDS
    static OncePerDevice attr_once;
    if (attr_once.first_on_device() != 0) {
        if (hipFuncSetAttribute((const void*)k_cholinv_block, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds_bytes) != hipSuccess) {
            attr_once.forget_current();
            return launch_status("cholinv_block attribute");
        }
    }
    hipLaunchKernelGGL(k_cholinv_block, dim3(1), dim3(256), lds_bytes, (hipStream_t)stream, a, lda, linv, ldo, (int)n,
                       info);
    return launch_status("cholinv_block");
}
