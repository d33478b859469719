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
#include <stdlib.h>

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

// ---------------------------------------------------------------------------------------------------------------------
// The same block in PANELS of 16 columns: two block-wide barriers per 16 columns instead of one per column (the
// per-column form above spends ~1200 cycles per column, most of them waiting: barrier, LDS round trip, sqrt, divide).
// Same ownership of A and X = L^-1 (thread (ti, tk) holds A[16 ii + ti][16 kk + tk]).  Per panel p:
//   (a) everybody publishes its element of the panel's columns (rows >= 16 p) to LDS;                       [barrier]
//   (b) every wave factors the 16 x 16 diagonal block D = Ld Ld^T by itself (lane = row, the column of the step
//       broadcast by v_readlane: no LDS, no barrier) and inverts Ld (lane = column of Ld^-1, forward substitution
//       with the entries of Ld read by v_readlane), then hands the rows tk and ti of Ld^-1 to its threads through a
//       wave-private LDS image;
//   (c) L of the rows below the panel: L[i][16 p + tk] = sum_k A[i][16 p + k] Ld^-1[tk][k], published to LDS;
//   (d) the panel's rows of X are final: X[16 p + ti][:] = sum_r Ld^-1[ti][r] X~[16 p + r][:] (r over the 16 lanes that
//       share tk: in-group shuffles), published to LDS (they are also the output);                            [barrier]
//   (e) rank-16 updates from LDS: A[i][k] -= sum_j L[i][j] L[k][j] below and right of the panel, X[i][:] -= sum_j
//       L[i][j] X[16 p + j][:] below it.
constexpr int PB = 16;  // panel width
constexpr int PP = 20;  // floats per row of the 16-wide LDS images (16-byte aligned rows, conflict-free pitch)

__device__ __forceinline__ float rl(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}

template <int P>
__device__ __forceinline__ void cholinv_panel(float (&a)[CR][CR], float (&x)[CR][CR], float* rows, float* pn, float* lp,
                                              float* ldw, int& bad, int ti, int tk, int lane, int l16) {
        // (a)
#pragma unroll
        for (int ii = P; ii < CR; ++ii) pn[(ii * 16 + ti) * PP + tk] = a[ii][P];
        __syncthreads();
        // (b) lane l16 holds row l16 of D (the four 16-lane groups of a wave compute the same thing)
        float d[PB], rinvs[PB], xi[PB];
        {
            const float4* src = (const float4*)(pn + (P * 16 + l16) * PP);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float4 v = src[c];
                d[4 * c] = v.x, d[4 * c + 1] = v.y, d[4 * c + 2] = v.z, d[4 * c + 3] = v.w;
            }
        }
#pragma unroll
        for (int j = 0; j < PB; ++j) {
            const float djj = rl(d[j], j);
            if (!(djj > 0.0f) && bad == 0) bad = P * PB + j + 1;
            // 1 / sqrt(djj): hardware estimate (1 ulp) + one Newton step; Ld[j][j] = djj * rinv is then the same
            // product every lane forms for its own entry (rows above the diagonal hold junk nobody reads)
            const float r0 = __builtin_amdgcn_rsqf(djj);
            const float rinv = r0 * __builtin_fmaf(-0.5f * djj * r0, r0, 1.5f);
            rinvs[j] = rinv;
            const float lj = d[j] * rinv;
            d[j] = lj;
#pragma unroll
            for (int k = j + 1; k < PB; ++k) d[k] = __builtin_fmaf(-lj, rl(lj, k), d[k]);
        }
        // Ld^-1, lane l16 = its column: x_i = (delta - sum_{k < i} Ld[i][k] x_k) / Ld[i][i]
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            float acc = (i == l16) ? 1.0f : 0.0f;
#pragma unroll
            for (int k = 0; k < i; ++k) acc = __builtin_fmaf(-rl(d[k], i), xi[k], acc);
            xi[i] = acc * rinvs[i];
        }
        if (lane < 16) {
#pragma unroll
            for (int i = 0; i < PB; ++i) ldw[i * PP + l16] = xi[i];
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        float ldk[PB], ldt[PB];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float4 u = ((const float4*)(ldw + tk * PP))[c], v = ((const float4*)(ldw + ti * PP))[c];
            ldk[4 * c] = u.x, ldk[4 * c + 1] = u.y, ldk[4 * c + 2] = u.z, ldk[4 * c + 3] = u.w;
            ldt[4 * c] = v.x, ldt[4 * c + 1] = v.y, ldt[4 * c + 2] = v.z, ldt[4 * c + 3] = v.w;
        }
        // (c)
#pragma unroll
        for (int ii = P + 1; ii < CR; ++ii) {
            const float4* src = (const float4*)(pn + (ii * 16 + ti) * PP);
            float sacc = 0.0f;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float4 v = src[c];
                sacc = __builtin_fmaf(v.x, ldk[4 * c], sacc);
                sacc = __builtin_fmaf(v.y, ldk[4 * c + 1], sacc);
                sacc = __builtin_fmaf(v.z, ldk[4 * c + 2], sacc);
                sacc = __builtin_fmaf(v.w, ldk[4 * c + 3], sacc);
            }
            lp[(ii * 16 + ti) * PP + tk] = sacc;
        }
        // (d)
#pragma unroll
        for (int kk = 0; kk <= P; ++kk) {
            const float v = x[P][kk];
            float sacc = 0.0f;
#pragma unroll
            for (int r = 0; r < PB; ++r) sacc = __builtin_fmaf(ldt[r], __shfl(v, r, 16), sacc);
            rows[(P * 16 + ti) * CLD + kk * 16 + tk] = sacc;
        }
        __syncthreads();
        // (e) the L rows of this thread's own rows, read once for both updates
        float li[CR][PB];
#pragma unroll
        for (int ii = P + 1; ii < CR; ++ii) {
            const float4* src = (const float4*)(lp + (ii * 16 + ti) * PP);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float4 v = src[c];
                li[ii][4 * c] = v.x, li[ii][4 * c + 1] = v.y, li[ii][4 * c + 2] = v.z, li[ii][4 * c + 3] = v.w;
            }
        }
#pragma unroll
        for (int kk = P + 1; kk < CR; ++kk) {
            float lk[PB];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float4 v = ((const float4*)(lp + (kk * 16 + tk) * PP))[c];
                lk[4 * c] = v.x, lk[4 * c + 1] = v.y, lk[4 * c + 2] = v.z, lk[4 * c + 3] = v.w;
            }
#pragma unroll
            for (int ii = kk; ii < CR; ++ii) {
                float sacc = 0.0f;
#pragma unroll
                for (int j = 0; j < PB; ++j) sacc = __builtin_fmaf(li[ii][j], lk[j], sacc);
                a[ii][kk] -= sacc;
            }
        }
#pragma unroll
        for (int kk = 0; kk <= P; ++kk) {
            float xr[PB];
#pragma unroll
            for (int j = 0; j < PB; ++j) xr[j] = rows[(P * 16 + j) * CLD + kk * 16 + tk];
#pragma unroll
            for (int ii = P + 1; ii < CR; ++ii) {
                float sacc = 0.0f;
#pragma unroll
                for (int j = 0; j < PB; ++j) sacc = __builtin_fmaf(li[ii][j], xr[j], sacc);
                x[ii][kk] -= sacc;
            }
        }
}

__global__ __launch_bounds__(256) void k_cholinv_block16(const float* __restrict__ a_in, int64_t lda,
                                                         float* __restrict__ linv_out, int64_t ldo, int n,
                                                         int* __restrict__ info) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* rows = sm;                  // [CB][CLD]: finished rows of X = L^-1; the staging image of A first
    float* pn = rows + CB * CLD + 3;   // [CB][PP]: the panel's columns of A as they are now   (+3: 16-byte aligned)
    float* lp = pn + CB * PP;          // [CB][PP]: L of the panel
    float* ldi = lp + CB * PP;         // [4][PB][PP]: Ld^-1, one image per wave
    const int t = threadIdx.x, ti = t & 15, tk = t >> 4, lane = t & 63, wave = t >> 6, l16 = lane & 15;
    for (int idx = t; idx < CB * CB; idx += 256) {
        const int i = idx / CB, k = idx % CB;
        float v = (i == k) ? 1.0f : 0.0f;
        if (i < n && k <= i) v = a_in[(int64_t)i * lda + k];
        rows[i * CLD + k] = v;
    }
    __syncthreads();
    float a[CR][CR], x[CR][CR];
#pragma unroll
    for (int ii = 0; ii < CR; ++ii)
#pragma unroll
        for (int kk = 0; kk < CR; ++kk) {
            const int i = ii * 16 + ti, k = kk * 16 + tk;
            a[ii][kk] = rows[i * CLD + k];
            x[ii][kk] = (i == k) ? 1.0f : 0.0f;
        }
    __syncthreads();
    int bad = 0;
    float* ldw = ldi + wave * PB * PP;
    // (the panels written out: every register index is static)
    cholinv_panel<0>(a, x, rows, pn, lp, ldw, bad, ti, tk, lane, l16);
    if (1 * PB < n) cholinv_panel<1>(a, x, rows, pn, lp, ldw, bad, ti, tk, lane, l16);
    if (2 * PB < n) cholinv_panel<2>(a, x, rows, pn, lp, ldw, bad, ti, tk, lane, l16);
    if (3 * PB < n) cholinv_panel<3>(a, x, rows, pn, lp, ldw, bad, ti, tk, lane, l16);
    if (4 * PB < n) cholinv_panel<4>(a, x, rows, pn, lp, ldw, bad, ti, tk, lane, l16);
    if (5 * PB < n) cholinv_panel<5>(a, x, rows, pn, lp, ldw, bad, ti, tk, lane, l16);
    if (6 * PB < n) cholinv_panel<6>(a, x, rows, pn, lp, ldw, bad, ti, tk, lane, l16);
    if (7 * PB < n) cholinv_panel<7>(a, x, rows, pn, lp, ldw, bad, ti, tk, lane, l16);
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
    static const bool per_column = [] { const char* e = getenv("NNCF_AMD_CHOLINV_PER_COLUMN"); return e && e[0] == '1'; }();
    if (per_column) {  // (developer knob: the one-barrier-per-column form)
        hipLaunchKernelGGL(k_cholinv_block, dim3(1), dim3(256), lds_bytes, (hipStream_t)stream, a, lda, linv, ldo, (int)n,
                           info);
        return launch_status("cholinv_block");
    }
    constexpr size_t lds16 = (size_t)(CB * CLD + 3 + 2 * CB * PP + 4 * PB * PP) * sizeof(float);
    static OncePerDevice attr16;
    if (attr16.first_on_device() != 0) {
        if (hipFuncSetAttribute((const void*)k_cholinv_block16, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds16) !=
            hipSuccess) {
            attr16.forget_current();
            return launch_status("cholinv_block16 attribute");
        }
    }
    hipLaunchKernelGGL(k_cholinv_block16, dim3(1), dim3(256), lds16, (hipStream_t)stream, a, lda, linv, ldo, (int)n, info);
    return launch_status("cholinv_block");
}
