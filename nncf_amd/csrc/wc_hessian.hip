// wc_hessian.hip -- GPTQ Hessian accumulation H = beta*H + alpha * X^T X on the gfx950 matrix cores.
//
// Reference: GPTQ._calculate_hessian (gptq.py:237-240) does, per calibration item,
//     H *= n/(n+b);  n += b;  X' = float32(X) * sqrt(2/n);  H += X'^T X'
// with an fp32 matmul.  Here X stays in its own dtype in HBM, is upcast while it is staged into LDS and
// multiplied with the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32: a k-ordered fp32 fma chain, no reduced
// precision), alpha = 2/n is applied once in the epilogue.  X^T X is symmetric: only tile pairs (ti <= tj) are
// computed, the mirror image is written from the same accumulators.
//
// Tiling: block = 128x128 of H, 4 waves in 2x2, each wave 64x64 = 2x2 MFMA tiles (64 accumulator VGPRs).
// K-step (tokens) = 16, double-buffered LDS image [16][128] per operand, k-major so that both MFMA operands
// (A[i][k] = X[t0+k][i0+i], B[k][j] = X[t0+k][j0+j]) are conflict-free ds_read_b32 of 32 consecutive floats.
// The token axis is split over gridDim.y when there are too few tile pairs to fill 256 CUs; partial products
// are combined with fp32 atomic adds (H is pre-scaled by beta in a separate pass), so the low-order bits of H
// can differ from run to run by summation order -- well inside the 1e-5 relative contract.
#include "wc_common.h"

namespace wc {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int HB = 128;  // tile edge
constexpr int HK = 16;   // tokens per step

__global__ __launch_bounds__(256) void k_scale_matrix(float* h, int64_t n, float beta) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) h[i] = beta == 0.0f ? 0.0f : h[i] * beta;
}

// stages X[t0 .. t0+HK) x [c0 .. c0+HB) into registers as floats (zero outside the matrix)
template <int DT, bool VEC>
struct TileLoader {
    static constexpr int E = VEC ? Elem<DT>::kPerVec : 1;
    static constexpr int PER_THREAD = HK * HB / E / 256;  // vectors (or elements) per thread
    float v[PER_THREAD][E];
    __device__ __forceinline__ void load(const void* x, int64_t t0, int64_t t_end, int64_t c0, int64_t K) {
#pragma unroll
        for (int u = 0; u < PER_THREAD; ++u) {
            const int idx = threadIdx.x + u * 256;
            const int row = idx / (HB / E), cv = idx % (HB / E);
            const int64_t t = t0 + row, c = c0 + (int64_t)cv * E;
            if (t < t_end && c < K) {
                if constexpr (VEC) {
                    unpack<DT>(*(const Vec16*)((const char*)x + (t * K + c) * Elem<DT>::kBytes), v[u]);
                } else {
                    v[u][0] = Elem<DT>::load(x, t * K + c);
                }
            } else {
#pragma unroll
                for (int i = 0; i < E; ++i) v[u][i] = 0.0f;
            }
        }
    }
    __device__ __forceinline__ void store(float (*lds)[HB]) const {
#pragma unroll
        for (int u = 0; u < PER_THREAD; ++u) {
            const int idx = threadIdx.x + u * 256;
            const int row = idx / (HB / E), cv = idx % (HB / E);
            if constexpr (E == 4) {
                *(float4*)&lds[row][cv * 4] = make_float4(v[u][0], v[u][1], v[u][2], v[u][3]);
            } else if constexpr (E == 8) {
                *(float4*)&lds[row][cv * 8] = make_float4(v[u][0], v[u][1], v[u][2], v[u][3]);
                *(float4*)&lds[row][cv * 8 + 4] = make_float4(v[u][4], v[u][5], v[u][6], v[u][7]);
            } else {
                lds[row][cv] = v[u][0];
            }
        }
    }
};

template <int DT, bool VEC>
__global__ __launch_bounds__(256) void k_syrk(float* __restrict__ h, const void* __restrict__ x, int64_t T, int64_t K,
                                              float alpha, int ntile, int64_t rows_per_split) {
    __shared__ __attribute__((aligned(16))) float sa[2][HK][HB];
    __shared__ __attribute__((aligned(16))) float sb[2][HK][HB];
    // decode the upper-triangular pair index
    int p = blockIdx.x, ti = 0;
    while (p >= ntile - ti) {
        p -= ntile - ti;
        ++ti;
    }
    const int tj = ti + p;
    const int64_t i0 = (int64_t)ti * HB, j0 = (int64_t)tj * HB;
    const int64_t t_begin = (int64_t)blockIdx.y * rows_per_split;
    const int64_t t_end = t_begin + rows_per_split < T ? t_begin + rows_per_split : T;
    if (t_begin >= t_end) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wi = wave >> 1, wj = wave & 1;  // this wave's 64x64 quadrant
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
    TileLoader<DT, VEC> la, lb;
    la.load(x, t_begin, t_end, i0, K);
    lb.load(x, t_begin, t_end, j0, K);
    la.store(sa[0]);
    lb.store(sb[0]);
    __syncthreads();
    int buf = 0;
    const int kl = lane >> 5, cl = lane & 31;
    for (int64_t t0 = t_begin; t0 < t_end; t0 += HK) {
        const bool more = t0 + HK < t_end;
        if (more) {
            la.load(x, t0 + HK, t_end, i0, K);
            lb.load(x, t0 + HK, t_end, j0, K);
        }
#pragma unroll
        for (int k = 0; k < HK; k += 2) {
            float a[2], b[2];
#pragma unroll
            for (int m = 0; m < 2; ++m) a[m] = sa[buf][k + kl][wi * 64 + m * 32 + cl];
#pragma unroll
            for (int n = 0; n < 2; ++n) b[n] = sb[buf][k + kl][wj * 64 + n * 32 + cl];
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], b[n], acc[m][n], 0, 0, 0);
        }
        if (more) {
            la.store(sa[buf ^ 1]);
            lb.store(sb[buf ^ 1]);
        }
        __syncthreads();
        buf ^= 1;
    }
    // epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t gi = i0 + wi * 64 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * kl;
                const int64_t gj = j0 + wj * 64 + n * 32 + cl;
                if (gi < K && gj < K) {
                    const float val = alpha * acc[m][n][r];
                    atomicAdd(&h[gi * K + gj], val);
                    if (ti != tj) atomicAdd(&h[gj * K + gi], val);
                }
            }
}

}  // namespace wc

extern "C" int nncf_amd_hessian_accumulate(float* h, const void* x, int x_dtype, int64_t tokens, int64_t k, float alpha,
                                           float beta, void* stream) {
    using namespace wc;
    if (!h || k <= 0 || tokens < 0 || (tokens > 0 && !x)) {
        set_error("hessian_accumulate: bad argument (k=%lld tokens=%lld)", (long long)k, (long long)tokens);
        return NNCF_AMD_ERR_BAD_ARG;
    }
    hipStream_t s = (hipStream_t)stream;
    if (beta != 1.0f) {
        const int64_t n = k * k;
        int64_t blocks = (n + 255) / 256;
        if (blocks > 8192) blocks = 8192;
        hipLaunchKernelGGL(k_scale_matrix, dim3((unsigned)blocks), dim3(256), 0, s, h, n, beta);
        int st = launch_status("hessian_scale");
        if (st != NNCF_AMD_OK) return st;
    }
    if (tokens == 0) return NNCF_AMD_OK;
    const int ntile = (int)((k + HB - 1) / HB);
    const int64_t pairs = (int64_t)ntile * (ntile + 1) / 2;
    // split the token axis until ~1024 blocks are in flight; every split is a multiple of the k-step
    int64_t splits = pairs >= 1024 ? 1 : (1024 + pairs - 1) / pairs;
    const int64_t max_splits = (tokens + 4 * HK - 1) / (4 * HK);
    if (splits > max_splits) splits = max_splits;
    if (splits > 65535) splits = 65535;
    int64_t rows_per_split = (tokens + splits - 1) / splits;
    rows_per_split = (rows_per_split + HK - 1) / HK * HK;
    splits = (tokens + rows_per_split - 1) / rows_per_split;
    dim3 grid((unsigned)pairs, (unsigned)splits);
    const int ev = x_dtype == NNCF_AMD_F32 ? 4 : 8;
    const bool vec = (((uintptr_t)x & 15u) == 0) && (k % ev == 0);
#define LAUNCH(DT)                                                                                                      \
    if (vec)                                                                                                            \
        hipLaunchKernelGGL((k_syrk<DT, true>), grid, dim3(256), 0, s, h, x, tokens, k, alpha, ntile, rows_per_split);   \
    else                                                                                                                \
        hipLaunchKernelGGL((k_syrk<DT, false>), grid, dim3(256), 0, s, h, x, tokens, k, alpha, ntile, rows_per_split)
    switch (x_dtype) {
        case NNCF_AMD_F32: LAUNCH(NNCF_AMD_F32); break;
        case NNCF_AMD_F16: LAUNCH(NNCF_AMD_F16); break;
        case NNCF_AMD_BF16: LAUNCH(NNCF_AMD_BF16); break;
        default:
            set_error("hessian_accumulate: unsupported dtype %d", x_dtype);
            return NNCF_AMD_ERR_DTYPE;
    }
#undef LAUNCH
    return launch_status("hessian_syrk");
}
