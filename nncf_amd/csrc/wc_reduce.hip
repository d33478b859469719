// wc_reduce.hip -- activation-statistics reducers for gfx950: one pass over x[batch, tokens, hidden] produces the
// per-channel token mean (AWQ / scale estimation) and the mixed-precision statistics (mean / max of the token
// variance, mean of the token abs-max, sum of squares for the HAWQ trace).  Column-parallel, 16-byte coalesced
// loads along `hidden`, the token axis split over blocks; partials go through a small workspace and are merged by
// a finalize kernel, so the result is deterministic (no float atomics).
//
// Variance is accumulated on values shifted by the first token of each (batch, channel): with d = x - x0,
// var = (sum d^2 - (sum d)^2 / T) / T.  The shift removes the catastrophic cancellation of the raw-moment form for
// channels with a large mean (activation outlier channels) at one extra subtract per element.
#include "wc_common.h"

namespace wc {

static inline int reduce_nsplit(int64_t batch, int64_t tokens, int64_t hidden) {
    const int64_t col_blocks = (hidden + 1023) / 1024;
    int64_t want = 2048 / (batch * col_blocks > 0 ? batch * col_blocks : 1);
    const int64_t max_split = (tokens + 31) / 32;
    if (want > max_split) want = max_split;
    if (want > 256) want = 256;
    if (want < 1) want = 1;
    return (int)want;
}

// partial layout: ws[((b * nsplit + s) * 3 + q) * hidden + k], q = 0: sum d, 1: sum d^2, 2: max |x|
template <int DT, bool VEC>
__global__ __launch_bounds__(256) void k_act_partials(const void* __restrict__ x, int64_t tokens, int64_t hidden, int nsplit,
                                                      float* __restrict__ ws) {
    constexpr int E = VEC ? Elem<DT>::kPerVec : 1;
    const int64_t col0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * E;
    if (col0 >= hidden) return;
    const int s = blockIdx.y;
    const int64_t b = blockIdx.z;
    const int64_t rows_per = (tokens + nsplit - 1) / nsplit;
    const int64_t t0 = s * rows_per;
    const int64_t t1 = t0 + rows_per < tokens ? t0 + rows_per : tokens;
    const int64_t base = b * tokens * hidden;
    float x0[E], sd[E], sq[E], am[E];
    if constexpr (VEC) {
        unpack<DT>(*(const Vec16*)((const char*)x + (base + col0) * Elem<DT>::kBytes), x0);
    } else {
        x0[0] = Elem<DT>::load(x, base + col0);
    }
#pragma unroll
    for (int i = 0; i < E; ++i) sd[i] = sq[i] = am[i] = 0.0f;
    int64_t t = t0;
    if constexpr (VEC) {
        // 4 rows in flight per thread
        for (; t + 4 <= t1; t += 4) {
            Vec16 r[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                r[u] = *(const Vec16*)((const char*)x + (base + (t + u) * hidden + col0) * Elem<DT>::kBytes);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float f[E];
                unpack<DT>(r[u], f);
#pragma unroll
                for (int i = 0; i < E; ++i) {
                    const float d = f[i] - x0[i];
                    sd[i] += d;
                    sq[i] = fmaf(d, d, sq[i]);
                    am[i] = fmaxf(am[i], fabsf(f[i]));
                }
            }
        }
    }
    for (; t < t1; ++t) {
        float f[E];
        if constexpr (VEC) {
            unpack<DT>(*(const Vec16*)((const char*)x + (base + t * hidden + col0) * Elem<DT>::kBytes), f);
        } else {
            f[0] = Elem<DT>::load(x, base + t * hidden + col0);
        }
#pragma unroll
        for (int i = 0; i < E; ++i) {
            const float d = f[i] - x0[i];
            sd[i] += d;
            sq[i] = fmaf(d, d, sq[i]);
            am[i] = fmaxf(am[i], fabsf(f[i]));
        }
    }
    float* p = ws + ((b * nsplit + s) * 3) * hidden + col0;
#pragma unroll
    for (int i = 0; i < E; ++i) {
        p[i] = sd[i];
        p[hidden + i] = sq[i];
        p[2 * hidden + i] = am[i];
    }
}

__device__ __forceinline__ float round_like(int dt, float v) {
    if (dt == NNCF_AMD_BF16) {
        uint32_t u = __builtin_bit_cast(uint32_t, v);
        if ((u & 0x7f800000u) != 0x7f800000u) u = (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u;
        return __builtin_bit_cast(float, u);
    }
    if (dt == NNCF_AMD_F16) return (float)(_Float16)v;
    return v;
}

// single block: merges the token splits, emits mean[hidden] and the four scalars
template <int DT>
__global__ __launch_bounds__(1024) void k_act_finalize(const void* __restrict__ x, int64_t batch, int64_t tokens,
                                                       int64_t hidden, int nsplit, const float* __restrict__ ws,
                                                       float* mean_out, float* stats_out) {
    __shared__ double red[4][16];
    double sum_var = 0.0, sum_amax = 0.0, sum_sq = 0.0;
    float max_var = -__builtin_inff();
    const float T = (float)tokens;
    for (int64_t k = threadIdx.x; k < hidden; k += 1024) {
        double col_sum = 0.0;
        for (int64_t b = 0; b < batch; ++b) {
            float sd = 0.0f, sq = 0.0f, am = 0.0f;
            for (int s = 0; s < nsplit; ++s) {
                const float* p = ws + ((b * nsplit + s) * 3) * hidden + k;
                sd += p[0];
                sq += p[hidden];
                am = fmaxf(am, p[2 * hidden]);
            }
            const float x0 = Elem<DT>::load(x, b * tokens * hidden + k);
            float var = (sq - sd * sd / T) / T;
            var = var < 0.0f ? 0.0f : var;
            var = round_like(DT, var);  // torch.var returns the input dtype before the outer mean / max
            sum_var += var;
            max_var = fmaxf(max_var, var);
            sum_amax += am;
            sum_sq += (double)sq + 2.0 * (double)x0 * (double)sd + (double)T * (double)x0 * (double)x0;
            col_sum += (double)sd + (double)T * (double)x0;
        }
        if (mean_out) mean_out[k] = (float)(col_sum / ((double)batch * (double)tokens));
    }
    // block reduction of the scalars
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int off = 32; off > 0; off >>= 1) {
        sum_var += __shfl_xor(sum_var, off, 64);
        sum_amax += __shfl_xor(sum_amax, off, 64);
        sum_sq += __shfl_xor(sum_sq, off, 64);
        max_var = fmaxf(max_var, __shfl_xor(max_var, off, 64));
    }
    if (lane == 0) {
        red[0][wave] = sum_var;
        red[1][wave] = sum_amax;
        red[2][wave] = sum_sq;
        red[3][wave] = (double)max_var;
    }
    __syncthreads();
    if (threadIdx.x == 0 && stats_out) {
        double a = 0, c = 0, d = 0, mx = red[3][0];
        for (int w = 0; w < 16; ++w) {
            a += red[0][w];
            c += red[1][w];
            d += red[2][w];
            mx = red[3][w] > mx ? red[3][w] : mx;
        }
        const double n = (double)batch * (double)hidden;
        stats_out[0] = (float)(a / n);
        stats_out[1] = (float)mx;
        stats_out[2] = (float)(c / n);
        stats_out[3] = (float)d;
    }
}

template <int DT>
static int run_reduce(const void* x, int64_t batch, int64_t tokens, int64_t hidden, float* mean_out, float* stats_out,
                      float* ws, hipStream_t s) {
    const int nsplit = reduce_nsplit(batch, tokens, hidden);
    const bool vec = (((uintptr_t)x & 15u) == 0) && (hidden % Elem<DT>::kPerVec == 0);
    if (vec) {
        const int64_t cols_per_block = 256 * Elem<DT>::kPerVec;
        dim3 grid((unsigned)((hidden + cols_per_block - 1) / cols_per_block), (unsigned)nsplit, (unsigned)batch);
        hipLaunchKernelGGL((k_act_partials<DT, true>), grid, dim3(256), 0, s, x, tokens, hidden, nsplit, ws);
    } else {
        dim3 grid((unsigned)((hidden + 255) / 256), (unsigned)nsplit, (unsigned)batch);
        hipLaunchKernelGGL((k_act_partials<DT, false>), grid, dim3(256), 0, s, x, tokens, hidden, nsplit, ws);
    }
    int st = launch_status("act_partials");
    if (st != NNCF_AMD_OK) return st;
    hipLaunchKernelGGL((k_act_finalize<DT>), dim3(1), dim3(1024), 0, s, x, batch, tokens, hidden, nsplit, (const float*)ws,
                       mean_out, stats_out);
    return launch_status("act_finalize");
}

}  // namespace wc

extern "C" size_t nncf_amd_reduce_workspace(int64_t batch, int64_t tokens, int64_t hidden) {
    if (batch <= 0 || tokens <= 0 || hidden <= 0) return 0;
    return (size_t)batch * (size_t)wc::reduce_nsplit(batch, tokens, hidden) * 3u * (size_t)hidden * sizeof(float);
}

extern "C" int nncf_amd_reduce_activation_stats(const void* x, int x_dtype, int64_t batch, int64_t tokens,
                                                int64_t hidden, float* mean_out, float* stats_out, void* workspace,
                                                size_t workspace_bytes, void* stream) {
    using namespace wc;
    if (batch <= 0 || tokens <= 0 || hidden <= 0 || !x) {
        set_error("reduce_activation_stats: empty or NULL input (batch=%lld tokens=%lld hidden=%lld)", (long long)batch,
                  (long long)tokens, (long long)hidden);
        return NNCF_AMD_ERR_BAD_ARG;
    }
    if (batch > 65535) {
        set_error("reduce_activation_stats: batch %lld exceeds the grid limit 65535", (long long)batch);
        return NNCF_AMD_ERR_BAD_ARG;
    }
    const size_t need = nncf_amd_reduce_workspace(batch, tokens, hidden);
    if (!workspace || workspace_bytes < need) {
        set_error("reduce_activation_stats: workspace needs %zu bytes", need);
        return NNCF_AMD_ERR_WORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    switch (x_dtype) {
        case NNCF_AMD_F32: return run_reduce<NNCF_AMD_F32>(x, batch, tokens, hidden, mean_out, stats_out, (float*)workspace, s);
        case NNCF_AMD_F16: return run_reduce<NNCF_AMD_F16>(x, batch, tokens, hidden, mean_out, stats_out, (float*)workspace, s);
        case NNCF_AMD_BF16: return run_reduce<NNCF_AMD_BF16>(x, batch, tokens, hidden, mean_out, stats_out, (float*)workspace, s);
        default:
            set_error("reduce_activation_stats: unsupported dtype %d", x_dtype);
            return NNCF_AMD_ERR_DTYPE;
    }
}
