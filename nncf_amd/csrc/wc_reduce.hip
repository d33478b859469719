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

// 16-byte streaming (non-temporal) load: activations are read once (+8 % on the token-mean kernel: 4.8 -> 5.2 TB/s at
// K = 4096, 5.65 -> 6.1 TB/s at K = 14336)
__device__ __forceinline__ Vec16 load16_stream(const char* p) {
    typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));
    const u32x4v q = __builtin_nontemporal_load((const u32x4v*)p);
    Vec16 v;
    v.w[0] = q[0];
    v.w[1] = q[1];
    v.w[2] = q[2];
    v.w[3] = q[3];
    return v;
}

static inline int reduce_nsplit(int64_t batch, int64_t tokens, int64_t hidden) {
    const int64_t col_blocks = (hidden + 511) / 512;  // one wave covers 512 bf16 / 256 fp32 channels
    int64_t want = 8192 / (batch * col_blocks > 0 ? batch * col_blocks : 1);
    const int64_t max_split = (tokens + 31) / 32;
    if (want > max_split) want = max_split;
    if (want > 256) want = 256;
    if (want < 1) want = 1;
    return (int)want;
}

// partial layout: ws[((b * nsplit + s) * 3 + q) * hidden + k], q = 0: sum d, 1: sum d^2, 2: max |x|
// SUB waves per block share one token slice (a quarter of its rows each, combined through LDS in wave order): a
// single calibration sample ([1, 2048, 4096] bf16 = 16 MB) is 512 slices x column blocks, two waves per CU, and a
// launch of it is three memory latencies long -- four times the waves, eight rows in flight per lane, same partials.
// SUB = 1 (short inputs) is the single-wave form: per element the rows of a slice are summed in row order.
template <int DT, bool VEC, int SUB>
__global__ __launch_bounds__(64 * SUB) void k_act_partials(const void* __restrict__ x, int64_t tokens, int64_t hidden,
                                                            int nsplit, float* __restrict__ ws) {
    constexpr int E = VEC ? Elem<DT>::kPerVec : 1;
    constexpr int U = SUB > 1 ? 8 : 4;  // rows in flight per lane
    __shared__ float part[SUB > 1 ? SUB - 1 : 1][3][E][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t col0 = ((int64_t)blockIdx.x * 64 + lane) * E;
    const bool live = col0 < hidden;
    const int s = blockIdx.y;
    const int64_t b = blockIdx.z;
    const int64_t rows_per = (tokens + nsplit - 1) / nsplit;
    int64_t t0 = s * rows_per;
    int64_t t1 = t0 + rows_per < tokens ? t0 + rows_per : tokens;
    if constexpr (SUB > 1) {
        const int64_t rows_w = t1 > t0 ? (t1 - t0 + SUB - 1) / SUB : 0;
        t0 += wv * rows_w;
        t1 = t0 + rows_w < t1 ? t0 + rows_w : t1;
    }
    const int64_t base = b * tokens * hidden;
    float x0[E], sd[E], sq[E], am[E];
#pragma unroll
    for (int i = 0; i < E; ++i) x0[i] = sd[i] = sq[i] = am[i] = 0.0f;
    if (live) {
        if constexpr (VEC) {
            unpack<DT>(*(const Vec16*)((const char*)x + (base + col0) * Elem<DT>::kBytes), x0);
        } else {
            x0[0] = Elem<DT>::load(x, base + col0);
        }
        int64_t t = t0;
        if constexpr (VEC) {
            for (; t + U <= t1; t += U) {
                Vec16 r[U];
#pragma unroll
                for (int u = 0; u < U; ++u)
                    r[u] = load16_stream((const char*)x + (base + (t + u) * hidden + col0) * Elem<DT>::kBytes);
#pragma unroll
                for (int u = 0; u < U; ++u) {
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
    }
    if constexpr (SUB > 1) {
        if (wv > 0) {
#pragma unroll
            for (int i = 0; i < E; ++i) {
                part[wv - 1][0][i][lane] = sd[i];
                part[wv - 1][1][i][lane] = sq[i];
                part[wv - 1][2][i][lane] = am[i];
            }
        }
        __syncthreads();
        if (wv > 0) return;
#pragma unroll
        for (int w = 0; w < SUB - 1; ++w) {
#pragma unroll
            for (int i = 0; i < E; ++i) {
                sd[i] += part[w][0][i][lane];
                sq[i] += part[w][1][i][lane];
                am[i] = fmaxf(am[i], part[w][2][i][lane]);
            }
        }
    }
    if (!live) return;
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

// stage 1 of the merge: 256 channels x 4 quarter-ranges of the token splits per block; thread (c, j) folds its
// quarter of the splits of one (batch, channel) in split order, the quarters are added in order through LDS ->
// column sum for the mean, and the block folds its variance / abs-max / square sums into one partial record
// (4 doubles; records == NULL: mean only, the d^2 / abs-max partials are not read).  With mean_direct (one batch
// item, no statistics) the mean is written here, by finalize's expression, and finalize is not launched.
// merge workspace layout (after the partials): colsum[batch][hidden] floats, then records[blocks][4] doubles.
template <int DT>
__global__ __launch_bounds__(1024) void k_act_merge(const void* __restrict__ x, int64_t tokens, int64_t hidden, int nsplit,
                                                    const float* __restrict__ ws, float* __restrict__ colsum,
                                                    double* __restrict__ records, float* __restrict__ mean_direct) {
    __shared__ float part[3][3][256];
    __shared__ double red[4][4];
    const int c = threadIdx.x & 255, j = threadIdx.x >> 8;
    const int64_t k = (int64_t)blockIdx.x * 256 + c;
    const int64_t b = blockIdx.y;
    const float T = (float)tokens;
    const bool want_stats = records != nullptr;
    const int per = (nsplit + 3) / 4;
    const int s0 = j * per, s1 = s0 + per < nsplit ? s0 + per : nsplit;
    float sd = 0.0f, sq = 0.0f, am = 0.0f;
    if (k < hidden) {
        if (want_stats) {
#pragma unroll 4
            for (int s = s0; s < s1; ++s) {
                const float* p = ws + ((b * nsplit + s) * 3) * hidden + k;
                sd += p[0];
                sq += p[hidden];
                am = fmaxf(am, p[2 * hidden]);
            }
        } else {
#pragma unroll 8
            for (int s = s0; s < s1; ++s) sd += ws[((b * nsplit + s) * 3) * hidden + k];
        }
    }
    if (j > 0) {
        part[j - 1][0][c] = sd;
        part[j - 1][1][c] = sq;
        part[j - 1][2][c] = am;
    }
    __syncthreads();
    double s_var = 0.0, s_amax = 0.0, s_sq = 0.0;
    float m_var = -__builtin_inff();
    if (j == 0 && k < hidden) {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            sd += part[q][0][c];
            sq += part[q][1][c];
            am = fmaxf(am, part[q][2][c]);
        }
        const float x0 = Elem<DT>::load(x, b * tokens * hidden + k);
        const float total = sd + T * x0;
        colsum[b * hidden + k] = total;
        if (mean_direct) mean_direct[k] = (float)((0.0 + (double)total) / ((double)1 * (double)tokens));
        if (want_stats) {
            float var = (sq - sd * sd / T) / T;
            var = var < 0.0f ? 0.0f : var;
            var = round_like(DT, var);  // torch.var returns the input dtype before the outer mean / max
            s_var = var;
            m_var = var;
            s_amax = am;
            s_sq = (double)sq + 2.0 * (double)x0 * (double)sd + (double)T * (double)x0 * (double)x0;
        }
    }
    if (!want_stats) return;  // uniform over the block
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (j == 0) {  // the four waves of quarter 0
        for (int off = 32; off > 0; off >>= 1) {
            s_var += __shfl_xor(s_var, off, 64);
            s_amax += __shfl_xor(s_amax, off, 64);
            s_sq += __shfl_xor(s_sq, off, 64);
            m_var = fmaxf(m_var, __shfl_xor(m_var, off, 64));
        }
        if (lane == 0) {
            red[0][wave] = s_var;
            red[1][wave] = s_amax;
            red[2][wave] = s_sq;
            red[3][wave] = (double)m_var;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double* r = records + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 4;
        r[0] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        r[1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        r[2] = red[2][0] + red[2][1] + red[2][2] + red[2][3];
        double mx = red[3][0];
        for (int w = 1; w < 4; ++w) mx = red[3][w] > mx ? red[3][w] : mx;
        r[3] = mx;
    }
}

// stage 2: mean[k] = sum_b colsum[b][k] / (B*T) over a grid of blocks; block 0 also folds the records
__global__ __launch_bounds__(256) void k_act_finalize(int64_t batch, int64_t tokens, int64_t hidden,
                                                      const float* __restrict__ colsum, const double* __restrict__ records,
                                                      int64_t n_records, float* mean_out, float* stats_out) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (mean_out && k < hidden) {
        double acc = 0.0;
        for (int64_t b = 0; b < batch; ++b) acc += (double)colsum[b * hidden + k];
        mean_out[k] = (float)(acc / ((double)batch * (double)tokens));
    }
    if (blockIdx.x == 0 && stats_out) {
        __shared__ double red[4][4];
        double a = 0.0, c = 0.0, d = 0.0, mx = -1.0;
        for (int64_t r = threadIdx.x; r < n_records; r += 256) {
            a += records[r * 4];
            c += records[r * 4 + 1];
            d += records[r * 4 + 2];
            mx = records[r * 4 + 3] > mx ? records[r * 4 + 3] : mx;
        }
        for (int off = 32; off > 0; off >>= 1) {
            a += __shfl_xor(a, off, 64);
            c += __shfl_xor(c, off, 64);
            d += __shfl_xor(d, off, 64);
            const double o = __shfl_xor(mx, off, 64);
            mx = o > mx ? o : mx;
        }
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        if (lane == 0) {
            red[0][wave] = a;
            red[1][wave] = c;
            red[2][wave] = d;
            red[3][wave] = mx;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            const double n = (double)batch * (double)hidden;
            double m = red[3][0];
            for (int w = 1; w < 4; ++w) m = red[3][w] > m ? red[3][w] : m;
            stats_out[0] = (float)((red[0][0] + red[0][1] + red[0][2] + red[0][3]) / n);
            stats_out[1] = (float)m;
            stats_out[2] = (float)((red[1][0] + red[1][1] + red[1][2] + red[1][3]) / n);
            stats_out[3] = (float)(red[2][0] + red[2][1] + red[2][2] + red[2][3]);
        }
    }
}

static inline size_t partial_floats(int64_t batch, int64_t tokens, int64_t hidden) {
    return (size_t)batch * (size_t)reduce_nsplit(batch, tokens, hidden) * 3u * (size_t)hidden;
}
static inline size_t align8(size_t n) { return (n + 7) / 8 * 8; }

template <int DT>
static int run_reduce(const void* x, int64_t batch, int64_t tokens, int64_t hidden, float* mean_out, float* stats_out,
                      float* ws, hipStream_t s) {
    const int nsplit = reduce_nsplit(batch, tokens, hidden);
    const bool vec = (((uintptr_t)x & 15u) == 0) && (hidden % Elem<DT>::kPerVec == 0);
    // four waves per token slice where a slice is long enough to share and the launch would otherwise leave the
    // chip short of waves; short inputs keep the single-wave form (and its summation order)
    const int64_t rows_per = (tokens + nsplit - 1) / nsplit;
    if (vec) {
        const int64_t cols_per_block = 64 * Elem<DT>::kPerVec;
        dim3 grid((unsigned)((hidden + cols_per_block - 1) / cols_per_block), (unsigned)nsplit, (unsigned)batch);
        const bool share = rows_per >= 32 && tokens >= 256 && (int64_t)grid.x * grid.y * grid.z <= 4096;
        if (share)
            hipLaunchKernelGGL((k_act_partials<DT, true, 4>), grid, dim3(256), 0, s, x, tokens, hidden, nsplit, ws);
        else
            hipLaunchKernelGGL((k_act_partials<DT, true, 1>), grid, dim3(64), 0, s, x, tokens, hidden, nsplit, ws);
    } else {
        dim3 grid((unsigned)((hidden + 63) / 64), (unsigned)nsplit, (unsigned)batch);
        hipLaunchKernelGGL((k_act_partials<DT, false, 1>), grid, dim3(64), 0, s, x, tokens, hidden, nsplit, ws);
    }
    int st = launch_status("act_partials");
    if (st != NNCF_AMD_OK) return st;
    float* colsum = ws + align8(partial_floats(batch, tokens, hidden));
    double* records = (double*)(colsum + align8((size_t)batch * (size_t)hidden));
    const unsigned kb = (unsigned)((hidden + 255) / 256);
    const bool direct = batch == 1 && mean_out && !stats_out;  // the mean of one sample leaves the merge kernel
    hipLaunchKernelGGL((k_act_merge<DT>), dim3(kb, (unsigned)batch), dim3(1024), 0, s, x, tokens, hidden, nsplit,
                       (const float*)ws, colsum, stats_out ? records : (double*)nullptr, direct ? mean_out : (float*)nullptr);
    st = launch_status("act_merge");
    if (st != NNCF_AMD_OK || direct) return st;
    hipLaunchKernelGGL(k_act_finalize, dim3(kb), dim3(256), 0, s, batch, tokens, hidden, (const float*)colsum,
                       (const double*)records, (int64_t)kb * batch, mean_out, stats_out);
    return launch_status("act_finalize");
}

// ------------------------------------------------------------------------------------------------ list of samples
// Token mean of MANY separately allocated activation samples x_i[tokens_i, hidden] in one pair of launches
// (WCTensorStatistic collection, gptq.py:381-396 / MeanReducer per calibration item: a launch triple per sample
// caps 16 MB samples at ~0.5 TB/s).  Up to kMaxSamples samples per launch, ragged token counts.
constexpr int kMaxSamples = 64;
struct SampleList {
    const void* x[kMaxSamples];
    int64_t tokens[kMaxSamples];
};

// ws[(z * nsplit + s) * hidden + k] = sum over the s-th token slice of sample z
template <int DT, bool VEC>
__global__ __launch_bounds__(64) void k_mean_partials(SampleList l, int64_t hidden, int nsplit, float* __restrict__ ws) {
    constexpr int E = VEC ? Elem<DT>::kPerVec : 1;
    const int64_t col0 = ((int64_t)blockIdx.x * 64 + threadIdx.x) * E;
    if (col0 >= hidden) return;
    const int s = blockIdx.y, z = blockIdx.z;
    const void* x = l.x[z];
    const int64_t tokens = l.tokens[z];
    const int64_t rows_per = (tokens + nsplit - 1) / nsplit;
    const int64_t t0 = s * rows_per;
    const int64_t t1 = t0 + rows_per < tokens ? t0 + rows_per : tokens;
    float acc[E];
#pragma unroll
    for (int i = 0; i < E; ++i) acc[i] = 0.0f;
    int64_t t = t0;
    if constexpr (VEC) {
        for (; t + 8 <= t1; t += 8) {  // 8 rows in flight per thread
            Vec16 r[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                r[u] = load16_stream((const char*)x + ((t + u) * hidden + col0) * Elem<DT>::kBytes);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                float f[E];
                unpack<DT>(r[u], f);
#pragma unroll
                for (int i = 0; i < E; ++i) acc[i] += f[i];
            }
        }
    }
    for (; t < t1; ++t) {
        float f[E];
        if constexpr (VEC) {
            unpack<DT>(*(const Vec16*)((const char*)x + (t * hidden + col0) * Elem<DT>::kBytes), f);
        } else {
            f[0] = Elem<DT>::load(x, t * hidden + col0);
        }
#pragma unroll
        for (int i = 0; i < E; ++i) acc[i] += f[i];
    }
    float* p = ws + ((int64_t)z * nsplit + s) * hidden + col0;
#pragma unroll
    for (int i = 0; i < E; ++i) p[i] = acc[i];
}

__global__ __launch_bounds__(256) void k_mean_merge(SampleList l, int64_t hidden, int nsplit, const float* __restrict__ ws,
                                                    float* __restrict__ means) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int z = blockIdx.y;
    if (k >= hidden) return;
    double acc = 0.0;
    for (int s = 0; s < nsplit; ++s) acc += (double)ws[((int64_t)z * nsplit + s) * hidden + k];
    means[(int64_t)z * hidden + k] = (float)(acc / (double)l.tokens[z]);
}

static inline int mean_nsplit(int n, int64_t max_tokens, int64_t hidden) {
    const int64_t col_blocks = (hidden + 511) / 512;
    int64_t want = 8192 / ((int64_t)n * col_blocks);
    const int64_t max_split = (max_tokens + 63) / 64;
    if (want > max_split) want = max_split;
    if (want > 128) want = 128;
    if (want < 1) want = 1;
    return (int)want;
}

template <int DT>
static int run_mean_list(const SampleList& l, int n, int64_t max_tokens, int64_t hidden, bool vec, float* means, float* ws,
                         hipStream_t s) {
    const int nsplit = mean_nsplit(n, max_tokens, hidden);
    if (vec) {
        const int64_t cpb = 64 * Elem<DT>::kPerVec;
        hipLaunchKernelGGL((k_mean_partials<DT, true>), dim3((unsigned)((hidden + cpb - 1) / cpb), (unsigned)nsplit, (unsigned)n),
                           dim3(64), 0, s, l, hidden, nsplit, ws);
    } else {
        hipLaunchKernelGGL((k_mean_partials<DT, false>), dim3((unsigned)((hidden + 63) / 64), (unsigned)nsplit, (unsigned)n),
                           dim3(64), 0, s, l, hidden, nsplit, ws);
    }
    int st = launch_status("mean_partials");
    if (st != NNCF_AMD_OK) return st;
    hipLaunchKernelGGL(k_mean_merge, dim3((unsigned)((hidden + 255) / 256), (unsigned)n), dim3(256), 0, s, l, hidden, nsplit,
                       (const float*)ws, means);
    return launch_status("mean_merge");
}

}  // namespace wc

extern "C" size_t nncf_amd_reduce_workspace(int64_t batch, int64_t tokens, int64_t hidden) {
    if (batch <= 0 || tokens <= 0 || hidden <= 0) return 0;
    const size_t kb = (size_t)((hidden + 255) / 256);
    return (wc::align8(wc::partial_floats(batch, tokens, hidden)) + wc::align8((size_t)batch * (size_t)hidden)) * sizeof(float) +
           kb * (size_t)batch * 4 * sizeof(double);
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

extern "C" size_t nncf_amd_token_mean_workspace(int n, int64_t max_tokens, int64_t hidden) {
    if (n <= 0 || max_tokens <= 0 || hidden <= 0) return 0;
    const int per = n < wc::kMaxSamples ? n : wc::kMaxSamples;
    return (size_t)per * (size_t)wc::mean_nsplit(per, max_tokens, hidden) * (size_t)hidden * sizeof(float);
}

extern "C" int nncf_amd_token_mean_batched(int n, const void* const* xs_host, int x_dtype, const int64_t* tokens_host,
                                           int64_t hidden, float* means_out, void* workspace, size_t workspace_bytes,
                                           void* stream) {
    using namespace wc;
    if (n < 0 || hidden <= 0 || (n > 0 && (!xs_host || !tokens_host || !means_out))) {
        set_error("token_mean_batched: bad argument (n=%d hidden=%lld)", n, (long long)hidden);
        return NNCF_AMD_ERR_BAD_ARG;
    }
    if (x_dtype != NNCF_AMD_F32 && x_dtype != NNCF_AMD_F16 && x_dtype != NNCF_AMD_BF16) {
        set_error("token_mean_batched: unsupported dtype %d", x_dtype);
        return NNCF_AMD_ERR_DTYPE;
    }
    int64_t max_tokens = 0;
    for (int i = 0; i < n; ++i) {
        if (!xs_host[i] || tokens_host[i] <= 0) {
            set_error("token_mean_batched: sample %d is NULL or empty", i);
            return NNCF_AMD_ERR_BAD_ARG;
        }
        max_tokens = tokens_host[i] > max_tokens ? tokens_host[i] : max_tokens;
    }
    if (n == 0) return NNCF_AMD_OK;
    const size_t need = nncf_amd_token_mean_workspace(n, max_tokens, hidden);
    if (!workspace || workspace_bytes < need) {
        set_error("token_mean_batched: workspace needs %zu bytes", need);
        return NNCF_AMD_ERR_WORKSPACE;
    }
    const int ev = x_dtype == NNCF_AMD_F32 ? 4 : 8;
    for (int first = 0; first < n; first += kMaxSamples) {
        const int cnt = n - first < kMaxSamples ? n - first : kMaxSamples;
        SampleList l = {};
        bool vec = hidden % ev == 0;
        int64_t mt = 0;
        for (int i = 0; i < cnt; ++i) {
            l.x[i] = xs_host[first + i];
            l.tokens[i] = tokens_host[first + i];
            vec = vec && (((uintptr_t)l.x[i] & 15u) == 0);
            mt = l.tokens[i] > mt ? l.tokens[i] : mt;
        }
        // the workspace was sized for min(n, kMaxSamples) samples of max_tokens tokens: every chunk fits
        (void)mt;
        float* out = means_out + (size_t)first * (size_t)hidden;
        int st;
        switch (x_dtype) {
            case NNCF_AMD_F32: st = run_mean_list<NNCF_AMD_F32>(l, cnt, max_tokens, hidden, vec, out, (float*)workspace, (hipStream_t)stream); break;
            case NNCF_AMD_F16: st = run_mean_list<NNCF_AMD_F16>(l, cnt, max_tokens, hidden, vec, out, (float*)workspace, (hipStream_t)stream); break;
            default: st = run_mean_list<NNCF_AMD_BF16>(l, cnt, max_tokens, hidden, vec, out, (float*)workspace, (hipStream_t)stream); break;
        }
        if (st != NNCF_AMD_OK) return st;
    }
    return NNCF_AMD_OK;
}
