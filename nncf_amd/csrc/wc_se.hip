// wc_se.hip -- scale estimation (scale_estimation.py:160-409) and the AWQ alpha search (awq.py:297-352) as fused
// kernels: ONE launch per candidate scale / per set of outlier groups instead of ~10 elementwise torch kernels, a
// quantize-dequantize pass that writes the weight-sized tensor and a batched matmul that reads it back.  Two kernel
// families behind each entry point: the tiled MFMA kernels (group sizes 32 / 64 / 128, <= 64 samples: the shapes of
// the reference's defaults) and the `_any` kernels (every other group size incl. whole rows, any sample count).
// Quantization is the exact policy arithmetic of the lowering kernels.
#include "wc_float_policy.h"
#include "wc_int_policy.h"

// ====================================================================================================================
// Fused search step of scale estimation (scale_estimation.py:268-360), one launch per candidate:
//     cand  = result_scale (rectification steps) | factor * scale0 (grid steps)
//     near  = sign(scale0) * sum_k |w_k| / (|q(w_k; cand) - zp| + zero_mask_k) * importance_k        (:380-409)
//     dq    = quantize-dequantize(w; near, zp)
//     diff  = mean_j (fp_outs[j] - sum_k dq_k X[k, j])^2  [+ weight_penalty * mean_k (dq_k - w_k)^2]
//     mask  = diff > best;  best = mask*best + (1-mask)*diff;  result = mask*result + (1-mask)*near
// instead of the estimate kernel + a QDQ pass that writes the dequantized weight + a batched matmul that reads it back
// (through a transposing copy) + ~8 elementwise launches.  The weight and the importance are read once per step
// (8 B per element); the group's slice of the activation subset X[g] (GS x <= 64 samples) sits in LDS; the small
// contraction runs on the exact-fp32 MFMA (v_mfma_f32_16x16x4_f32), 16 rows of one group per wave:
//   A operand  lane (r = lane % 16, q = lane / 16) supplies dq[row r][k = q*GS/4 + p] in MFMA number p, i.e. a lane owns
//              one quarter of a row's group (GS/4 consecutive elements) -- the k order of the sum is ours to choose;
//   B operand  lane (c = lane % 16, q) supplies X[g][q*GS/4 + p][t*16 + c] for the sample tiles t = 0..3 with ONE
//              16-byte LDS read (the four tiles of a (k, c) pair are stored next to each other);
//   C          4 registers per 16-sample tile: rows 4q + i, column c.
// Rows arrive coalesced (whole 4*GS-byte row segments per load instruction) and are transposed to the lane-owns-
// quarter-row layout through a wave-private LDS buffer.  A block serves one group column (X slice staged once) and
// strides over the row tiles.  (A 32x32x2 tiling with half rows per lane needs 2x the registers: measured slower.)
namespace wc {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct SeStepArgs {
    const void* w;          // [batch, rows, groups, GS] in the weight's own dtype (fp32 or bf16)
    const uint32_t* zmask;  // [batch, rows, groups, 4]: bit i of word q = "min-max target of element q*GS/4 + i is zero"
    const float* rden;      // [batch, rows, groups] = 1 / (sum of the unmasked s + eps)
    const float* s;         // [batch, groups * GS] per-channel activation magnitude
    const float* x;         // [batch, groups, GS, ns]
    float* fp_outs;         // [batch, groups, rows, ns]: written by kind 0, read by kinds 1 and 2
    const float* scale0;    // [batch, rows, groups] min-max scale
    const float* zp;        // same shape or NULL
    float* best;            // same shape, in/out (out only for kind 0)
    float* result;          // same shape, in/out (out only for kind 0)
    int64_t rows, groups;
    int ns, mode, kind;
    float factor, zero_scale, weight_penalty;
    int64_t batch;
    uint32_t* zmask_out;    // kind 0 of k_se_step16 only, or NULL: the baseline step also writes zmask / rden (it
    float* rden_out;        // quantizes with the min-max parameters anyway: no separate pass over the weight)
};

template <int M>
__device__ __forceinline__ typename IntPolicyT<M>::Par se_par(const IntPolicyT<M>& pol, float sc, float zpf, bool has_zp) {
    typename IntPolicyT<M>::Par p;
    p.scale = sc;
    p.has_zp = has_zp;
    p.zpf = has_zp ? zpf : 0.0f;
    p.zpi = (int)p.zpf;
    pol.finish(p);
    return p;
}
template <bool NF4, int M>
__device__ __forceinline__ typename FloatPolicy<NF4, M>::Par se_par(const FloatPolicy<NF4, M>& pol, float sc, float, bool) {
    return pol.with_reciprocal(sc);
}
// values (the unscaled quantized tensor as float) of N elements sharing one parameter set; in and out may alias
template <int N, int M>
__device__ __forceinline__ void se_values(const IntPolicyT<M>& pol, const typename IntPolicyT<M>::Par& p, const float* in,
                                          float* out) {
    pol.template quant_n<N>(p, in, out);
}
template <int N, bool NF4, int M>
__device__ __forceinline__ void se_values(const FloatPolicy<NF4, M>& pol, const typename FloatPolicy<NF4, M>::Par& p,
                                          const float* in, float* out) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
        int code;
        float v;
        pol.quant(in[i], p, code, v);
        out[i] = v;
    }
}

// 16 rows of one group, coalesced (whole row segments per load instruction, 16 bytes per lane) -> registers of the
// lane that owns the quarter row (lane % 16 = row, lane / 16 = quarter), through the wave's staging buffer (raw
// bytes, row pitch + 16 B), upcast to fp32 on the way out.  The group of row n starts at element (gbase + n * G) * GS.
template <int GS, int DT>
__device__ __forceinline__ void se_load_rows(const void* __restrict__ base, float (&out)[GS / 4], float* stage_f, int64_t n0,
                                             int64_t N, int64_t gbase, int64_t G, int lane) {
    constexpr int EB = Elem<DT>::kBytes, EPV = Elem<DT>::kPerVec;  // bytes per element, elements per 16-byte vector
    constexpr int QUART = GS / 4, PITCH = GS * EB + 16, LPR = GS / EPV, RPI = 64 / LPR, NLD = 16 / RPI;
    static_assert(NLD >= 1 && QUART % EPV == 0, "group size too small for this dtype");
    char* stage = (char*)stage_f;
    const int r = lane & 15, q = lane >> 4;
    const int lr = lane / LPR, lc = (lane % LPR) * 16;  // row inside the instruction, byte offset inside the row
    Vec16 v[NLD];
#pragma unroll
    for (int t = 0; t < NLD; ++t) {
        const int64_t gr = n0 + t * RPI + lr;
        const int64_t grc = gr < N ? gr : N - 1;
        v[t] = *(const Vec16*)((const char*)base + (gbase + grc * G) * GS * EB + lc);
    }
#pragma unroll
    for (int t = 0; t < NLD; ++t) *(Vec16*)(stage + (t * RPI + lr) * PITCH + lc) = v[t];
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const char* src = stage + r * PITCH + q * QUART * EB;
#pragma unroll
    for (int i = 0; i < QUART / EPV; ++i) {
        const Vec16 x = *(const Vec16*)(src + 16 * i);
        unpack<DT>(x, out + EPV * i);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

constexpr int kSeWaves = 12;  // 133 KiB of LDS: one block per CU, three waves per SIMD
constexpr int kSeThreads = kSeWaves * 64;
template <int GS>
constexpr int se_lds_floats() {
    return GS * 64 + kSeWaves * 16 * (GS + 4) + kSeWaves * 16 + GS + kPolicyLds;
}

// NT: 16-sample tiles of the MFMA (1, 2 or 4)
template <typename Policy, int DT, int GS, int NT>
__global__ __launch_bounds__(kSeThreads) void k_se_step(SeStepArgs a) {
    constexpr int QUART = GS / 4;  // elements a lane owns
    constexpr int SP = GS + 4;     // pitch (floats) of a staged row: +16 B
    extern __shared__ __align__(16) float se_smem[];
    float* Xs = se_smem;  // [GS][16 columns][4 tiles]: sample column j of row k at k*64 + (j%16)*4 + j/16
    float* ssh = Xs + GS * 64 + kSeWaves * 16 * SP + kSeWaves * 16;  // per-channel activation magnitudes of the group
    float* lut = ssh + GS;
    Policy::init_shared(lut);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* stage = Xs + GS * 64 + wave * 16 * SP;
    float* dsh = Xs + GS * 64 + kSeWaves * 16 * SP + wave * 16;
    const int64_t g = blockIdx.y, e = blockIdx.z, N = a.rows, G = a.groups;
    const int ns = a.ns;
    {
        const float* xg = a.x + ((e * G + g) * GS) * ns;
        if (ns == 64 && (((uintptr_t)xg) & 15u) == 0) {
            for (int idx = threadIdx.x; idx < GS * 16; idx += kSeThreads) {
                const float4 v = ((const float4*)xg)[idx];
                const int row = idx >> 4, j0 = (idx & 15) * 4;  // columns j0 .. j0+3
                float* dst = Xs + row * 64 + (j0 & 15) * 4 + (j0 >> 4);
                dst[0] = v.x, dst[4] = v.y, dst[8] = v.z, dst[12] = v.w;
            }
        } else {
            for (int idx = threadIdx.x; idx < GS * 64; idx += kSeThreads) {
                const int row = idx >> 6, j = idx & 63;
                Xs[row * 64 + (j & 15) * 4 + (j >> 4)] = j < ns ? xg[row * ns + j] : 0.0f;
            }
        }
    }
    if (threadIdx.x < GS) ssh[threadIdx.x] = a.kind != 0 ? a.s[(e * G + g) * GS + threadIdx.x] : 0.0f;
    __syncthreads();
    GroupArgs ga = {};
    ga.mode = a.mode;
    const Policy pol(ga, lut);
    const int r = lane & 15, q = lane >> 4;
    const bool has_zp = a.zp != nullptr;
    float* fo = a.fp_outs + ((e * G + g) * N) * ns;
    // a block serves ONE group column g (its X slice stays in LDS) and every gridDim.x-th set of four 16-row tiles
    for (int64_t n0 = ((int64_t)blockIdx.x * kSeWaves + wave) * 16; n0 < N; n0 += (int64_t)gridDim.x * kSeWaves * 16) {
        const int64_t row = n0 + r;
        const int64_t rowc = row < N ? row : N - 1;
        const int64_t pidx = (e * N + rowc) * G + g;
        const float sc0 = a.scale0[pidx];
        const float zpf = has_zp ? a.zp[pidx] : 0.0f;
        float wv[QUART];
        se_load_rows<GS, DT>(a.w, wv, stage, n0, N, (e * N) * G + g, G, lane);
        float near = sc0;
        constexpr int C = QUART < 8 ? QUART : 8;
        // kind 0 also produces fp_outs = w @ X (scale_estimation.py:250) for the later steps, on the same MFMA path
        f32x4 facc[NT];
        if (a.kind == 0) {
#pragma unroll
            for (int t = 0; t < NT; ++t) facc[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            const float* xrow0 = Xs + (q * QUART) * 64 + r * 4;
#pragma unroll
            for (int p = 0; p < QUART; ++p) {
                float b[4];
                if constexpr (NT == 4) {
                    const float4 x = *(const float4*)(xrow0 + p * 64);
                    b[0] = x.x, b[1] = x.y, b[2] = x.z, b[3] = x.w;
                } else {
                    const float2 x = *(const float2*)(xrow0 + p * 64);
                    b[0] = x.x, b[1] = x.y;
                }
#pragma unroll
                for (int t = 0; t < NT; ++t) facc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[p], b[t], facc[t], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int64_t orow = n0 + 4 * q + i;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int col = t * 16 + r;
                    if (col < ns && orow < N) fo[orow * ns + col] = facc[t][i];
                }
            }
        }
        if (a.kind != 0) {
            // importance_k = (mask_k ? 0 : s_k) / (sum + eps)  (scale_estimation.py:236-247): the sum over the group of
            // ideal_k * s_k is taken over the unmasked elements and scaled by 1 / (sum + eps) once
            const uint32_t mw = a.zmask[pidx * 4 + q];
            const float rd = a.rden[pidx];
            const float cand = a.kind == 1 ? a.result[pidx] : a.factor * sc0;
            const typename Policy::Par pc = se_par(pol, cand, zpf, has_zp);
            float acc = 0.0f;
#pragma unroll
            for (int c0 = 0; c0 < QUART; c0 += C) {
                float val[C], sch[C];
                se_values<C>(pol, pc, wv + c0, val);
#pragma unroll
                for (int i = 0; i < C; i += 4) *(float4*)(sch + i) = *(const float4*)(ssh + q * QUART + c0 + i);
#pragma unroll
                for (int i = 0; i < C; ++i) {
                    const float target = pol.target(val[i], pc);
                    const float zm = fabsf(target) <= 1e-8f ? a.zero_scale : 0.0f;  // isclose(target, 0): atol 1e-8
                    const float sv = ((mw >> (c0 + i)) & 1u) ? 0.0f : sch[i];
                    acc += fabsf(wv[c0 + i]) / (fabsf(target) + zm) * sv;
                }
            }
            acc += __shfl_xor(acc, 16, 64);
            acc += __shfl_xor(acc, 32, 64);
            near = (acc * rd) * (sc0 / fabsf(sc0));  // scale_sign = scale / |scale|  (scale_estimation.py:264)
        }
        // quantize-dequantize with `near`, in place; optional weight penalty
        const typename Policy::Par pn = se_par(pol, near, zpf, has_zp);
        float pen = 0.0f;
        const bool want_pen = a.weight_penalty > 0.0f;  // uniform
#pragma unroll
        for (int c0 = 0; c0 < QUART; c0 += C) {
            float val[C];
            se_values<C>(pol, pn, wv + c0, val);
#pragma unroll
            for (int i = 0; i < C; ++i) {
                const float dq = pol.dq(0, val[i], pn);
                if (want_pen) {
                    const float d = dq - wv[c0 + i];
                    pen += d * d;
                }
                wv[c0 + i] = dq;
            }
        }
        // q_outs[row][j] = sum_k dq[row][k] X[k][j] on the fp32 MFMA: MFMA number p multiplies the k = {p, QUART+p,
        // 2 QUART+p, 3 QUART+p} slices (lane quarter q supplies k = q*QUART + p of its row / of sample column c)
        f32x4 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        const float* xrow = Xs + (q * QUART) * 64 + r * 4;
#pragma unroll
        for (int p = 0; p < QUART; ++p) {
            float b[4];
            if constexpr (NT == 4) {
                const float4 x = *(const float4*)(xrow + p * 64);
                b[0] = x.x, b[1] = x.y, b[2] = x.z, b[3] = x.w;
            } else {
                const float2 x = *(const float2*)(xrow + p * 64);
                b[0] = x.x, b[1] = x.y;
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[p], b[t], acc[t], 0, 0, 0);
        }
        // C layout: lane (c = lane % 16, q) holds rows 4q + i (i = 0..3) of sample column t*16 + c
        float sq[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t orow = n0 + 4 * q + i;
            float sum = 0.0f;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int col = t * 16 + r;
                float f;
                if (a.kind == 0) f = (col < ns && orow < N) ? facc[t][i] : 0.0f;
                else f = (col < ns && orow < N) ? fo[orow * ns + col] : 0.0f;
                const float d = f - acc[t][i];
                sum += (col < ns) ? d * d : 0.0f;
            }
            sq[i] = lanes_allreduce<16>(sum, OpAdd());
        }
        if (r == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) dsh[4 * q + i] = sq[i];
        }
        if (want_pen) {
            pen += __shfl_xor(pen, 16, 64);
            pen += __shfl_xor(pen, 32, 64);
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (q == 0 && row < N) {
            float diff = dsh[r] / (float)ns;
            if (want_pen) diff = diff + a.weight_penalty * (pen / (float)GS);
            if (a.kind == 0) {
                a.best[pidx] = diff;
                a.result[pidx] = near;
            } else {
                const float b = a.best[pidx], rs = a.result[pidx];
                const float mask = diff > b ? 1.0f : 0.0f;
                a.best[pidx] = mask * b + (1.0f - mask) * diff;
                a.result[pidx] = mask * rs + (1.0f - mask) * near;
            }
        }
        __builtin_amdgcn_wave_barrier();  // dsh and the staging buffer are rewritten by the next tile
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The tiled step for policies whose unscaled values are exact bf16 numbers (integer codes minus the zero point: |t| <=
// 255; e2m1 values).  Same ownership as k_se_step (a wave = 16 rows of one group, lane = row x quarter), but
//   * q_outs = near * (t @ X): the contraction runs on v_mfma_f32_16x16x32_bf16 with t as an exact bf16 operand against
//     a THREE-WAY bf16 split of X (x = h + m + l exactly: 3 x 8 significand bits), i.e. 3 MFMAs of K = 32 instead of
//     24 of K = 4; every product is exact in fp32, the sums are fp32 as before.  fp_outs (step 0) the same way: a bf16
//     weight is one exact operand, an fp32 weight is split like X (9 products);
//   * both quantization passes use the correctly rounded quotient (quant_n_div) -- straight-line packed fp32 math
//     instead of the tie test and its divergent redo; the scale is GIVEN here, so a lane first checks that its own
//     elements are finite and their quotients far from overflow (`se_guard`), else it takes the IEEE divide;
//   * the ideal scale |w| / (|t| + zero_mask) multiplies by a reciprocal from a small LDS table indexed by |t| (t takes
//     <= 256 values; entry 0 holds 1 / zero_scale), within 1 ulp of the reference's quotient, instead of an IEEE
//     divide per element.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma_bf16_k32(const i32x4v& a, const i32x4v& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// the upper halves of two floats as a bf16 pair (element 0 = lo): exact for values with <= 8 significant bits
__device__ __forceinline__ int pack_hi16(float lo, float hi) {
    return (int)__builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, hi), __builtin_bit_cast(uint32_t, lo), 0x07060302u);
}
__device__ __forceinline__ float hi16_of(float x) {
    return __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, x) & 0xffff0000u);
}

template <typename Policy>
struct SeExact16 {
    static constexpr bool value = false;
};
template <int M>
struct SeExact16<IntPolicyT<M>> {
    static constexpr bool value = true;
    static constexpr int kIdx = 1;  // table index = |t| * kIdx
};
template <int M>
struct SeExact16<FloatPolicy<false, M>> {
    static constexpr bool value = true;
    static constexpr int kIdx = 2;  // e2m1 magnitudes are multiples of 0.5
};

template <int M>
__device__ __forceinline__ void se_guard(const IntPolicyT<M>&, typename IntPolicyT<M>::Par& p, float amax) {
    if (!(amax * fabsf(p.r) <= 1.2676506002282294e+30f)) p.fast = false;  // inf / huge quotients: the IEEE divide
}
template <bool NF4, int M>
__device__ __forceinline__ void se_guard(const FloatPolicy<NF4, M>&, typename FloatPolicy<NF4, M>::Par&, float) {}
// IntPolicyT::quant_n_div restated for this kernel's issue order: the five multiply-adds of the correctly rounded
// quotient (see wc_int_policy.h) run stage by stage over the N / 2 element pairs, so consecutive packed instructions
// are independent (a dependent v_pk_fma chain costs a wait state per link), the zero point is always added (+0 for
// symmetric modes: rne and the clip see the same value) and the choice between this path and the IEEE divide is made
// once per WAVE (any lane whose group is "not fast" sends the wave through the divide, which is exact for everyone).
template <int N, int M>
__device__ __forceinline__ void se_values_div(const IntPolicyT<M>& pol, const typename IntPolicyT<M>::Par& p, const float* in,
                                              float* out) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    static_assert(N % 2 == 0, "pairs of elements");
    if (__builtin_amdgcn_ballot_w64(!p.fast) != 0) {
#pragma unroll
        for (int i = 0; i < N; ++i) out[i] = int_quant(in[i], p.scale, p.zpf, p.has_zp, pol.m);
        return;
    }
    const f32x2 r = {p.r, p.r}, ns = {-p.scale, -p.scale};
    const float zf = p.has_zp ? p.zpf : 0.0f;
    const f32x2 z = {zf, zf};
    f32x2 w[N / 2], q[N / 2], res[N / 2];
#pragma unroll
    for (int j = 0; j < N / 2; ++j) w[j] = f32x2{in[2 * j], in[2 * j + 1]};
#pragma unroll
    for (int j = 0; j < N / 2; ++j) q[j] = w[j] * r;
#pragma unroll
    for (int j = 0; j < N / 2; ++j) res[j] = __builtin_elementwise_fma(q[j], ns, w[j]);
#pragma unroll
    for (int j = 0; j < N / 2; ++j) q[j] = __builtin_elementwise_fma(res[j], r, q[j]);
#pragma unroll
    for (int j = 0; j < N / 2; ++j) res[j] = __builtin_elementwise_fma(q[j], ns, w[j]);
#pragma unroll
    for (int j = 0; j < N / 2; ++j) q[j] = __builtin_elementwise_fma(res[j], r, q[j]);
#pragma unroll
    for (int j = 0; j < N / 2; ++j) q[j] = q[j] + z;
#pragma unroll
    for (int j = 0; j < N / 2; ++j) {
        out[2 * j] = __builtin_amdgcn_fmed3f(rintf(q[j].x), pol.m.level_low, pol.m.level_high);
        out[2 * j + 1] = __builtin_amdgcn_fmed3f(rintf(q[j].y), pol.m.level_low, pol.m.level_high);
    }
}
template <int N, bool NF4, int M>
__device__ __forceinline__ void se_values_div(const FloatPolicy<NF4, M>& pol, const typename FloatPolicy<NF4, M>::Par& p,
                                              const float* in, float* out) {
    se_values<N>(pol, p, in, out);
}

// acc[t] += A[m] x (piece p of X, sample tile t) over the 3 pieces and M k-blocks.  The piece loop stays rolled: fully
// unrolled the compiler hoists all 12 * M operand reads (192 registers) in front of the first MFMA.  (Measured and
// dropped: a rolling / double-buffered operand pipeline behind scheduling barriers and an early request of the
// tile's fp_outs -- both cost more in registers than they hide in latency, 108 / 137 us against 100 us per step on a
// 14336 x 4096 weight.)
template <typename SH, int NT>
__device__ __forceinline__ void se_mfma_pipeline(const i32x4v* A, const char* bb, f32x4* acc) {
    constexpr int M = SH::M;
#pragma unroll 1
    for (int p = 0; p < 3; ++p) {
        const char* bp = bb + p * SH::kPiece;
#pragma unroll
        for (int m = 0; m < M; ++m)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = mfma_bf16_k32(A[m], *(const i32x4v*)(bp + (t * M + m) * 1024), acc[t]);
    }
}

template <int DT, int GS>
struct Se16Shape {
        static constexpr int kWaves = GS == 128 ? 12 : 16;  // 170 / 128 registers per lane (16 waves at GS 128: spills, +12 %)
    static constexpr int kThreads = kWaves * 64;
    static constexpr int M = GS / 32;                       // MFMAs (K = 32) per operand piece and sample tile
    static constexpr int kPiece = 4 * M * 1024;             // bytes of one bf16 piece of X (4 sample tiles)
    static constexpr int kPitch = GS * Elem<DT>::kBytes + 16;
    static constexpr int kStage = 3 * kPiece;
    static constexpr int kDsh = kStage + kWaves * 16 * kPitch;
    static constexpr int kSsh = kDsh + kWaves * 16 * 4;
    static constexpr int kRtab = kSsh + GS * 4;
    static constexpr int kLut = kRtab + 256 * 4;
    static constexpr int kBytes = kLut + kPolicyLds * 4;
};

// BASE: step 0 (min-max baseline + fp_outs); the search steps (kinds 1, 2) are the other instantiation
template <typename Policy, int DT, int GS, int NT, bool BASE>
__global__ __launch_bounds__((Se16Shape<DT, GS>::kThreads)) void k_se_step16(SeStepArgs a) {
    using SH = Se16Shape<DT, GS>;
    constexpr int QUART = GS / 4, M = SH::M, THREADS = SH::kThreads;
    constexpr int KIDX = SeExact16<Policy>::kIdx;
    extern __shared__ __align__(16) char se16_smem[];
    char* Bx = se16_smem;  // [piece][tile][m][q][c] x 16 bytes = the 8 bf16 of k = q*QUART + 8m .. +7, sample t*16 + c
    float* ssh = (float*)(se16_smem + SH::kSsh);
    float* rtab = (float*)(se16_smem + SH::kRtab);
    float* lut = (float*)(se16_smem + SH::kLut);
    Policy::init_shared(lut);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* stage = (float*)(se16_smem + SH::kStage + wave * 16 * SH::kPitch);
    float* dsh = (float*)(se16_smem + SH::kDsh) + wave * 16;
    const int64_t e = blockIdx.z, N = a.rows, G = a.groups;
    const int ns = a.ns;
    if (threadIdx.x < 256)
        rtab[threadIdx.x] = threadIdx.x == 0 ? 1.0f / a.zero_scale : (float)KIDX / (float)threadIdx.x;
    GroupArgs ga = {};
    ga.mode = a.mode;
    const Policy pol(ga, lut);
    const int r = lane & 15, q = lane >> 4;
    const bool has_zp = a.zp != nullptr;
    const bool want_pen = a.weight_penalty > 0.0f;  // uniform
    const char* bb = Bx + (q * 16 + r) * 16;
    constexpr int C = QUART < 8 ? QUART : 8;
    // The 16-row tiles of the weight, group column by group column, are cut into gridDim.x equal runs: every block
    // does the same number of tiles whatever the shape (a grid of whole columns left the last round of blocks half
    // empty: 336 blocks on 256 CUs for a 4096 x 14336 weight), and re-stages the X slice when its run crosses into
    // the next column (at most twice for G <= 256).
    const int64_t tpc = (N + 15) / 16, total = G * tpc;
    const int64_t run_lo = total * blockIdx.x / gridDim.x, run_hi = total * (blockIdx.x + 1) / gridDim.x;
    for (int64_t seg = run_lo; seg < run_hi;) {
      const int64_t g = seg / tpc;
      const int64_t seg_end = (g + 1) * tpc < run_hi ? (g + 1) * tpc : run_hi;
      __syncthreads();  // the previous column's readers are done with Bx / ssh
      {
        const float* xg = a.x + ((e * G + g) * GS) * ns;
        for (int idx = threadIdx.x; idx < GS * 16 * NT; idx += THREADS) {
            const int k = idx / (16 * NT), col = idx % (16 * NT);
            const float x = col < ns ? xg[k * ns + col] : 0.0f;
            const float h = hi16_of(x), r1 = x - h, m = hi16_of(r1), l = r1 - m;  // exact: 8 + 8 + 8 significand bits
            const int off = (((((col >> 4) * M + (k % QUART) / 8) * 4 + k / QUART) * 16 + (col & 15)) * 8 + (k & 7)) * 2;
            *(uint16_t*)(Bx + off) = (uint16_t)(__builtin_bit_cast(uint32_t, h) >> 16);
            *(uint16_t*)(Bx + SH::kPiece + off) = (uint16_t)(__builtin_bit_cast(uint32_t, m) >> 16);
            *(uint16_t*)(Bx + 2 * SH::kPiece + off) = (uint16_t)(__builtin_bit_cast(uint32_t, l) >> 16);
        }
        if (threadIdx.x < GS) ssh[threadIdx.x] = (!BASE || a.zmask_out) ? a.s[(e * G + g) * GS + threadIdx.x] : 0.0f;
      }
      __syncthreads();
      float* fo = a.fp_outs + ((e * G + g) * N) * ns;
      for (int64_t tile = seg + wave; tile < seg_end; tile += SH::kWaves) {
        const int64_t n0 = (tile - g * tpc) * 16;
        const int64_t row = n0 + r;
        const int64_t rowc = row < N ? row : N - 1;
        const int64_t pidx = (e * N + rowc) * G + g;
        const float sc0 = a.scale0[pidx];
        const float zsub = has_zp ? a.zp[pidx] : 0.0f;
        float wv[QUART];
        se_load_rows<GS, DT>(a.w, wv, stage, n0, N, (e * N) * G + g, G, lane);
        float amax = 0.0f;
#pragma unroll
        for (int i = 0; i < QUART; i += 2) amax = max3_abs_raw(amax, wv[i], wv[i + 1]);
        float near = sc0;
        if constexpr (!BASE) {
            // importance_k = (mask_k ? 0 : s_k) / (sum + eps)  (scale_estimation.py:236-247)
            const uint32_t keep = ~a.zmask[pidx * 4 + q];
            const float rd = a.rden[pidx];
            typename Policy::Par pc = se_par(pol, a.kind == 1 ? a.result[pidx] : a.factor * sc0, zsub, has_zp);
            se_guard(pol, pc, amax);
            float acc1 = 0.0f;
#pragma unroll
            for (int c0 = 0; c0 < QUART; c0 += C) {
                float val[C], sch[C];
                se_values_div<C>(pol, pc, wv + c0, val);
#pragma unroll
                for (int i = 0; i < C; i += 4) *(float4*)(sch + i) = *(const float4*)(ssh + q * QUART + c0 + i);
#pragma unroll
                for (int i = 0; i < C; ++i) {
                    const float t = fabsf(val[i] - zsub);
                    const float rc = rtab[(int)(KIDX == 1 ? t : t * (float)KIDX)];
                    // bit c0 + i of `keep` spread over the word: the channel's s or 0
                    const int sel = __builtin_amdgcn_sbfe(keep, c0 + i, 1);
                    const float sv = __builtin_bit_cast(float, __builtin_bit_cast(int, sch[i]) & sel);
                    acc1 = fmaf(fabsf(wv[c0 + i]) * rc, sv, acc1);
                }
            }
            acc1 += __shfl_xor(acc1, 16, 64);
            acc1 += __shfl_xor(acc1, 32, 64);
            near = (acc1 * rd) * (sc0 / fabsf(sc0));  // scale_sign = scale / |scale|  (scale_estimation.py:264)
        }
        // the unscaled quantization with `near` as exact bf16 operands (dq = t * near is only formed for the penalty)
        typename Policy::Par pn = se_par(pol, near, zsub, has_zp);
        se_guard(pol, pn, amax);
        i32x4v Aq[M];
        float pen = 0.0f;
        uint32_t zbits = 0;   // BASE with zmask_out: the zero mask of this min-max quantization and the unmasked sum of s
        float ssum = 0.0f;
#pragma unroll
        for (int c0 = 0; c0 < QUART; c0 += C) {
            float val[C];
            se_values_div<C>(pol, pn, wv + c0, val);
#pragma unroll
            for (int i = 0; i < C; ++i) {
                val[i] = val[i] - zsub;
                if (want_pen) {
                    const float d = val[i] * near - wv[c0 + i];
                    pen += d * d;
                }
            }
            if constexpr (BASE) {
                if (a.zmask_out) {  // uniform
                    float sch[C];
#pragma unroll
                    for (int i = 0; i < C; i += 4) *(float4*)(sch + i) = *(const float4*)(ssh + q * QUART + c0 + i);
#pragma unroll
                    for (int i = 0; i < C; ++i) {
                        const bool zero = val[i] == 0.0f;  // isclose(target, 0) for integer / e2m1 targets
                        zbits |= zero ? (1u << (c0 + i)) : 0u;
                        ssum += zero ? 0.0f : sch[i];
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < C; i += 2) Aq[(c0 + i) / 8][((c0 + i) & 7) / 2] = pack_hi16(val[i], val[i + 1]);
        }
        if constexpr (BASE) {
            if (a.zmask_out) {
                ssum += __shfl_xor(ssum, 16, 64);
                ssum += __shfl_xor(ssum, 32, 64);
                if (row < N) {
                    a.zmask_out[pidx * 4 + q] = zbits;
                    if (q == 0) a.rden_out[pidx] = 1.0f / (ssum + kEps);
                }
            }
        }
        f32x4 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        se_mfma_pipeline<SH, NT>(Aq, bb, acc);
        // fp_outs = w @ X (scale_estimation.py:250), written by step 0 for the later ones: the weight as exact bf16 pieces
        f32x4 facc[NT];
        if constexpr (BASE) {
#pragma unroll
            for (int t = 0; t < NT; ++t) facc[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            constexpr int WP = DT == NNCF_AMD_F32 ? 3 : 1;
#pragma unroll 1
            for (int wp = 0; wp < WP; ++wp) {
                i32x4v Aw[M];
#pragma unroll
                for (int i = 0; i < QUART; i += 2) {
                    Aw[i / 8][(i & 7) / 2] = pack_hi16(wv[i], wv[i + 1]);
                    if (WP > 1) wv[i] -= hi16_of(wv[i]), wv[i + 1] -= hi16_of(wv[i + 1]);  // next piece (exact)
                }
                se_mfma_pipeline<SH, NT>(Aw, bb, facc);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int64_t orow = n0 + 4 * q + i;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int col = t * 16 + r;
                    if (col < ns && orow < N) fo[orow * ns + col] = facc[t][i];
                }
            }
        }
        // C layout: lane (c = lane % 16, q) holds rows 4q + i (i = 0..3) of sample column t*16 + c; the rows' `near`
        // sits in lanes 4q + i (quarter 0 of those rows)
        float sq[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t orow = n0 + 4 * q + i;
            const float near_i = __shfl(near, 4 * q + i, 64);
            float sum = 0.0f;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int col = t * 16 + r;
                float f;
                if constexpr (BASE) f = (col < ns && orow < N) ? facc[t][i] : 0.0f;
                else f = (col < ns && orow < N) ? fo[orow * ns + col] : 0.0f;
                const float d = f - near_i * acc[t][i];
                sum += (col < ns) ? d * d : 0.0f;
            }
            sq[i] = lanes_allreduce<16>(sum, OpAdd());
        }
        if (r == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) dsh[4 * q + i] = sq[i];
        }
        if (want_pen) {
            pen += __shfl_xor(pen, 16, 64);
            pen += __shfl_xor(pen, 32, 64);
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (q == 0 && row < N) {
            float diff = dsh[r] / (float)ns;
            if (want_pen) diff = diff + a.weight_penalty * (pen / (float)GS);
            if constexpr (BASE) {
                a.best[pidx] = diff;
                a.result[pidx] = near;
            } else {
                const float b = a.best[pidx], rs = a.result[pidx];
                const float mask = diff > b ? 1.0f : 0.0f;
                a.best[pidx] = mask * b + (1.0f - mask) * diff;
                a.result[pidx] = mask * rs + (1.0f - mask) * near;
            }
        }
        __builtin_amdgcn_wave_barrier();  // dsh and the staging buffer are rewritten by the next tile
      }
      seg = seg_end;
    }
}

// Any group size / any number of samples (the tiled kernel above takes 32 / 64 / 128 x <= 64): a wave owns R = 4 rows
// of one group.  Pass 1, lanes stride over the group's elements: zero mask of the min-max quantization and the sum of
// the unmasked s (the importance of scale_estimation.py:236-247, rebuilt per step: no weight-sized mask is kept),
// candidate quantization, ideal-scale sum -> `near`.  Pass 2, 64 elements at a time: every lane dequantizes one
// element per row with `near`, then the 64 values are broadcast one after the other (v_readlane) against the lane's
// sample column(s) of X[g] (coalesced 256-byte reads, L1 / L2 resident: the group's slice is shared by all rows).
// More than 64 * NT samples: further passes over the group.  Same blends as the tiled kernel.
template <typename Policy, int DT, int NT>
__global__ __launch_bounds__(256) void k_se_step_any(SeStepArgs a, int64_t gs) {
    constexpr int R = 4;
    __shared__ float lut[kPolicyLds];
    Policy::init_shared(lut);
    __syncthreads();
    GroupArgs ga = {};
    ga.mode = a.mode;
    const Policy pol(ga, lut);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t N = a.rows, G = a.groups;
    const int ns = a.ns;
    const int64_t n0 = ((int64_t)blockIdx.x * 4 + wave) * R;
    if (n0 >= N) return;  // (no block barrier below)
    const bool has_zp = a.zp != nullptr;
    const bool want_pen = a.weight_penalty > 0.0f;
    for (int64_t e = blockIdx.z; e < a.batch; e += gridDim.z)
        for (int64_t g = blockIdx.y; g < G; g += gridDim.y) {
            const float* xg = a.x + ((e * G + g) * gs) * ns;
            float* fo = a.fp_outs + ((e * G + g) * N) * ns;
            int64_t pidx[R], wbase[R];
            float sc0[R], zpf[R], near[R];
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const int64_t row = n0 + i < N ? n0 + i : N - 1;
                pidx[i] = (e * N + row) * G + g;
                wbase[i] = pidx[i] * gs;
                sc0[i] = a.scale0[pidx[i]];
                zpf[i] = has_zp ? a.zp[pidx[i]] : 0.0f;
                near[i] = sc0[i];
            }
            if (a.kind != 0) {
                const float* sg = a.s + (e * G + g) * gs;
                typename Policy::Par p0[R], pc[R];
                float den[R], acc[R];
#pragma unroll
                for (int i = 0; i < R; ++i) {
                    p0[i] = se_par(pol, sc0[i], zpf[i], has_zp);
                    pc[i] = se_par(pol, a.kind == 1 ? a.result[pidx[i]] : a.factor * sc0[i], zpf[i], has_zp);
                    den[i] = acc[i] = 0.0f;
                }
                for (int64_t k = lane; k < gs; k += 64) {
                    const float sv = sg[k];
#pragma unroll
                    for (int i = 0; i < R; ++i) {
                        const float w = Elem<DT>::load(a.w, wbase[i] + k);
                        int code;
                        float v0, v;
                        pol.quant(w, p0[i], code, v0);
                        const float imp = fabsf(pol.target(v0, p0[i])) <= 1e-8f ? 0.0f : sv;  // isclose(target, 0): atol 1e-8
                        den[i] += imp;
                        pol.quant(w, pc[i], code, v);
                        const float target = pol.target(v, pc[i]);
                        const float zm = fabsf(target) <= 1e-8f ? a.zero_scale : 0.0f;
                        acc[i] += fabsf(w) / (fabsf(target) + zm) * imp;
                    }
                }
#pragma unroll
                for (int i = 0; i < R; ++i) {
                    const float rd = 1.0f / (lanes_allreduce<64>(den[i], OpAdd()) + kEps);
                    near[i] = (lanes_allreduce<64>(acc[i], OpAdd()) * rd) * (sc0[i] / fabsf(sc0[i]));
                }
            }
            typename Policy::Par pn[R];
            float sq[R], pen[R];
#pragma unroll
            for (int i = 0; i < R; ++i) {
                pn[i] = se_par(pol, near[i], zpf[i], has_zp);
                sq[i] = pen[i] = 0.0f;
            }
            for (int j0 = 0; j0 < ns; j0 += 64 * NT) {
                float qa[R][NT], fa[R][NT];
#pragma unroll
                for (int i = 0; i < R; ++i)
#pragma unroll
                    for (int t = 0; t < NT; ++t) qa[i][t] = fa[i][t] = 0.0f;
                for (int64_t k0 = 0; k0 < gs; k0 += 64) {
                    const int64_t k = k0 + lane;
                    float dqv[R], wv[R];
#pragma unroll
                    for (int i = 0; i < R; ++i) {
                        wv[i] = dqv[i] = 0.0f;
                        if (k < gs) {
                            const float w = Elem<DT>::load(a.w, wbase[i] + k);
                            int code;
                            float v;
                            pol.quant(w, pn[i], code, v);
                            const float dq = pol.dq(0, v, pn[i]);
                            if (want_pen && j0 == 0) pen[i] += (dq - w) * (dq - w);
                            wv[i] = w;
                            dqv[i] = dq;
                        }
                    }
                    const int kn = gs - k0 < 64 ? (int)(gs - k0) : 64;
                    const float* xr = xg + k0 * ns + j0 + lane;
                    for (int kk = 0; kk < kn; ++kk) {
                        float xv[NT];
#pragma unroll
                        for (int t = 0; t < NT; ++t) xv[t] = (j0 + t * 64 + lane) < ns ? xr[(int64_t)kk * ns + t * 64] : 0.0f;
#pragma unroll
                        for (int i = 0; i < R; ++i) {
                            const float d = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, dqv[i]), kk));
#pragma unroll
                            for (int t = 0; t < NT; ++t) qa[i][t] = fmaf(d, xv[t], qa[i][t]);
                            if (a.kind == 0) {
                                const float wl = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wv[i]), kk));
#pragma unroll
                                for (int t = 0; t < NT; ++t) fa[i][t] = fmaf(wl, xv[t], fa[i][t]);
                            }
                        }
                    }
                }
#pragma unroll
                for (int i = 0; i < R; ++i) {
                    const int64_t row = n0 + i;
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const int col = j0 + t * 64 + lane;
                        if (col < ns && row < N) {
                            float f;
                            if (a.kind == 0) {
                                f = fa[i][t];
                                fo[row * ns + col] = f;
                            } else {
                                f = fo[row * ns + col];
                            }
                            const float d = f - qa[i][t];
                            sq[i] += d * d;
                        }
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < R; ++i) {
                float diff = lanes_allreduce<64>(sq[i], OpAdd()) / (float)ns;
                if (want_pen) diff = diff + a.weight_penalty * (lanes_allreduce<64>(pen[i], OpAdd()) / (float)gs);
                if (lane == 0 && n0 + i < N) {
                    if (a.kind == 0) {
                        a.best[pidx[i]] = diff;
                        a.result[pidx[i]] = near[i];
                    } else {
                        const float b = a.best[pidx[i]], rs = a.result[pidx[i]];
                        const float mask = diff > b ? 1.0f : 0.0f;
                        a.best[pidx[i]] = mask * b + (1.0f - mask) * diff;
                        a.result[pidx[i]] = mask * rs + (1.0f - mask) * near[i];
                    }
                }
            }
        }
}

// Zero mask + normaliser of the importance (scale_estimation.py:236-247): importance_k = (target0_k ~ 0 ? 0 : s_k) /
// (sum + eps) with target0 the min-max quantization.  Instead of the importance (4 B per element, re-read by every
// step) the steps get 1 bit per element (word q of a group = its q-th quarter) and 1 / (sum + eps) per group.
// One lane per 16-byte vector of the weight (consecutive lanes read consecutive vectors); gs in {32, 64, 128}.
template <typename Policy, int DT>
__global__ __launch_bounds__(256) void k_se_zero_mask(const void* __restrict__ w, const float* __restrict__ s, GroupArgs a,
                                                      int gs, int64_t groups_per_row, int64_t rows,
                                                      uint32_t* __restrict__ zmask, float* __restrict__ rden) {
    constexpr int E = Elem<DT>::kPerVec, EB = Elem<DT>::kBytes;
    __shared__ float lds[kPolicyLds];
    Policy::init_shared(lds);
    __syncthreads();
    const Policy pol(a, lds);
    const int lpg = gs / E, lpq = lpg >> 2, quart = gs >> 2;  // lanes per group / per quarter (powers of two, lpq >= 1)
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool live = idx / lpg < a.n_groups;
    const int64_t gg = live ? idx / lpg : a.n_groups - 1;  // idle lanes shadow the last group (they take part in the reductions)
    const int l = (int)(idx % lpg), q = l / lpq, bit0 = l * E - q * quart;
    const typename Policy::Par par = pol.load(a, gg);
    const float* sg = s + ((gg / (rows * groups_per_row)) * groups_per_row + gg % groups_per_row) * gs + l * E;
    float f[E], sv[E];
    unpack<DT>(*(const Vec16*)((const char*)w + (gg * gs + (int64_t)l * E) * EB), f);
#pragma unroll
    for (int i = 0; i < E; i += 4) *(float4*)(sv + i) = *(const float4*)(sg + i);
    uint32_t bits = 0;
    float sum = 0.0f;
#pragma unroll
    for (int i = 0; i < E; ++i) {
        int code;
        float val;
        pol.quant(f[i], par, code, val);
        const bool zero = fabsf(pol.target(val, par)) <= 1e-8f;  // isclose(target, 0): atol 1e-8
        bits |= zero ? (1u << (bit0 + i)) : 0u;
        sum += zero ? 0.0f : sv[i];
    }
    for (int m = 1; m < lpq; m <<= 1) bits |= (uint32_t)__shfl_xor((int)bits, m, 64);
    for (int m = 1; m < lpg; m <<= 1) sum += __shfl_xor(sum, m, 64);
    if (!live) return;
    if (l % lpq == 0) zmask[gg * 4 + q] = bits;
    if (l == 0) rden[gg] = 1.0f / (sum + kEps);
}

template <typename Policy, int DT, int GS>
static int se_step_launch(const SeStepArgs& a, int64_t batch, hipStream_t s) {
    // blocks per group column: one resident set (one block per CU), at most one per set of row tiles a block's
    // waves take at a time; the X slice of the group is staged once per block
    auto grid_of = [&](int waves) {
        const int64_t tile_sets = (a.rows + 16 * waves - 1) / (16 * waves);
        int64_t per_col = (256 + a.groups * batch - 1) / (a.groups * batch);
        if (per_col > tile_sets) per_col = tile_sets;
        if (per_col < 1) per_col = 1;
        return dim3((unsigned)per_col, (unsigned)a.groups, (unsigned)batch);
    };
    static OncePerDevice attr_once;
    if constexpr (SeExact16<Policy>::value) {
        using SH = Se16Shape<DT, GS>;
        if (attr_once.first_on_device() != 0) {
            if (hipFuncSetAttribute((const void*)k_se_step16<Policy, DT, GS, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, SH::kBytes) != hipSuccess ||
                hipFuncSetAttribute((const void*)k_se_step16<Policy, DT, GS, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, SH::kBytes) != hipSuccess ||
                hipFuncSetAttribute((const void*)k_se_step16<Policy, DT, GS, 2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, SH::kBytes) != hipSuccess ||
                hipFuncSetAttribute((const void*)k_se_step16<Policy, DT, GS, 4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, SH::kBytes) != hipSuccess) {
                attr_once.forget_current();
                return launch_status("se_step attribute");
            }
        }
        // equal runs of tiles over one resident set of blocks (one block per CU)
        const int64_t total = a.groups * ((a.rows + 15) / 16);
        int64_t nb = 256 / batch > 0 ? 256 / batch : 1;
        if (nb > (total + SH::kWaves - 1) / SH::kWaves) nb = (total + SH::kWaves - 1) / SH::kWaves;
        const dim3 grid((unsigned)nb, 1, (unsigned)batch), block(SH::kThreads);
        if (a.ns <= 32 && a.kind == 0)
            hipLaunchKernelGGL((k_se_step16<Policy, DT, GS, 2, true>), grid, block, SH::kBytes, s, a);
        else if (a.ns <= 32)
            hipLaunchKernelGGL((k_se_step16<Policy, DT, GS, 2, false>), grid, block, SH::kBytes, s, a);
        else if (a.kind == 0)
            hipLaunchKernelGGL((k_se_step16<Policy, DT, GS, 4, true>), grid, block, SH::kBytes, s, a);
        else
            hipLaunchKernelGGL((k_se_step16<Policy, DT, GS, 4, false>), grid, block, SH::kBytes, s, a);
    } else {
        constexpr size_t lds_bytes = (size_t)se_lds_floats<GS>() * sizeof(float);
        if (attr_once.first_on_device() != 0) {
            if (hipFuncSetAttribute((const void*)k_se_step<Policy, DT, GS, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess ||
                hipFuncSetAttribute((const void*)k_se_step<Policy, DT, GS, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess) {
                attr_once.forget_current();
                return launch_status("se_step attribute");
            }
        }
        if (a.ns <= 32)
            hipLaunchKernelGGL((k_se_step<Policy, DT, GS, 2>), grid_of(kSeWaves), dim3(kSeThreads), lds_bytes, s, a);
        else
            hipLaunchKernelGGL((k_se_step<Policy, DT, GS, 4>), grid_of(kSeWaves), dim3(kSeThreads), lds_bytes, s, a);
    }
    return launch_status("se_step");
}

// the `_any` kernel: R = 4 rows per wave, 4 waves per block; groups / batch entries beyond the grid limits are looped over
template <typename Policy, int DT>
static int se_step_any_launch(const SeStepArgs& a, int64_t gs, hipStream_t s) {
    const dim3 grid((unsigned)((a.rows + 15) / 16), (unsigned)(a.groups < 65535 ? a.groups : 65535),
                    (unsigned)(a.batch < 65535 ? a.batch : 65535));
    if (a.ns <= 64)
        hipLaunchKernelGGL((k_se_step_any<Policy, DT, 1>), grid, dim3(256), 0, s, a, gs);
    else if (a.ns <= 128)
        hipLaunchKernelGGL((k_se_step_any<Policy, DT, 2>), grid, dim3(256), 0, s, a, gs);
    else
        hipLaunchKernelGGL((k_se_step_any<Policy, DT, 4>), grid, dim3(256), 0, s, a, gs);
    return launch_status("se_step (any group size)");
}

// shapes of the tiled MFMA kernels: the reference's default group sizes and sample subsets, vector-aligned rows
static bool se_tiled_shape(int64_t gs, int64_t ns, int64_t groups, int64_t batch, const void* w) {
    return (gs == 32 || gs == 64 || gs == 128) && ns >= 1 && ns <= 64 && groups <= 65535 && batch <= 65535 &&
           ((uintptr_t)w & 15u) == 0;
}

template <typename Policy, int DT>
static int se_step_gs(const SeStepArgs& a, int64_t batch, int64_t gs, hipStream_t s) {
    if (!se_tiled_shape(gs, a.ns, a.groups, batch, a.w)) return se_step_any_launch<Policy, DT>(a, gs, s);
    switch (gs) {
        case 32: return se_step_launch<Policy, DT, 32>(a, batch, s);
        case 64: return se_step_launch<Policy, DT, 64>(a, batch, s);
        default: return se_step_launch<Policy, DT, 128>(a, batch, s);
    }
}

template <typename Policy>
static int se_step_dt(const SeStepArgs& a, int w_dtype, int64_t batch, int64_t gs, hipStream_t s) {
    if (w_dtype == NNCF_AMD_F32) return se_step_gs<Policy, NNCF_AMD_F32>(a, batch, gs, s);
    if (w_dtype == NNCF_AMD_BF16) return se_step_gs<Policy, NNCF_AMD_BF16>(a, batch, gs, s);
    set_error("se_step: weight dtype %d (fp32 and bf16 are taken; upcast fp16 first)", w_dtype);
    return NNCF_AMD_ERR_DTYPE;
}

}  // namespace wc

// 1: the shape runs on the tiled MFMA kernels (and nncf_amd_se_zero_mask fills zero_mask / rden for it); 0: on the
// `_any` kernels, which rebuild the mask themselves.  Every shape is served; this is information, not a gate.
extern "C" int nncf_amd_se_step_supported(int64_t group_size, int64_t n_samples) {
    return (group_size == 32 || group_size == 64 || group_size == 128) && n_samples >= 1 && n_samples <= 64;
}

static int se_step_impl(const void* weight, int w_dtype, const uint32_t* zero_mask, const float* rden,
                        const float* s_per_channel, const float* x, float* fp_outs, const float* scale0,
                        const float* zero_point, float* best_diffs, float* result_scale, int64_t batch, int64_t rows,
                        int64_t groups, int64_t group_size, int64_t n_samples, int mode, int kind, float factor,
                        float zero_scale, float weight_penalty, void* stream, uint32_t* zmask_out, float* rden_out) {
    using namespace wc;
    if (batch < 0 || rows < 0 || groups < 0 || kind < 0 || kind > 2 || group_size < 1 || n_samples < 1 ||
        n_samples > 0x7fffffff) {
        set_error("se_step: bad argument");
        return NNCF_AMD_ERR_BAD_ARG;
    }
    if (batch * rows * groups == 0) return NNCF_AMD_OK;
    const bool tiled = se_tiled_shape(group_size, n_samples, groups, batch, weight);
    if (!weight || !x || !fp_outs || !scale0 || !best_diffs || !result_scale ||
        (kind != 0 && (!s_per_channel || (tiled && (!zero_mask || !rden))))) {
        set_error("se_step: NULL buffer");
        return NNCF_AMD_ERR_BAD_ARG;
    }
    SeStepArgs a = {weight, zero_mask, rden, s_per_channel, x, fp_outs, scale0, zero_point, best_diffs, result_scale,
                    rows, groups, (int)n_samples, mode, kind, factor, zero_scale, weight_penalty, batch, zmask_out, rden_out};
    hipStream_t s = (hipStream_t)stream;
    if (mode >= NNCF_AMD_INT8_ASYM && mode <= NNCF_AMD_INT4_SYM) {
        const bool asym = mode == NNCF_AMD_INT8_ASYM || mode == NNCF_AMD_INT4_ASYM;
        if (asym && !zero_point) {
            set_error("se_step: asymmetric mode needs the zero points");
            return NNCF_AMD_ERR_BAD_ARG;
        }
        if (!asym) a.zp = nullptr;
        return se_step_dt<IntPolicy>(a, w_dtype, batch, group_size, s);
    }
    a.zp = nullptr;
    if (mode == NNCF_AMD_NF4) return se_step_dt<FloatPolicy<true>>(a, w_dtype, batch, group_size, s);
    if (mode == NNCF_AMD_MXFP4 || mode == NNCF_AMD_FP4) return se_step_dt<FloatPolicy<false>>(a, w_dtype, batch, group_size, s);
    set_error("se_step: unsupported mode %d", mode);
    return NNCF_AMD_ERR_MODE;
}

extern "C" int nncf_amd_se_step(const void* weight, int w_dtype, const uint32_t* zero_mask, const float* rden,
                                const float* s_per_channel, const float* x, float* fp_outs, const float* scale0,
                                const float* zero_point, float* best_diffs, float* result_scale, int64_t batch,
                                int64_t rows, int64_t groups, int64_t group_size, int64_t n_samples, int mode, int kind,
                                float factor, float zero_scale, float weight_penalty, void* stream) {
    return se_step_impl(weight, w_dtype, zero_mask, rden, s_per_channel, x, fp_outs, scale0, zero_point, best_diffs,
                        result_scale, batch, rows, groups, group_size, n_samples, mode, kind, factor, zero_scale,
                        weight_penalty, stream, nullptr, nullptr);
}

template <typename Policy>
static int se_zero_mask_dt(const void* w, int w_dtype, const float* s, const wc::GroupArgs& a, int gs, int64_t groups,
                           int64_t rows, uint32_t* zmask, float* rden, hipStream_t st) {
    using namespace wc;
    const int64_t lanes = a.n_groups * (gs / (w_dtype == NNCF_AMD_F32 ? 4 : 8));
    const dim3 grid((unsigned)((lanes + 255) / 256));
    if (w_dtype == NNCF_AMD_F32)
        hipLaunchKernelGGL((k_se_zero_mask<Policy, NNCF_AMD_F32>), grid, dim3(256), 0, st, w, s, a, gs, groups, rows, zmask, rden);
    else if (w_dtype == NNCF_AMD_BF16)
        hipLaunchKernelGGL((k_se_zero_mask<Policy, NNCF_AMD_BF16>), grid, dim3(256), 0, st, w, s, a, gs, groups, rows, zmask, rden);
    else if (w_dtype == NNCF_AMD_F16)
        hipLaunchKernelGGL((k_se_zero_mask<Policy, NNCF_AMD_F16>), grid, dim3(256), 0, st, w, s, a, gs, groups, rows, zmask, rden);
    else {
        set_error("se_zero_mask: unsupported weight dtype %d", w_dtype);
        return NNCF_AMD_ERR_DTYPE;
    }
    return launch_status("se_zero_mask");
}

extern "C" int nncf_amd_se_zero_mask(const void* weight, int w_dtype, const float* s_per_channel, const float* scale0,
                                     const float* zero_point, int64_t batch, int64_t rows, int64_t groups,
                                     int64_t group_size, int mode, uint32_t* zero_mask, float* rden, void* stream) {
    using namespace wc;
    if (batch < 0 || rows < 0 || groups < 0 || group_size < 1) {
        set_error("se_zero_mask: bad argument");
        return NNCF_AMD_ERR_BAD_ARG;
    }
    const int64_t n_groups = batch * rows * groups;
    if (n_groups == 0) return NNCF_AMD_OK;
    // shapes the `_any` step kernel serves need no mask: it rebuilds the importance itself (same rule as se_step's)
    if (!se_tiled_shape(group_size, 1, groups, batch, weight)) return NNCF_AMD_OK;
    if ((uintptr_t)s_per_channel & 15u) {
        set_error("se_zero_mask: s_per_channel must be 16-byte aligned");
        return NNCF_AMD_ERR_BAD_ARG;
    }
    if (!weight || !s_per_channel || !scale0 || !zero_mask || !rden) {
        set_error("se_zero_mask: NULL buffer");
        return NNCF_AMD_ERR_BAD_ARG;
    }
    GroupArgs a = {};
    a.scale_in = scale0;
    a.zp_in = zero_point;
    a.zp_dtype = NNCF_AMD_ZP_F32;
    a.n_groups = n_groups;
    a.gs = group_size;
    a.inner = 1;
    a.mode = mode;
    hipStream_t st = (hipStream_t)stream;
    if (mode >= NNCF_AMD_INT8_ASYM && mode <= NNCF_AMD_INT4_SYM) {
        const bool asym = mode == NNCF_AMD_INT8_ASYM || mode == NNCF_AMD_INT4_ASYM;
        if (asym && !zero_point) {
            set_error("se_zero_mask: asymmetric mode needs the zero points");
            return NNCF_AMD_ERR_BAD_ARG;
        }
        if (!asym) a.zp_in = nullptr;
        return se_zero_mask_dt<IntPolicy>(weight, w_dtype, s_per_channel, a, (int)group_size, groups, rows, zero_mask, rden, st);
    }
    a.zp_in = nullptr;
    if (mode == NNCF_AMD_NF4)
        return se_zero_mask_dt<FloatPolicy<true>>(weight, w_dtype, s_per_channel, a, (int)group_size, groups, rows, zero_mask, rden, st);
    if (mode == NNCF_AMD_MXFP4 || mode == NNCF_AMD_FP4)
        return se_zero_mask_dt<FloatPolicy<false>>(weight, w_dtype, s_per_channel, a, (int)group_size, groups, rows, zero_mask, rden, st);
    set_error("se_zero_mask: unsupported mode %d", mode);
    return NNCF_AMD_ERR_MODE;
}

// The whole search of one weight in ONE call: zero mask, baseline, `initial_steps` rectifications, `scale_steps` grid
// candidates (factor 1 - 0.05 k, scale_estimation.py:301) -- the same launches nncf_amd_se_zero_mask / nncf_amd_se_step
// make, queued back to back from here instead of one FFI round trip each (12 per weight with the reference's defaults).
// (Zero mask inside the baseline step of k_se_step16: 49.7 -> 57.9 us for that step, a 37.9 us pass less, per weight
// of the Llama-3-8B layer mix.)
extern "C" int nncf_amd_se_search(const void* weight, int w_dtype, uint32_t* zero_mask, float* rden,
                                  const float* s_per_channel, const float* x, float* fp_outs, const float* scale0,
                                  const float* zero_point, float* best_diffs, float* result_scale, int64_t batch,
                                  int64_t rows, int64_t groups, int64_t group_size, int64_t n_samples, int mode,
                                  int initial_steps, int scale_steps, float zero_scale, float weight_penalty, void* stream) {
    if (initial_steps < 0 || scale_steps < 0) {
        wc::set_error("se_search: negative step count");
        return NNCF_AMD_ERR_BAD_ARG;
    }
    // k_se_step16's baseline step (tiled shapes, integer modes and e2m1) writes the zero mask itself; the other kernels
    // take it from the separate pass (NF4) or rebuild it (`_any`)
    const bool fused_mask = wc::se_tiled_shape(group_size, n_samples, groups, batch, weight) && mode != NNCF_AMD_NF4 &&
                            (w_dtype == NNCF_AMD_F32 || w_dtype == NNCF_AMD_BF16) && s_per_channel && zero_mask && rden;
    int st = NNCF_AMD_OK;
    if (!fused_mask)
        st = nncf_amd_se_zero_mask(weight, w_dtype, s_per_channel, scale0, zero_point, batch, rows, groups, group_size, mode,
                                   zero_mask, rden, stream);
    if (st != NNCF_AMD_OK) return st;
    auto step = [&](int kind, float factor) {
        const bool base = kind == 0 && fused_mask;
        return se_step_impl(weight, w_dtype, zero_mask, rden, s_per_channel, x, fp_outs, scale0, zero_point, best_diffs,
                            result_scale, batch, rows, groups, group_size, n_samples, mode, kind, factor, zero_scale,
                            weight_penalty, stream, base ? zero_mask : nullptr, base ? rden : nullptr);
    };
    if ((st = step(0, 1.0f)) != NNCF_AMD_OK) return st;
    for (int i = 0; i < initial_steps; ++i)
        if ((st = step(1, 1.0f)) != NNCF_AMD_OK) return st;
    for (int k = 0; k < scale_steps; ++k)
        if ((st = step(2, (float)(1.0 - 0.05 * k))) != NNCF_AMD_OK) return st;
    return NNCF_AMD_OK;
}

// ====================================================================================================================
// AWQ alpha-grid search (awq.py:297-352), every outlier group and every alpha step in ONE launch.  The reference walks
// the groups and, per group, 100 candidate scales cur = clip(s_g)^alpha; per candidate:
//     wq = gweight * cur  ->  per-channel quantize-dequantize (one quantization group per row of the [N, gs] slice)
//     cur_out = dq @ (gacts / cur[:, None]);   cur_diff = mean |cur_out - fp32_out|
// (~10 launches and ~1 GB of traffic per group on a Llama down_proj).  Here a wave keeps 16 rows of the group's weight
// slice in registers (lane = row x quarter, as in k_se_step), and for each of its alpha steps scales them, reduces
// min/max over the row (4 lanes), quantizes, and multiplies by the pre-divided activations on the fp32 MFMA; what
// leaves the kernel is one partial sum of |cur_out - fp32_out| per (group, step, 16-row tile), summed by the host in
// a fixed order (no atomics: the argmin over alpha must not depend on the order of a floating-point reduction).
namespace wc {

struct AwqArgs {
    const float* w;        // [.., rows, K] fp32, row pitch K
    const int64_t* goff;   // [groups] element offset of (row 0, first column) of each group's slice
    const float* scales;   // [groups, steps, GS] candidate scales
    const float* sacts;    // [groups, steps, GS, 16, NT] = gacts[k][t*16 + c] / scale[k]  (zero beyond the samples)
    const float* fp_out;   // [groups, rows, ns] = gweight @ gacts
    float* partial;        // [groups, steps, tiles] sums of |cur_out - fp32_out| over a 16-row tile
    int64_t rows, K;
    int steps, ns, mode, steps_per_block;
};

template <int M>
__device__ __forceinline__ void awq_stat_xor(const IntPolicyT<M>&, typename IntPolicyT<M>::Stat& s, int mask) {
    s.lo = fminf(s.lo, __shfl_xor(s.lo, mask, 64));
    s.hi = fmaxf(s.hi, __shfl_xor(s.hi, mask, 64));
}
template <bool NF4, int M>
__device__ __forceinline__ void awq_stat_xor(const FloatPolicy<NF4, M>&, typename FloatPolicy<NF4, M>::Stat& s, int mask) {
    s.amax = fmaxf(s.amax, __shfl_xor(s.amax, mask, 64));
}

template <typename Policy, int GS, int NT>
__global__ __launch_bounds__(256) void k_awq_search(AwqArgs a) {
    constexpr int QUART = GS / 4, SP = GS + 4;
    __shared__ __align__(16) float stage_all[4 * 16 * SP];
    __shared__ float lut[kPolicyLds];
    Policy::init_shared(lut);
    __syncthreads();
    GroupArgs ga = {};
    ga.mode = a.mode;
    const Policy pol(ga, lut);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* stage = stage_all + wave * 16 * SP;
    const int r = lane & 15, q = lane >> 4;
    const int64_t N = a.rows;
    const int g = blockIdx.z;
    const int64_t tiles = (N + 15) / 16;
    const int64_t tile = (int64_t)blockIdx.x * 4 + wave;
    if (tile >= tiles) return;  // (no block barrier below)
    const int64_t n0 = tile * 16;
    const int s_begin = blockIdx.y * a.steps_per_block;
    const int s_end = s_begin + a.steps_per_block < a.steps ? s_begin + a.steps_per_block : a.steps;
    // rows n0 .. n0+15 of the group's [rows, GS] slice (row pitch K): se_load_rows addresses base + (gbase + n*G)*GS
    float wv[QUART];
    {
        const float* base = a.w + a.goff[g];
        constexpr int LPR = GS / 4, RPI = 64 / LPR, NLD = 16 / RPI;
        const int lr = lane / LPR, lc = (lane % LPR) * 4;
#pragma unroll
        for (int t = 0; t < NLD; ++t) {
            const int64_t gr = n0 + t * RPI + lr;
            const int64_t grc = gr < N ? gr : N - 1;
            const float* src = base + grc * a.K + lc;
            float4 v;
            if ((((uintptr_t)src) & 15u) == 0) {
                v = *(const float4*)src;
            } else {
                v.x = src[0], v.y = src[1], v.z = src[2], v.w = src[3];
            }
            *(float4*)(stage + (t * RPI + lr) * SP + lc) = v;
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const float* src = stage + r * SP + q * QUART;
#pragma unroll
        for (int i = 0; i < QUART / 4; ++i) {
            const float4 x = *(const float4*)(src + 4 * i);
            wv[4 * i] = x.x, wv[4 * i + 1] = x.y, wv[4 * i + 2] = x.z, wv[4 * i + 3] = x.w;
        }
    }
    // fp32_out of this tile in the accumulator layout: rows 4q + i, sample column t*16 + r
    const float* fo = a.fp_out + (int64_t)g * N * a.ns;
    float fpv[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t orow = n0 + 4 * q + i;
            const int col = t * 16 + r;
            fpv[t][i] = (col < a.ns && orow < N) ? fo[orow * a.ns + col] : 0.0f;
        }
    constexpr int C = QUART < 8 ? QUART : 8;
    for (int s = s_begin; s < s_end; ++s) {
        const float* sc = a.scales + ((int64_t)g * a.steps + s) * GS + q * QUART;
        float wq[QUART];
        typename Policy::Stat st = pol.stat_init();
#pragma unroll
        for (int i = 0; i < QUART; i += 4) {
            const float4 c4 = *(const float4*)(sc + i);
            wq[i] = wv[i] * c4.x, wq[i + 1] = wv[i + 1] * c4.y, wq[i + 2] = wv[i + 2] * c4.z, wq[i + 3] = wv[i + 3] * c4.w;
        }
#pragma unroll
        for (int i = 0; i < QUART; ++i) pol.stat_add(st, wq[i]);
        awq_stat_xor(pol, st, 16);
        awq_stat_xor(pol, st, 32);
        const typename Policy::Par par = pol.compute(st);
#pragma unroll
        for (int c0 = 0; c0 < QUART; c0 += C) {
            float val[C];
            se_values<C>(pol, par, wq + c0, val);
#pragma unroll
            for (int i = 0; i < C; ++i) wq[c0 + i] = pol.dq(0, val[i], par);
        }
        f32x4 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        const float* xb = a.sacts + (((int64_t)g * a.steps + s) * GS + q * QUART) * 16 * NT + r * NT;
#pragma unroll
        for (int p = 0; p < QUART; ++p) {
            float b[4];
            if constexpr (NT == 4) {
                const float4 x = *(const float4*)(xb + p * 16 * NT);
                b[0] = x.x, b[1] = x.y, b[2] = x.z, b[3] = x.w;
            } else if constexpr (NT == 2) {
                const float2 x = *(const float2*)(xb + p * 16 * NT);
                b[0] = x.x, b[1] = x.y;
            } else {
                b[0] = xb[p * 16 * NT];
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[p], b[t], acc[t], 0, 0, 0);
        }
        float sum = 0.0f;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool live = (t * 16 + r) < a.ns && (n0 + 4 * q + i) < N;
                sum += live ? fabsf(acc[t][i] - fpv[t][i]) : 0.0f;
            }
        sum = lanes_allreduce<64>(sum, OpAdd());
        if (lane == 0) a.partial[((int64_t)g * a.steps + s) * tiles + tile] = sum;
    }
}

// The same search for policies whose unscaled values are exact bf16 numbers (SeExact16: integer codes minus the zero
// point, e2m1 values): cur_out = scale_row * (t @ sacts) on v_mfma_f32_16x16x32_bf16 against the three-way bf16 split of
// sacts that k_awq_prepare wrote in the MFMA's B-operand order (3 x NT x GS/32 MFMAs of K = 32 per step instead of
// GS/4 x NT of K = 4), and the quantization through the correctly rounded quotient (se_values_div) instead of the tie
// test: the parameters come from these very elements, so `fast` of Policy::compute covers them.
template <typename Par>
__device__ __forceinline__ float awq_zsub_of(const Par& p, int) {
    return 0.0f;
}
template <typename Par>
__device__ __forceinline__ auto awq_zsub_of(const Par& p, long) -> decltype(p.zpf) {
    return p.has_zp ? p.zpf : 0.0f;
}

// A block is AWQ16_WAVES row tiles of one group and a run of alpha steps: sacts[g, s] (the same for every tile) is
// staged in LDS once per block and step -- read per wave from L2 it was the bound of this kernel (24 KB per wave and
// step: 17 GB per 28 groups x 100 steps of a Llama down_proj) -- and the next step's copy is requested into registers
// before the current step's arithmetic.
// (4 / 8 / 8 at a forced 128 registers / 16 waves per block: 856 / 1121 / 881 / 890 us on that shape, 1156 us with
// per-wave L2 reads; what is left is the per-element VALU work of the quantization.)
constexpr int AWQ16_WAVES = 4;

template <typename Policy, int GS, int NT>
__global__ __launch_bounds__(AWQ16_WAVES * 64) void k_awq_search16(AwqArgs a) {
    constexpr int QUART = GS / 4, M = GS / 32, THREADS = AWQ16_WAVES * 64;
    constexpr int PIECE = NT * M * 1024;  // bytes of one bf16 piece of sacts[g, s]
    constexpr int UNITS = 3 * PIECE / 16, NU = (UNITS + THREADS - 1) / THREADS;
    __shared__ __align__(16) char buf[3 * PIECE];
    __shared__ float lut[kPolicyLds];
    Policy::init_shared(lut);
    GroupArgs ga = {};
    ga.mode = a.mode;
    const Policy pol(ga, lut);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int64_t N = a.rows;
    const int g = blockIdx.z;
    const int64_t tiles = (N + 15) / 16;
    const int64_t tile_raw = (int64_t)blockIdx.x * AWQ16_WAVES + wave;
    const bool active = tile_raw < tiles;  // (an idle wave still takes part in the staging and the barriers)
    const int64_t tile = active ? tile_raw : tiles - 1;
    const int64_t n0 = tile * 16;
    const int s_begin = blockIdx.y * a.steps_per_block;
    const int s_end = s_begin + a.steps_per_block < a.steps ? s_begin + a.steps_per_block : a.steps;
    // lane (row r, quarter q) reads its QUART weights straight from the row (once per block: no LDS transpose)
    float wv[QUART];
    {
        const int64_t row = n0 + r < N ? n0 + r : N - 1;
        const float* src = a.w + a.goff[g] + row * a.K + q * QUART;
        if ((((uintptr_t)src) & 15u) == 0) {
#pragma unroll
            for (int i = 0; i < QUART / 4; ++i) {
                const float4 x = *(const float4*)(src + 4 * i);
                wv[4 * i] = x.x, wv[4 * i + 1] = x.y, wv[4 * i + 2] = x.z, wv[4 * i + 3] = x.w;
            }
        } else {
#pragma unroll
            for (int i = 0; i < QUART; ++i) wv[i] = src[i];
        }
    }
    const float* fo = a.fp_out + (int64_t)g * N * a.ns;
    float fpv[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t orow = n0 + 4 * q + i;
            const int col = t * 16 + r;
            fpv[t][i] = (col < a.ns && orow < N) ? fo[orow * a.ns + col] : 0.0f;
        }
    constexpr int C = QUART < 8 ? QUART : 8;
    const char* sa = (const char*)a.sacts + (int64_t)g * a.steps * (3 * PIECE);
    i32x4v pre[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int unit = u * THREADS + threadIdx.x;
        if (unit < UNITS && s_begin < s_end) pre[u] = *(const i32x4v*)(sa + (int64_t)s_begin * (3 * PIECE) + unit * 16);
    }
    const char* bb = buf + (q * 16 + r) * 16;
    for (int s = s_begin; s < s_end; ++s) {
        __syncthreads();  // the previous step's readers are done (first step: the policy table is written)
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int unit = u * THREADS + threadIdx.x;
            if (unit < UNITS) *(i32x4v*)(buf + unit * 16) = pre[u];
        }
        __syncthreads();
        if (s + 1 < s_end) {
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int unit = u * THREADS + threadIdx.x;
                if (unit < UNITS) pre[u] = *(const i32x4v*)(sa + (int64_t)(s + 1) * (3 * PIECE) + unit * 16);
            }
        }
        const float* sc = a.scales + ((int64_t)g * a.steps + s) * GS + q * QUART;
        float wq[QUART];
        typename Policy::Stat st = pol.stat_init();
#pragma unroll
        for (int i = 0; i < QUART; i += 4) {
            const float4 c4 = *(const float4*)(sc + i);
            wq[i] = wv[i] * c4.x, wq[i + 1] = wv[i + 1] * c4.y, wq[i + 2] = wv[i + 2] * c4.z, wq[i + 3] = wv[i + 3] * c4.w;
        }
#pragma unroll
        for (int i = 0; i < QUART; ++i) pol.stat_add(st, wq[i]);
        awq_stat_xor(pol, st, 16);
        awq_stat_xor(pol, st, 32);
        const typename Policy::Par par = pol.compute(st);
        const float zsub = awq_zsub_of(par, 0L);
        i32x4v Aq[M];
#pragma unroll
        for (int c0 = 0; c0 < QUART; c0 += C) {
            float val[C];
            se_values_div<C>(pol, par, wq + c0, val);
#pragma unroll
            for (int i = 0; i < C; i += 2) Aq[(c0 + i) / 8][((c0 + i) & 7) / 2] = pack_hi16(val[i] - zsub, val[i + 1] - zsub);
        }
        f32x4 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int m = 0; m < M; ++m)
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    acc[t] = mfma_bf16_k32(Aq[m], *(const i32x4v*)(bb + p * PIECE + (t * M + m) * 1024), acc[t]);
        // C layout: lane (c = lane % 16, q) holds rows 4q + i of sample column t*16 + c; row 4q + i's scale sits in
        // lane 4q + i (every quarter of a row computed the same parameters)
        float sum = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float scale_i = __shfl(par.scale, 4 * q + i, 64);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const bool live = (t * 16 + r) < a.ns && (n0 + 4 * q + i) < N;
                sum += live ? fabsf(scale_i * acc[t][i] - fpv[t][i]) : 0.0f;
            }
        }
        sum = lanes_allreduce<64>(sum, OpAdd());
        if (lane == 0 && active) a.partial[((int64_t)g * a.steps + s) * tiles + tile] = sum;
    }
}

// Preparation of the search for a set of outlier groups (awq.py:311-342), one launch: the candidate scale vector of every
// alpha step, cur = gscale^alpha (sqrt for alpha == 0.5, as `tensor ** 0.5` evaluates), the fp16-overflow clamp when the
// previous matmul absorbs 1 / scale (:315-335: where the scaled previous weight would leave a quarter of the fp16 range,
// those channels take prev_w * prev_s * n / threshold), and the scaled activations gacts / cur in the layout the search
// kernel reads.  Replaces ~15 small torch kernels and two 46 MB temporaries per 28 groups of a Llama down_proj.
struct AwqPrepArgs {
    const float* gscale;  // [groups, gs], already clipped to [quantile 0.1, 100]
    const float* alphas;  // [steps]
    const float* gacts;   // [groups, gs, ns]
    const float* gprev;   // [groups, gs] mean |previous weight| per channel, or NULL (no merged matmul)
    float* scales;        // [groups, steps, gs]
    float* sacts;         // AwqLayout of the shape and mode
    int gs, ns, steps, tiles, split;
    float prev_s, n_prev;
};

// Layout of `sacts` per (group, step):
//   tiles == 0           plain [gs, ns] fp32 (k_awq_search_any)
//   tiles > 0, !split    [gs, 16, tiles] fp32 (k_awq_search: NF4)
//   tiles > 0, split     [3 pieces][tiles][gs / 32][4 quarters][16 columns] x 8 bf16: the exact three-way bf16 split in
//                        the B-operand order of v_mfma_f32_16x16x32_bf16 (k_awq_search16: integer modes, e2m1)
struct AwqLayout {
    int tiles;
    bool split;
    int64_t bytes(int64_t gs, int64_t ns) const {
        if (tiles == 0) return gs * ns * 4;
        return split ? 3 * tiles * gs * 32 : gs * 16 * tiles * 4;
    }
};
static AwqLayout awq_layout(int64_t gs, int64_t ns, int mode) {
    if (!nncf_amd_se_step_supported(gs, ns)) return {0, false};
    return {ns <= 16 ? 1 : (ns <= 32 ? 2 : 4), mode != NNCF_AMD_NF4};
}

__global__ __launch_bounds__(256) void k_awq_prepare(AwqPrepArgs a) {
    extern __shared__ float cur[];  // [gs]
    __shared__ float red[4];
    const int s = blockIdx.x, g = blockIdx.y, gs = a.gs, ns = a.ns;
    const float alpha = a.alphas[s];
    const float threshold = 0.25f * 65504.0f;
    float worst = 0.0f;
    for (int k = threadIdx.x; k < gs; k += 256) {
        const float x = a.gscale[(int64_t)g * gs + k];
        const float c = alpha == 0.5f ? sqrtf(x) : powf(x, alpha);
        cur[k] = c;
        if (a.gprev) worst = fmaxf(worst, ((a.gprev[(int64_t)g * gs + k] / c) * a.prev_s) * a.n_prev);
    }
    if (a.gprev) {  // uniform
        worst = block_allreduce<4>(worst, OpMax(), red);
        const bool clip_row = worst >= threshold;
        __syncthreads();
        for (int k = threadIdx.x; k < gs; k += 256) {
            const float pw = a.gprev[(int64_t)g * gs + k];
            const float mag = ((pw / cur[k]) * a.prev_s) * a.n_prev;
            if (clip_row && !(mag < threshold)) cur[k] = ((pw * a.prev_s) * a.n_prev) / threshold;
        }
    }
    __syncthreads();
    float* so = a.scales + ((int64_t)g * a.steps + s) * gs;
    for (int k = threadIdx.x; k < gs; k += 256) so[k] = cur[k];
    const float* ga = a.gacts + (int64_t)g * gs * ns;
    if (a.tiles > 0 && a.split) {
        // one thread writes the bf16 pair (k, k + 1) of a sample column in each of the three pieces (4-byte stores,
        // consecutive threads consecutive words)
        const int M = gs / 32, quart = gs / 4, piece = a.tiles * M * 1024;
        char* out = (char*)a.sacts + ((int64_t)g * a.steps + s) * 3 * piece;
        for (int idx = threadIdx.x; idx < piece / 4; idx += 256) {
            const int i2 = idx & 3, c = (idx >> 2) & 15, qq = (idx >> 6) & 3, m = (idx >> 8) % M, t = (idx >> 8) / M;
            const int k = qq * quart + m * 8 + i2 * 2, j = t * 16 + c;
            uint32_t word[3] = {0u, 0u, 0u};
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const float x = j < ns ? ga[(k + e) * ns + j] / cur[k + e] : 0.0f;
                const float h = hi16_of(x), r1 = x - h, mm = hi16_of(r1), l = r1 - mm;  // exact: 8 + 8 + 8 significand bits
                word[0] |= (__builtin_bit_cast(uint32_t, h) >> 16) << (16 * e);
                word[1] |= (__builtin_bit_cast(uint32_t, mm) >> 16) << (16 * e);
                word[2] |= (__builtin_bit_cast(uint32_t, l) >> 16) << (16 * e);
            }
#pragma unroll
            for (int p = 0; p < 3; ++p) *(uint32_t*)(out + p * piece + idx * 4) = word[p];
        }
    } else if (a.tiles > 0) {
        const int width = 16 * a.tiles;  // per k: [16 columns][tiles], sample t*16 + c at c * tiles + t
        float* out = a.sacts + ((int64_t)g * a.steps + s) * gs * width;
        for (int idx = threadIdx.x; idx < gs * width; idx += 256) {
            const int k = idx / width, r = idx % width, c = r / a.tiles, t = r % a.tiles, j = t * 16 + c;
            out[idx] = j < ns ? ga[k * ns + j] / cur[k] : 0.0f;
        }
    } else {
        float* out = a.sacts + ((int64_t)g * a.steps + s) * gs * ns;
        for (int64_t idx = threadIdx.x; idx < (int64_t)gs * ns; idx += 256) out[idx] = ga[idx] / cur[idx / ns];
    }
}

}  // namespace wc

extern "C" int64_t nncf_amd_awq_sacts_bytes(int64_t group_size, int64_t n_samples, int mode) {
    if (group_size < 1 || n_samples < 1 || n_samples > 0x7fffffff / group_size) return -1;
    return wc::awq_layout(group_size, n_samples, mode).bytes(group_size, n_samples);
}

extern "C" int nncf_amd_awq_prepare(const float* gscale, const float* alphas, const float* gacts, const float* gprev,
                                    float prev_s, float n_prev, int groups, int steps, int64_t group_size,
                                    int64_t n_samples, int mode, float* scales, float* sacts, void* stream) {
    using namespace wc;
    if (groups < 0 || steps < 0 || group_size < 1 || n_samples < 1 || n_samples > 0x7fffffff / group_size) {
        set_error("awq_prepare: bad argument");
        return NNCF_AMD_ERR_BAD_ARG;
    }
    if (group_size > 40000) {  // one group's candidate scale sits in LDS (160 KB)
        set_error("awq_prepare: group size %lld exceeds 40000", (long long)group_size);
        return NNCF_AMD_ERR_UNSUPPORTED;
    }
    if (groups == 0 || steps == 0) return NNCF_AMD_OK;
    if (!gscale || !alphas || !gacts || !scales || !sacts || groups > 65535) {
        set_error("awq_prepare: NULL buffer or more than 65535 groups");
        return NNCF_AMD_ERR_BAD_ARG;
    }
    const AwqLayout lay = awq_layout(group_size, n_samples, mode);
    AwqPrepArgs a = {gscale, alphas, gacts, gprev, scales, sacts, (int)group_size, (int)n_samples, steps,
                     lay.tiles, lay.split ? 1 : 0, prev_s, n_prev};
    static OncePerDevice attr_once;
    if (attr_once.first_on_device() != 0) {
        if (hipFuncSetAttribute((const void*)k_awq_prepare, hipFuncAttributeMaxDynamicSharedMemorySize, 160000) != hipSuccess) {
            attr_once.forget_current();
            return launch_status("awq_prepare attribute");
        }
    }
    hipLaunchKernelGGL(k_awq_prepare, dim3((unsigned)steps, (unsigned)groups), dim3(256), (size_t)group_size * sizeof(float),
                       (hipStream_t)stream, a);
    return launch_status("awq_prepare");
}

namespace wc {

// Any group size / sample count: a wave owns a 16-row tile (four sets of R = 4 rows, so `partial` keeps the tiled
// kernel's [groups, steps, ceil(rows / 16)] shape) of one group for a run of alpha steps.  Per step and row set: pass 1
// strides the lanes over the group (scaled weight -> row statistics -> parameters), pass 2 dequantizes 64 elements at
// a time and broadcasts them against the lane's sample column(s) of sacts[g, s] in the PLAIN layout
// [groups, steps, group_size, n_samples].
template <typename Policy, int NT>
__global__ __launch_bounds__(256) void k_awq_search_any(AwqArgs a, int64_t gs) {
    constexpr int R = 4;
    __shared__ float lut[kPolicyLds];
    Policy::init_shared(lut);
    __syncthreads();
    GroupArgs ga = {};
    ga.mode = a.mode;
    const Policy pol(ga, lut);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t N = a.rows;
    const int g = blockIdx.z;
    const int ns = a.ns;
    const int64_t tiles = (N + 15) / 16;
    const int64_t tile = (int64_t)blockIdx.x * 4 + wave;
    if (tile >= tiles) return;  // (no block barrier below)
    const int s_begin = blockIdx.y * a.steps_per_block;
    const int s_end = s_begin + a.steps_per_block < a.steps ? s_begin + a.steps_per_block : a.steps;
    const float* base = a.w + a.goff[g];
    const float* fo = a.fp_out + (int64_t)g * N * ns;
    for (int s = s_begin; s < s_end; ++s) {
        const float* sc = a.scales + ((int64_t)g * a.steps + s) * gs;
        const float* sa = a.sacts + ((int64_t)g * a.steps + s) * gs * ns;
        float total = 0.0f;
        for (int sub = 0; sub < 4; ++sub) {
            const int64_t n0 = tile * 16 + sub * R;
            if (n0 >= N) break;  // uniform
            const float* wr[R];
            typename Policy::Stat st[R];
#pragma unroll
            for (int i = 0; i < R; ++i) {
                wr[i] = base + (n0 + i < N ? n0 + i : N - 1) * a.K;
                st[i] = pol.stat_init();
            }
            for (int64_t k = lane; k < gs; k += 64) {
                const float c = sc[k];
#pragma unroll
                for (int i = 0; i < R; ++i) pol.stat_add(st[i], wr[i][k] * c);
            }
            typename Policy::Par par[R];
#pragma unroll
            for (int i = 0; i < R; ++i) {
#pragma unroll
                for (int m = 1; m < 64; m <<= 1) awq_stat_xor(pol, st[i], m);
                par[i] = pol.compute(st[i]);
            }
            for (int j0 = 0; j0 < ns; j0 += 64 * NT) {
                float qa[R][NT];
#pragma unroll
                for (int i = 0; i < R; ++i)
#pragma unroll
                    for (int t = 0; t < NT; ++t) qa[i][t] = 0.0f;
                for (int64_t k0 = 0; k0 < gs; k0 += 64) {
                    const int64_t k = k0 + lane;
                    float dqv[R];
#pragma unroll
                    for (int i = 0; i < R; ++i) {
                        dqv[i] = 0.0f;
                        if (k < gs) {
                            int code;
                            float v;
                            pol.quant(wr[i][k] * sc[k], par[i], code, v);
                            dqv[i] = pol.dq(0, v, par[i]);
                        }
                    }
                    const int kn = gs - k0 < 64 ? (int)(gs - k0) : 64;
                    const float* xr = sa + k0 * ns + j0 + lane;
                    for (int kk = 0; kk < kn; ++kk) {
                        float xv[NT];
#pragma unroll
                        for (int t = 0; t < NT; ++t) xv[t] = (j0 + t * 64 + lane) < ns ? xr[(int64_t)kk * ns + t * 64] : 0.0f;
#pragma unroll
                        for (int i = 0; i < R; ++i) {
                            const float d = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, dqv[i]), kk));
#pragma unroll
                            for (int t = 0; t < NT; ++t) qa[i][t] = fmaf(d, xv[t], qa[i][t]);
                        }
                    }
                }
#pragma unroll
                for (int i = 0; i < R; ++i)
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const int col = j0 + t * 64 + lane;
                        if (col < ns && n0 + i < N) total += fabsf(qa[i][t] - fo[(n0 + i) * ns + col]);
                    }
            }
        }
        total = lanes_allreduce<64>(total, OpAdd());
        if (lane == 0) a.partial[((int64_t)g * a.steps + s) * tiles + tile] = total;
    }
}

template <typename Policy>
static int awq_any_launch(const AwqArgs& a, int groups, int64_t gs, hipStream_t s) {
    const int64_t tiles = (a.rows + 15) / 16;
    const dim3 grid((unsigned)((tiles + 3) / 4), (unsigned)((a.steps + a.steps_per_block - 1) / a.steps_per_block), (unsigned)groups);
    if (a.ns <= 64)
        hipLaunchKernelGGL((k_awq_search_any<Policy, 1>), grid, dim3(256), 0, s, a, gs);
    else if (a.ns <= 128)
        hipLaunchKernelGGL((k_awq_search_any<Policy, 2>), grid, dim3(256), 0, s, a, gs);
    else
        hipLaunchKernelGGL((k_awq_search_any<Policy, 4>), grid, dim3(256), 0, s, a, gs);
    return launch_status("awq_search (any group size)");
}

template <typename Policy, int GS>
static int awq_launch(const AwqArgs& a, int groups, hipStream_t s) {
    const int64_t tiles = (a.rows + 15) / 16;
    const dim3 grid((unsigned)((tiles + 3) / 4), (unsigned)((a.steps + a.steps_per_block - 1) / a.steps_per_block), (unsigned)groups);
    if constexpr (SeExact16<Policy>::value) {
        const dim3 grid16((unsigned)((tiles + AWQ16_WAVES - 1) / AWQ16_WAVES), grid.y, grid.z), block16(AWQ16_WAVES * 64);
        if (a.ns <= 16)
            hipLaunchKernelGGL((k_awq_search16<Policy, GS, 1>), grid16, block16, 0, s, a);
        else if (a.ns <= 32)
            hipLaunchKernelGGL((k_awq_search16<Policy, GS, 2>), grid16, block16, 0, s, a);
        else
            hipLaunchKernelGGL((k_awq_search16<Policy, GS, 4>), grid16, block16, 0, s, a);
        return launch_status("awq_search (bf16 MFMA)");
    }
    if (a.ns <= 16)
        hipLaunchKernelGGL((k_awq_search<Policy, GS, 1>), grid, dim3(256), 0, s, a);
    else if (a.ns <= 32)
        hipLaunchKernelGGL((k_awq_search<Policy, GS, 2>), grid, dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL((k_awq_search<Policy, GS, 4>), grid, dim3(256), 0, s, a);
    return launch_status("awq_search");
}

template <typename Policy>
static int awq_gs(const AwqArgs& a, int groups, int64_t gs, hipStream_t s) {
    if (a.ns > 64) return awq_any_launch<Policy>(a, groups, gs, s);
    switch (gs) {
        case 32: return awq_launch<Policy, 32>(a, groups, s);
        case 64: return awq_launch<Policy, 64>(a, groups, s);
        case 128: return awq_launch<Policy, 128>(a, groups, s);
        default: return awq_any_launch<Policy>(a, groups, gs, s);
    }
}

}  // namespace wc

extern "C" int nncf_amd_awq_search(const float* weight, int64_t rows, int64_t row_pitch, const int64_t* group_offsets,
                                   int groups, int64_t group_size, const float* scales, int steps, const float* sacts,
                                   const float* fp_out, int64_t n_samples, int mode, float* partial, void* stream) {
    using namespace wc;
    if (rows < 0 || groups < 0 || steps < 0 || group_size < 1 || row_pitch < group_size || n_samples < 1 ||
        n_samples > 0x7fffffff) {
        set_error("awq_search: bad argument");
        return NNCF_AMD_ERR_BAD_ARG;
    }
    if (rows == 0 || groups == 0 || steps == 0) return NNCF_AMD_OK;
    if (!weight || !group_offsets || !scales || !sacts || !fp_out || !partial) {
        set_error("awq_search: NULL buffer");
        return NNCF_AMD_ERR_BAD_ARG;
    }
    if (groups > 65535) {
        set_error("awq_search: more than 65535 groups per launch (%d): split the list", groups);
        return NNCF_AMD_ERR_UNSUPPORTED;
    }
    if (awq_layout(group_size, n_samples, mode).tiles != 0 && (((uintptr_t)scales | (uintptr_t)sacts) & 15u)) {  // (tiled kernels)
        set_error("awq_search: scales / sacts must be 16-byte aligned");
        return NNCF_AMD_ERR_BAD_ARG;
    }
    // enough blocks to fill the chip: split the alpha steps of a tile over blocks when there are few row tiles
    const int64_t tiles = (rows + 15) / 16;
    const AwqLayout lay = awq_layout(group_size, n_samples, mode);
    const bool wide = lay.tiles != 0 && lay.split;  // k_awq_search16: AWQ16_WAVES tiles per block
    const int64_t blocks_per_chunk = (wide ? (tiles + AWQ16_WAVES - 1) / AWQ16_WAVES : (tiles + 3) / 4) * groups;
    int chunks = (int)(((wide ? 1024 : 2048) + blocks_per_chunk - 1) / blocks_per_chunk);
    if (chunks < 1) chunks = 1;
    if (chunks > steps) chunks = steps;
    AwqArgs a = {weight, group_offsets, scales, sacts, fp_out, partial, rows, row_pitch, steps, (int)n_samples, mode,
                 (steps + chunks - 1) / chunks};
    hipStream_t s = (hipStream_t)stream;
    if (mode >= NNCF_AMD_INT8_ASYM && mode <= NNCF_AMD_INT4_SYM) return awq_gs<IntPolicy>(a, groups, group_size, s);
    if (mode == NNCF_AMD_NF4) return awq_gs<FloatPolicy<true>>(a, groups, group_size, s);
    if (mode == NNCF_AMD_MXFP4 || mode == NNCF_AMD_FP4) return awq_gs<FloatPolicy<false>>(a, groups, group_size, s);
    set_error("awq_search: unsupported mode %d", mode);
    return NNCF_AMD_ERR_MODE;
}
