// wc_float_policy.h -- arithmetic policy of the NF4 / MXFP4 / FP4 codebook lowering (shared by the lowering
// kernels and the GPTQ column sweep).  See wc_group_kernels.h for the policy interface.
#pragma once
#include "wc_e8m0_table.h"
#include "wc_group_kernels.h"

namespace wc {

// constants.py:17-58, 84-106 (data, bit exact)
__device__ static const float kNF4Q[16] = {-1.0f, -0.6961928009986877f, -0.5250730514526367f, -0.39491748809814453f,
                                           -0.28444138169288635f, -0.18477343022823334f, -0.09105003625154495f, 0.0f,
                                           0.07958029955625534f, 0.16093020141124725f, 0.24611230194568634f,
                                           0.33791524171829224f, 0.44070982933044434f, 0.5626170039176941f,
                                           0.7229568362236023f, 1.0f};
__device__ static const float kNF4C[15] = {-0.84809643f, -0.6106329f, -0.45999527f, -0.33967942f, -0.2346074f,
                                           -0.13791174f, -0.045525018f, 0.03979015f, 0.120255254f, 0.20352125f,
                                           0.29201376f, 0.38931254f, 0.5016634f, 0.6427869f, 0.8614784f};
__device__ static const float kE2M1Q[15] = {-6.0f, -4.0f, -3.0f, -2.0f, -1.5f, -1.0f, -0.5f, -0.0f,
                                            0.5f, 1.0f, 1.5f, 2.0f, 3.0f, 4.0f, 6.0f};
// index in kE2M1Q -> OCP e2m1 bit pattern (sign<<3 | exp<<1 | man); the reference's single (negative) zero -> 0b0000
__device__ static const uint8_t kE2M1Code[15] = {15, 14, 13, 12, 11, 10, 9, 0, 1, 2, 3, 4, 5, 6, 7};

// LDS table layout (floats): [0,80) per bucket the pair (center, base index as int bits), [80,96) quantiles,
// [96,112) codes (as int bits)
constexpr int kLutPair = 0, kLutQ = 80, kLutCode = 96;

// Fast exact NF4 (multi-tensor kernel).  The reference takes n = RN(w/s) per element and counts the centers below
// n.  Here r ~ 1/s once per group (v_rcp_f32 + one Newton step) and t = RN(w*r): |t - n| <= 3.01*2^-24*|t| < 2^-22
// (|t| <= 1), so the count can differ only if a center lies that close to t.  No center is within 0.02/16 of a
// bucket boundary (table property, checked by tests/test_oracle_known_answers.py), so testing the center of t's
// own bucket is enough; elements inside the margin are recomputed with the IEEE divide.
constexpr float kNF4Margin = 2.384185791015625e-07f;  // 2^-22

// MODE >= 0 fixes the mode at compile time (multi-tensor kernel); -1 reads it from GroupArgs.
template <bool NF4, int MODE = -1>
struct FloatPolicy {
    // bucket = clamp(floor(n * K) + OFF, 0, NB-1)
    static constexpr float K = NF4 ? 16.0f : 2.0f;
    static constexpr int OFF = NF4 ? 17 : 14;
    static constexpr int NB = NF4 ? 35 : 29;
    static constexpr int NQ = NF4 ? 16 : 15;
    const float* lut;
    int mode;
    struct Stat {
        float amax;
    };
    struct Par {
        float scale;
        float r;    // ~1/scale (NF4 fast path)
        bool fast;  // scale inside [2^-100, 2^100]
    };

    static __device__ __forceinline__ float center(int j) {
        if constexpr (NF4) return kNF4C[j];
        return (kE2M1Q[j + 1] + kE2M1Q[j]) / 2;  // constants.py:106
    }
    static __device__ __forceinline__ int bucket_of(float n) {
        int b = __float2int_rd(n * K) + OFF;
        return b < 0 ? 0 : (b > NB - 1 ? NB - 1 : b);
    }
    static __device__ void init_shared(float* lds) {
        const int t = threadIdx.x;
        if (t < NB) {
            // centers falling into bucket t (at most one by construction) and the number of centers below it
            float c = __builtin_inff();
            int base = 0;
            for (int j = 0; j < NQ - 1; ++j) {
                const int bj = bucket_of(center(j));
                if (bj < t) ++base;
                if (bj == t) c = center(j);
            }
            lds[kLutPair + 2 * t] = c;
            lds[kLutPair + 2 * t + 1] = __builtin_bit_cast(float, base);
        }
        if (t < NQ) {
            lds[kLutQ + t] = NF4 ? kNF4Q[t] : kE2M1Q[t];
            lds[kLutCode + t] = __builtin_bit_cast(float, NF4 ? t : (int)kE2M1Code[t]);
        }
    }
    __device__ __forceinline__ FloatPolicy(const GroupArgs& a, const float* lds) : lut(lds), mode(MODE >= 0 ? MODE : a.mode) {}
    __device__ __forceinline__ Stat stat_init() const { return {0.0f}; }
    __device__ __forceinline__ void stat_add(Stat& s, float f) const { s.amax = fmaxf(s.amax, fabsf(f)); }
    template <int E>
    __device__ __forceinline__ void stat_add_vec(Stat& s, const float* f) const {
#pragma unroll
        for (int i = 0; i < E; i += 2) s.amax = max3_abs_raw(s.amax, f[i], f[i + 1]);
    }
    template <int L>
    __device__ __forceinline__ void stat_lanes(Stat& s) const {
        max_dpp16<(L < 16 ? L : 16)>(s.amax);
        if constexpr (L >= 32) s.amax = fmaxf(s.amax, __shfl_xor(s.amax, 16, 64));
        if constexpr (L >= 64) s.amax = fmaxf(s.amax, __shfl_xor(s.amax, 32, 64));
    }
    template <int E>
    __device__ __forceinline__ void emit(const GroupArgs& a, const Par& p, const float* f, int64_t v) const {
        emit_vec<E>(a, *this, p, f, v);
    }
    // multi-tensor kernel: NE consecutive elements of one lane -> NE/8 dwords of nibble codes
    template <int NE, int GPL>
    __device__ __forceinline__ void pack_lane(const Par* par, float* f, uint32_t* out) const {
        constexpr int SG = NE / GPL;
        if constexpr (!NF4 && MODE == NNCF_AMD_MXFP4) {
            // The scale is a power of two, so w/scale is exact and v_cvt_scalef32_pk_fp4_f32 (divide by the scale's
            // power of two, round to nearest even in e2m1, saturate at +-6) IS the reference's rounding
            // (weight_lowering.py:577-583: ties go to the even mantissa).  Its -0 (code 8) becomes the
            // reference's single zero (code 0).
#pragma unroll
            for (int j = 0; j < NE / 8; ++j) {
                const float* e = f + 8 * j;
                uint32_t w = 0;
                w = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(w, e[0], e[1], par[(8 * j + 0) / SG].scale, 0);
                w = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(w, e[2], e[3], par[(8 * j + 2) / SG].scale, 1);
                w = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(w, e[4], e[5], par[(8 * j + 4) / SG].scale, 2);
                w = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(w, e[6], e[7], par[(8 * j + 6) / SG].scale, 3);
                const uint32_t mag = w & 0x77777777u;
                const uint32_t nz = (mag | (mag >> 1) | (mag >> 2)) & 0x11111111u;  // bit 0 of a nibble: magnitude != 0
                out[j] = w & (0x77777777u | (nz << 3));
            }
        } else if constexpr (NF4) {
#pragma unroll
            for (int j = 0; j < NE / 8; ++j) {
                int idx[8];
                float d[8];
                bool fast = true;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const Par& p = par[(8 * j + q) / SG];
                    fast = fast && p.fast;
                    const float t = f[8 * j + q] * p.r;
                    int bk = __float2int_rd(t * K) + OFF;
                    bk = bk < 0 ? 0 : (bk > NB - 1 ? NB - 1 : bk);
                    const float2 pr = *(const float2*)(lut + kLutPair + 2 * bk);
                    idx[q] = __builtin_bit_cast(int, pr.y) + (t > pr.x ? 1 : 0);
                    d[q] = t - pr.x;
                }
                float viol = __builtin_inff();
#pragma unroll
                for (int q = 0; q < 8; q += 2) viol = min3_abs_raw(viol, d[q], d[q + 1]);
                if (!fast || !(viol > kNF4Margin)) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        if (!fast || !(fabsf(d[q]) > kNF4Margin)) {
                            float val;
                            quant(f[8 * j + q], par[(8 * j + q) / SG], idx[q], val);
                        }
                    }
                }
                uint32_t w = 0;
#pragma unroll
                for (int q = 0; q < 8; ++q) w |= (uint32_t)idx[q] << (4 * q);
                out[j] = w;
            }
        } else {
#pragma unroll
            for (int j = 0; j < NE / 8; ++j) {
                uint32_t w = 0;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    int code;
                    float val;
                    quant(f[8 * j + q], par[(8 * j + q) / SG], code, val);
                    w |= (uint32_t)code << (4 * q);
                }
                out[j] = w;
            }
        }
    }
    template <int NE, int GPL, bool NT, bool DIV = false>
    __device__ __forceinline__ void emit_lane(const GroupArgs& a, const Par* par, float* f, int64_t e0) const {
        constexpr int SG = NE / GPL;
        int code[NE];
#pragma unroll
        for (int gi = 0; gi < GPL; ++gi) {
#pragma unroll
            for (int i = 0; i < SG; ++i) {
                float val;
                quant(f[gi * SG + i], par[gi], code[gi * SG + i], val);
                f[gi * SG + i] = val;
            }
        }
        if (a.packed) {
            uint32_t w[NE / 8];
#pragma unroll
            for (int j = 0; j < NE / 8; ++j) {
                w[j] = 0;
#pragma unroll
                for (int i = 0; i < 8; ++i) w[j] |= (uint32_t)code[8 * j + i] << (4 * i);
            }
            uint32_t* dst = (uint32_t*)(a.packed + e0 / 2);
            if constexpr (NE == 16) st<NT>((u32x2*)dst, u32x2{w[0], w[1]});
            if constexpr (NE == 32) st<NT>((u32x4*)dst, u32x4{w[0], w[1], w[2], w[3]});
        }
        if (a.codes) {
            uint4* dst = (uint4*)((uint8_t*)a.codes + e0);
#pragma unroll
            for (int j = 0; j < NE / 16; ++j) {
                uint32_t b[4];
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    b[q] = (uint32_t)code[16 * j + 4 * q] | ((uint32_t)code[16 * j + 4 * q + 1] << 8) |
                           ((uint32_t)code[16 * j + 4 * q + 2] << 16) | ((uint32_t)code[16 * j + 4 * q + 3] << 24);
                dst[j] = make_uint4(b[0], b[1], b[2], b[3]);
            }
        }
        if (a.values) {
            float4* o = (float4*)(a.values + e0);
#pragma unroll
            for (int j = 0; j < NE / 4; ++j) o[j] = make_float4(f[4 * j], f[4 * j + 1], f[4 * j + 2], f[4 * j + 3]);
        }
        if (a.dq) {
            float4* o = (float4*)(a.dq + e0);
#pragma unroll
            for (int gi = 0; gi < GPL; ++gi) {
                const float sc = par[gi].scale;
#pragma unroll
                for (int j = 0; j < SG / 4; ++j) {
                    const int q = gi * SG + 4 * j;
                    o[q / 4] = make_float4(f[q] * sc, f[q + 1] * sc, f[q + 2] * sc, f[q + 3] * sc);
                }
            }
        }
    }
    __device__ __forceinline__ void stat_merge(Stat& s, const Stat& o) const { s.amax = fmaxf(s.amax, o.amax); }
    __device__ __forceinline__ void stat_block(Stat& s, float* smem) const {
        s.amax = block_allreduce<4>(s.amax, OpMax(), smem);
    }
    // weight_lowering.py:106-126
    __device__ __forceinline__ Par compute(const Stat& s) const {
        float sc = s.amax;
        if (!NF4) sc = sc / 6.0f;
        if (fabsf(sc) < kEps) sc = kEps;
        if (mode == NNCF_AMD_MXFP4) {
            // 2 ** clip(ceil(log2_f32(sc)), -127, 127): exponent k, bumped unless the correctly rounded
            // fp32 log2 ties back to k (wc_e8m0_table.h).  sc >= 2^-23 here, so it is a normal number.
            const uint32_t bits = __builtin_bit_cast(uint32_t, sc);
            int k = (int)((bits >> 23) & 0xffu) - 127;
            const uint32_t mant = bits & 0x7fffffu;
            if (k >= -127 && k <= 127 && mant > kLog2SlackMant[k + 127]) ++k;
            k = k < -127 ? -127 : (k > 127 ? 127 : k);
            sc = ldexpf(1.0f, k);
        }
        return with_reciprocal(sc);
    }
    __device__ __forceinline__ Par with_reciprocal(float sc) const {
        Par p;
        p.scale = sc;
        const float r0 = __builtin_amdgcn_rcpf(sc);
        p.r = fmaf(fmaf(-sc, r0, 1.0f), r0, r0);
        const float as = fabsf(sc);
        p.fast = (as >= 7.888609052210118e-31f) && (as <= 1.2676506002282294e+30f);  // [2^-100, 2^100]
        return p;
    }
    __device__ __forceinline__ Par load(const GroupArgs& a, int64_t g) const { return with_reciprocal(a.scale_in[g]); }
    template <bool NT>
    __device__ __forceinline__ void store(const GroupArgs& a, int64_t g, const Par& p) const {
        if (a.scale_out) st<NT>(a.scale_out + g, p.scale);
        if (a.e8m0_out) {
            const uint32_t bits = __builtin_bit_cast(uint32_t, p.scale);
            int e = (int)((bits >> 23) & 0xffu);  // biased exponent == log2(scale) + 127 for normal powers of two
            if (e == 0) e = 0;                      // 2^-127 is an fp32 denormal: E8M0 byte 0
            a.e8m0_out[g] = (uint8_t)e;
        }
    }
    __device__ __forceinline__ void quant(float w, const Par& p, int& code, float& value) const {
        const float n = w / p.scale;  // weight_lowering.py:626
        const int b = bucket_of(n);
        int idx = __builtin_bit_cast(int, lut[kLutPair + 2 * b + 1]) + (n > lut[kLutPair + 2 * b] ? 1 : 0);
        if constexpr (!NF4) {
            // weight_lowering.py:577-583
            const int r = idx + 1 > NQ - 1 ? NQ - 1 : idx + 1;
            const float dl = fabsf(n - lut[kLutQ + idx]);
            const float dr = fabsf(n - lut[kLutQ + r]);
            const bool right = (dr < dl) || ((dl == dr) && (((r + 1) & 1) == 0));
            idx = right ? r : idx;
        }
        value = lut[kLutQ + idx];
        code = __builtin_bit_cast(int, lut[kLutCode + idx]);
    }
    __device__ __forceinline__ float dq(int, float value, const Par& p) const { return value * p.scale; }
    __device__ __forceinline__ int nibble(int code) const { return code; }
    __device__ __forceinline__ float target(float value, const Par&) const { return value; }
};

}  // namespace wc
