// wc_int_policy.h -- arithmetic policy of the INT4/INT8 lowering (shared by the lowering kernels and the
// GPTQ column sweep).  See wc_group_kernels.h for the policy interface.
#pragma once
#include "wc_group_kernels.h"

namespace wc {

// Fast exact rounding.  The reference needs k = rne(RN(w/s) [+ z]) with an IEEE fp32 divide per element
// (~10 VALU ops, one of them a quarter-rate v_rcp).  Per group we take r = RN(1/s) once and per element
//     t = RN(w*r),  t2 = RN(t + z),  k = rne(t2).
// With x = RN(w/s), x2 = RN(x + z):  |t - x| <= 3.01*2^-24*|t|  and  |t2 - x2| <= 3.01*2^-24*(|t| + |t2|)
// (one rounding each in r, w*r and w/s, plus the two sums' roundings), so k can differ from the reference
// only if a half-integer lies within that distance of t2 (a constant per mode, see quant_n).  (r itself comes from v_rcp_f32 plus one Newton step:
// within 0.5 ulp + 2^-46 of 1/s, which the 0.01 of slack in the 3.01 absorbs.)  Such elements (|t2 - k| > 0.5 - 2^-21*(|t|+|t2|),
// probability ~2^-17 per int4 element), NaN/inf intermediates and groups whose scale is outside
// [2^-100, 2^100] are recomputed with the IEEE divide, so the result is bit-exact by construction.

// MODE >= 0 fixes the mode at compile time (the multi-tensor kernel: levels and branches fold away);
// MODE == -1 reads it from GroupArgs.
template <int MODE>
struct IntPolicyT {
    IntMode m;
    float pack_bias;  // 8 * 0x1111 for signed nibbles, 0 for unsigned
    struct Stat {
        float lo, hi;
    };
    struct Par {
        float scale, zpf, r;
        int zpi;
        bool has_zp, fast;
    };
    static __device__ __forceinline__ void init_shared(float*) {}
    __device__ __forceinline__ IntPolicyT(const GroupArgs& a, const float*) : m(int_mode(MODE >= 0 ? MODE : a.mode)) {
        pack_bias = m.asym ? 0.0f : 34952.0f;
    }
    __device__ __forceinline__ Stat stat_init() const { return {__builtin_inff(), -__builtin_inff()}; }
    __device__ __forceinline__ void stat_add(Stat& s, float f) const {
        s.lo = fminf(s.lo, f);
        s.hi = fmaxf(s.hi, f);
    }
    template <int E>
    __device__ __forceinline__ void stat_add_vec(Stat& s, const float* f) const {
#pragma unroll
        for (int i = 0; i < E; i += 2) {
            s.lo = min3_raw(s.lo, f[i], f[i + 1]);
            s.hi = max3_raw(s.hi, f[i], f[i + 1]);
        }
    }
    template <int L>
    __device__ __forceinline__ void stat_lanes(Stat& s) const {
        minmax_dpp16<(L < 16 ? L : 16)>(s.lo, s.hi);
        if constexpr (L >= 32) {
            s.lo = fminf(s.lo, __shfl_xor(s.lo, 16, 64));
            s.hi = fmaxf(s.hi, __shfl_xor(s.hi, 16, 64));
        }
        if constexpr (L >= 64) {
            s.lo = fminf(s.lo, __shfl_xor(s.lo, 32, 64));
            s.hi = fmaxf(s.hi, __shfl_xor(s.hi, 32, 64));
        }
    }
    __device__ __forceinline__ void stat_merge(Stat& s, const Stat& o) const {
        s.lo = fminf(s.lo, o.lo);
        s.hi = fmaxf(s.hi, o.hi);
    }
    __device__ __forceinline__ void stat_block(Stat& s, float* smem) const {
        s.lo = block_allreduce<4>(s.lo, OpMin(), smem);
        s.hi = block_allreduce<4>(s.hi, OpMax(), smem + 8);
    }
    // r = RN(1/scale), correctly rounded (quant_n_div depends on it): v_rcp_f32 + one Newton step lands on RN(1/s) or
    // on the float next to it; the residual e = 1 - s*r is exactly representable then (|e| < 2^-23, a multiple of
    // 2^-47 relative to s*r ~ 1), and r is the nearest float iff |1/s - r| = |e| / |s| <= ulp(r) / 2, an exact
    // comparison (the right side is |s| times a power of two).  Otherwise the nearest float is one step further in
    // the direction of sign(e) (in magnitude: |1/s| - |r| = e / |s|).
    __device__ __forceinline__ void finish(Par& p) const {
        const float r0 = __builtin_amdgcn_rcpf(p.scale);
        float r = fmaf(fmaf(-p.scale, r0, 1.0f), r0, r0);
        const float e = fmaf(-p.scale, r, 1.0f);
        const float half_ulp_s = ldexpf(fabsf(p.scale), __builtin_amdgcn_frexp_expf(r) - 25);
        if (fabsf(e) > half_ulp_s) r = __builtin_bit_cast(float, __builtin_bit_cast(int, r) + (e > 0.0f ? 1 : -1));
        p.r = r;
        const float as = fabsf(p.scale);
        p.fast = (as >= 8.673617379884035e-19f) && (as <= 1.152921504606847e+18f);  // [2^-60, 2^60]: no under/overflow below
        if (!(fabsf(p.zpf) <= m.level_high)) p.fast = false;  // the constant tie margins assume |z| <= levels
    }
    __device__ __forceinline__ Par compute(const Stat& s) const {
        Par p;
        int_params(s.lo, s.hi, m, p.scale, p.zpf, p.zpi);
        p.has_zp = m.asym;
        finish(p);
        // quant_n_div: every quotient of the group must stay far from overflow (constant groups have scale = eps)
        if (!(fmaxf(fabsf(s.lo), fabsf(s.hi)) * fabsf(p.r) <= 1.2676506002282294e+30f)) p.fast = false;
        return p;
    }
    __device__ __forceinline__ Par load(const GroupArgs& a, int64_t g) const {
        Par p;
        p.scale = a.scale_in[g];
        p.has_zp = a.zp_in != nullptr;
        p.zpi = 0;
        p.zpf = 0.0f;
        if (p.has_zp) {
            if (a.zp_dtype == NNCF_AMD_ZP_I32) {
                p.zpi = ((const int32_t*)a.zp_in)[g];
                p.zpf = (float)p.zpi;
            } else {
                p.zpf = ((const float*)a.zp_in)[g];  // weight_lowering.py:679-680: added as float
                p.zpi = (int)p.zpf;
            }
        }
        finish(p);
        return p;
    }
    template <bool NT>
    __device__ __forceinline__ void store(const GroupArgs& a, int64_t g, const Par& p) const {
        if (a.scale_out) st<NT>(a.scale_out + g, p.scale);
        if (a.zp_out) st<NT>(a.zp_out + g, p.zpi);
    }
    // scalar form (generic kernel): the reference's operation order verbatim
    __device__ __forceinline__ void quant(float w, const Par& p, int& code, float& value) const {
        value = int_quant(w, p.scale, p.zpf, p.has_zp, m);
        code = (int)value;
    }
    // weight_lowering.py:442-443: (int32(q) - zp).astype(f32) * scale; q, zp small integers -> exact subtraction
    __device__ __forceinline__ float dq(int, float value, const Par& p) const {
        return p.has_zp ? (value - p.zpf) * p.scale : value * p.scale;
    }
    __device__ __forceinline__ int nibble(int code) const { return code + (m.asym ? 0 : 8); }
    // scale_estimation.py:380-393: the unscaled tensor minus the zero point
    __device__ __forceinline__ float target(float value, const Par& p) const { return p.has_zp ? value - p.zpf : value; }

    // k[i] <- clip(rne(w/s [+ z])) for N elements sharing one parameter set; exact (see the header comment).
    // f and k may alias: elements are processed in chunks of C whose inputs stay intact until the chunk's
    // rare exact re-computation has been decided.  The fast path's t2 = fma(w, r, z) (one rounding) is
    // covered by the same error bound as RN(RN(w*r) + z).  The tie test keeps a running maximum of the
    // violation instead of a per-element flag: one v_max3 per two elements.
    template <int N>
    __device__ __forceinline__ void quant_n(const Par& p, const float* f, float* k) const {
        constexpr int C = N < 8 ? N : 8;
        const float z = p.has_zp ? p.zpf : 0.0f;
        // Wherever the clip does not decide anyway, |t2| <= levels + 1.5 and |w*r| <= |t2| + |z| with |z| <= levels
        // (finish() clears `fast` otherwise), so the error bound -- 3.01*2^-24*|t2| without a zero point,
        // 2^-24*(2.01*|w*r| + 2*|t2|) with one -- is a constant per mode:
        //   4-bit sym 1.6e-6 < 2^-18, asym 5.7e-6 < 2^-16;   8-bit sym 2.3e-5 < 2^-15, asym 9.2e-5 < 2^-12
        const float margin = m.bits == 4 ? (p.has_zp ? 1.52587890625e-05f : 3.814697265625e-06f)
                                         : (p.has_zp ? 2.44140625e-04f : 3.0517578125e-05f);
        const float lim = 0.5f - margin;
#pragma unroll
        for (int c0 = 0; c0 < N; c0 += C) {
            float kc[C], d[C];
#pragma unroll
            for (int i = 0; i < C; ++i) {
                const float t2 = fmaf(f[c0 + i], p.r, z);
                const float kk = rintf(t2);
                d[i] = t2 - kk;
                kc[i] = __builtin_amdgcn_fmed3f(kk, m.level_low, m.level_high);
            }
            float viol = 0.0f;
            if constexpr (C >= 2) {
#pragma unroll
                for (int i = 0; i < C; i += 2) viol = max3_abs_raw(viol, d[i], d[i + 1]);
            } else {
                viol = fabsf(d[0]);
            }
            // +-inf weights give d = NaN, which v_max3 drops: fine, the fast path's clip(+-inf) already equals the
            // reference's result.  NaN weights are outside the contract (the reference's cast of NaN is undefined).
            const bool redo = !p.fast || !(viol <= lim);
            if (redo) {
                // divergent, element by element, only where the tie test failed.  Exact ties are NOT rare for
                // bf16/f16 weights (8..11-bit significands: ~0.25 % of the elements of a N(0, 0.02) bf16 tensor have
                // w/s exactly on x.5), so they are settled without a divide: with b the half-integer next to t2,
                // rho = fma(-(b - z), s, w) is zero iff w/s == b - z exactly; then RN(w/s) + z == b exactly and the
                // reference rounds the tie to even, rintf(b).  Everything else takes the IEEE divide.
#pragma unroll
                for (int i = 0; i < C; ++i) {
                    if (!p.fast) {
                        kc[i] = int_quant(f[c0 + i], p.scale, p.zpf, p.has_zp, m);
                    } else if (!(fabsf(d[i]) <= lim)) {
                        const float w = f[c0 + i];
                        const float b = floorf(fmaf(w, p.r, z)) + 0.5f;
                        const float rho = fmaf(-(b - z), p.scale, w);
                        if (rho == 0.0f)
                            kc[i] = __builtin_amdgcn_fmed3f(rintf(b), m.level_low, m.level_high);
                        else
                            kc[i] = int_quant(w, p.scale, p.zpf, p.has_zp, m);
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < C; ++i) k[c0 + i] = kc[i];
        }
    }
    // k[i] <- clip(rne(RN(w/s) [+ z])) for N elements sharing parameters that were COMPUTED from these very elements
    // (so a non-finite element has already made the scale non-finite and the group "not fast").  Straight-line and
    // exact: the IEEE quotient comes from the correctly rounded reciprocal r = RN(1/s) by two correction steps,
    //     q0 = RN(w r);  q1 = RN(q0 + (w - s q0) r);  q2 = RN(q1 + (w - s q1) r)
    // (the residuals are exact in one fma).  q1 is within half an ulp + 2^-23 ulp of w/s, hence a faithful rounding,
    // and Markstein's theorem (r correctly rounded, q faithful => RN(q + (w - s q) r) = RN(w/s)) makes q2 the
    // correctly rounded quotient, so the sum with z and the rounding see what the reference's divide gives them.  No tie test, no branch: exact ties (w/s on x.5)
    // are NOT rare for bf16 / f16 weights (8..11-bit significands: ~0.25 % of the elements of a N(0, 0.02) bf16
    // tensor), which made 70 % of the 8-element chunks of a wave take the tie-test path's divergent re-computation
    // -- a quarter of the kernel's issue slots.  The five multiply-adds are packed (v_pk_mul_f32 / v_pk_fma_f32:
    // two elements per issue slot).  Scales outside [2^-60, 2^60] (residual under/overflow) take the IEEE divide.
    template <int N>
    __device__ __forceinline__ void quant_n_div(const Par& p, const float* f, float* k) const {
        static_assert(N % 2 == 0, "pairs of elements");
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const f32x2 r = {p.r, p.r}, ns = {-p.scale, -p.scale};
        const float zf = p.has_zp ? p.zpf : 0.0f;
        const f32x2 z = {zf, zf};
        if (!p.fast) {  // (per lane; the wave skips the block unless one of its groups is out of range)
#pragma unroll
            for (int i = 0; i < N; ++i) k[i] = int_quant(f[i], p.scale, p.zpf, p.has_zp, m);
        }
        if (p.fast) {
#pragma unroll
            for (int i = 0; i < N; i += 2) {
                const f32x2 w = {f[i], f[i + 1]};
                f32x2 q = w * r;
                f32x2 res = __builtin_elementwise_fma(q, ns, w);
                q = __builtin_elementwise_fma(res, r, q);
                res = __builtin_elementwise_fma(q, ns, w);
                q = __builtin_elementwise_fma(res, r, q);
                if (p.has_zp) q = q + z;
                k[i] = __builtin_amdgcn_fmed3f(rintf(q.x), m.level_low, m.level_high);
                k[i + 1] = __builtin_amdgcn_fmed3f(rintf(q.y), m.level_low, m.level_high);
            }
        }
    }
    // 8 clipped codes -> one dword of nibbles, assembled in the float domain (exact integers < 2^16 per half)
    __device__ __forceinline__ uint32_t nibbles8(const float* k) const {
        const float p01 = fmaf(k[1], 16.0f, k[0]), p23 = fmaf(k[3], 16.0f, k[2]);
        const float p45 = fmaf(k[5], 16.0f, k[4]), p67 = fmaf(k[7], 16.0f, k[6]);
        const float lo = fmaf(p23, 256.0f, p01) + pack_bias, hi = fmaf(p67, 256.0f, p45) + pack_bias;
        return (uint32_t)lo | ((uint32_t)hi << 16);
    }
    __device__ __forceinline__ uint32_t bytes4(const float* k) const {
        const float off = m.asym ? 0.0f : 128.0f;
        uint32_t w = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) w = __builtin_amdgcn_cvt_pk_u8_f32(k[b] + off, b, w);
        return m.asym ? w : (w ^ 0x80808080u);
    }
    // N consecutive elements starting at element index e0 (N = 4, 8, 16 or 32; e0 % N == 0)
    template <int N, bool NT = false>
    __device__ __forceinline__ void store_codes(const GroupArgs& a, const float* k, int64_t e0) const {
        if (a.packed) {
            if constexpr (N == 4) {
                const float p01 = fmaf(k[1], 16.0f, k[0]), p23 = fmaf(k[3], 16.0f, k[2]);
                *(uint16_t*)(a.packed + e0 / 2) = (uint16_t)(uint32_t)(fmaf(p23, 256.0f, p01) + pack_bias);
            } else {
                uint32_t w[N / 8];
#pragma unroll
                for (int j = 0; j < N / 8; ++j) w[j] = nibbles8(k + 8 * j);
                uint32_t* dst = (uint32_t*)(a.packed + e0 / 2);
                if constexpr (N == 8) st<NT>(dst, w[0]);
                if constexpr (N == 16) st<NT>((u32x2*)dst, u32x2{w[0], w[1]});
                if constexpr (N == 32) st<NT>((u32x4*)dst, u32x4{w[0], w[1], w[2], w[3]});
            }
        }
        if (a.codes) {
            uint32_t* dst = (uint32_t*)((uint8_t*)a.codes + e0);
            if constexpr (N == 4) dst[0] = bytes4(k);
            if constexpr (N == 8) st<NT>((u32x2*)dst, u32x2{bytes4(k), bytes4(k + 4)});
            if constexpr (N >= 16) {
#pragma unroll
                for (int j = 0; j < N / 16; ++j)
                    st<NT>((u32x4*)dst + j, u32x4{bytes4(k + 16 * j), bytes4(k + 16 * j + 4), bytes4(k + 16 * j + 8),
                                                  bytes4(k + 16 * j + 12)});
            }
        }
    }
    template <int N>
    __device__ __forceinline__ void store_dq(const GroupArgs& a, const Par& p, const float* k, int64_t e0) const {
        if (a.dq) {
            float4* o = (float4*)(a.dq + e0);
#pragma unroll
            for (int j = 0; j < N / 4; ++j)
                o[j] = make_float4(dq(0, k[4 * j], p), dq(0, k[4 * j + 1], p), dq(0, k[4 * j + 2], p), dq(0, k[4 * j + 3], p));
        }
    }
    // row kernel: one vector of E elements
    template <int E>
    __device__ __forceinline__ void emit(const GroupArgs& a, const Par& p, const float* f, int64_t v) const {
        float k[E];
        quant_n<E>(p, f, k);
        store_codes<E>(a, k, v * E);
        store_dq<E>(a, p, k, v * E);
    }
    // multi-tensor kernel: NE consecutive elements of one lane -> NE/8 dwords of nibbles (f is clobbered)
    template <int NE, int GPL>
    __device__ __forceinline__ void pack_lane(const Par* par, float* f, uint32_t* out) const {
        constexpr int SG = NE / GPL;
#pragma unroll
        for (int gi = 0; gi < GPL; ++gi) quant_n_div<SG>(par[gi], f + gi * SG, f + gi * SG);
#pragma unroll
        for (int j = 0; j < NE / 8; ++j) out[j] = nibbles8(f + 8 * j);
    }
    // grouped kernel: NE consecutive elements of one lane, covering GPL whole groups (or a part of one)
    template <int NE, int GPL, bool NT, bool DIV = false>
    __device__ __forceinline__ void emit_lane(const GroupArgs& a, const Par* par, float* f, int64_t e0) const {
        constexpr int SG = NE / GPL;
#pragma unroll
        for (int gi = 0; gi < GPL; ++gi) {  // in place; DIV: the parameters were computed from these elements
            if constexpr (DIV) quant_n_div<SG>(par[gi], f + gi * SG, f + gi * SG);
            else quant_n<SG>(par[gi], f + gi * SG, f + gi * SG);
        }
        store_codes<NE, NT>(a, f, e0);
#pragma unroll
        for (int gi = 0; gi < GPL; ++gi) store_dq<SG>(a, par[gi], f + gi * SG, e0 + gi * SG);
    }
};
using IntPolicy = IntPolicyT<-1>;

}  // namespace wc
