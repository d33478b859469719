// wc_decompress.hip -- the decompressor forward: packed codes -> weights in the layer's dtype, one HBM pass.
// Reference: INT8/INT4 {Asymmetric,Symmetric}WeightsDecompressor.forward (src/nncf/torch/quantization/layers.py
// :1303-1460) = unpack_uint4 / unpack_int4 + decompress_asymmetric / decompress_symmetric
// (src/nncf/torch/quantization/quantize_functions.py:389-473) + reshape + cast: five to eight elementwise torch
// kernels and as many intermediate tensors; here 0.5 (or 1) byte in and 2 (or 4) bytes out per element.
//
// Arithmetic as in the reference: everything happens in the SCALE's dtype (fp16 in the reference's modules):
//     d = code - zero_point (exact small integer), v = RN_scale_dtype(d * scale), out = cast(v, out dtype).
// NF4 / MXFP4 have no torch decompressor in the reference; they follow the same scheme with
// the code book value in place of d (MXFP4: scale = 2^(E8M0 byte - 127), so the product is exact in fp32).
#include "wc_float_policy.h"

namespace wc {

enum { kScaleF32 = 0, kScaleF16 = 1, kScaleE8M0 = 3 };

template <int OUT_DT>
__device__ __forceinline__ void store8(void* out, int64_t e0, const float* v) {
    if constexpr (OUT_DT == NNCF_AMD_F32) {
        float4* o = (float4*)((float*)out + e0);
        __builtin_nontemporal_store(u32x4{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])}, (u32x4*)o);
        __builtin_nontemporal_store(u32x4{__float_as_uint(v[4]), __float_as_uint(v[5]), __float_as_uint(v[6]), __float_as_uint(v[7])}, (u32x4*)(o + 1));
    } else {
        uint32_t w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint32_t lo, hi;
            if constexpr (OUT_DT == NNCF_AMD_F16) {
                lo = __builtin_bit_cast(uint16_t, (_Float16)v[2 * i]);
                hi = __builtin_bit_cast(uint16_t, (_Float16)v[2 * i + 1]);
            } else {
                lo = __builtin_bit_cast(uint16_t, (__bf16)v[2 * i]);
                hi = __builtin_bit_cast(uint16_t, (__bf16)v[2 * i + 1]);
            }
            w[i] = lo | (hi << 16);
        }
        __builtin_nontemporal_store(u32x4{w[0], w[1], w[2], w[3]}, (u32x4*)((uint16_t*)out + e0));
    }
}

// E8M0 scale byte -> 2^(byte - 127): byte 0 is the subnormal 2^-127, byte 255 stays +inf (the format's NaN code is
// never written by the lowering)
__device__ __forceinline__ float e8m0_scale(uint32_t byte) {
    return __builtin_bit_cast(float, byte ? byte << 23 : 0x00400000u);
}

// one thread = 8 consecutive elements of one group (group_size % 8 == 0)
template <int OUT_DT>
__global__ __launch_bounds__(256) void k_decompress8(const uint8_t* __restrict__ codes, int mode, int64_t n_elems, int64_t gs,
                                                     const void* __restrict__ scale, int scale_kind,
                                                     const uint8_t* __restrict__ zp, void* __restrict__ out) {
    const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;  // vector of 8 elements
    const int64_t e0 = v * 8;
    if (e0 >= n_elems) return;
    const int64_t g = e0 / gs;
    float s;
    if (scale_kind == kScaleF16) s = (float)__builtin_bit_cast(_Float16, ((const uint16_t*)scale)[g]);
    else if (scale_kind == kScaleE8M0) s = e8m0_scale(((const uint8_t*)scale)[g]);
    else s = ((const float*)scale)[g];
    float d[8];
    const bool four = mode >= NNCF_AMD_INT4_ASYM;
    if (four) {
        const uint32_t p = *(const uint32_t*)(codes + e0 / 2);
        if (mode == NNCF_AMD_NF4) {
#pragma unroll
            for (int i = 0; i < 8; ++i) d[i] = kNF4Q[(p >> (4 * i)) & 15];
        } else if (mode == NNCF_AMD_MXFP4 || mode == NNCF_AMD_FP4) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const uint32_t c = (p >> (4 * i)) & 15;
                const float mag = (c & 6) == 0 ? 0.5f * (float)(c & 1) : ldexpf(1.0f + 0.5f * (float)(c & 1), (int)((c >> 1) & 3) - 1);
                d[i] = (c & 8) ? -mag : mag;
            }
        } else {
            float z = 8.0f;  // INT4_SYM: codes are stored with a +8 bias (pack_int4)
            if (mode == NNCF_AMD_INT4_ASYM) z = (float)((zp[g >> 1] >> (4 * (g & 1))) & 15);
#pragma unroll
            for (int i = 0; i < 8; ++i) d[i] = (float)((p >> (4 * i)) & 15) - z;
        }
    } else {
        const uint2 p = *(const uint2*)(codes + e0);
        const uint32_t w[2] = {p.x, p.y};
        const float z = mode == NNCF_AMD_INT8_ASYM ? (float)zp[g] : 0.0f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint32_t b = (w[i >> 2] >> (8 * (i & 3))) & 255u;
            d[i] = (mode == NNCF_AMD_INT8_ASYM ? (float)b : (float)(int8_t)b) - z;
        }
    }
    float r[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float prod = d[i] * s;
        if (scale_kind == kScaleF16) {
            if (mode == NNCF_AMD_NF4) prod = (float)(_Float16)d[i] * s;  // the code book value lives in the scale's dtype
            prod = (float)(_Float16)prod;                                  // the product is rounded to the scale's dtype
        }
        r[i] = prod;
    }
    store8<OUT_DT>(out, e0, r);
}

// The modules' own configurations (fp16 scale: INT8 / INT4 sym / asym, NF4; E8M0 scale: MXFP4), mode and output
// dtype fixed at compile time.  The generic kernel above spends ~12 VALU slots per element (a 64-bit division per
// thread, conversions through fp32, a constant-memory table for NF4) and is issue-bound at 3.2-4.5 TB/s; here
//   * the arithmetic of the fp16 configurations is packed fp16 on exact operands -- an integer code c enters as the
//     fp16 number 1024 + c (its bits OR-ed under the exponent 0x6400), d = (1024 + c) - (1024 + z) is exact and
//     RN_fp16(d * s) is what `v_pk_mul_f16` returns (the reference multiplies in fp16 as well) -- two elements per
//     issue slot and no conversion on the way in;
//   * e2m1 codes enter as the fp16 number value * 2^-14 (exponent and mantissa bits shifted into place, zero and
//     the subnormal 0.5 included), the factor goes into the power-of-two scale;
//   * NF4 reads a 16-entry fp16 table in LDS;
//   * the group of a thread's 8 elements comes from a 2-D grid: blockIdx.y counts spans of `span_groups` groups, and
//     inside a span (< 2^24 vectors) the division is a shift or a float multiply with one correction.
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

// what a thread reads for one vector of 8 elements: the codes (p1: INT8 only), the raw scale (fp16 bits or E8M0 byte) and
// the raw zero-point byte of its group
struct HVec {
    uint32_t p0, p1, sraw, zraw;
    uint64_t g, e0;
    bool ok;
};

template <int MODE, int OUT_DT>
__device__ __forceinline__ void decompress_h_vec(const HVec& in, const h2* lut, void* __restrict__ out) {
    float r[8];
    if constexpr (MODE == NNCF_AMD_MXFP4) {
        // scale 2^(byte - 127) times the 2^14 of the code's fp16 image: one exact power of two while it is finite
        // (byte <= 240); the few bytes above take the two factors one after the other
        const uint32_t byte = in.sraw;
        const bool wide = byte > 240u;
        const float s = wide ? e8m0_scale(byte) : __builtin_bit_cast(float, (byte + 14u) << 23);
        const float pre = wide ? 16384.0f : 1.0f;
        const uint32_t p = in.p0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t q = p >> (4 * k);
            const h2 h = __builtin_bit_cast(h2, ((q & 0x00070007u) << 9) | ((q & 0x00080008u) << 12));
            r[k] = (float)h.x * s;
            r[k + 4] = (float)h.y * s;
        }
        if (wide) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t q = p >> (4 * k);
                const h2 h = __builtin_bit_cast(h2, ((q & 0x00070007u) << 9) | ((q & 0x00080008u) << 12));
                r[k] = ((float)h.x * pre) * s;
                r[k + 4] = ((float)h.y * pre) * s;
            }
        }
    } else {
        const _Float16 s = __builtin_bit_cast(_Float16, (uint16_t)in.sraw);
        const h2 s2 = {s, s};
        h2 h[4];
        if constexpr (MODE == NNCF_AMD_NF4) {
            const uint32_t p = in.p0;
#pragma unroll
            for (int k = 0; k < 4; ++k) h[k] = lut[(p >> (8 * k)) & 255u] * s2;  // elements 2k, 2k + 1
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                r[2 * k] = (float)h[k].x;
                r[2 * k + 1] = (float)h[k].y;
            }
        } else if constexpr (MODE == NNCF_AMD_INT4_SYM || MODE == NNCF_AMD_INT4_ASYM) {
            const uint32_t p = in.p0;
            uint32_t z = 8;  // INT4_SYM: codes are stored with a +8 bias (pack_int4)
            if constexpr (MODE == NNCF_AMD_INT4_ASYM) z = (in.zraw >> (4 * (in.g & 1))) & 15;
            const _Float16 zb = __builtin_bit_cast(_Float16, (uint16_t)(0x6400u | z));
            const h2 zb2 = {zb, zb};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const h2 c = __builtin_bit_cast(h2, ((p >> (4 * k)) & 0x000f000fu) | 0x64006400u);  // elements k, k + 4
                h[k] = (c - zb2) * s2;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                r[k] = (float)h[k].x;
                r[k + 4] = (float)h[k].y;
            }
        } else {
            uint32_t w[2] = {in.p0, in.p1};
            uint32_t z = 128;  // INT8_SYM: the sign bit is flipped, code + 128
            if constexpr (MODE == NNCF_AMD_INT8_ASYM) z = in.zraw;
            else {
                w[0] ^= 0x80808080u;
                w[1] ^= 0x80808080u;
            }
            const _Float16 zb = __builtin_bit_cast(_Float16, (uint16_t)(0x6400u | z));
            const h2 zb2 = {zb, zb};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                // bytes (2k & 3), (2k & 3) + 1 of dword k / 2 under the exponent byte 0x64: elements 2k, 2k + 1
                const uint32_t sel = (k & 1) ? 0x03070106u : 0x03050104u;
                const h2 c = __builtin_bit_cast(h2, __builtin_amdgcn_perm(w[k >> 1], 0x64006400u, sel));
                h[k] = (c - zb2) * s2;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                r[2 * k] = (float)h[k].x;
                r[2 * k + 1] = (float)h[k].y;
            }
        }
    }
    store8<OUT_DT>(out, (int64_t)in.e0, r);
}

// ITERS chunks of 256 vectors per block, all of a thread's reads requested before the first is used: with one vector
// per thread (4 bytes in, 16 out) the kernel is bound by the latency of that one read times the resident waves --
// 5.2-5.3 TB/s on a 28672 x 8192 weight (out of the caches; a plain fill of the output reaches 6.9), NF4 4.4 with its
// table prologue (constant loads, LDS writes and a barrier in front of 2048 elements of work).
template <int MODE, int OUT_DT, int ITERS>
__global__ __launch_bounds__(256) void k_decompress_h(const uint8_t* __restrict__ codes, const void* __restrict__ scale,
                                                      const uint8_t* __restrict__ zp, void* __restrict__ out, uint64_t n_vec,
                                                      uint32_t d, uint32_t span_groups, float rcp_d, int shift) {
    // NF4: one LDS entry per code BYTE (the fp16 pair of its two nibbles): one 4-byte read per two elements
    __shared__ h2 lut[MODE == NNCF_AMD_NF4 ? 256 : 1];
    if constexpr (MODE == NNCF_AMD_NF4) {
        lut[threadIdx.x] = h2{(_Float16)kNF4Q[threadIdx.x & 15], (_Float16)kNF4Q[threadIdx.x >> 4]};
        __syncthreads();
    }
    const uint32_t span_vec = span_groups * d;
    HVec in[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const uint32_t vl = (blockIdx.x * ITERS + it) * 256 + threadIdx.x;  // vector of 8 elements inside the span
        uint32_t gl;
        if (shift >= 0) {
            gl = vl >> shift;
        } else {
            gl = (uint32_t)((float)vl * rcp_d);  // within 1 of vl / d (vl < 2^24)
            const uint32_t r = vl - gl * d;
            if ((int32_t)r < 0) --gl;
            else if (r >= d) ++gl;
        }
        const uint64_t v = (uint64_t)blockIdx.y * span_vec + vl;
        HVec& x = in[it];
        x.ok = vl < span_vec && v < n_vec;
        x.g = x.ok ? (uint64_t)blockIdx.y * span_groups + gl : 0;  // (a vector past the end reads vector 0: no branch)
        x.e0 = (x.ok ? v : 0) * 8;
        x.p1 = x.zraw = 0;
        if constexpr (MODE == NNCF_AMD_INT8_ASYM || MODE == NNCF_AMD_INT8_SYM) {
            const uint2 pp = *(const uint2*)(codes + x.e0);
            x.p0 = pp.x, x.p1 = pp.y;
        } else {
            x.p0 = *(const uint32_t*)(codes + x.e0 / 2);
        }
        if constexpr (MODE == NNCF_AMD_MXFP4) x.sraw = ((const uint8_t*)scale)[x.g];
        else x.sraw = ((const uint16_t*)scale)[x.g];
        if constexpr (MODE == NNCF_AMD_INT4_ASYM) x.zraw = zp[x.g >> 1];
        if constexpr (MODE == NNCF_AMD_INT8_ASYM) x.zraw = zp[x.g];
    }
#pragma unroll
    for (int it = 0; it < ITERS; ++it)
        if (in[it].ok) decompress_h_vec<MODE, OUT_DT>(in[it], lut, out);
}

// any group size / alignment: one thread per element (correctness path)
template <int OUT_DT>
__global__ __launch_bounds__(256) void k_decompress1(const uint8_t* __restrict__ codes, int mode, int64_t n_elems, int64_t gs,
                                                     const void* __restrict__ scale, int scale_kind,
                                                     const uint8_t* __restrict__ zp, void* __restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n_elems) return;
    const int64_t g = e / gs;
    float s;
    if (scale_kind == kScaleF16) s = (float)__builtin_bit_cast(_Float16, ((const uint16_t*)scale)[g]);
    else if (scale_kind == kScaleE8M0) s = e8m0_scale(((const uint8_t*)scale)[g]);
    else s = ((const float*)scale)[g];
    float d;
    if (mode >= NNCF_AMD_INT4_ASYM) {
        const uint32_t c = (codes[e >> 1] >> (4 * (e & 1))) & 15u;
        if (mode == NNCF_AMD_NF4) {
            d = kNF4Q[c];
        } else if (mode == NNCF_AMD_MXFP4 || mode == NNCF_AMD_FP4) {
            const float mag = (c & 6) == 0 ? 0.5f * (float)(c & 1) : ldexpf(1.0f + 0.5f * (float)(c & 1), (int)((c >> 1) & 3) - 1);
            d = (c & 8) ? -mag : mag;
        } else {
            const float z = mode == NNCF_AMD_INT4_ASYM ? (float)((zp[g >> 1] >> (4 * (g & 1))) & 15) : 8.0f;
            d = (float)c - z;
        }
    } else {
        d = mode == NNCF_AMD_INT8_ASYM ? (float)codes[e] - (float)zp[g] : (float)(int8_t)codes[e];
    }
    float prod = d * s;
    if (scale_kind == kScaleF16) {
        if (mode == NNCF_AMD_NF4) prod = (float)(_Float16)d * s;
        // (kept from fusing into a multiply-add with the conversion: d * s + 0 loses the sign of a zero product)
        asm volatile("" : "+v"(prod));
        prod = (float)(_Float16)prod;
    }
    if constexpr (OUT_DT == NNCF_AMD_F32) ((float*)out)[e] = prod;
    else if constexpr (OUT_DT == NNCF_AMD_F16) ((_Float16*)out)[e] = (_Float16)prod;
    else ((__bf16*)out)[e] = (__bf16)prod;
}

}  // namespace wc

extern "C" int nncf_amd_decompress(const uint8_t* codes, int mode, int64_t n_groups, int64_t group_size, const void* scale,
                                   int scale_kind, const uint8_t* zero_point, void* out, int out_dtype, void* stream) {
    using namespace wc;
    if (n_groups < 0 || group_size < 1 || (n_groups > 0 && (!codes || !scale || !out))) {
        set_error("decompress: bad argument (n_groups=%lld group_size=%lld)", (long long)n_groups, (long long)group_size);
        return NNCF_AMD_ERR_BAD_ARG;
    }
    if (mode < NNCF_AMD_INT8_ASYM || mode > NNCF_AMD_FP4) {
        set_error("decompress: unsupported mode %d", mode);
        return NNCF_AMD_ERR_MODE;
    }
    if ((mode == NNCF_AMD_INT8_ASYM || mode == NNCF_AMD_INT4_ASYM) && !zero_point) {
        set_error("decompress: asymmetric mode needs the zero points");
        return NNCF_AMD_ERR_BAD_ARG;
    }
    if (scale_kind != kScaleF32 && scale_kind != kScaleF16 && !(scale_kind == kScaleE8M0 && mode == NNCF_AMD_MXFP4)) {
        set_error("decompress: scale kind %d invalid for mode %d (0 fp32, 1 fp16, 3 E8M0 bytes for MXFP4)", scale_kind, mode);
        return NNCF_AMD_ERR_DTYPE;
    }
    if (n_groups == 0) return NNCF_AMD_OK;
    const int64_t n = n_groups * group_size;
    const bool fast = group_size % 8 == 0 && (((uintptr_t)codes | (uintptr_t)out) & 15u) == 0;
    hipStream_t s = (hipStream_t)stream;
    const bool half_cfg = scale_kind == kScaleF16 && (mode <= NNCF_AMD_INT4_SYM || mode == NNCF_AMD_NF4);
    if (fast && (half_cfg || (mode == NNCF_AMD_MXFP4 && scale_kind == kScaleE8M0))) {
        const uint64_t n_vec = (uint64_t)n / 8;
        const uint64_t d = (uint64_t)group_size / 8;                    // vectors per group
        const uint64_t span_groups = d >= 4096 ? 1 : 4096 / d;           // >= 4096 vectors (16 blocks) per span
        const uint64_t spans = ((uint64_t)n_groups + span_groups - 1) / span_groups;
        if (d < ((uint64_t)1 << 23) && spans <= 65535) {
            int shift = -1;
            if ((d & (d - 1)) == 0) {
                shift = 0;
                while (((uint64_t)1 << shift) < d) ++shift;
            }
            const uint64_t chunks = (span_groups * d + 255) / 256;  // of 256 vectors, per span
            // with enough blocks to go round: 4 chunks per block
            const int iters = chunks * spans >= 8192 ? 4 : 1;
            const dim3 grid((unsigned)((chunks + iters - 1) / iters), (unsigned)spans);
#define LAUNCH_H(M, DT)                                                                                                     \
    if (iters == 4)                                                                                                         \
        hipLaunchKernelGGL((k_decompress_h<M, DT, 4>), grid, dim3(256), 0, s, codes, scale, zero_point, out, n_vec,         \
                           (uint32_t)d, (uint32_t)span_groups, 1.0f / (float)d, shift);                                    \
    else                                                                                                                    \
        hipLaunchKernelGGL((k_decompress_h<M, DT, 1>), grid, dim3(256), 0, s, codes, scale, zero_point, out, n_vec,         \
                           (uint32_t)d, (uint32_t)span_groups, 1.0f / (float)d, shift)
#define LAUNCH_HM(DT)                                                                                                       \
    switch (mode) {                                                                                                         \
        case NNCF_AMD_INT8_ASYM: LAUNCH_H(NNCF_AMD_INT8_ASYM, DT); break;                                                    \
        case NNCF_AMD_INT8_SYM: LAUNCH_H(NNCF_AMD_INT8_SYM, DT); break;                                                      \
        case NNCF_AMD_INT4_ASYM: LAUNCH_H(NNCF_AMD_INT4_ASYM, DT); break;                                                    \
        case NNCF_AMD_INT4_SYM: LAUNCH_H(NNCF_AMD_INT4_SYM, DT); break;                                                      \
        case NNCF_AMD_NF4: LAUNCH_H(NNCF_AMD_NF4, DT); break;                                                                \
        default: LAUNCH_H(NNCF_AMD_MXFP4, DT); break;                                                                        \
    }
            switch (out_dtype) {
                case NNCF_AMD_F32: LAUNCH_HM(NNCF_AMD_F32); break;
                case NNCF_AMD_F16: LAUNCH_HM(NNCF_AMD_F16); break;
                case NNCF_AMD_BF16: LAUNCH_HM(NNCF_AMD_BF16); break;
                default:
                    set_error("decompress: unsupported output dtype %d", out_dtype);
                    return NNCF_AMD_ERR_DTYPE;
            }
#undef LAUNCH_HM
#undef LAUNCH_H
            return launch_status("decompress");
        }
    }
    const dim3 grid((unsigned)(((fast ? n / 8 : n) + 255) / 256));
#define LAUNCH(DT)                                                                                                         \
    if (fast)                                                                                                              \
        hipLaunchKernelGGL((k_decompress8<DT>), grid, dim3(256), 0, s, codes, mode, n, group_size, scale, scale_kind,     \
                           zero_point, out);                                                                               \
    else                                                                                                                   \
        hipLaunchKernelGGL((k_decompress1<DT>), grid, dim3(256), 0, s, codes, mode, n, group_size, scale, scale_kind,     \
                           zero_point, out)
    switch (out_dtype) {
        case NNCF_AMD_F32: LAUNCH(NNCF_AMD_F32); break;
        case NNCF_AMD_F16: LAUNCH(NNCF_AMD_F16); break;
        case NNCF_AMD_BF16: LAUNCH(NNCF_AMD_BF16); break;
        default:
            set_error("decompress: unsupported output dtype %d", out_dtype);
            return NNCF_AMD_ERR_DTYPE;
    }
#undef LAUNCH
    return launch_status("decompress");
}
