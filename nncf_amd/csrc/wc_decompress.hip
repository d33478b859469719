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
    else if (scale_kind == kScaleE8M0) s = __builtin_bit_cast(float, (uint32_t)((const uint8_t*)scale)[g] << 23);  // 2^(byte - 127)
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
    else if (scale_kind == kScaleE8M0) s = __builtin_bit_cast(float, (uint32_t)((const uint8_t*)scale)[g] << 23);
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
    const dim3 grid((unsigned)(((fast ? n / 8 : n) + 255) / 256));
    hipStream_t s = (hipStream_t)stream;
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
