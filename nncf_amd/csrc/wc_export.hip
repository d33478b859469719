// wc_export.hip -- export of per-channel INT8 weights into the GPTQ v2 tensor layout, and the forward of that layout.
// Reference: convert_linear_module (src/nncf/experimental/torch/gptqmodel/convertor.py:35-84) decompresses the
// weight, hands it to gptqmodel's TritonV2QuantLinear.pack on the CPU (which re-derives the integer codes column by
// column with round((W + zero * scale) / scale) and then ORs them into int32 words in numpy) and moves the module
// back.  gptqmodel is a third-party dependency that is absent here (tests/integration/gptq_model/requirements_extra.txt
// pins gptqmodel==2.2.0); the layout restated: bits = 8, group_size = -1, desc_act = False,
//     qweight[in / 4, out] int32 : byte j of word (i, o) = code of weight[o, 4 i + j]
//     qzeros [1, out / 4]  int32 : byte j of word c      = zero point of channel 4 c + j   (v2: stored as it is)
//     scales [1, out]      fp16,  g_idx[in] int32 = 0.
// The codes are already in HBM, so nothing is re-derived: a row of `codes` read as 32-bit words IS the column of
// qweight, i.e. the pack is one 32-bit transpose [out, in / 4] -> [in / 4, out] through LDS (signed codes get their
// sign bit flipped: code + 128 with zero point 128, the symmetric convention of the GPTQ formats).
// The forward (`k_gptq_dequant8`) is the inverse transpose with (byte - zero) * scale in fp16 arithmetic.
#include "wc_common.h"

namespace wc {

// 64 x 64 words per block, 256 threads; tile[r][c] padded to 65 columns: both phases are conflict free
__global__ __launch_bounds__(256) void k_transpose32(const uint32_t* __restrict__ in, uint32_t* __restrict__ out,
                                                     int64_t rows, int64_t cols, uint32_t flip) {
    __shared__ uint32_t tile[64][65];
    const int64_t r0 = (int64_t)blockIdx.y * 64, c0 = (int64_t)blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int64_t r = r0 + ty + 4 * i, c = c0 + tx;
        if (r < rows && c < cols) tile[ty + 4 * i][tx] = in[r * cols + c] ^ flip;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int64_t c = c0 + ty + 4 * i, r = r0 + tx;
        if (r < rows && c < cols) out[c * rows + r] = tile[tx][ty + 4 * i];
    }
}

// qweight[in / 4, out] -> weight[out, in]: the block reads 64 words of 64 k-rows and writes rows of 256 elements
template <int OUT_DT>
__global__ __launch_bounds__(256) void k_gptq_dequant8(const uint32_t* __restrict__ qweight, const uint8_t* __restrict__ qzeros,
                                                       const uint16_t* __restrict__ scales, int64_t in4, int64_t out_f,
                                                       void* __restrict__ w) {
    __shared__ uint32_t tile[64][65];
    const int64_t k0 = (int64_t)blockIdx.y * 64, o0 = (int64_t)blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int64_t k = k0 + ty + 4 * i, o = o0 + tx;
        if (k < in4 && o < out_f) tile[ty + 4 * i][tx] = qweight[k * out_f + o];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int64_t o = o0 + ty + 4 * i, k = k0 + tx;
        if (o >= out_f || k >= in4) continue;
        const uint32_t word = tile[tx][ty + 4 * i];
        const int z = qzeros[o];
        const _Float16 s = __builtin_bit_cast(_Float16, scales[o]);
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const _Float16 d = (_Float16)(int)(((word >> (8 * j)) & 255u) - z);  // |d| <= 255: exact in fp16
            v[j] = (float)(_Float16)(d * s);
        }
        const int64_t e = o * in4 * 4 + k * 4;
        if constexpr (OUT_DT == NNCF_AMD_F32) {
            *(float4*)((float*)w + e) = float4{v[0], v[1], v[2], v[3]};
        } else {
            uint32_t p[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                uint32_t lo, hi;
                if constexpr (OUT_DT == NNCF_AMD_F16) {
                    lo = __builtin_bit_cast(uint16_t, (_Float16)v[2 * j]);
                    hi = __builtin_bit_cast(uint16_t, (_Float16)v[2 * j + 1]);
                } else {
                    lo = __builtin_bit_cast(uint16_t, (__bf16)v[2 * j]);
                    hi = __builtin_bit_cast(uint16_t, (__bf16)v[2 * j + 1]);
                }
                p[j] = lo | (hi << 16);
            }
            *(uint2*)((uint16_t*)w + e) = uint2{p[0], p[1]};
        }
    }
}

static int check_shape(const char* who, int64_t out_f, int64_t in_f) {
    if (out_f < 0 || in_f < 0) {
        set_error("%s: bad shape", who);
        return NNCF_AMD_ERR_BAD_ARG;
    }
    if (in_f % 4 || out_f % 4) {
        set_error("%s: in_features %lld and out_features %lld must be multiples of 4 (8-bit codes in 32-bit words)", who,
                  (long long)in_f, (long long)out_f);
        return NNCF_AMD_ERR_UNSUPPORTED;
    }
    if ((in_f / 4 + 63) / 64 > 65535) {
        set_error("%s: in_features %lld too large", who, (long long)in_f);
        return NNCF_AMD_ERR_UNSUPPORTED;
    }
    return NNCF_AMD_OK;
}

}  // namespace wc

using namespace wc;

extern "C" int nncf_amd_gptq_pack8(const void* codes, int codes_signed, int64_t out_features, int64_t in_features,
                                   int32_t* qweight, void* stream) {
    if (int st = check_shape("gptq_pack8", out_features, in_features)) return st;
    if (out_features == 0 || in_features == 0) return NNCF_AMD_OK;
    if (!codes || !qweight || ((uintptr_t)codes & 3)) {
        set_error("gptq_pack8: NULL or unaligned buffer");
        return NNCF_AMD_ERR_BAD_ARG;
    }
    const int64_t in4 = in_features / 4;
    dim3 grid((unsigned)((in4 + 63) / 64), (unsigned)((out_features + 63) / 64));
    if (grid.y > 65535) {
        set_error("gptq_pack8: out_features %lld too large", (long long)out_features);
        return NNCF_AMD_ERR_UNSUPPORTED;
    }
    hipLaunchKernelGGL(k_transpose32, grid, dim3(256), 0, (hipStream_t)stream, (const uint32_t*)codes, (uint32_t*)qweight,
                       out_features, in4, codes_signed ? 0x80808080u : 0u);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("gptq_pack8: %s", hipGetErrorString(e));
        return NNCF_AMD_ERR_LAUNCH;
    }
    return NNCF_AMD_OK;
}

extern "C" int nncf_amd_gptq_dequant8(const int32_t* qweight, const int32_t* qzeros, const void* scales,
                                      int64_t in_features, int64_t out_features, void* weight, int out_dtype,
                                      void* stream) {
    if (int st = check_shape("gptq_dequant8", out_features, in_features)) return st;
    if (out_dtype != NNCF_AMD_F32 && out_dtype != NNCF_AMD_F16 && out_dtype != NNCF_AMD_BF16) {
        set_error("gptq_dequant8: out_dtype %d invalid", out_dtype);
        return NNCF_AMD_ERR_DTYPE;
    }
    if (out_features == 0 || in_features == 0) return NNCF_AMD_OK;
    if (!qweight || !qzeros || !scales || !weight) {
        set_error("gptq_dequant8: NULL buffer");
        return NNCF_AMD_ERR_BAD_ARG;
    }
    const int64_t in4 = in_features / 4;
    dim3 grid((unsigned)((out_features + 63) / 64), (unsigned)((in4 + 63) / 64));
    auto launch = [&](auto kern) {
        hipLaunchKernelGGL(kern, grid, dim3(256), 0, (hipStream_t)stream, (const uint32_t*)qweight, (const uint8_t*)qzeros,
                           (const uint16_t*)scales, in4, out_features, weight);
    };
    if (out_dtype == NNCF_AMD_F32) launch(k_gptq_dequant8<NNCF_AMD_F32>);
    else if (out_dtype == NNCF_AMD_F16) launch(k_gptq_dequant8<NNCF_AMD_F16>);
    else launch(k_gptq_dequant8<NNCF_AMD_BF16>);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("gptq_dequant8: %s", hipGetErrorString(e));
        return NNCF_AMD_ERR_LAUNCH;
    }
    return NNCF_AMD_OK;
}
