// wc_misc.hip -- status plumbing, nibble (un)packing, standalone dequantization and the
// integer quantization-error metric.
#include <stdarg.h>
#include <stdio.h>

#include <mutex>

#include "wc_common.h"

namespace wc {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int launch_status(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return NNCF_AMD_ERR_LAUNCH;
    }
    return NNCF_AMD_OK;
}

// ------------------------------------------------------------------ pack / unpack (quantize_functions.py:417-473)
// 16 codes (16 bytes) in -> 8 bytes out per thread step; tail handled by a scalar loop.
template <int BIAS>
__global__ __launch_bounds__(256) void k_pack_u4(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int64_t n_pairs,
                                                 bool vec_ok) {
    const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t nthreads = (int64_t)gridDim.x * 256;
    const int64_t nv = vec_ok ? n_pairs / 8 : 0;
    for (int64_t v = tid; v < nv; v += nthreads) {
        const uint4 x = ((const uint4*)in)[v];
        const uint32_t w[4] = {x.x, x.y, x.z, x.w};
        uint32_t o[2] = {0, 0};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t b0 = ((w[i] & 0xffu) + BIAS) & 15u, b1 = (((w[i] >> 8) & 0xffu) + BIAS) & 15u;
            const uint32_t b2 = (((w[i] >> 16) & 0xffu) + BIAS) & 15u, b3 = ((w[i] >> 24) + BIAS) & 15u;
            const uint32_t two = b0 | (b1 << 4) | (b2 << 8) | (b3 << 12);
            o[i >> 1] |= two << (16 * (i & 1));
        }
        ((uint2*)out)[v] = make_uint2(o[0], o[1]);
    }
    for (int64_t k = nv * 8 + tid; k < n_pairs; k += nthreads) {
        const uint32_t lo = ((uint32_t)in[2 * k] + BIAS) & 15u, hi = ((uint32_t)in[2 * k + 1] + BIAS) & 15u;
        out[k] = (uint8_t)(lo | (hi << 4));
    }
}

template <int BIAS>
__global__ __launch_bounds__(256) void k_unpack_u4(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int64_t n_packed) {
    const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t nthreads = (int64_t)gridDim.x * 256;
    for (int64_t k = tid; k < n_packed; k += nthreads) {
        const uint32_t b = in[k];
        out[2 * k] = (uint8_t)((int)(b & 15u) - BIAS);
        out[2 * k + 1] = (uint8_t)((int)(b >> 4) - BIAS);
    }
}

static inline int grid_for(int64_t work_items) {
    int64_t b = (work_items + 255) / 256;
    if (b < 1) b = 1;
    if (b > 4096) b = 4096;
    return (int)b;
}

template <int BIAS>
static int pack_impl(const uint8_t* in, int64_t n, uint8_t* out, hipStream_t s) {
    if (n < 0 || (n & 1)) {
        set_error("pack: element count %lld must be even and non-negative", (long long)n);
        return NNCF_AMD_ERR_BAD_ARG;
    }
    if (n == 0) return NNCF_AMD_OK;
    if (!in || !out) {
        set_error("pack: NULL buffer");
        return NNCF_AMD_ERR_BAD_ARG;
    }
    const bool vec_ok = (((uintptr_t)in & 15u) == 0) && (((uintptr_t)out & 7u) == 0);
    hipLaunchKernelGGL((k_pack_u4<BIAS>), dim3(grid_for(n / 16 + 1)), dim3(256), 0, s, in, out, n / 2, vec_ok);
    return launch_status("pack_u4");
}

template <int BIAS>
static int unpack_impl(const uint8_t* in, int64_t n_packed, uint8_t* out, hipStream_t s) {
    if (n_packed < 0) {
        set_error("unpack: negative size");
        return NNCF_AMD_ERR_BAD_ARG;
    }
    if (n_packed == 0) return NNCF_AMD_OK;
    if (!in || !out) {
        set_error("unpack: NULL buffer");
        return NNCF_AMD_ERR_BAD_ARG;
    }
    hipLaunchKernelGGL((k_unpack_u4<BIAS>), dim3(grid_for(n_packed)), dim3(256), 0, s, in, out, n_packed);
    return launch_status("unpack_u4");
}

// ------------------------------------------------------------------ standalone dequantization (weight_lowering.py:428-448)
__global__ __launch_bounds__(256) void k_int_dequant(const void* codes, int codes_signed, int64_t n, int64_t gs, int64_t r,
                                                     int64_t inner, const float* scale, const void* zp, int zp_dtype,
                                                     float* out) {
    const int64_t nthreads = (int64_t)gridDim.x * 256;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < n; idx += nthreads) {
        // idx = (o*r + rr)*inner + i  ->  group = (o*(r/gs) + rr/gs)*inner + i
        const int64_t i = idx % inner;
        const int64_t orr = idx / inner;
        const int64_t o = orr / r, rr = orr % r;
        const int64_t g = (o * (r / gs) + rr / gs) * inner + i;
        const float q = codes_signed ? (float)((const int8_t*)codes)[idx] : (float)((const uint8_t*)codes)[idx];
        float v;
        if (zp) {
            const float z = zp_dtype == NNCF_AMD_ZP_I32 ? (float)((const int32_t*)zp)[g] : ((const float*)zp)[g];
            v = (q - z) * scale[g];
        } else {
            v = q * scale[g];
        }
        out[idx] = v;
    }
}

// ------------------------------------------------------------------ error metric (weight_lowering.py:329-374)
// pass 1: one block per channel (o, i): per-group INT Q->DQ and sum of squared differences over the
//         whole reduction axis r -> ws[channel].   pass 2: single block: max(ws)/r or sqrt(sum(ws)).
template <int DT>
__global__ __launch_bounds__(256) void k_err_channel(const void* w, int64_t r, int64_t inner, int64_t gs, int mode,
                                                     float* ws) {
    __shared__ float smem[8];
    const IntMode m = int_mode(mode);
    const int64_t ch = blockIdx.x;  // o*inner + i
    const int64_t o = ch / inner, i = ch % inner;
    const int64_t base = o * r * inner + i;
    float acc = 0.0f;
    for (int64_t g0 = 0; g0 < r; g0 += gs) {
        float lo = __builtin_inff(), hi = -__builtin_inff();
        for (int64_t k = threadIdx.x; k < gs; k += 256) {
            const float f = Elem<DT>::load(w, base + (g0 + k) * inner);
            lo = fminf(lo, f);
            hi = fmaxf(hi, f);
        }
        lo = block_allreduce<4>(lo, OpMin(), smem);
        hi = block_allreduce<4>(hi, OpMax(), smem + 4);
        float scale, zpf;
        int zpi;
        int_params(lo, hi, m, scale, zpf, zpi);
        for (int64_t k = threadIdx.x; k < gs; k += 256) {
            const float f = Elem<DT>::load(w, base + (g0 + k) * inner);
            const float q = int_quant(f, scale, zpf, m.asym, m);
            const float dq = m.asym ? (q - zpf) * scale : q * scale;
            const float d = dq - f;
            acc += d * d;
        }
    }
    acc = block_allreduce<4>(acc, OpAdd(), smem);
    if (threadIdx.x == 0) ws[ch] = acc;
}

__global__ __launch_bounds__(256) void k_err_final(const float* ws, int64_t channels, int64_t r, int reduction, float* out) {
    __shared__ float smem[8];
    float mx = 0.0f, sum = 0.0f;
    for (int64_t c = threadIdx.x; c < channels; c += 256) {
        mx = fmaxf(mx, ws[c]);
        sum += ws[c];
    }
    mx = block_allreduce<4>(mx, OpMax(), smem);
    sum = block_allreduce<4>(sum, OpAdd(), smem + 4);
    if (threadIdx.x == 0) out[0] = reduction == NNCF_AMD_ERR_MAX_MEAN ? mx / (float)r : sqrtf(sum);
}

// ------------------------------------------------------------------ error metric of MANY weights in two launches
// (the mixed-precision scoring loop, mixed_precision.py:161-199, walks the weights one by one).  2-D [rows, cols]
// weights reduced along the columns.  Pass 1: one block per row of any tensor -> ws[global row]; pass 2: one block per
// tensor -> results[i].

template <int DT>
__global__ __launch_bounds__(256) void k_err_rows_batch(ErrBatchArgs a, float* ws) {
    __shared__ float smem[8];
    const IntMode m = int_mode(a.mode);
    const uint32_t grow = blockIdx.x;
    int i = 0;
    while (grow >= a.row_start[i + 1]) ++i;  // block-uniform
    const int64_t r = a.cols[i];
    const int64_t gs = a.gs == -1 ? r : a.gs;
    const int64_t base = (int64_t)(grow - a.row_start[i]) * r;
    const void* w = a.w[i];
    float acc = 0.0f;
    for (int64_t g0 = 0; g0 < r; g0 += gs) {
        float lo = __builtin_inff(), hi = -__builtin_inff();
        for (int64_t k = threadIdx.x; k < gs; k += 256) {
            const float f = Elem<DT>::load(w, base + g0 + k);
            lo = fminf(lo, f);
            hi = fmaxf(hi, f);
        }
        lo = block_allreduce<4>(lo, OpMin(), smem);
        hi = block_allreduce<4>(hi, OpMax(), smem + 4);
        float scale, zpf;
        int zpi;
        int_params(lo, hi, m, scale, zpf, zpi);
        for (int64_t k = threadIdx.x; k < gs; k += 256) {
            const float f = Elem<DT>::load(w, base + g0 + k);
            const float q = int_quant(f, scale, zpf, m.asym, m);
            const float dq = m.asym ? (q - zpf) * scale : q * scale;
            const float d = dq - f;
            acc += d * d;
        }
    }
    acc = block_allreduce<4>(acc, OpAdd(), smem);
    if (threadIdx.x == 0) ws[grow] = acc;
}

__global__ __launch_bounds__(256) void k_err_final_batch(ErrBatchArgs a, const float* ws, float* out) {
    __shared__ float smem[8];
    const int i = blockIdx.x;
    float mx = 0.0f, sum = 0.0f;
    for (uint32_t c = a.row_start[i] + threadIdx.x; c < a.row_start[i + 1]; c += 256) {
        mx = fmaxf(mx, ws[c]);
        sum += ws[c];
    }
    mx = block_allreduce<4>(mx, OpMax(), smem);
    sum = block_allreduce<4>(sum, OpAdd(), smem + 4);
    if (threadIdx.x == 0) out[i] = a.reduction == NNCF_AMD_ERR_MAX_MEAN ? mx / (float)a.cols[i] : sqrtf(sum);
}

}  // namespace wc

using namespace wc;

extern "C" int nncf_amd_version(void) { return 100; }
extern "C" const char* nncf_amd_last_error(void) { return g_err; }
extern "C" const char* nncf_amd_status_string(int st) {
    switch (st) {
        case NNCF_AMD_OK: return "ok";
        case NNCF_AMD_ERR_BAD_ARG: return "bad argument";
        case NNCF_AMD_ERR_GROUP_SIZE: return "channel size not divisible by group size";
        case NNCF_AMD_ERR_DTYPE: return "unsupported dtype";
        case NNCF_AMD_ERR_MODE: return "unsupported mode";
        case NNCF_AMD_ERR_LAUNCH: return "kernel launch failed";
        case NNCF_AMD_ERR_WORKSPACE: return "workspace missing or too small";
        case NNCF_AMD_ERR_UNSUPPORTED: return "unsupported request";
        default: return "unknown status";
    }
}

extern "C" int nncf_amd_pack_u4(const uint8_t* in, int64_t n, uint8_t* out, void* stream) {
    return pack_impl<0>(in, n, out, (hipStream_t)stream);
}
extern "C" int nncf_amd_pack_i4(const int8_t* in, int64_t n, uint8_t* out, void* stream) {
    return pack_impl<8>((const uint8_t*)in, n, out, (hipStream_t)stream);
}
extern "C" int nncf_amd_unpack_u4(const uint8_t* in, int64_t n_packed, uint8_t* out, void* stream) {
    return unpack_impl<0>(in, n_packed, out, (hipStream_t)stream);
}
extern "C" int nncf_amd_unpack_i4(const uint8_t* in, int64_t n_packed, int8_t* out, void* stream) {
    return unpack_impl<8>(in, n_packed, (uint8_t*)out, (hipStream_t)stream);
}

extern "C" int nncf_amd_int_dequantize(const void* codes, int codes_signed, int64_t outer, int64_t r, int64_t inner,
                                       int64_t group_size, const float* scale, const void* zp, int zp_dtype,
                                       float* dq_out, void* stream) {
    if (outer < 0 || r < 0 || inner < 0 || (group_size < 1 && group_size != -1)) {
        set_error("int_dequantize: bad shape");
        return NNCF_AMD_ERR_BAD_ARG;
    }
    const int64_t n = outer * r * inner;
    if (n == 0) return NNCF_AMD_OK;
    const int64_t gs = group_size == -1 ? r : group_size;
    if (r % gs) {
        set_error("Channel size %lld should be divisible by size of group %lld.", (long long)r, (long long)gs);
        return NNCF_AMD_ERR_GROUP_SIZE;
    }
    if (!codes || !scale || !dq_out) {
        set_error("int_dequantize: NULL buffer");
        return NNCF_AMD_ERR_BAD_ARG;
    }
    if (zp && zp_dtype != NNCF_AMD_ZP_I32 && zp_dtype != NNCF_AMD_ZP_F32) {
        set_error("int_dequantize: zp_dtype %d invalid", zp_dtype);
        return NNCF_AMD_ERR_DTYPE;
    }
    hipLaunchKernelGGL(k_int_dequant, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, codes, codes_signed, n, gs, r,
                       inner, scale, zp, zp_dtype, dq_out);
    return launch_status("int_dequant");
}

extern "C" size_t nncf_amd_int_quant_error_workspace(int64_t outer, int64_t inner) {
    return (size_t)(outer * inner) * sizeof(float);
}

// ------------------------------------------------------------------ no-allocation variant of the error metric
// Per-channel partial sums of nncf_amd_int_quant_error_static live in this file-scope device array instead of a
// caller workspace.  Every GPU the library runs on holds its own copy (1 MiB) at its own address until the process
// ends; the host-side name `scratch` is only a handle for hipGetSymbolAddress, never a kernel argument.
constexpr int64_t kStaticErrChannels = NNCF_AMD_STATIC_ERR_MAX_CHANNELS;
static __device__ __align__(256) float scratch[kStaticErrChannels];
constexpr int kScratchDevices = 64;
static float* g_scratch_ptr[kScratchDevices];  // device address of `scratch` on device i (NULL: not asked yet)
static std::mutex g_scratch_mutex;

// device address of `scratch` on the CURRENT device, resolved once per device
static float* scratch_on_current_device() {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kScratchDevices) {
        set_error("int_quant_error_static: no current device or device index %d beyond %d", dev, kScratchDevices - 1);
        return nullptr;
    }
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[dev]) {
        float* ptr = nullptr;
        const hipError_t e = hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !ptr) {
            set_error("int_quant_error_static: hipGetSymbolAddress failed: %s", hipGetErrorString(e));
            return nullptr;
        }
        g_scratch_ptr[dev] = ptr;
    }
    return g_scratch_ptr[dev];
}

extern "C" int nncf_amd_int_quant_error(const void* weight, int w_dtype, int64_t outer, int64_t r, int64_t inner,
                                        int64_t group_size, int mode, int reduction, float* result, void* workspace,
                                        size_t workspace_bytes, void* stream);

extern "C" int nncf_amd_int_quant_error_static(const void* weight, int w_dtype, int64_t outer, int64_t r, int64_t inner,
                                               int64_t group_size, int mode, int reduction, float* result,
                                               void* stream) {
    if (outer < 1 || inner < 1 || outer > kStaticErrChannels || inner > kStaticErrChannels ||
        outer * inner > kStaticErrChannels) {
        set_error("int_quant_error_static: %lld x %lld channels; this entry point takes 1 .. %lld (use "
                  "nncf_amd_int_quant_error with a workspace beyond that)", (long long)outer, (long long)inner,
                  (long long)kStaticErrChannels);
        return NNCF_AMD_ERR_WORKSPACE;
    }
    float* ptr = scratch_on_current_device();
    if (!ptr) return NNCF_AMD_ERR_LAUNCH;
    // every channel entry that pass 2 reads is written by pass 1 of the same call: nothing to clear
    return nncf_amd_int_quant_error(weight, w_dtype, outer, r, inner, group_size, mode, reduction, result, ptr,
                                    (size_t)kStaticErrChannels * sizeof(float), stream);
}

extern "C" int nncf_amd_int_quant_error(const void* weight, int w_dtype, int64_t outer, int64_t r, int64_t inner,
                                        int64_t group_size, int mode, int reduction, float* result, void* workspace,
                                        size_t workspace_bytes, void* stream) {
    if (mode < NNCF_AMD_INT8_ASYM || mode > NNCF_AMD_INT4_SYM) {
        set_error("int_quant_error: mode %d is not an integer mode", mode);
        return NNCF_AMD_ERR_MODE;
    }
    if (reduction != NNCF_AMD_ERR_MAX_MEAN && reduction != NNCF_AMD_ERR_FROBENIUS) {
        set_error("Unsupported aggregation mode: %d.", reduction);
        return NNCF_AMD_ERR_BAD_ARG;
    }
    if (outer < 1 || r < 1 || inner < 1 || !weight || !result || (group_size < 1 && group_size != -1)) {
        set_error("int_quant_error: bad argument");
        return NNCF_AMD_ERR_BAD_ARG;
    }
    const int64_t gs = group_size == -1 ? r : group_size;
    if (r % gs) {
        set_error("Channel size %lld should be divisible by size of group %lld.", (long long)r, (long long)gs);
        return NNCF_AMD_ERR_GROUP_SIZE;
    }
    const int64_t channels = outer * inner;
    if (!workspace || workspace_bytes < (size_t)channels * sizeof(float)) {
        set_error("int_quant_error: workspace needs %lld bytes", (long long)(channels * sizeof(float)));
        return NNCF_AMD_ERR_WORKSPACE;
    }
    if (channels > 0x7fffffff) {
        set_error("int_quant_error: too many channels");
        return NNCF_AMD_ERR_BAD_ARG;
    }
    hipStream_t s = (hipStream_t)stream;
    float* ws = (float*)workspace;
    if (w_dtype != NNCF_AMD_F32 && w_dtype != NNCF_AMD_F16 && w_dtype != NNCF_AMD_BF16) {
        set_error("int_quant_error: unsupported weight dtype %d", w_dtype);
        return NNCF_AMD_ERR_DTYPE;
    }
    if (inner == 1 && r <= 0x7fffffff) {  // a [rows, cols] weight: the tile / row kernel when its shape conditions hold
        ErrBatchArgs a = {};
        a.n = 1;
        a.mode = mode;
        a.reduction = reduction;
        a.gs = group_size;
        a.row_start[0] = 0;
        a.row_start[1] = (uint32_t)channels;
        a.cols[0] = (uint32_t)r;
        a.w[0] = weight;
        if (err_tiles_launch(w_dtype, a, ws, s) || err_rows_launch(w_dtype, a, ws, s)) {
            int st = launch_status("err_tiles");
            if (st != NNCF_AMD_OK) return st;
            hipLaunchKernelGGL(k_err_final, dim3(1), dim3(256), 0, s, (const float*)ws, channels, r, reduction, result);
            return launch_status("err_final");
        }
    }
    switch (w_dtype) {
        case NNCF_AMD_F32:
            hipLaunchKernelGGL((k_err_channel<NNCF_AMD_F32>), dim3((unsigned)channels), dim3(256), 0, s, weight, r, inner, gs, mode, ws);
            break;
        case NNCF_AMD_F16:
            hipLaunchKernelGGL((k_err_channel<NNCF_AMD_F16>), dim3((unsigned)channels), dim3(256), 0, s, weight, r, inner, gs, mode, ws);
            break;
        case NNCF_AMD_BF16:
            hipLaunchKernelGGL((k_err_channel<NNCF_AMD_BF16>), dim3((unsigned)channels), dim3(256), 0, s, weight, r, inner, gs, mode, ws);
            break;
        default:
            set_error("int_quant_error: unsupported weight dtype %d", w_dtype);
            return NNCF_AMD_ERR_DTYPE;
    }
    int st = launch_status("err_channel");
    if (st != NNCF_AMD_OK) return st;
    hipLaunchKernelGGL(k_err_final, dim3(1), dim3(256), 0, s, (const float*)ws, channels, r, reduction, result);
    return launch_status("err_final");
}

extern "C" size_t nncf_amd_int_quant_error_batched_workspace(int n, const int64_t* rows_host) {
    size_t total = 0;
    for (int i = 0; i < n; ++i) total += rows_host && rows_host[i] > 0 ? (size_t)rows_host[i] : 0;
    return total * sizeof(float);
}

extern "C" int nncf_amd_int_quant_error_batched(int n, const void* const* weights_host, int w_dtype,
                                                const int64_t* rows_host, const int64_t* cols_host, int64_t group_size,
                                                int mode, int reduction, float* results, void* workspace,
                                                size_t workspace_bytes, void* stream) {
    if (mode < NNCF_AMD_INT8_ASYM || mode > NNCF_AMD_INT4_SYM) {
        set_error("int_quant_error_batched: mode %d is not an integer mode", mode);
        return NNCF_AMD_ERR_MODE;
    }
    if (reduction != NNCF_AMD_ERR_MAX_MEAN && reduction != NNCF_AMD_ERR_FROBENIUS) {
        set_error("Unsupported aggregation mode: %d.", reduction);
        return NNCF_AMD_ERR_BAD_ARG;
    }
    if (n < 0 || (n > 0 && (!weights_host || !rows_host || !cols_host || !results)) || (group_size < 1 && group_size != -1)) {
        set_error("int_quant_error_batched: bad argument");
        return NNCF_AMD_ERR_BAD_ARG;
    }
    if (w_dtype != NNCF_AMD_F32 && w_dtype != NNCF_AMD_F16 && w_dtype != NNCF_AMD_BF16) {
        set_error("int_quant_error_batched: unsupported weight dtype %d", w_dtype);
        return NNCF_AMD_ERR_DTYPE;
    }
    for (int i = 0; i < n; ++i) {
        if (!weights_host[i] || rows_host[i] < 1 || cols_host[i] < 1 || cols_host[i] > 0x7fffffff) {
            set_error("int_quant_error_batched: tensor %d: NULL or empty", i);
            return NNCF_AMD_ERR_BAD_ARG;
        }
        if (group_size != -1 && cols_host[i] % group_size) {
            set_error("Channel size %lld should be divisible by size of group %lld.", (long long)cols_host[i],
                      (long long)group_size);
            return NNCF_AMD_ERR_GROUP_SIZE;
        }
    }
    if (workspace_bytes < nncf_amd_int_quant_error_batched_workspace(n, rows_host) || (n > 0 && !workspace)) {
        set_error("int_quant_error_batched: workspace needs %lld bytes",
                  (long long)nncf_amd_int_quant_error_batched_workspace(n, rows_host));
        return NNCF_AMD_ERR_WORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    float* ws = (float*)workspace;
    for (int first = 0; first < n;) {
        ErrBatchArgs a = {};
        a.mode = mode;
        a.reduction = reduction;
        a.gs = group_size;
        uint64_t acc = 0;
        int j = 0;
        for (; j < kErrBatch && first + j < n; ++j) {
            if (acc + (uint64_t)rows_host[first + j] > 0x7fffffffull) break;  // 32-bit row ids / grid
            a.row_start[j] = (uint32_t)acc;
            a.cols[j] = (uint32_t)cols_host[first + j];
            a.w[j] = weights_host[first + j];
            acc += (uint64_t)rows_host[first + j];
        }
        if (j == 0) {
            set_error("int_quant_error_batched: tensor %d has too many rows", first);
            return NNCF_AMD_ERR_UNSUPPORTED;
        }
        a.row_start[j] = (uint32_t)acc;
        a.n = j;
        const dim3 grid((unsigned)acc);
        if (err_tiles_launch(w_dtype, a, ws, s)) {  // grouped modes on 16-byte rows: the tile kernel (wc_int_err.hip)
        } else if (err_rows_launch(w_dtype, a, ws, s)) {  // whole-row groups: the row kernel without its stores
        } else
        switch (w_dtype) {
            case NNCF_AMD_F32: hipLaunchKernelGGL((k_err_rows_batch<NNCF_AMD_F32>), grid, dim3(256), 0, s, a, ws); break;
            case NNCF_AMD_F16: hipLaunchKernelGGL((k_err_rows_batch<NNCF_AMD_F16>), grid, dim3(256), 0, s, a, ws); break;
            default: hipLaunchKernelGGL((k_err_rows_batch<NNCF_AMD_BF16>), grid, dim3(256), 0, s, a, ws); break;
        }
        int st = launch_status("err_rows_batch");
        if (st != NNCF_AMD_OK) return st;
        hipLaunchKernelGGL(k_err_final_batch, dim3((unsigned)j), dim3(256), 0, s, a, (const float*)ws, results + first);
        st = launch_status("err_final_batch");
        if (st != NNCF_AMD_OK) return st;
        ws += acc;
        first += j;
    }
    return NNCF_AMD_OK;
}
