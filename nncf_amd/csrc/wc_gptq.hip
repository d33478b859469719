// wc_gptq.hip -- GPTQ in-block column sweep for gfx950 (gptq.py:298-361, one block of <= 128 columns).
//
// Rows of the weight are independent in GPTQ: the quantization error of W[r, c] only propagates along row r
// (W[r, c:] -= err * Hinv[c, c:]).  One lane owns one row: its slice of the current block lives in LDS,
// column-major [128][64] so that the 64 lanes of the wave touch 64 consecutive floats (conflict free); the
// Hinv row is wave-uniform and comes through the scalar cache.  The ~K sequential Python iterations x ~10 kernel
// launches of the reference become K/128 launches; the trailing update W[:, i2:] -= Err @ Hinv[i1:i2, i2:]
// (gptq.py:363) stays a plain GEMM at the caller.
//
// Numerics follow the reference step by step: group parameters are computed from `weight` (the tensor WITHOUT
// this block's in-block updates, gptq.py:315-333) with the same policy arithmetic as the lowering kernels, the
// column is quantize-dequantized with the IEEE divide, err = (w - q) / d, and every update is
// w = w - fl(err * h) (separately rounded product and difference, as matmul of a [N,1] by a [1,M] gives).
#include "wc_float_policy.h"
#include "wc_int_policy.h"

namespace wc {

constexpr int GB = 128;  // max columns per call
constexpr int GR = 64;   // rows per workgroup (one wave)

template <typename Policy>
__global__ __launch_bounds__(GR) void k_gptq_block(const float* __restrict__ weight, float* __restrict__ quantized,
                                                   float* __restrict__ err_block, const float* __restrict__ hinv,
                                                   int64_t rows, int64_t K, int64_t i1, int count, int64_t gs, int mode,
                                                   float* scales, int32_t* zps, int given) {
    __shared__ float wb[GB][GR];
    __shared__ float lds[kPolicyLds];
    Policy::init_shared(lds);
    __syncthreads();
    GroupArgs a = {};
    a.mode = mode;
    a.scale_in = scales;
    a.zp_in = zps;
    a.zp_dtype = NNCF_AMD_ZP_I32;
    a.scale_out = scales;
    a.zp_out = zps;
    const Policy pol(a, lds);
    const int tid = threadIdx.x;
    const int64_t r = (int64_t)blockIdx.x * GR + tid;
    if (r >= rows) return;  // no block-wide synchronisation below
    const int64_t ngroups = K / gs;
    const float* wrow = weight + r * K;
    for (int j = 0; j < count; ++j) wb[j][tid] = wrow[i1 + j];
    typename Policy::Par par;
    if (i1 % gs != 0) par = pol.load(a, r * ngroups + i1 / gs);  // the group began in an earlier block
    for (int i = 0; i < count; ++i) {
        const int64_t c = i1 + i;
        if (c % gs == 0) {
            if (given) {  // parameters supplied by the caller (scale estimation inside GPTQ, gptq.py:319-327)
                par = pol.load(a, r * ngroups + c / gs);
            } else {
                typename Policy::Stat st = pol.stat_init();
                for (int64_t k = 0; k < gs; ++k) pol.stat_add(st, wrow[c + k]);
                par = pol.compute(st);
                pol.template store<false>(a, r * ngroups + c / gs, par);
            }
        }
        const float w = wb[i][tid];
        int code;
        float val;
        pol.quant(w, par, code, val);
        const float q = pol.dq(code, val, par);
        quantized[r * K + c] = q;
        const float* hrow = hinv + c * K + i1;  // wave-uniform
        const float err = (w - q) / hrow[i];
        for (int j = i; j < count; ++j) {
            const float prod = err * hrow[j];
            wb[j][tid] = wb[j][tid] - prod;
        }
        err_block[r * count + i] = err;
    }
}

}  // namespace wc

extern "C" int nncf_amd_gptq_block(const float* weight, float* quantized, float* err_block, const float* hinv,
                                   int64_t rows, int64_t k, int64_t i1, int64_t count, int64_t group_size, int mode,
                                   float* scales, int32_t* zps, int use_given_params, void* stream) {
    using namespace wc;
    if (!weight || !quantized || !err_block || !hinv || !scales || rows <= 0 || k <= 0 || i1 < 0 || count <= 0 ||
        count > GB || i1 + count > k || (group_size < 1 && group_size != -1)) {
        set_error("gptq_block: bad argument (rows=%lld k=%lld i1=%lld count=%lld group_size=%lld)", (long long)rows,
                  (long long)k, (long long)i1, (long long)count, (long long)group_size);
        return NNCF_AMD_ERR_BAD_ARG;
    }
    const int64_t gs = group_size == -1 ? k : group_size;
    if (k % gs != 0) {
        set_error("Channel size %lld should be divisible by size of group %lld.", (long long)k, (long long)gs);
        return NNCF_AMD_ERR_GROUP_SIZE;
    }
    const bool asym = mode == NNCF_AMD_INT4_ASYM || mode == NNCF_AMD_INT8_ASYM;
    if (asym && !zps) {
        set_error("gptq_block: asymmetric mode needs the zps buffer");
        return NNCF_AMD_ERR_BAD_ARG;
    }
    if (!asym) zps = nullptr;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)((rows + GR - 1) / GR));
    if (mode >= NNCF_AMD_INT8_ASYM && mode <= NNCF_AMD_INT4_SYM) {
        hipLaunchKernelGGL((k_gptq_block<IntPolicy>), grid, dim3(GR), 0, s, weight, quantized, err_block, hinv, rows, k, i1,
                           (int)count, gs, mode, scales, zps, use_given_params);
    } else if (mode == NNCF_AMD_NF4) {
        hipLaunchKernelGGL((k_gptq_block<FloatPolicy<true>>), grid, dim3(GR), 0, s, weight, quantized, err_block, hinv, rows,
                           k, i1, (int)count, gs, mode, scales, zps, use_given_params);
    } else if (mode == NNCF_AMD_MXFP4 || mode == NNCF_AMD_FP4) {
        hipLaunchKernelGGL((k_gptq_block<FloatPolicy<false>>), grid, dim3(GR), 0, s, weight, quantized, err_block, hinv, rows,
                           k, i1, (int)count, gs, mode, scales, zps, use_given_params);
    } else {
        set_error("gptq_block: unsupported mode %d", mode);
        return NNCF_AMD_ERR_MODE;
    }
    return launch_status("gptq_block");
}
