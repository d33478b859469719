// wc_float_quant.hip -- fused NF4 / MXFP4 / FP4 (e2m1) codebook lowering for gfx950 (policy + C entry point).
//
// group abs-max -> scale (/6 for e2m1; E8M0 power of two for MXFP4) -> w/scale -> nearest quantile.
// The quantile search is a bucket LUT in LDS: floor(n*K) selects a bucket that contains at most one
// decision boundary (NF4: K=16 over [-1,1]; e2m1: K=2 over [-6.5,6.5]), so
//     idx = base[bucket] + (n > center[bucket])
// equals searchsorted(centers, n, side='left') of weight_lowering.py:574 for every float n, and the e2m1
// tie rule of weight_lowering.py:577-583 is then applied literally on the two neighbouring quantiles.
#include <vector>

#include "wc_float_policy.h"

namespace wc {
static int run_float_batch(int w_dtype, const BatchItem* items, int n, int vpg, int mode, const char* what, hipStream_t s) {
    if (mode == NNCF_AMD_NF4) return run_batch<FloatPolicy<true>>(w_dtype, items, n, vpg, mode, what, s);
    if (mode == NNCF_AMD_MXFP4) return run_batch<FloatPolicy<false, NNCF_AMD_MXFP4>>(w_dtype, items, n, vpg, mode, what, s);
    return run_batch<FloatPolicy<false>>(w_dtype, items, n, vpg, mode, what, s);
}
}  // namespace wc

extern "C" int nncf_amd_float_quantize(const void* weight, int w_dtype, int64_t outer, int64_t r, int64_t inner,
                                       int64_t group_size, int mode, const float* scale_in, float* values_out,
                                       uint8_t* codes_out, uint8_t* packed_out, float* scale_out,
                                       uint8_t* scale_e8m0_out, float* dq_out, void* stream) {
    using namespace wc;
    if (mode != NNCF_AMD_NF4 && mode != NNCF_AMD_MXFP4 && mode != NNCF_AMD_FP4) {
        set_error("float_quantize: mode %d is not a float codebook mode", mode);
        return NNCF_AMD_ERR_MODE;
    }
    int64_t gs;
    int st = check_group_shape("float_quantize", weight, outer, r, inner, group_size, gs);
    if (st != NNCF_AMD_OK) return st > 0 ? NNCF_AMD_OK : st;
    if (packed_out && ((outer * r * inner) & 1)) {
        set_error("float_quantize: packed_out needs an even number of elements");
        return NNCF_AMD_ERR_BAD_ARG;
    }
    if (scale_e8m0_out && mode != NNCF_AMD_MXFP4) {
        set_error("float_quantize: scale_e8m0_out is for MXFP4 only");
        return NNCF_AMD_ERR_BAD_ARG;
    }
    // data-free packed compression of one tensor: the mode-specialised multi-tensor kernel with one item
    if (inner == 1 && packed_out && scale_out && !scale_in && !codes_out && !values_out && !dq_out) {
        BatchItem it = {weight, outer, r, packed_out, scale_out, (void*)scale_e8m0_out};
        int vpg;
        if (batch_fits<FloatPolicy<true>>(w_dtype, &it, 1, group_size, &vpg))
            return run_float_batch(w_dtype, &it, 1, vpg, mode, "float_quantize", (hipStream_t)stream);
    }
    GroupArgs a = {};
    a.w = weight;
    a.scale_in = scale_in;
    a.codes = codes_out;
    a.packed = packed_out;
    a.values = values_out;
    a.scale_out = scale_out;
    a.e8m0_out = scale_e8m0_out;
    a.dq = dq_out;
    a.gs = gs;
    a.inner = inner;
    a.n_groups = outer * (r / gs) * inner;
    a.mode = mode;
    if (mode == NNCF_AMD_NF4) return run_group_op_dtype<FloatPolicy<true>>(w_dtype, a, 0, "float_quantize", (hipStream_t)stream);
    return run_group_op_dtype<FloatPolicy<false>>(w_dtype, a, 0, "float_quantize", (hipStream_t)stream);
}

extern "C" int nncf_amd_float_quantize_batched(int n, const void* const* weights_host, int w_dtype,
                                               const int64_t* rows_host, const int64_t* cols_host,
                                               int64_t group_size, int mode, uint8_t* const* packed_out_host,
                                               float* const* scale_out_host, uint8_t* const* scale_e8m0_out_host,
                                               void* stream) {
    using namespace wc;
    if (n < 0 || (n > 0 && (!weights_host || !rows_host || !cols_host || !packed_out_host || !scale_out_host))) {
        set_error("float_quantize_batched: bad descriptor arrays");
        return NNCF_AMD_ERR_BAD_ARG;
    }
    if (mode != NNCF_AMD_NF4 && mode != NNCF_AMD_MXFP4 && mode != NNCF_AMD_FP4) {
        set_error("float_quantize_batched: mode %d is not a float codebook mode", mode);
        return NNCF_AMD_ERR_MODE;
    }
    if (scale_e8m0_out_host && mode != NNCF_AMD_MXFP4) {
        set_error("float_quantize_batched: scale_e8m0_out is for MXFP4 only");
        return NNCF_AMD_ERR_BAD_ARG;
    }
    if (n > 0 && (w_dtype == NNCF_AMD_F32 || w_dtype == NNCF_AMD_F16 || w_dtype == NNCF_AMD_BF16)) {
        std::vector<BatchItem> items(n);
        for (int i = 0; i < n; ++i)
            items[i] = {weights_host[i], rows_host[i], cols_host[i], packed_out_host[i], scale_out_host[i],
                        scale_e8m0_out_host ? (void*)scale_e8m0_out_host[i] : nullptr};
        int vpg;
        if (batch_fits<FloatPolicy<true>>(w_dtype, items.data(), n, group_size, &vpg)) {
            return run_float_batch(w_dtype, items.data(), n, vpg, mode, "float_quantize_batched", (hipStream_t)stream);
        }
    }
    for (int i = 0; i < n; ++i) {
        int st = nncf_amd_float_quantize(weights_host[i], w_dtype, rows_host[i], cols_host[i], 1, group_size, mode, nullptr,
                                         nullptr, nullptr, packed_out_host[i], scale_out_host[i],
                                         scale_e8m0_out_host ? scale_e8m0_out_host[i] : nullptr, nullptr, stream);
        if (st != NNCF_AMD_OK) return st;
    }
    return NNCF_AMD_OK;
}
