// wc_int_quant.hip -- fused INT4/INT8 group-wise lowering for gfx950 (policy + C entry points).
//
// One pass over the weight: 16-byte coalesced loads, wave-64 DPP min/max per group, scale / zero point,
// IEEE divide -> add -> round-half-even -> clip, and the stores the caller asked for (unpacked codes,
// nibble-packed codes, scale, zero point, dequantized fp32).  HBM-bound by construction: every input byte
// is read once, every output byte written once.  Kernel skeletons: wc_group_kernels.h.
#include <vector>

#include "wc_int_policy.h"

namespace wc {

// wc_int_rows.hip: whole-row groups (per-channel) over many tensors
bool int_rows_fit(int w_dtype, const BatchItem* items, int n, int64_t group_size);
int int_rows_run(int w_dtype, const BatchItem* items, int n, int mode, const char* what, hipStream_t s);

static int int_quantize_impl(const void* weight, int w_dtype, int64_t outer, int64_t r, int64_t inner,
                             int64_t group_size, int mode, const float* scale_in, const void* zp_in, int zp_dtype,
                             void* codes_out, uint8_t* packed_out, float* scale_out, int32_t* zp_out, float* dq_out,
                             hipStream_t s) {
    if (mode < NNCF_AMD_INT8_ASYM || mode > NNCF_AMD_INT4_SYM) {
        set_error("int_quantize: mode %d is not an integer mode", mode);
        return NNCF_AMD_ERR_MODE;
    }
    int64_t gs;
    int st = check_group_shape("int_quantize", weight, outer, r, inner, group_size, gs);
    if (st != NNCF_AMD_OK) return st > 0 ? NNCF_AMD_OK : st;
    const IntMode m = int_mode(mode);
    if (m.asym && ((scale_in == nullptr) != (zp_in == nullptr))) {
        set_error("If precomputed quantization parameters are provided, both scale and zero point are required "
                  "for asymmetric quantization.");
        return NNCF_AMD_ERR_BAD_ARG;
    }
    if (zp_in && zp_dtype != NNCF_AMD_ZP_I32 && zp_dtype != NNCF_AMD_ZP_F32) {
        set_error("int_quantize: zp_dtype %d invalid", zp_dtype);
        return NNCF_AMD_ERR_DTYPE;
    }
    if (zp_in && !scale_in) {
        set_error("int_quantize: zp_in without scale_in");
        return NNCF_AMD_ERR_BAD_ARG;
    }
    if (packed_out && m.bits != 4) {
        set_error("int_quantize: packed_out is for 4-bit modes only");
        return NNCF_AMD_ERR_BAD_ARG;
    }
    if (packed_out && ((outer * r * inner) & 1)) {
        set_error("int_quantize: packed_out needs an even number of elements");
        return NNCF_AMD_ERR_BAD_ARG;
    }
    // data-free packed 4-bit compression of one tensor: the mode-specialised multi-tensor kernel with one item
    if (m.bits == 4 && inner == 1 && packed_out && scale_out && !scale_in && !codes_out && !dq_out && (!m.asym || zp_out)) {
        BatchItem it = {weight, outer, r, packed_out, scale_out, m.asym ? (void*)zp_out : nullptr};
        int vpg;
        if (batch_fits<IntPolicy>(w_dtype, &it, 1, group_size, &vpg))
            return m.asym ? run_batch<IntPolicyT<NNCF_AMD_INT4_ASYM>>(w_dtype, &it, 1, vpg, mode, "int_quantize", s)
                          : run_batch<IntPolicyT<NNCF_AMD_INT4_SYM>>(w_dtype, &it, 1, vpg, mode, "int_quantize", s);
    }
    // data-free per-channel compression of one tensor (codes for 8-bit, nibbles for 4-bit): the row kernel
    if (inner == 1 && scale_out && !scale_in && !dq_out && (!m.asym || zp_out) &&
        (m.bits == 4 ? (packed_out && !codes_out) : (codes_out && !packed_out))) {
        BatchItem it = {weight, outer, r, m.bits == 4 ? packed_out : (uint8_t*)codes_out, scale_out,
                        m.asym ? (void*)zp_out : nullptr};
        if (gs == r && int_rows_fit(w_dtype, &it, 1, -1)) return int_rows_run(w_dtype, &it, 1, mode, "int_quantize", s);
    }
    GroupArgs a = {};
    a.w = weight;
    a.scale_in = scale_in;
    a.zp_in = zp_in;
    a.zp_dtype = zp_dtype;
    a.codes = codes_out;
    a.packed = packed_out;
    a.scale_out = scale_out;
    a.zp_out = (scale_in == nullptr && m.asym) ? zp_out : nullptr;
    a.dq = dq_out;
    a.gs = gs;
    a.inner = inner;
    a.n_groups = outer * (r / gs) * inner;
    a.mode = mode;
    return run_group_op_dtype<IntPolicy>(w_dtype, a, m.asym ? 0 : 8, "int_quantize", s);
}

}  // namespace wc

extern "C" int nncf_amd_int_quantize(const void* weight, int w_dtype, int64_t outer, int64_t r, int64_t inner,
                                     int64_t group_size, int mode, const float* scale_in, const void* zp_in,
                                     int zp_dtype, void* codes_out, uint8_t* packed_out, float* scale_out,
                                     int32_t* zp_out, float* dq_out, void* stream) {
    return wc::int_quantize_impl(weight, w_dtype, outer, r, inner, group_size, mode, scale_in, zp_in, zp_dtype,
                                 codes_out, packed_out, scale_out, zp_out, dq_out, (hipStream_t)stream);
}

extern "C" int nncf_amd_int_quantize_batched(int n, const void* const* weights_host, int w_dtype,
                                             const int64_t* rows_host, const int64_t* cols_host, int64_t group_size,
                                             int mode, void* const* codes_out_host, uint8_t* const* packed_out_host,
                                             float* const* scale_out_host, int32_t* const* zp_out_host, void* stream) {
    using namespace wc;
    if (n < 0 || (n > 0 && (!weights_host || !rows_host || !cols_host))) {
        set_error("int_quantize_batched: bad descriptor arrays");
        return NNCF_AMD_ERR_BAD_ARG;
    }
    if (mode < NNCF_AMD_INT8_ASYM || mode > NNCF_AMD_INT4_SYM) {
        set_error("int_quantize_batched: mode %d is not an integer mode", mode);
        return NNCF_AMD_ERR_MODE;
    }
    const IntMode m = int_mode(mode);
    // one multi-tensor launch per 48 tensors when every tensor takes the packed 4-bit grouped path
    if (n > 0 && m.bits == 4 && packed_out_host && scale_out_host && !codes_out_host && (!m.asym || zp_out_host) &&
        (w_dtype == NNCF_AMD_F32 || w_dtype == NNCF_AMD_F16 || w_dtype == NNCF_AMD_BF16)) {
        std::vector<BatchItem> items(n);
        for (int i = 0; i < n; ++i)
            items[i] = {weights_host[i], rows_host[i], cols_host[i], packed_out_host[i], scale_out_host[i],
                        m.asym ? (void*)zp_out_host[i] : nullptr};
        int vpg;
        bool fits = batch_fits<IntPolicy>(w_dtype, items.data(), n, group_size, &vpg);
        if (fits && m.asym)
            for (int i = 0; i < n; ++i) fits = fits && items[i].aux != nullptr;
        if (fits) {
            const char* what = "int_quantize_batched";
            return m.asym ? run_batch<IntPolicyT<NNCF_AMD_INT4_ASYM>>(w_dtype, items.data(), n, vpg, mode, what, (hipStream_t)stream)
                          : run_batch<IntPolicyT<NNCF_AMD_INT4_SYM>>(w_dtype, items.data(), n, vpg, mode, what, (hipStream_t)stream);
        }
    }
    // per-channel groups (whole rows): the multi-tensor row kernel
    if (n > 0 && scale_out_host && (!m.asym || zp_out_host) &&
        (m.bits == 4 ? (packed_out_host && !codes_out_host) : (codes_out_host && !packed_out_host))) {
        std::vector<BatchItem> items(n);
        for (int i = 0; i < n; ++i)
            items[i] = {weights_host[i], rows_host[i], cols_host[i],
                        m.bits == 4 ? packed_out_host[i] : (uint8_t*)codes_out_host[i], scale_out_host[i],
                        m.asym ? (void*)zp_out_host[i] : nullptr};
        bool fits = int_rows_fit(w_dtype, items.data(), n, group_size);
        if (fits && m.asym)
            for (int i = 0; i < n; ++i) fits = fits && items[i].aux != nullptr;
        if (fits) return int_rows_run(w_dtype, items.data(), n, mode, "int_quantize_batched", (hipStream_t)stream);
    }
    for (int i = 0; i < n; ++i) {
        int st = int_quantize_impl(weights_host[i], w_dtype, rows_host[i], cols_host[i], 1, group_size, mode, nullptr,
                                   nullptr, NNCF_AMD_ZP_NONE, codes_out_host ? codes_out_host[i] : nullptr,
                                   packed_out_host ? packed_out_host[i] : nullptr,
                                   scale_out_host ? scale_out_host[i] : nullptr, zp_out_host ? zp_out_host[i] : nullptr,
                                   nullptr, (hipStream_t)stream);
        if (st != NNCF_AMD_OK) return st;
    }
    return NNCF_AMD_OK;
}
