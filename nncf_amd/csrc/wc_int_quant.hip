// wc_int_quant.hip -- fused INT4/INT8 group-wise lowering for gfx950 (policy + C entry points).
//
// One pass over the weight: 16-byte coalesced loads, wave-64 DPP min/max per group, scale / zero point,
// IEEE divide -> add -> round-half-even -> clip, and the stores the caller asked for (unpacked codes,
// nibble-packed codes, scale, zero point, dequantized fp32).  HBM-bound by construction: every input byte
// is read once, every output byte written once.  Kernel skeletons: wc_group_kernels.h.
#include "wc_group_kernels.h"

namespace wc {

struct IntPolicy {
    IntMode m;
    int bias;
    struct Stat {
        float lo, hi;
    };
    struct Par {
        float scale, zpf;
        int zpi;
        bool has_zp;
    };
    static __device__ __forceinline__ void init_shared(float*) {}
    __device__ __forceinline__ IntPolicy(const GroupArgs& a, float*) : m(int_mode(a.mode)) { bias = m.asym ? 0 : 8; }
    __device__ __forceinline__ Stat stat_init() const { return {__builtin_inff(), -__builtin_inff()}; }
    __device__ __forceinline__ void stat_add(Stat& s, float f) const {
        s.lo = fminf(s.lo, f);
        s.hi = fmaxf(s.hi, f);
    }
    template <int L>
    __device__ __forceinline__ void stat_lanes(Stat& s) const {
        s.lo = lanes_allreduce<L>(s.lo, OpMin());
        s.hi = lanes_allreduce<L>(s.hi, OpMax());
    }
    __device__ __forceinline__ void stat_block(Stat& s, float* smem) const {
        s.lo = block_allreduce<4>(s.lo, OpMin(), smem);
        s.hi = block_allreduce<4>(s.hi, OpMax(), smem + 8);
    }
    __device__ __forceinline__ Par compute(const Stat& s) const {
        Par p;
        int_params(s.lo, s.hi, m, p.scale, p.zpf, p.zpi);
        p.has_zp = m.asym;
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
        return p;
    }
    __device__ __forceinline__ void store(const GroupArgs& a, int64_t g, const Par& p) const {
        if (a.scale_out) a.scale_out[g] = p.scale;
        if (a.zp_out) a.zp_out[g] = p.zpi;
    }
    __device__ __forceinline__ void quant(float w, const Par& p, int& code, float& value) const {
        value = int_quant(w, p.scale, p.zpf, p.has_zp, m);
        code = (int)value;
    }
    // weight_lowering.py:442-443: (int32(q) - zp).astype(f32) * scale; q, zp small integers -> exact subtraction
    __device__ __forceinline__ float dq(int, float value, const Par& p) const {
        return p.has_zp ? (value - p.zpf) * p.scale : value * p.scale;
    }
    __device__ __forceinline__ int nibble(int code) const { return code + bias; }
};

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
    if (n < 0 || (n > 0 && (!weights_host || !rows_host || !cols_host))) {
        wc::set_error("int_quantize_batched: bad descriptor arrays");
        return NNCF_AMD_ERR_BAD_ARG;
    }
    for (int i = 0; i < n; ++i) {
        int st = wc::int_quantize_impl(weights_host[i], w_dtype, rows_host[i], cols_host[i], 1, group_size, mode, nullptr,
                                       nullptr, NNCF_AMD_ZP_NONE, codes_out_host ? codes_out_host[i] : nullptr,
                                       packed_out_host ? packed_out_host[i] : nullptr,
                                       scale_out_host ? scale_out_host[i] : nullptr,
                                       zp_out_host ? zp_out_host[i] : nullptr, nullptr, (hipStream_t)stream);
        if (st != NNCF_AMD_OK) return st;
    }
    return NNCF_AMD_OK;
}
