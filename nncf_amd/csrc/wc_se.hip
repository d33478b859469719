// wc_se.hip -- fused inner step of scale estimation (scale_estimation.py:300-360 with :380-409): for a candidate
// scale per group, quantize the group, form target = q - zp and zero_mask = 0.001 * isclose(target, 0), and return
//     sum_k |w_k| / (|target_k| + zero_mask_k) * importance_k        (estimate_scales, :396-409)
// in ONE pass over the weight and the importance (8 B per element) instead of ~10 elementwise torch kernels plus a
// reduction (~80 B per element).  Quantization is the exact policy arithmetic of the lowering kernels.
#include "wc_float_policy.h"
#include "wc_int_policy.h"

namespace wc {

template <typename Policy>
__global__ __launch_bounds__(256) void k_se_estimate(const float* __restrict__ w, const float* __restrict__ imp, GroupArgs a,
                                                     int64_t gs, float zero_scale, float* __restrict__ out) {
    __shared__ float lds[kPolicyLds];
    Policy::init_shared(lds);
    __syncthreads();
    const Policy pol(a, lds);
    const int l = threadIdx.x & 15;
    const int64_t g = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4;
    const bool live = g < a.n_groups;
    const int64_t gg = live ? g : a.n_groups - 1;  // idle lanes shadow the last group (they take part in the reduction)
    const typename Policy::Par par = pol.load(a, gg);
    const float* wg = w + gg * gs;
    const float* ig = imp + gg * gs;
    float acc = 0.0f;
    auto term = [&](float wv, float iv) {
        int code;
        float val;
        pol.quant(wv, par, code, val);
        const float target = pol.target(val, par);
        const float zm = fabsf(target) <= 1e-8f ? zero_scale : 0.0f;  // isclose(target, 0): atol 1e-8
        acc += fabsf(wv) / (fabsf(target) + zm) * iv;
    };
    if ((gs & 3) == 0 && ((((uintptr_t)w | (uintptr_t)imp) & 15u) == 0)) {
        for (int64_t k = l * 4; k < gs; k += 64) {
            const float4 wv = *(const float4*)(wg + k);
            const float4 iv = *(const float4*)(ig + k);
            term(wv.x, iv.x);
            term(wv.y, iv.y);
            term(wv.z, iv.z);
            term(wv.w, iv.w);
        }
    } else {
        for (int64_t k = l; k < gs; k += 16) term(wg[k], ig[k]);
    }
    acc = lanes_allreduce<16>(acc, OpAdd());
    if (live && l == 0) out[g] = acc;
}

}  // namespace wc

extern "C" int nncf_amd_se_estimate(const float* weight, const float* importance, int64_t n_groups, int64_t group_size,
                                    int mode, const float* scale, const float* zero_point, float zero_scale, float* out,
                                    void* stream) {
    using namespace wc;
    if (n_groups < 0 || group_size < 1 || (n_groups > 0 && (!weight || !importance || !scale || !out))) {
        set_error("se_estimate: bad argument (n_groups=%lld group_size=%lld)", (long long)n_groups, (long long)group_size);
        return NNCF_AMD_ERR_BAD_ARG;
    }
    if (n_groups == 0) return NNCF_AMD_OK;
    GroupArgs a = {};
    a.scale_in = scale;
    a.zp_in = zero_point;
    a.zp_dtype = NNCF_AMD_ZP_F32;
    a.n_groups = n_groups;
    a.gs = group_size;
    a.inner = 1;
    a.mode = mode;
    const dim3 grid((unsigned)((n_groups * 16 + 255) / 256));
    hipStream_t s = (hipStream_t)stream;
    if (mode >= NNCF_AMD_INT8_ASYM && mode <= NNCF_AMD_INT4_SYM) {
        const bool asym = mode == NNCF_AMD_INT8_ASYM || mode == NNCF_AMD_INT4_ASYM;
        if (asym && !zero_point) {
            set_error("se_estimate: asymmetric mode needs the zero points");
            return NNCF_AMD_ERR_BAD_ARG;
        }
        if (!asym) a.zp_in = nullptr;
        hipLaunchKernelGGL((k_se_estimate<IntPolicy>), grid, dim3(256), 0, s, weight, importance, a, group_size, zero_scale, out);
    } else if (mode == NNCF_AMD_NF4) {
        a.zp_in = nullptr;
        hipLaunchKernelGGL((k_se_estimate<FloatPolicy<true>>), grid, dim3(256), 0, s, weight, importance, a, group_size, zero_scale, out);
    } else if (mode == NNCF_AMD_MXFP4 || mode == NNCF_AMD_FP4) {
        a.zp_in = nullptr;
        hipLaunchKernelGGL((k_se_estimate<FloatPolicy<false>>), grid, dim3(256), 0, s, weight, importance, a, group_size, zero_scale, out);
    } else {
        set_error("se_estimate: unsupported mode %d", mode);
        return NNCF_AMD_ERR_MODE;
    }
    return launch_status("se_estimate");
}
