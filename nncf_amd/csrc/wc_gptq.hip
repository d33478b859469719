// wc_gptq.hip -- GPTQ in-block column sweep for gfx950 (gptq.py:298-361, one block of <= 128 columns).
//
// Rows of the weight are independent in GPTQ: the quantization error of W[r, c] only propagates along row r
// (W[r, c:] -= err * Hinv[c, c:]).  Sixteen lanes own one row: column j of the block lives in register j / 16 of
// lane j % 16 (interleaved, so that every lane has columns left to update until the end of the sweep).  Per
// column: the owner's value is broadcast over the 16 lanes (ds_bpermute), every lane quantizes it (same parameters,
// no divergence) and updates its own remaining columns with the Hinv row from LDS (the 128x128 Hinv block is staged
// once per workgroup of 32 rows).  Loads and stores of W, Q and Err are coalesced along the row.  The ~K
// sequential Python iterations x ~10 kernel launches of the reference become K/128 launches; the trailing update
// W[:, i2:] -= Err @ Hinv[i1:i2, i2:] (gptq.py:363) stays a plain GEMM at the caller.
//
// Numerics follow the reference step by step: group parameters are computed from `weight` (the tensor WITHOUT
// this block's in-block updates, gptq.py:315-333) with the same policy arithmetic as the lowering kernels, the
// column is quantize-dequantized with the IEEE divide, err = (w - q) / d, and every update is
// w = w - fl(err * h) (separately rounded product and difference, as matmul of a [N,1] by a [1,M] gives).
#include <stdlib.h>

#include "wc_float_policy.h"
#include "wc_int_policy.h"

namespace wc {

constexpr int GB = 128;   // max columns per call
constexpr int GTHR = 256;  // threads per workgroup (64 KiB of LDS each: two workgroups per CU)

// GLPR lanes own one row (GB / GLPR columns each).  Every lane of a row repeats the column's quantization, so the issue
// slots per row and column are GLPR x (quantization + its share of the updates): 16 lanes per row spend ~850, 4 lanes
// per row ~310; the launcher picks by the row count.
template <typename Policy, int GLPR>
__global__ __launch_bounds__(GTHR) void k_gptq_block(const float* __restrict__ weight, float* __restrict__ quantized,
                                                    float* __restrict__ err_block, const float* __restrict__ hinv,
                                                    int64_t rows, int64_t K, int64_t i1, int count, int64_t gs, int mode,
                                                    float* scales, int32_t* zps, int given) {
    constexpr int GRPB = GTHR / GLPR;  // rows per workgroup
    constexpr int GCPL = GB / GLPR;    // columns per lane
    extern __shared__ float hb[];  // [GB][GB]: Hinv[i1 + i][i1 + j], zero outside the block
    __shared__ float lds[kPolicyLds];
    Policy::init_shared(lds);
    for (int idx = threadIdx.x; idx < GB * GB; idx += GTHR) {
        const int i = idx / GB, j = idx % GB;
        hb[idx] = (i < count && j < count) ? hinv[(i1 + i) * K + i1 + j] : 0.0f;
    }
    __syncthreads();
    GroupArgs a = {};
    a.mode = mode;
    a.scale_in = scales;
    a.zp_in = zps;
    a.zp_dtype = NNCF_AMD_ZP_I32;
    a.scale_out = scales;
    a.zp_out = zps;
    const Policy pol(a, lds);
    const int tid = threadIdx.x, l = tid % GLPR, lane = tid & 63;
    const int64_t row = (int64_t)blockIdx.x * GRPB + tid / GLPR;
    const bool active = row < rows;
    const int64_t r = active ? row : rows - 1;  // idle lanes shadow the last row (they take part in the shuffles)
    const int64_t ngroups = K / gs;
    const float* wrow = weight + r * K;
    float w[GCPL], qv[GCPL], ev[GCPL];
#pragma unroll
    for (int k = 0; k < GCPL; ++k) {
        const int j = k * GLPR + l;
        w[k] = j < count ? wrow[i1 + j] : 0.0f;
        qv[k] = 0.0f;
        ev[k] = 0.0f;
    }
    typename Policy::Par par;
    // position of the sweep inside its quantization group, kept as counters: a 64-bit modulo per column would cost
    // more than the column's whole dependent chain (no integer divider on the device)
    int64_t gidx = r * ngroups + i1 / gs;
    int64_t in_group = i1 % gs;
    if (in_group != 0) par = pol.load(a, gidx);  // the group began in an earlier block
    // the Hinv row of the NEXT column (its diagonal entry and this lane's 8 columns) is fetched from LDS while the
    // current column is being quantized: two LDS round trips less in the per-column dependent chain
    float hn[GCPL], hdn;
    auto prefetch = [&](int i) {
        const float* hr = hb + i * GB;
        hdn = hr[i];
#pragma unroll
        for (int k = 0; k < GCPL; ++k) hn[k] = hr[k * GLPR + l];
    };
    prefetch(0);
#pragma unroll
    for (int k0 = 0; k0 < GCPL; ++k0) {
        for (int lo = 0; lo < GLPR; ++lo) {
            const int i = k0 * GLPR + lo;
            if (i >= count) break;
            if (in_group == 0) {
                if (given) {  // parameters supplied by the caller (scale estimation inside GPTQ, gptq.py:319-327)
                    par = pol.load(a, gidx);
                } else {
                    const int64_t c = i1 + i;
                    typename Policy::Stat st = pol.stat_init();
                    for (int64_t k = l; k < gs; k += GLPR) pol.stat_add(st, wrow[c + k]);
                    pol.template stat_lanes<GLPR>(st);
                    par = pol.compute(st);
                    if (active && l == 0) pol.template store<false>(a, gidx, par);
                }
            }
            if (++in_group == gs) {
                in_group = 0;
                ++gidx;
            }
            float hc[GCPL];
#pragma unroll
            for (int k = 0; k < GCPL; ++k) hc[k] = hn[k];
            const float hd = hdn;
            prefetch(i + 1 < GB ? i + 1 : GB - 1);
            // the owner (lane lo of the row) broadcasts column i
            const float wv = __builtin_bit_cast(
                float, __builtin_amdgcn_ds_bpermute(((lane & ~(GLPR - 1)) + lo) * 4, __builtin_bit_cast(int, w[k0])));
            int code;
            float val;
            pol.quant(wv, par, code, val);
            const float q = pol.dq(code, val, par);
            const float err = (wv - q) / hd;
            if (l == lo) {
                qv[k0] = q;
                ev[k0] = err;
            }
            {
                const float prod = err * hc[k0];
                w[k0] = l >= lo ? w[k0] - prod : w[k0];
            }
#pragma unroll
            for (int k = k0 + 1; k < GCPL; ++k) {
                const float prod = err * hc[k];
                w[k] = w[k] - prod;
            }
        }
    }
    if (!active) return;
#pragma unroll
    for (int k = 0; k < GCPL; ++k) {
        const int j = k * GLPR + l;
        if (j < count) {
            quantized[r * K + i1 + j] = qv[k];
            err_block[r * count + j] = ev[k];
        }
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
    constexpr size_t hb_bytes = (size_t)GB * GB * sizeof(float);  // 64 KiB of dynamic LDS
    // Few rows: the per-column dependent chain sets the time and 16 lanes per row keep it shortest (52 us per block at
    // 8192 rows against 66 us with 4 lanes).  Many rows: the launch is bound by issue slots, which 4 lanes per row
    // need 2.7x fewer of (28672 rows: 174 -> 84 us per block; 8 lanes: 123).  NNCF_AMD_GPTQ_LANES overrides (developer knob).
    static const int forced_lanes = [] {
        const char* e = getenv("NNCF_AMD_GPTQ_LANES");
        const int v = e ? atoi(e) : 0;
        return (v == 4 || v == 8 || v == 16) ? v : 0;
    }();
    const int lanes_per_row = forced_lanes ? forced_lanes : (rows >= 16384 ? 4 : 16);
    auto launch = [&](auto kernel, int glpr) {
        const int rpb = GTHR / glpr;
        hipLaunchKernelGGL(kernel, dim3((unsigned)((rows + rpb - 1) / rpb)), dim3(GTHR), hb_bytes, s, weight, quantized,
                           err_block, hinv, rows, k, i1, (int)count, gs, mode, scales, zps, use_given_params);
    };
    static OncePerDevice attr_once;
    if (attr_once.first_on_device() != 0) {
        bool ok = true;
        auto opt_in = [&](auto kernel) {
            ok = ok && hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)hb_bytes) == hipSuccess;
        };
        opt_in(k_gptq_block<IntPolicy, 4>), opt_in(k_gptq_block<IntPolicy, 8>), opt_in(k_gptq_block<IntPolicy, 16>);
        opt_in(k_gptq_block<FloatPolicy<true>, 4>), opt_in(k_gptq_block<FloatPolicy<true>, 8>), opt_in(k_gptq_block<FloatPolicy<true>, 16>);
        opt_in(k_gptq_block<FloatPolicy<false>, 4>), opt_in(k_gptq_block<FloatPolicy<false>, 8>), opt_in(k_gptq_block<FloatPolicy<false>, 16>);
        if (!ok) {
            attr_once.forget_current();
            return launch_status("gptq_block attribute");
        }
    }
    auto by_lanes = [&](auto k4, auto k8, auto k16) {
        if (lanes_per_row == 4) launch(k4, 4);
        else if (lanes_per_row == 8) launch(k8, 8);
        else launch(k16, 16);
    };
    if (mode >= NNCF_AMD_INT8_ASYM && mode <= NNCF_AMD_INT4_SYM) {
        by_lanes(k_gptq_block<IntPolicy, 4>, k_gptq_block<IntPolicy, 8>, k_gptq_block<IntPolicy, 16>);
    } else if (mode == NNCF_AMD_NF4) {
        by_lanes(k_gptq_block<FloatPolicy<true>, 4>, k_gptq_block<FloatPolicy<true>, 8>, k_gptq_block<FloatPolicy<true>, 16>);
    } else if (mode == NNCF_AMD_MXFP4 || mode == NNCF_AMD_FP4) {
        by_lanes(k_gptq_block<FloatPolicy<false>, 4>, k_gptq_block<FloatPolicy<false>, 8>, k_gptq_block<FloatPolicy<false>, 16>);
    } else {
        set_error("gptq_block: unsupported mode %d", mode);
        return NNCF_AMD_ERR_MODE;
    }
    return launch_status("gptq_block");
}
