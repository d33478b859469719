// wc_common.h -- shared device helpers for the gfx950 weight-compression kernels.
// Build with -ffp-contract=off: the reference computes round(w/scale + zp) with separately rounded
// fp32 division and addition (weight_lowering.py:678-682); nothing here may be contracted into an fma.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>

#include "../../include/nncf_amd_wc.h"

namespace wc {

constexpr float kEps = 1.1920928955078125e-07f;  // 2^-23 == fns.finfo(float32).eps

void set_error(const char* fmt, ...);
int launch_status(const char* what);

// Per-device "done once" flag for attribute opt-ins (hipFuncSetAttribute is per device): first_on_device() is true
// the first time it is asked on the current device and false afterwards; thread-safe.
struct OncePerDevice {
    std::mutex m;
    bool seen[64] = {};
    // returns -1: cannot tell the device, 1: first call on it, 0: seen before
    int first_on_device() {
        int dev = -1;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return -1;
        std::lock_guard<std::mutex> lock(m);
        if (seen[dev]) return 0;
        seen[dev] = true;
        return 1;
    }
    void forget_current() {  // the opt-in failed: try again next time
        int dev = -1;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return;
        std::lock_guard<std::mutex> lock(m);
        seen[dev] = false;
    }
};

// Batched error metric (wc_misc.hip: entry point and the row-per-block kernels; wc_int_err.hip: the tile kernel)
constexpr int kErrBatch = 48;
struct ErrBatchArgs {
    int n, mode, reduction;
    int64_t gs;                         // -1: the whole row
    uint32_t row_start[kErrBatch + 1];  // prefix sums of the row counts
    uint32_t cols[kErrBatch];
    const void* w[kErrBatch];
};
// Launches the tile kernel over the rows of `a` when its shape conditions hold (returns false otherwise, nothing
// launched): ws[global row] = sum of squared quantization errors of the row.
bool err_tiles_launch(int w_dtype, const ErrBatchArgs& a, float* ws, hipStream_t s);
bool err_rows_launch(int w_dtype, const ErrBatchArgs& a, float* ws, hipStream_t s);  // whole-row groups (wc_int_rows.hip)

// ---------------------------------------------------------------- 16-byte vectors of the input dtype
struct alignas(16) Vec16 {
    uint32_t w[4];
};

template <int DT>
struct Elem;
template <>
struct Elem<NNCF_AMD_F32> {
    static constexpr int kBytes = 4;
    static constexpr int kPerVec = 4;
    using raw_t = float;
    static __device__ __forceinline__ float load(const void* p, int64_t i) { return ((const float*)p)[i]; }
};
template <>
struct Elem<NNCF_AMD_F16> {
    static constexpr int kBytes = 2;
    static constexpr int kPerVec = 8;
    using raw_t = uint16_t;
    static __device__ __forceinline__ float load(const void* p, int64_t i) {
        return (float)__builtin_bit_cast(_Float16, ((const uint16_t*)p)[i]);
    }
};
template <>
struct Elem<NNCF_AMD_BF16> {
    static constexpr int kBytes = 2;
    static constexpr int kPerVec = 8;
    using raw_t = uint16_t;
    static __device__ __forceinline__ float load(const void* p, int64_t i) {
        return __builtin_bit_cast(float, (uint32_t)((const uint16_t*)p)[i] << 16);
    }
};

// unpack one 16-byte vector into kPerVec floats (exact upcast, weight_lowering.py:493-494)
template <int DT>
__device__ __forceinline__ void unpack(const Vec16& v, float* f);
template <>
__device__ __forceinline__ void unpack<NNCF_AMD_F32>(const Vec16& v, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = __builtin_bit_cast(float, v.w[i]);
}
template <>
__device__ __forceinline__ void unpack<NNCF_AMD_BF16>(const Vec16& v, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = __builtin_bit_cast(float, v.w[i] << 16);
        f[2 * i + 1] = __builtin_bit_cast(float, v.w[i] & 0xffff0000u);
    }
}
template <>
__device__ __forceinline__ void unpack<NNCF_AMD_F16>(const Vec16& v, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = (float)__builtin_bit_cast(_Float16, (uint16_t)(v.w[i] & 0xffffu));
        f[2 * i + 1] = (float)__builtin_bit_cast(_Float16, (uint16_t)(v.w[i] >> 16));
    }
}

// ---------------------------------------------------------------- cross-lane reductions (wave = 64)
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float x) {
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
}

// all-reduce over aligned sets of LANES consecutive lanes; every lane ends with the result.
template <int LANES, typename Op>
__device__ __forceinline__ float lanes_allreduce(float x, Op op) {
    if constexpr (LANES >= 2) x = op(x, dpp_f32<0xB1>(x));   // quad_perm [1,0,3,2]
    if constexpr (LANES >= 4) x = op(x, dpp_f32<0x4E>(x));   // quad_perm [2,3,0,1]
    if constexpr (LANES >= 8) x = op(x, dpp_f32<0x141>(x));  // row_half_mirror
    if constexpr (LANES >= 16) x = op(x, dpp_f32<0x140>(x)); // row_mirror
    if constexpr (LANES >= 32) x = op(x, __shfl_xor(x, 16, 64));
    if constexpr (LANES >= 64) x = op(x, __shfl_xor(x, 32, 64));
    return x;
}
// v_min3/v_max3 and DPP min/max through inline asm: llvm.minnum on values of unknown provenance (bit-cast
// loads) is lowered with a v_max x,x canonicalisation per operand in IEEE mode, doubling the VALU cost of
// the statistics pass.  NaN weights are outside the contract (the reference's int cast of NaN is undefined).
__device__ __forceinline__ float min3_raw(float a, float b, float c) {
    float d;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ float max3_raw(float a, float b, float c) {
    float d;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ float min3_abs_raw(float a, float b, float c) {
    float d;
    asm("v_min3_f32 %0, %1, |%2|, |%3|" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ float max3_abs_raw(float a, float b, float c) {
    float d;
    asm("v_max3_f32 %0, %1, |%2|, |%3|" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
// (lo, hi) <- (min, max) over aligned sets of LANES <= 16 lanes, every lane gets the result.
// s_nop 1: a DPP read of a VGPR written by the previous VALU instruction needs 2 wait states and hipcc
// pads nothing inside asm.
template <int LANES>
__device__ __forceinline__ void minmax_dpp16(float& lo, float& hi) {
    static_assert(LANES <= 16, "row-local DPP only");
    if constexpr (LANES >= 2)
        asm("s_nop 1\n\tv_min_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
            "v_max_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(lo), "+v"(hi));
    if constexpr (LANES >= 4)
        asm("s_nop 1\n\tv_min_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
            "v_max_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "+v"(lo), "+v"(hi));
    if constexpr (LANES >= 8)
        asm("s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
            "v_max_f32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf" : "+v"(lo), "+v"(hi));
    if constexpr (LANES >= 16)
        asm("s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
            "v_max_f32_dpp %1, %1, %1 row_mirror row_mask:0xf bank_mask:0xf" : "+v"(lo), "+v"(hi));
}
template <int LANES>
__device__ __forceinline__ void max_dpp16(float& hi) {
    static_assert(LANES <= 16, "row-local DPP only");
    if constexpr (LANES >= 2)
        asm("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(hi));
    if constexpr (LANES >= 4)
        asm("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "+v"(hi));
    if constexpr (LANES >= 8)
        asm("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf" : "+v"(hi));
    if constexpr (LANES >= 16)
        asm("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf" : "+v"(hi));
}

struct OpMin {
    __device__ __forceinline__ float operator()(float a, float b) const { return fminf(a, b); }
};
struct OpMax {
    __device__ __forceinline__ float operator()(float a, float b) const { return fmaxf(a, b); }
};
struct OpAdd {
    __device__ __forceinline__ float operator()(float a, float b) const { return a + b; }
};

// block-wide all-reduce for blocks of NW waves through LDS (smem: NW floats); every thread gets the result
template <int NW, typename Op>
__device__ __forceinline__ float block_allreduce(float x, Op op, float* smem) {
    x = lanes_allreduce<64>(x, op);
    if constexpr (NW == 1) return x;
    const int wave = threadIdx.x >> 6;
    __syncthreads();  // protect smem reuse
    if ((threadIdx.x & 63) == 0) smem[wave] = x;
    __syncthreads();
    float r = smem[0];
#pragma unroll
    for (int i = 1; i < NW; ++i) r = op(r, smem[i]);
    return r;
}

// ---------------------------------------------------------------- quantization parameter math (bit exact)
struct IntMode {
    int bits;
    int asym;
    float level_low, level_high;
    float inv_half;  // 1 / 2^(bits-1), exact
};
__host__ __device__ inline IntMode int_mode(int mode) {
    IntMode m;
    m.bits = (mode == NNCF_AMD_INT8_ASYM || mode == NNCF_AMD_INT8_SYM) ? 8 : 4;
    m.asym = (mode == NNCF_AMD_INT8_ASYM || mode == NNCF_AMD_INT4_ASYM) ? 1 : 0;
    m.level_low = m.asym ? 0.0f : -(float)(1 << (m.bits - 1));
    m.level_high = m.asym ? (float)((1 << m.bits) - 1) : (float)((1 << (m.bits - 1)) - 1);
    m.inv_half = m.bits == 8 ? 0.0078125f : 0.125f;
    return m;
}

// x86 cvttss2si semantics: numpy / torch-CPU astype(int32) of out-of-range or NaN gives INT_MIN
// (fake_quantize.py:349 casts before clipping).
__device__ __forceinline__ int cvtt_x86(float x) { return (fabsf(x) < 2147483648.0f) ? (int)x : (int)0x80000000; }

// asym: fake_quantize.py:342-350;  sym: weight_lowering.py:638-647
__device__ __forceinline__ void int_params(float lo, float hi, const IntMode& m, float& scale, float& zpf, int& zpi) {
    if (m.asym) {
        float s = (hi - lo) / m.level_high;  // levels-1 == level_high for level_low == 0
        if (fabsf(s) < kEps) s = kEps;
        float z = 0.0f - rintf(lo / s);
        int zi = cvtt_x86(z);
        zi = zi < 0 ? 0 : (zi > (int)m.level_high ? (int)m.level_high : zi);
        scale = s;
        zpi = zi;
        zpf = (float)zi;
    } else {
        float a = fabsf(lo);
        float s = (a >= hi) ? a : -hi;
        s = s * m.inv_half;  // == s / 2^(bits-1) bit for bit: both are the correctly rounded value of the same real
        if (fabsf(s) < kEps) s = kEps;
        scale = s;
        zpi = 0;
        zpf = 0.0f;
    }
}

// weight_lowering.py:678-682: round(w/scale [+ zp]) -> clip.  Returns the clipped value as float.
__device__ __forceinline__ float int_quant(float w, float scale, float zpf, bool has_zp, const IntMode& m) {
    float t = w / scale;
    if (has_zp) t = t + zpf;
    t = rintf(t);
    return fminf(fmaxf(t, m.level_low), m.level_high);
}

}  // namespace wc
