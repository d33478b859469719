// wc_int_err.hip -- the quantization-error metric of grouped integer modes at the rate of the lowering kernels.
// Reference: get_integer_quantization_error (src/nncf/quantization/algorithms/weight_compression/weight_lowering.py
// :329-374) inside the mixed-precision scoring loop (mixed_precision.py:161-199): quantize -> dequantize -> squared
// difference -> mean over the row -> max over the rows (or the Frobenius norm).
// One block per row (ws[row] keeps the layout of the row-per-block kernel in wc_misc.hip); its WPB waves (as many as
// the shortest row of the launch has tiles, at most four: a [N, 4096] bf16 row is two tiles) take the
// row's 4-KiB tiles in turn with the tile kernel's structure (wc_group_kernels.h: LDS-DMA, a lane owns 64
// contiguous bytes, DPP min/max over the group's lanes, quant_n_div) and add d^2 = (dequantized - w)^2 per lane; lanes
// -> wave -> block sums run in a fixed order, so the result does not depend on scheduling.  The kernel it replaces
// for these shapes read 2 bytes per lane, took two block-wide barriers per 128-element group and an IEEE divide per
// element: 0.38 TB/s of weights on Llama-3-8B shapes (INT4 g128).
#include "wc_int_policy.h"

namespace wc {

template <typename Policy, int DT, int VPG, int WPB>
__global__ __launch_bounds__(64 * WPB) void k_err_tiles(ErrBatchArgs a, float* __restrict__ ws) {
    constexpr int VPL = kVPL;
    constexpr int E = Elem<DT>::kPerVec;
    constexpr int NE = VPL * E;
    constexpr int GPL = VPG >= VPL ? 1 : VPL / VPG;
    constexpr int LPG = VPG >= VPL ? VPG / VPL : 1;
    constexpr int SG = NE / GPL;
    constexpr int TILE = 64 * VPL;
    __shared__ Vec16 stage[WPB][TILE];
    __shared__ float wsum[WPB];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t grow = blockIdx.x;
    int lo_i = 0, hi_i = a.n;
    while (hi_i - lo_i > 1) {
        const int mid = (lo_i + hi_i) >> 1;
        if (grow >= a.row_start[mid]) lo_i = mid;
        else hi_i = mid;
    }
    const uint32_t vpr = a.cols[lo_i] / E;  // 16-byte vectors per row (host: cols % (E * VPL) == 0)
    const char* row = (const char*)a.w[lo_i] + (uint64_t)(grow - a.row_start[lo_i]) * vpr * 16;
    const uint32_t lds_base = (uint32_t)(uintptr_t)stage[wave];
    uint32_t src_off[VPL];
#pragma unroll
    for (int u = 0; u < VPL; ++u) {
        const int sidx = u * 64 + lane;
        src_off[u] = (uint32_t)((sidx & ~3) + ((sidx & 3) ^ ((sidx >> 4) & 3))) * 16u;
    }
    const int rot = (lane >> 2) & 3;
    GroupArgs o = {};
    o.mode = a.mode;
    const Policy pol(o, nullptr);
    float acc = 0.0f;
    for (uint32_t v0 = (uint32_t)wave * TILE; v0 < vpr; v0 += WPB * TILE) {
        const uint32_t in_tile = vpr - v0 < (uint32_t)TILE ? vpr - v0 : (uint32_t)TILE;
        const uint32_t last = (in_tile - 1) * 16u;
#pragma unroll
        for (int u = 0; u < VPL; ++u)
            dma16_nt(row + (uint64_t)v0 * 16, src_off[u] < last ? src_off[u] : last, lds_base + u * 1024);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        float f[NE], w0[NE];
#pragma unroll
        for (int k = 0; k < VPL; ++k) {
            const Vec16 raw = stage[wave][lane * VPL + (k ^ rot)];
            unpack<DT>(raw, f + k * E);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the image may be overwritten by the next round's DMA
#pragma unroll
        for (int i = 0; i < NE; ++i) w0[i] = f[i];
        typename Policy::Par par[GPL];
#pragma unroll
        for (int gi = 0; gi < GPL; ++gi) {
            typename Policy::Stat st = pol.stat_init();
            pol.template stat_add_vec<SG>(st, f + gi * SG);
            if constexpr (LPG > 1) pol.template stat_lanes<LPG>(st);
            par[gi] = pol.compute(st);
            pol.template quant_n_div<SG>(par[gi], f + gi * SG, f + gi * SG);
        }
        float part = 0.0f;
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const float d = pol.dq(0, f[i], par[i / SG]) - w0[i];
            part = fmaf(d, d, part);
        }
        if ((uint32_t)(lane * VPL) < in_tile) acc += part;  // (a lane is all live or all dead)
    }
    acc = lanes_allreduce<64>(acc, OpAdd());
    if constexpr (WPB == 1) {
        if (lane == 0) ws[grow] = acc;
    } else {
        if (lane == 0) wsum[wave] = acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            if constexpr (WPB == 2) ws[grow] = wsum[0] + wsum[1];
            else ws[grow] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
        }
    }
}

template <typename Policy, int DT, int WPB>
static bool err_tiles_vpg(int vpg, const ErrBatchArgs& a, float* ws, hipStream_t s) {
    const dim3 grid(a.row_start[a.n]);
    switch (vpg) {
        case 4: hipLaunchKernelGGL((k_err_tiles<Policy, DT, 4, WPB>), grid, dim3(64 * WPB), 0, s, a, ws); return true;
        case 8: hipLaunchKernelGGL((k_err_tiles<Policy, DT, 8, WPB>), grid, dim3(64 * WPB), 0, s, a, ws); return true;
        case 16: hipLaunchKernelGGL((k_err_tiles<Policy, DT, 16, WPB>), grid, dim3(64 * WPB), 0, s, a, ws); return true;
        case 32: hipLaunchKernelGGL((k_err_tiles<Policy, DT, 32, WPB>), grid, dim3(64 * WPB), 0, s, a, ws); return true;
        default: return false;
    }
}

template <typename Policy, int DT>
static bool err_tiles_wpb(int vpg, int wpb, const ErrBatchArgs& a, float* ws, hipStream_t s) {
    if (wpb >= 4) return err_tiles_vpg<Policy, DT, 4>(vpg, a, ws, s);
    if (wpb >= 2) return err_tiles_vpg<Policy, DT, 2>(vpg, a, ws, s);
    return err_tiles_vpg<Policy, DT, 1>(vpg, a, ws, s);
}

template <int DT>
static bool err_tiles_mode(int vpg, int wpb, const ErrBatchArgs& a, float* ws, hipStream_t s) {
    switch (a.mode) {
        case NNCF_AMD_INT8_ASYM: return err_tiles_wpb<IntPolicyT<NNCF_AMD_INT8_ASYM>, DT>(vpg, wpb, a, ws, s);
        case NNCF_AMD_INT8_SYM: return err_tiles_wpb<IntPolicyT<NNCF_AMD_INT8_SYM>, DT>(vpg, wpb, a, ws, s);
        case NNCF_AMD_INT4_ASYM: return err_tiles_wpb<IntPolicyT<NNCF_AMD_INT4_ASYM>, DT>(vpg, wpb, a, ws, s);
        case NNCF_AMD_INT4_SYM: return err_tiles_wpb<IntPolicyT<NNCF_AMD_INT4_SYM>, DT>(vpg, wpb, a, ws, s);
        default: return false;
    }
}

bool err_tiles_launch(int w_dtype, const ErrBatchArgs& a, float* ws, hipStream_t s) {
    const int eb = w_dtype == NNCF_AMD_F32 ? 4 : 2;
    if (a.gs < 1 || a.n < 1 || a.row_start[a.n] == 0 || (a.gs * eb) % 16 != 0) return false;
    const int64_t vpg = a.gs * eb / 16;
    if (vpg != 4 && vpg != 8 && vpg != 16 && vpg != 32) return false;
    int64_t min_tiles = 4;  // waves per block: the tiles of the shortest row, so that no wave of a block is idle
    for (int i = 0; i < a.n; ++i) {
        // rows of whole lanes (64 bytes), 16-byte aligned tensors
        if (((int64_t)a.cols[i] * eb) % (16 * kVPL) != 0 || !is16(a.w[i])) return false;
        const int64_t tiles = ((int64_t)a.cols[i] * eb + 4095) / 4096;
        if (tiles < min_tiles) min_tiles = tiles;
    }
    const int wpb = (int)min_tiles;
    switch (w_dtype) {
        case NNCF_AMD_F32: return err_tiles_mode<NNCF_AMD_F32>((int)vpg, wpb, a, ws, s);
        case NNCF_AMD_F16: return err_tiles_mode<NNCF_AMD_F16>((int)vpg, wpb, a, ws, s);
        case NNCF_AMD_BF16: return err_tiles_mode<NNCF_AMD_BF16>((int)vpg, wpb, a, ws, s);
        default: return false;
    }
}

}  // namespace wc
