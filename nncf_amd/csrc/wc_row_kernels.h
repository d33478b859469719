// wc_row_kernels.h -- multi-tensor per-channel lowering: the quantization group is a whole row of the weight
// (INT8 per-channel, the reference's default mode; 4-bit with group_size = -1), too long for one wave tile.
//
// A team (one wave for rows up to 512 vectors, one 256-thread block above) owns ONE row.  The row is read once (LDS-DMA
// for rows up to 32 KiB, non-temporal 16-byte loads above) and stays in registers (KEEP raw vectors per thread)
// between the min/max pass and the quantize pass; what hides the load latency is occupancy (every resident team
// has a whole row in flight).  Wave-owned rows come out of LDS so that a lane owns PAIRS of adjacent vectors and
// its codes leave as 16-byte streaming stores.
#pragma once
#include <type_traits>

#include "wc_group_kernels.h"

namespace wc {

__device__ __forceinline__ Vec16 load16_nt(const Vec16* p) {
    Vec16 v;
    const u32x4 r = __builtin_nontemporal_load((const u32x4*)p);
    v.w[0] = r[0];
    v.w[1] = r[1];
    v.w[2] = r[2];
    v.w[3] = r[3];
    return v;
}

// BatchArgs use: tile_start = prefix sums of ROWS, nfull[i] = vectors per row of tensor i, packed[i] = output
// (1-byte codes or nibble-packed), scale[i], aux[i] = int32 zero points or NULL.
// One row per team, a grid of rows / teams-per-block blocks: the workgroup dispatcher hands out the rows in address
// order, occupancy hides the load latency (same finding as for the tile kernel, wc_group_kernels.h: 0.775 of the
// HBM roofline for Llama-3-8B INT8_ASYM; a resident set of teams claiming rows from atomic counters and
// prefetching the next row stopped at 0.68).
// ERR: nothing is stored but the row's sum of squared quantization errors, (float*)scale[i])[row] (the error metric
// of the mixed-precision scoring, weight_lowering.py:329-374 with the per-channel backup configuration): lanes ->
// waves -> block in a fixed order.
template <typename Policy, int DT, int TEAM, int KEEP, bool PACKED, bool ERR = false>
__global__ __launch_bounds__(256) void k_rows_batch(BatchArgs b) {
    constexpr int E = Elem<DT>::kPerVec;
    constexpr int TPB = 256 / TEAM;
    __shared__ float lds[kPolicyLds];
    __shared__ typename Policy::Stat wave_stat[4];
    __shared__ float wave_err[4];
    constexpr bool DMA = TEAM == 64 || KEEP <= 8;
    __shared__ Vec16 stage[TEAM == 64 ? 4 : 1][DMA ? TEAM * KEEP : 1];
    const int lane = threadIdx.x & 63;
    const int tl = threadIdx.x % TEAM;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t total = b.tile_start[b.n];
    const uint32_t r = TEAM == 64 ? blockIdx.x * TPB + wave : blockIdx.x;
    const bool have = r < total;
    int lo_i = 0, hi_i = b.n;
    while (have && hi_i - lo_i > 1) {
        const int mid = (lo_i + hi_i) >> 1;
        if (r >= b.tile_start[mid]) lo_i = mid;
        else hi_i = mid;
    }
    const uint32_t vpr = b.nfull[lo_i];
    const uint64_t row = have ? r - b.tile_start[lo_i] : 0;
    constexpr bool PAIRS = DMA && KEEP >= 2;
    auto vec_of = [&](int k) -> uint32_t {
        return PAIRS ? (uint32_t)(2 * TEAM * (k >> 1) + 2 * tl + (k & 1)) : (uint32_t)(tl + k * TEAM);
    };
    Vec16 cur[KEEP];
    if (have) {
        if constexpr (DMA) {
            const uint32_t lds_base = (uint32_t)(uintptr_t)stage[TEAM == 64 ? wave : 0];
            const char* tbase = (const char*)b.w[lo_i] + row * vpr * 16;
            const uint32_t last = (vpr - 1) * 16u;
            const uint32_t w0 = TEAM == 64 ? 0u : (uint32_t)wave * 64u;
#pragma unroll
            for (int u = 0; u < KEEP; ++u) {
                const uint32_t v = (uint32_t)(u * TEAM) + w0;
                const uint32_t off = (v + lane) * 16u;
                dma16_nt(tbase, off < last ? off : last, lds_base + v * 16u);
            }
        } else {
            const Vec16* src = (const Vec16*)b.w[lo_i] + row * vpr;
#pragma unroll
            for (int k = 0; k < KEEP; ++k) {
                const uint32_t j = tl + k * TEAM;
                if (j < vpr) cur[k] = load16_nt(src + j);
            }
        }
    }
    Policy::init_shared(lds);
    if constexpr (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // policy tables; block teams: the whole image has landed
    if (!have) return;  // (a whole block for block teams; wave teams have no barrier below)
    GroupArgs o = {};
    o.mode = b.mode;
    const Policy pol(o, lds);
    if constexpr (DMA) {
#pragma unroll
        for (int k = 0; k < KEEP; ++k) cur[k] = stage[TEAM == 64 ? wave : 0][vec_of(k)];
    }
    typename Policy::Stat st = pol.stat_init();
#pragma unroll
    for (int k = 0; k < KEEP; ++k) {
        const uint32_t j = vec_of(k);
        __builtin_amdgcn_sched_barrier(0);
        if (j < vpr) {
            float f[E];
            unpack<DT>(cur[k], f);
            pol.template stat_add_vec<E>(st, f);
        }
    }
    pol.template stat_lanes<64>(st);
    if constexpr (TEAM != 64) {
        if (lane == 0) wave_stat[wave] = st;
        __syncthreads();
        st = wave_stat[0];
#pragma unroll
        for (int q = 1; q < 4; ++q) pol.stat_merge(st, wave_stat[q]);
    }
    const typename Policy::Par par = pol.compute(st);
    if constexpr (!ERR) {
        o.scale_out = b.scale[lo_i];
        o.zp_out = (int32_t*)b.aux[lo_i];
        if (tl == 0) pol.template store<true>(o, (int64_t)row, par);
        if constexpr (PACKED) o.packed = b.packed[lo_i];
        else o.codes = b.packed[lo_i];
    }
    float err = 0.0f;
    // N elements: quantize in place, then codes out (or, ERR, the squared differences of the dequantized values)
    auto emit = [&](auto n_tag, float* f, int64_t e0) {
        constexpr int N = decltype(n_tag)::value;
        if constexpr (ERR) {
            float w0[N];
#pragma unroll
            for (int i = 0; i < N; ++i) w0[i] = f[i];
            pol.template quant_n_div<N>(par, f, f);
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const float d = pol.dq(0, f[i], par) - w0[i];
                err = fmaf(d, d, err);
            }
        } else {
            pol.template quant_n_div<N>(par, f, f);
            pol.template store_codes<N, true>(o, f, e0);
        }
    };
    auto opaque = [](Vec16 raw) {
        asm volatile("" : "+v"(raw.w[0]), "+v"(raw.w[1]), "+v"(raw.w[2]), "+v"(raw.w[3]));
        return raw;
    };
    if constexpr (PAIRS) {
#pragma unroll
        for (int k = 0; k < KEEP; k += 2) {
            const uint32_t j = vec_of(k);
            __builtin_amdgcn_sched_barrier(0);
            if (j + 1 < vpr) {
                float f[2 * E];
                unpack<DT>(opaque(cur[k]), f);
                unpack<DT>(opaque(cur[k + 1]), f + E);
                emit(std::integral_constant<int, 2 * E>(), f, (int64_t)((row * vpr + j) * E));
            } else if (j < vpr) {
                float f[E];
                unpack<DT>(opaque(cur[k]), f);
                emit(std::integral_constant<int, E>(), f, (int64_t)((row * vpr + j) * E));
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < KEEP; ++k) {
            const uint32_t j = vec_of(k);
            __builtin_amdgcn_sched_barrier(0);
            if (j < vpr) {
                float f[E];
                unpack<DT>(opaque(cur[k]), f);
                emit(std::integral_constant<int, E>(), f, (int64_t)((row * vpr + j) * E));
            }
        }
    }
    if constexpr (ERR) {
        float* dst = (float*)b.scale[lo_i];
        err = lanes_allreduce<64>(err, OpAdd());
        if constexpr (TEAM == 64) {
            if (lane == 0) dst[row] = err;
        } else {
            if (lane == 0) wave_err[wave] = err;
            __syncthreads();  // (block teams: `have` is uniform over the block)
            if (threadIdx.x == 0) dst[row] = (wave_err[0] + wave_err[1]) + (wave_err[2] + wave_err[3]);
        }
    }
}

// team/keep class of a row of `vpr` 16-byte vectors; 0 = does not fit (longer than 4096 vectors)
struct RowClass {
    int team, keep;
};
static inline RowClass row_class(int64_t vpr) {
    if (vpr <= 64) return {64, 1};
    if (vpr <= 128) return {64, 2};
    if (vpr <= 256) return {64, 4};
    if (vpr <= 512) return {64, 8};
    if (vpr <= 1024) return {256, 4};
    if (vpr <= 2048) return {256, 8};
    if (vpr <= 4096) return {256, 16};
    return {0, 0};
}

template <typename Policy, int DT, int TEAM, int KEEP, bool PACKED>
static hipError_t launch_rows(BatchArgs& b, hipStream_t s) {
    constexpr int TPB = 256 / TEAM;
    const int64_t blocks = ((int64_t)b.tile_start[b.n] + TPB - 1) / TPB;  // rows <= 2^30 per launch
    if (blocks > 0)
        hipLaunchKernelGGL((k_rows_batch<Policy, DT, TEAM, KEEP, PACKED>), dim3((unsigned)blocks), dim3(256), 0, s, b);
    return hipPeekAtLastError();
}

template <typename Policy, int DT, bool PACKED>
static hipError_t dispatch_rows(RowClass c, BatchArgs& b, hipStream_t s) {
    if (c.team == 64) {
        switch (c.keep) {
            case 1: return launch_rows<Policy, DT, 64, 1, PACKED>(b, s);
            case 2: return launch_rows<Policy, DT, 64, 2, PACKED>(b, s);
            case 4: return launch_rows<Policy, DT, 64, 4, PACKED>(b, s);
            default: return launch_rows<Policy, DT, 64, 8, PACKED>(b, s);
        }
    }
    switch (c.keep) {
        case 4: return launch_rows<Policy, DT, 256, 4, PACKED>(b, s);
        case 8: return launch_rows<Policy, DT, 256, 8, PACKED>(b, s);
        default: return launch_rows<Policy, DT, 256, 16, PACKED>(b, s);
    }
}

// The error metric of whole-row groups for the tensors of one ErrBatchArgs launch (class by class, like run_rows).
// false: some tensor does not fit (the caller falls back to the generic kernel for the whole launch).
template <typename Policy, int DT>
static bool run_err_rows(const ErrBatchArgs& a, float* ws, hipStream_t s) {
    constexpr int eb = DT == NNCF_AMD_F32 ? 4 : 2;
    static_assert(kErrBatch <= kMaxBatch, "one BatchArgs takes a whole ErrBatchArgs launch");
    for (int i = 0; i < a.n; ++i)
        if (((int64_t)a.cols[i] * eb) % 16 != 0 || !is16(a.w[i]) || row_class((int64_t)a.cols[i] * eb / 16).team == 0) return false;
    char done[kErrBatch] = {};
    for (int first = 0; first < a.n; ++first) {
        if (done[first]) continue;
        const RowClass c = row_class((int64_t)a.cols[first] * eb / 16);
        BatchArgs b = {};
        b.mode = a.mode;
        int j = 0;
        uint32_t acc = 0;
        for (int k = first; k < a.n; ++k) {
            const RowClass ck = row_class((int64_t)a.cols[k] * eb / 16);
            if (done[k] || ck.team != c.team || ck.keep != c.keep) continue;
            b.tile_start[j] = acc;
            b.nfull[j] = (uint32_t)((int64_t)a.cols[k] * eb / 16);
            b.w[j] = a.w[k];
            b.scale[j] = ws + a.row_start[k];
            acc += a.row_start[k + 1] - a.row_start[k];
            b.tile_start[++j] = acc;
            done[k] = 1;
        }
        b.n = j;
        const unsigned blocks = c.team == 64 ? (acc + 3) / 4 : acc;
        const dim3 grid(blocks), blk(256);
        if (c.team == 64) {
            switch (c.keep) {
                case 1: hipLaunchKernelGGL((k_rows_batch<Policy, DT, 64, 1, false, true>), grid, blk, 0, s, b); break;
                case 2: hipLaunchKernelGGL((k_rows_batch<Policy, DT, 64, 2, false, true>), grid, blk, 0, s, b); break;
                case 4: hipLaunchKernelGGL((k_rows_batch<Policy, DT, 64, 4, false, true>), grid, blk, 0, s, b); break;
                default: hipLaunchKernelGGL((k_rows_batch<Policy, DT, 64, 8, false, true>), grid, blk, 0, s, b); break;
            }
        } else {
            switch (c.keep) {
                case 4: hipLaunchKernelGGL((k_rows_batch<Policy, DT, 256, 4, false, true>), grid, blk, 0, s, b); break;
                case 8: hipLaunchKernelGGL((k_rows_batch<Policy, DT, 256, 8, false, true>), grid, blk, 0, s, b); break;
                default: hipLaunchKernelGGL((k_rows_batch<Policy, DT, 256, 16, false, true>), grid, blk, 0, s, b); break;
            }
        }
    }
    return true;
}

// Can every item run as whole-row groups?  (16-byte rows, aligned pointers, rows of at most 4096 vectors)
static inline bool rows_fit(int w_dtype, const BatchItem* items, int n, int64_t group_size) {
    const int eb = w_dtype == NNCF_AMD_F32 ? 4 : 2;
    for (int i = 0; i < n; ++i) {
        const BatchItem& it = items[i];
        if (it.rows <= 0 || it.cols <= 0 || (group_size != -1 && group_size != it.cols)) return false;
        if ((it.cols * eb) % 16 != 0 || row_class(it.cols * eb / 16).team == 0) return false;
        if (it.rows > ((int64_t)1 << 30)) return false;
        if (!is16(it.w) || !it.packed || !is16(it.packed) || !it.scale) return false;
    }
    return n > 0;
}

// Runs the items class by class (tensors of one team/keep class share launches of up to kMaxBatch tensors).
template <typename Policy, bool PACKED>
static int run_rows(int w_dtype, const BatchItem* items, int n, int mode, const char* what, hipStream_t s) {
    const int eb = w_dtype == NNCF_AMD_F32 ? 4 : 2;
    std::vector<char> done(n, 0);
    for (int first = 0; first < n; ++first) {
        if (done[first]) continue;
        const RowClass c = row_class(items[first].cols * eb / 16);
        BatchArgs b = {};
        b.mode = mode;
        int j = 0;
        uint32_t acc = 0;
        auto flush = [&]() -> int {
            if (j == 0) return NNCF_AMD_OK;
            b.n = j;
            hipError_t e;
            switch (w_dtype) {
                case NNCF_AMD_F32: e = dispatch_rows<Policy, NNCF_AMD_F32, PACKED>(c, b, s); break;
                case NNCF_AMD_F16: e = dispatch_rows<Policy, NNCF_AMD_F16, PACKED>(c, b, s); break;
                case NNCF_AMD_BF16: e = dispatch_rows<Policy, NNCF_AMD_BF16, PACKED>(c, b, s); break;
                default:
                    set_error("%s: unsupported weight dtype %d", what, w_dtype);
                    return NNCF_AMD_ERR_DTYPE;
            }
            if (e != hipSuccess) {
                set_error("%s: launch failed: %s", what, hipGetErrorString(e));
                return NNCF_AMD_ERR_LAUNCH;
            }
            j = 0;
            acc = 0;
            b = BatchArgs{};
            b.mode = mode;
            return launch_status(what);
        };
        for (int k = first; k < n; ++k) {
            if (done[k]) continue;
            const RowClass ck = row_class(items[k].cols * eb / 16);
            if (ck.team != c.team || ck.keep != c.keep) continue;
            if (j == kMaxBatch || (int64_t)acc + items[k].rows > ((int64_t)1 << 30)) {
                int st = flush();
                if (st != NNCF_AMD_OK) return st;
            }
            b.tile_start[j] = acc;
            b.nfull[j] = (uint32_t)(items[k].cols * eb / 16);
            b.w[j] = items[k].w;
            b.packed[j] = items[k].packed;
            b.scale[j] = items[k].scale;
            b.aux[j] = items[k].aux;
            acc += (uint32_t)items[k].rows;
            b.tile_start[j + 1] = acc;
            ++j;
            done[k] = 1;
        }
        int st = flush();
        if (st != NNCF_AMD_OK) return st;
    }
    return NNCF_AMD_OK;
}

}  // namespace wc
