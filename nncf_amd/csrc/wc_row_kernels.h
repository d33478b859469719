// wc_row_kernels.h -- multi-tensor per-channel lowering: the quantization group is a whole row of the weight
// (INT8 per-channel, the reference's default mode; 4-bit with group_size = -1), too long for one wave tile.
//
// A team (one wave for rows up to 512 vectors, one 256-thread block above) owns one row at a time, claimed from
// the same counters as the tile kernel (wc_group_kernels.h).  The row is read ONCE, with non-temporal 16-byte
// loads, and stays in registers (KEEP raw vectors per thread) between the min/max pass and the quantize pass;
// what hides the load latency is occupancy (every resident wave has a whole row in flight), not a prefetch.
// Codes leave as 8 bytes (or 4 bytes of nibbles) per lane and vector, streaming stores.
#pragma once
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
template <typename Policy, int DT, int TEAM, int KEEP, bool PACKED>
__global__ __launch_bounds__(256) void k_rows_batch(BatchArgs b) {
    constexpr int E = Elem<DT>::kPerVec;
    __shared__ float lds[kPolicyLds];
    __shared__ typename Policy::Stat wave_stat[2][4];  // block teams: per-wave statistics, double-buffered
    __shared__ uint32_t claim_sh[2];                    // block teams: the claimed row, double-buffered
    int parity = 0;
    // rows are staged by LDS-DMA: the next row lands in LDS while this one is processed from registers
    // (wave-owned rows: a private image per wave; block-owned rows up to 32 KiB: one image per block)
    constexpr bool DMA = TEAM == 64 || KEEP <= 8;
    __shared__ Vec16 stage[TEAM == 64 ? 4 : 1][DMA ? TEAM * KEEP : 1];
    Policy::init_shared(lds);
    __syncthreads();
    GroupArgs o = {};
    o.mode = b.mode;
    const Policy pol(o, lds);
    const int lane = threadIdx.x & 63;
    const int tl = threadIdx.x % TEAM;  // thread inside the team
    const uint32_t total = b.tile_start[b.n];
    const uint32_t part = (blockIdx.x >> 3) & ((1u << b.log2_groups) - 1);
    const uint32_t lo = b.part_lo[part], hi = b.part_lo[part + 1];
    (void)lo;
    (void)total;
    uint32_t* ctr = b.ctr + part * kClaimStride;
    uint32_t* done = b.done + part * kClaimStride;
    const uint32_t cbase = lo;
    // team-uniform claim: a wave claims for itself, a block through thread 0 and LDS
    claim_init();
    auto claim_issue = [&]() {
        if (TEAM == 64 ? lane == 0 : threadIdx.x == 0) claim_async(ctr);
    };
    auto claim_take = [&]() -> uint32_t {  // after the caller's vmcnt(0)
        if constexpr (TEAM == 64) {
            return cbase + (uint32_t)__builtin_amdgcn_readfirstlane((int)claim_value());
        } else {
            // one barrier: the slot is rewritten two rounds later, with the statistics barrier in between
            if (threadIdx.x == 0) claim_sh[parity] = claim_value();
            __syncthreads();
            return cbase + claim_sh[parity];
        }
    };
    // The claim of the next row is always in flight while this one is processed.  Block-owned rows are loaded
    // straight into registers afterwards (a second register set for prefetching costs more in occupancy than it
    // gains: measured); wave-owned rows (<= 8 KiB) are prefetched through LDS instead.
    struct Row {
        uint32_t vpr;
        uint64_t row;
        int i;
    };
    int scan = 0;
    auto locate = [&](uint32_t r) {
        while (r >= b.tile_start[scan + 1]) ++scan;
        Row d;
        d.i = scan;
        d.vpr = b.nfull[scan];
        d.row = r - b.tile_start[scan];
        return d;
    };
    auto load_row = [&](const Row& d, Vec16* dst) {
        const Vec16* src = (const Vec16*)b.w[d.i] + d.row * d.vpr;
#pragma unroll
        for (int k = 0; k < KEEP; ++k) {
            const uint32_t j = tl + k * TEAM;
            if (j < d.vpr) dst[k] = load16_nt(src + j);
        }
    };
    // vector owned by this thread in register k.  Block-owned rows: tl + k * TEAM (what the coalesced register
    // loads give).  Wave-owned rows come out of LDS, so a lane can own PAIRS of adjacent vectors: its codes then
    // leave as one 16-byte store per pair instead of two 8-byte ones.
    constexpr bool PAIRS = DMA && KEEP >= 2;
    auto vec_of = [&](int k) -> uint32_t {
        return PAIRS ? (uint32_t)(2 * TEAM * (k >> 1) + 2 * tl + (k & 1)) : (uint32_t)(tl + k * TEAM);
    };
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lds_base = (uint32_t)(uintptr_t)stage[TEAM == 64 ? wave : 0];  // (unused without DMA)
    auto issue_dma = [&](const Row& d) {  // every wave brings its 64-vector slices of the team's image
        const char* tbase = (const char*)b.w[d.i] + d.row * d.vpr * 16;  // wave-uniform
        const uint32_t last = (d.vpr - 1) * 16u;
        const uint32_t w0 = TEAM == 64 ? 0u : (uint32_t)wave * 64u;
#pragma unroll
        for (int u = 0; u < KEEP; ++u) {
            const uint32_t v = (uint32_t)(u * TEAM) + w0;  // first vector of this instruction
            const uint32_t off = (v + lane) * 16u;
            dma16_nt(tbase, off < last ? off : last, lds_base + v * 16u);
        }
    };
    claim_issue();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    uint32_t r = claim_take();
    parity ^= 1;  // (block teams) the slot just read must not be rewritten before everybody has read it
    if (r >= hi) {  // (a team makes exactly one failing claim, the last thing it does with the counter)
        if (TEAM == 64 ? lane == 0 : threadIdx.x == 0) claim_done(ctr, done, b.teams_on[part]);
        return;
    }
    Row cd = locate(r), nd = cd;
    Vec16 cur[KEEP];
    if constexpr (DMA) issue_dma(cd);
    else load_row(cd, cur);
    claim_issue();
    for (;;) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint32_t rn = claim_take();  // (block teams: contains the barrier after which the whole image has landed)
        if constexpr (DMA) {
#pragma unroll
            for (int k = 0; k < KEEP; ++k) cur[k] = stage[TEAM == 64 ? wave : 0][vec_of(k)];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if constexpr (TEAM == 64) {
                if (rn < hi) {
                    nd = locate(rn);
                    issue_dma(nd);
                    claim_issue();
                }
            }
        }
        typename Policy::Stat st = pol.stat_init();
#pragma unroll
        for (int k = 0; k < KEEP; ++k) {
            const uint32_t j = vec_of(k);
            __builtin_amdgcn_sched_barrier(0);
            if (j < cd.vpr) {
                float f[E];
                unpack<DT>(cur[k], f);
                pol.template stat_add_vec<E>(st, f);
            }
        }
        pol.template stat_lanes<64>(st);
        if constexpr (TEAM != 64) {
            if (lane == 0) wave_stat[parity][wave] = st;
            __syncthreads();  // also: everybody has copied the LDS image to registers, it may be overwritten
            st = wave_stat[parity][0];
#pragma unroll
            for (int q = 1; q < 4; ++q) pol.stat_merge(st, wave_stat[parity][q]);
            parity ^= 1;
            if constexpr (DMA) {
                if (rn < hi) {
                    nd = locate(rn);
                    issue_dma(nd);
                    claim_issue();
                }
            }
        }
        const typename Policy::Par par = pol.compute(st);
        o.scale_out = b.scale[cd.i];
        o.zp_out = (int32_t*)b.aux[cd.i];
        if (tl == 0) pol.template store<true>(o, (int64_t)cd.row, par);
        if constexpr (PACKED) o.packed = b.packed[cd.i];
        else o.codes = b.packed[cd.i];
        // second unpack of the raw vectors (opaque, or the compiler keeps all KEEP*E floats of the first one
        // alive across the reduction) and one vector at a time (no interleaving: registers are what buys occupancy)
        auto opaque = [](Vec16 raw) {
            asm volatile("" : "+v"(raw.w[0]), "+v"(raw.w[1]), "+v"(raw.w[2]), "+v"(raw.w[3]));
            return raw;
        };
        if constexpr (PAIRS) {
#pragma unroll
            for (int k = 0; k < KEEP; k += 2) {
                const uint32_t j = vec_of(k);
                __builtin_amdgcn_sched_barrier(0);
                if (j + 1 < cd.vpr) {
                    float f[2 * E];
                    unpack<DT>(opaque(cur[k]), f);
                    unpack<DT>(opaque(cur[k + 1]), f + E);
                    pol.template quant_n_div<2 * E>(par, f, f);
                    pol.template store_codes<2 * E, true>(o, f, (int64_t)((cd.row * cd.vpr + j) * E));
                } else if (j < cd.vpr) {
                    float f[E];
                    unpack<DT>(opaque(cur[k]), f);
                    pol.template quant_n_div<E>(par, f, f);
                    pol.template store_codes<E, true>(o, f, (int64_t)((cd.row * cd.vpr + j) * E));
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < KEEP; ++k) {
                const uint32_t j = vec_of(k);
                __builtin_amdgcn_sched_barrier(0);
                if (j < cd.vpr) {
                    float f[E];
                    unpack<DT>(opaque(cur[k]), f);
                    pol.template quant_n_div<E>(par, f, f);
                    pol.template store_codes<E, true>(o, f, (int64_t)((cd.row * cd.vpr + j) * E));
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (rn >= hi) break;
        if constexpr (DMA) {
            cd = nd;
        } else {
            cd = locate(rn);
            load_row(cd, cur);
            claim_issue();
        }
    }
    if (TEAM == 64 ? lane == 0 : threadIdx.x == 0) claim_done(ctr, done, b.teams_on[part]);  // rn was the failing claim
}

// One row per team, one launch-wide grid: the workgroup dispatcher hands out the rows in address order, occupancy
// hides the load latency (experimental alternative to the claimed kernel above, NNCF_AMD_ROW_ONCE=1).
template <typename Policy, int DT, int TEAM, int KEEP, bool PACKED>
__global__ __launch_bounds__(256) void k_rows_once(BatchArgs b) {
    constexpr int E = Elem<DT>::kPerVec;
    constexpr int TPB = 256 / TEAM;
    __shared__ float lds[kPolicyLds];
    __shared__ typename Policy::Stat wave_stat[4];
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
    o.scale_out = b.scale[lo_i];
    o.zp_out = (int32_t*)b.aux[lo_i];
    if (tl == 0) pol.template store<true>(o, (int64_t)row, par);
    if constexpr (PACKED) o.packed = b.packed[lo_i];
    else o.codes = b.packed[lo_i];
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
                pol.template quant_n_div<2 * E>(par, f, f);
                pol.template store_codes<2 * E, true>(o, f, (int64_t)((row * vpr + j) * E));
            } else if (j < vpr) {
                float f[E];
                unpack<DT>(opaque(cur[k]), f);
                pol.template quant_n_div<E>(par, f, f);
                pol.template store_codes<E, true>(o, f, (int64_t)((row * vpr + j) * E));
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
                pol.template quant_n_div<E>(par, f, f);
                pol.template store_codes<E, true>(o, f, (int64_t)((row * vpr + j) * E));
            }
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

constexpr int kRowBlocksPerCU = 0;  // 0: as many as fit (occupancy)

template <typename Policy, int DT, int TEAM, int KEEP, bool PACKED>
static hipError_t launch_rows(BatchArgs& b, hipStream_t s) {
    static const int per_cu = tile_knob("NNCF_AMD_ROW_BLOCKS_PER_CU", 1, 16, kRowBlocksPerCU);
    static const int resident = resident_blocks(k_rows_batch<Policy, DT, TEAM, KEEP, PACKED>, per_cu);
    // Parts of the row range (sequential streams): as few as the claim counters allow -- measured on Llama-3-8B INT8
    // (8 KiB and 28 KiB rows): 32 parts 0.63, 16: 0.66, 8: 0.68 of the roofline, 4: 0.52 (one address sustains ~10^8
    // atomics/s) -- so the count follows the shortest row of the launch: ~6 x 10^7 claims/s per counter at the
    // ~3.6 TB/s the rows are read with.
    static const int forced = tile_knob("NNCF_AMD_ROW_CLAIM_GROUPS", 1, kClaimGroupsUsed, 0);
    uint32_t min_vpr = 0xffffffffu;
    for (int j = 0; j < b.n; ++j) min_vpr = b.nfull[j] < min_vpr ? b.nfull[j] : min_vpr;
    int groups = 8;
    while (groups < kClaimGroupsUsed && (double)groups * (double)min_vpr * 16.0 * 6e7 < 3.6e12) groups *= 2;
    if (forced > 0 && (forced & (forced - 1)) == 0) groups = forced;
    constexpr int TPB = 256 / TEAM;
    const int64_t blocks = ((int64_t)b.tile_start[b.n] + TPB - 1) / TPB;
    static const int once = tile_knob("NNCF_AMD_ROW_ONCE", 0, 1, 0);
    if (once) {
        hipLaunchKernelGGL((k_rows_once<Policy, DT, TEAM, KEEP, PACKED>), dim3((unsigned)blocks), dim3(256), 0, s, b);
        return hipPeekAtLastError();
    }
    return launch_claimed(k_rows_batch<Policy, DT, TEAM, KEEP, PACKED>, (int)(blocks < resident ? blocks : resident), TPB, b, s,
                          1, groups);
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
