// wc_group_kernels.h -- the three group-wise lowering kernel skeletons, shared by the integer
// (wc_int_quant.hip) and float-codebook (wc_float_quant.hip) policies.
//
//   grouped : one 4-KiB tile per wave, staged through LDS by DMA so that a lane owns 64 contiguous bytes; a group is
//             a power-of-two number of 16-byte vectors inside the tile (DPP reductions over its lanes).  Single-tensor
//             form with any output mix (k_grouped) and multi-tensor packed form (k_grouped_batch).
//   row     : group = one wave or one 256-thread block (long rows: per-channel INT8, fp32 g >= 512 ...);
//             up to KEEP vectors per thread stay in registers between the statistics and quantize passes
//             (k_row here; the multi-tensor form is in wc_row_kernels.h).
//   generic : one thread per group, any group size / inner stride / alignment (small or odd shapes).
//
// A Policy supplies the arithmetic:
//   Policy(const GroupArgs&, float* lds)        per-thread constructor (lds: kPolicyLds floats, block shared)
//   static void init_shared(float* lds)          fills block-shared tables (followed by __syncthreads())
//   Stat stat_init(); void stat_add(Stat&, float); template<int L> void stat_lanes(Stat&);
//   void stat_block(Stat&, float* smem16)        block (4 waves) all-reduce
//   Par compute(const Stat&); Par load(const GroupArgs&, int64 g); void store(const GroupArgs&, int64 g, const Par&)
//   void quant(float w, const Par&, int& code, float& value)   scalar form (generic kernel)
//   float dq(int code, float value, const Par&);  int nibble(int code)
//   template<int E> void stat_add_vec(Stat&, const float*); template<int E> void emit(args, par, f, v)  vector forms
//   emit_lane<NE, GPL, NT, DIV> / pack_lane<NE, GPL>: a lane's NE consecutive elements (tile kernels)
#pragma once
#include "wc_common.h"

namespace wc {

// streaming (non-temporal) store for outputs nobody re-reads soon: keeps them out of L2/MALL's way
template <bool NT, typename T>
__device__ __forceinline__ void st(T* p, T v) {
    if constexpr (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct GroupArgs {
    const void* w;
    const float* scale_in;
    const void* zp_in;
    int zp_dtype;
    void* codes;        // 1 byte per element
    uint8_t* packed;    // 2 elements per byte
    float* values;      // float policies: fp32 quantile values
    float* scale_out;
    int32_t* zp_out;
    uint8_t* e8m0_out;
    float* dq;
    int64_t n_groups;
    int64_t gs;
    int64_t inner;
    int mode;
};

static inline bool is16(const void* p) { return ((uintptr_t)p & 15u) == 0; }

constexpr int kPolicyLds = 128;  // floats of block-shared table space handed to a policy

template <int E, typename Policy, typename Par>
__device__ __forceinline__ void emit_vec(const GroupArgs& a, const Policy& pol, const Par& par, const float* f,
                                         int64_t v) {
    int code[E];
    float val[E];
#pragma unroll
    for (int i = 0; i < E; ++i) pol.quant(f[i], par, code[i], val[i]);
    if (a.codes) {
        uint32_t pk[E / 4];
#pragma unroll
        for (int i = 0; i < E / 4; ++i) {
            pk[i] = ((uint32_t)code[4 * i] & 0xffu) | (((uint32_t)code[4 * i + 1] & 0xffu) << 8) |
                    (((uint32_t)code[4 * i + 2] & 0xffu) << 16) | (((uint32_t)code[4 * i + 3] & 0xffu) << 24);
        }
        if constexpr (E == 8) {
            ((uint2*)a.codes)[v] = make_uint2(pk[0], pk[1]);
        } else {
            ((uint32_t*)a.codes)[v] = pk[0];
        }
    }
    if (a.packed) {
        uint32_t pk = 0;
#pragma unroll
        for (int i = 0; i < E; ++i) pk |= ((uint32_t)pol.nibble(code[i]) & 0xfu) << (4 * i);
        if constexpr (E == 8) {
            ((uint32_t*)a.packed)[v] = pk;
        } else {
            ((uint16_t*)a.packed)[v] = (uint16_t)pk;
        }
    }
    if (a.values) {
        float4* o = (float4*)a.values;
#pragma unroll
        for (int i = 0; i < E / 4; ++i)
            o[v * (E / 4) + i] = make_float4(val[4 * i], val[4 * i + 1], val[4 * i + 2], val[4 * i + 3]);
    }
    if (a.dq) {
        float4* o = (float4*)a.dq;
#pragma unroll
        for (int i = 0; i < E / 4; ++i)
            o[v * (E / 4) + i] =
                make_float4(pol.dq(code[4 * i], val[4 * i], par), pol.dq(code[4 * i + 1], val[4 * i + 1], par),
                            pol.dq(code[4 * i + 2], val[4 * i + 2], par), pol.dq(code[4 * i + 3], val[4 * i + 3], par));
    }
}

// ------------------------------------------------------------------------------------------------ grouped
// LDS-staged groups.  A wave owns tiles of 64*VPL consecutive 16-byte vectors (VPL KiB).  The tile is brought
// in by VPL fully coalesced LDS-DMA loads (global_load_lds_dwordx4: no VGPR round trip) into a wave-private
// LDS image, from which every lane reads VPL *consecutive* vectors (VPL*E elements of one group, or of VPL/VPG
// whole groups).  So the per-group work (cross-lane reduction, scale / reciprocal / zero point) is amortised over
// VPL*E elements per lane instead of E, and packed output leaves as one 16-byte store per lane.
//   LDS image: vector (l, k) (lane l, k-th vector of that lane) sits in 16-byte slot l*VPL + (k ^ ((l>>2)&3));
//   the XOR spreads the 16 lanes that one ds_read_b128 services together over all 64 banks (conflict free).
//   The DMA writes slots linearly (slot = u*64 + lane), so the permutation is applied to the global SOURCE
//   address; one DMA instruction still reads one contiguous KiB.
// One tile per wave, a grid of tiles/4 blocks (see the multi-tensor kernel below for why).
// GIVEN: the group parameters come from the caller (scale_in [, zp_in]); otherwise they are computed from the tile and
// the codes may take the correctly-rounded-quotient path (Policy::emit_lane<..., DIV = true>).
constexpr int kVPL = 4;

template <typename Policy, int DT, int VPG, bool GIVEN>
__global__ __launch_bounds__(256) void k_grouped(GroupArgs a) {
    constexpr int VPL = kVPL;
    constexpr int E = Elem<DT>::kPerVec;
    constexpr int NE = VPL * E;                         // elements per lane
    constexpr int GPL = VPG >= VPL ? 1 : VPL / VPG;     // groups per lane
    constexpr int LPG = VPG >= VPL ? VPG / VPL : 1;     // lanes per group
    constexpr int SG = NE / GPL;                        // elements of one group held by one lane
    constexpr int TILE = 64 * VPL;                      // vectors per tile
    static_assert(LPG <= 64, "group must fit in one wave");
    __shared__ Vec16 stage[4][TILE];
    __shared__ __align__(16) float lds[kPolicyLds];
    const int64_t nvec = a.n_groups * VPG;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // scalar: tile bases stay in SGPRs
    Vec16* my = stage[wave];
    const int64_t base = ((int64_t)blockIdx.x * 4 + wave) * TILE;  // first vector of this wave's tile
    const bool have = base < nvec;
    if (have) {
        // source permutation of the DMA (see above): slot s = u*64 + lane holds vector (l = s>>2, k = (s&3)^((s>>4)&3))
        const char* tbase = (const char*)a.w + base * 16;  // wave-uniform
        const uint32_t last = base + TILE <= nvec ? 0xffffffffu : (uint32_t)(nvec - base - 1) * 16u;  // partial tile: clamp
#pragma unroll
        for (int u = 0; u < VPL; ++u) {
            const int sidx = u * 64 + lane;
            const uint32_t off = (uint32_t)((sidx & ~3) + ((sidx & 3) ^ ((sidx >> 4) & 3))) * 16u;
            const uint32_t o2 = off < last ? off : last;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(tbase + o2),
                                             (__attribute__((address_space(3))) void*)(my + u * 64), 16, 0, 0);
        }
    }
    Policy::init_shared(lds);
    __syncthreads();
    if (!have) return;
    const Policy pol(a, lds);
    const int rot = (lane >> 2) & 3;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float f[NE];
#pragma unroll
    for (int k = 0; k < VPL; ++k) {
        const Vec16 raw = my[lane * VPL + (k ^ rot)];
        unpack<DT>(raw, f + k * E);
    }
    const int64_t v0 = base + (int64_t)lane * VPL;  // first vector of this lane
    const bool live = v0 < nvec;                    // host guarantees nvec % VPL == 0: lanes are all-or-nothing
    const int64_t g0 = v0 / VPG;
    typename Policy::Par par[GPL];
    if constexpr (!GIVEN) {
#pragma unroll
        for (int gi = 0; gi < GPL; ++gi) {
            typename Policy::Stat st = pol.stat_init();
            pol.template stat_add_vec<SG>(st, f + gi * SG);
            if constexpr (LPG > 1) pol.template stat_lanes<LPG>(st);  // all 64 lanes take part
            par[gi] = pol.compute(st);
        }
    } else {
#pragma unroll
        for (int gi = 0; gi < GPL; ++gi) par[gi] = pol.load(a, live ? g0 + gi : 0);
    }
    if (live) {
        if ((lane & (LPG - 1)) == 0) {
#pragma unroll
            for (int gi = 0; gi < GPL; ++gi) pol.template store<false>(a, g0 + gi, par[gi]);
        }
        pol.template emit_lane<NE, GPL, false, !GIVEN>(a, par, f, v0 * E);
    }
}

// ------------------------------------------------------------------------------------------------ grouped, multi-tensor
// The data-free compress path over MANY tensors in one launch (a launch boundary costs ~10 us of ramp-up, drain and
// dispatch on MI355X, as much as compressing a 20 MB tensor).  Tensors are cut into the same 64*VPL-vector tiles;
// tile ids run through all tensors of the batch (tile_start = prefix sums).  Always "lean": packed codes +
// computed parameters only.
//
// ONE TILE PER WAVE, a grid of tiles/4 blocks: the hardware's workgroup dispatcher hands out the tiles in address
// order as waves retire, and occupancy (6 blocks = 24 waves = 96 KiB of tiles in flight per CU) hides the load
// latency.  Measured on MI355X (bench.py, Llama-3-8B INT4_SYM g128): 0.762 of the 8 TB/s roofline, the rate of a
// pure-traffic probe of the same shape (tools/probe.hip: 0.77).  The kernel this replaces kept one resident set of
// waves alive over the whole launch -- tiles claimed from per-part atomic counters, the next tile prefetched by
// LDS-DMA under the arithmetic, a second LDS image per wave in its last version -- and stopped at 0.71 whatever
// was tuned (blocks per CU, parts, tiles per claim): ablation builds showed its data movement alone at 0.72-0.73
// with the packed stores and at 0.77 without them, i.e. the persistent waves' store stream, not their loads or
// their arithmetic, held it back; short-lived waves in dispatch order do not have that problem.
constexpr int kMaxBatch = 92;  // sizeof(BatchArgs) must stay under the 4 KiB of kernel arguments
struct BatchArgs {
    int n;
    int mode;
    uint32_t tile_start[kMaxBatch + 1];  // 32-bit tile ids: every per-tile decision is wave-uniform SALU work
    uint32_t nfull[kMaxBatch];           // whole tiles of the tensor
    uint32_t rem[kMaxBatch];             // vectors in its trailing partial tile (0: none)
    const void* w[kMaxBatch];
    uint8_t* packed[kMaxBatch];
    float* scale[kMaxBatch];
    void* aux[kMaxBatch];  // int32 zero points (asymmetric integer modes) or uint8 E8M0 scales (MXFP4); may be NULL
};
static_assert(sizeof(BatchArgs) <= 4096, "BatchArgs is passed by value as the kernel argument");
constexpr int64_t kMaxBatchVec = (int64_t)1 << 30;  // vectors per tensor and per launch (32-bit tile / group indices)

// appends tensor j (nvec 16-byte vectors) to a batch; returns the running tile count
static inline uint32_t batch_set(BatchArgs& b, int j, uint32_t acc, int64_t nvec, const void* w, uint8_t* packed,
                                 float* scale, void* aux) {
    constexpr int64_t TILE = 64 * kVPL;
    b.tile_start[j] = acc;
    b.nfull[j] = (uint32_t)(nvec / TILE);
    b.rem[j] = (uint32_t)(nvec % TILE);
    b.w[j] = w;
    b.packed[j] = packed;
    b.scale[j] = scale;
    b.aux[j] = aux;
    acc += (uint32_t)((nvec + TILE - 1) / TILE);
    b.tile_start[j + 1] = acc;
    return acc;
}

// one 16-byte-per-lane LDS-DMA load, non-temporal: global (SGPR base + 32-bit VGPR offset) -> LDS (M0 + lane*16).
__device__ __forceinline__ void dma16_nt(const char* sbase, uint32_t voff, uint32_t lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt"
                 :
                 : "v"(voff), "s"(sbase), "s"(lds_addr)
                 : "memory", "m0");
}

// Per wave: locate the tensor of its tile (binary search over the prefix sums, scalar), 4 DMA instructions bring the
// tile into the wave's LDS image (source addresses permuted as in k_grouped; a partial tile clamps to its last
// vector), every lane reads its 64 contiguous bytes back, min/max over the group's lanes by DPP, parameters,
// exact quantization (Policy::pack_lane), one 16-byte store of packed codes per lane, parameter stores by the
// group's first lane.
template <typename Policy, int DT, int VPG>
__global__ __launch_bounds__(256) void k_grouped_batch(BatchArgs b) {
    constexpr int VPL = kVPL;
    constexpr int E = Elem<DT>::kPerVec;
    constexpr int NE = VPL * E;
    constexpr int GPL = VPG >= VPL ? 1 : VPL / VPG;
    constexpr int LPG = VPG >= VPL ? VPG / VPL : 1;
    constexpr int SG = NE / GPL;
    constexpr int TILE = 64 * VPL;
    constexpr int NW = NE / 8;
    __shared__ Vec16 stage[4][TILE];
    __shared__ __align__(16) float lds[kPolicyLds];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t t = blockIdx.x * 4 + wave;
    const uint32_t total = b.tile_start[b.n];
    const uint32_t lds_base = (uint32_t)(uintptr_t)stage[wave];
    // tensor of this tile: binary search over the prefix sums (wave-uniform, scalar loads)
    int lo_i = 0, hi_i = b.n;
    if (t < total) {
        while (hi_i - lo_i > 1) {
            const int mid = (lo_i + hi_i) >> 1;
            if (t >= b.tile_start[mid]) lo_i = mid;
            else hi_i = mid;
        }
        const uint32_t lt = t - b.tile_start[lo_i];
        const char* tbase = (const char*)b.w[lo_i] + (uint64_t)lt * (TILE * 16);
        const uint32_t nfull = b.nfull[lo_i];
        const uint32_t last = lt < nfull ? 0xffffffffu : (b.rem[lo_i] - 1) * 16u;
#pragma unroll
        for (int u = 0; u < VPL; ++u) {
            const int sidx = u * 64 + lane;
            const uint32_t off = (uint32_t)((sidx & ~3) + ((sidx & 3) ^ ((sidx >> 4) & 3))) * 16u;
            dma16_nt(tbase, off < last ? off : last, lds_base + u * 1024);
        }
    }
    Policy::init_shared(lds);
    __syncthreads();
    if (t >= total) return;
    GroupArgs o = {};
    o.mode = b.mode;
    const Policy pol(o, lds);
    const int rot = (lane >> 2) & 3;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float f[NE];
#pragma unroll
    for (int k = 0; k < VPL; ++k) {
        const Vec16 raw = stage[wave][lane * VPL + (k ^ rot)];
        unpack<DT>(raw, f + k * E);
    }
    const uint32_t lt = t - b.tile_start[lo_i];
    const uint32_t in_tile = lt < b.nfull[lo_i] ? (uint32_t)TILE : b.rem[lo_i];
    const bool live = (uint32_t)(lane * VPL) < in_tile;
    const uint32_t v0 = lt * TILE + lane * VPL;
    typename Policy::Par par[GPL];
#pragma unroll
    for (int gi = 0; gi < GPL; ++gi) {
        typename Policy::Stat st = pol.stat_init();
        pol.template stat_add_vec<SG>(st, f + gi * SG);
        if constexpr (LPG > 1) pol.template stat_lanes<LPG>(st);
        par[gi] = pol.compute(st);
    }
    uint32_t out[NW];
    pol.template pack_lane<NE, GPL>(par, f, out);
    if (live) {
        uint8_t* dst = b.packed[lo_i] + (uint64_t)v0 * (E / 2);
        if constexpr (NW == 1) st<true>((uint32_t*)dst, out[0]);
        if constexpr (NW == 2) st<true>((u32x2*)dst, u32x2{out[0], out[1]});
        if constexpr (NW == 4) st<true>((u32x4*)dst, u32x4{out[0], out[1], out[2], out[3]});
        if ((lane & (LPG - 1)) == 0) {
            o.scale_out = b.scale[lo_i];
            o.zp_out = (int32_t*)b.aux[lo_i];
            o.e8m0_out = (uint8_t*)b.aux[lo_i];
#pragma unroll
            for (int gi = 0; gi < GPL; ++gi) pol.template store<true>(o, v0 / VPG + gi, par[gi]);
        }
    }
}


template <typename Policy, int DT, int VPG>
static void launch_batch(BatchArgs& b, hipStream_t s) {
    const int64_t blocks = ((int64_t)b.tile_start[b.n] + 3) / 4;  // < 2^29
    if (blocks > 0)
        hipLaunchKernelGGL((k_grouped_batch<Policy, DT, VPG>), dim3((unsigned)blocks), dim3(256), 0, s, b);
    // errors surface through launch_status()
}

template <typename Policy, int DT>
static bool dispatch_batch(int vpg, BatchArgs& b, hipStream_t s) {
    switch (vpg) {
        case 1: launch_batch<Policy, DT, 1>(b, s); return true;
        case 2: launch_batch<Policy, DT, 2>(b, s); return true;
        case 4: launch_batch<Policy, DT, 4>(b, s); return true;
        case 8: launch_batch<Policy, DT, 8>(b, s); return true;
        case 16: launch_batch<Policy, DT, 16>(b, s); return true;
        case 32: launch_batch<Policy, DT, 32>(b, s); return true;
        case 64: launch_batch<Policy, DT, 64>(b, s); return true;
        case 128: launch_batch<Policy, DT, 128>(b, s); return true;
        case 256: launch_batch<Policy, DT, 256>(b, s); return true;
        default: return false;
    }
}

// One tensor of a data-free batch.  aux: int32 zero points / uint8 E8M0 scales / NULL.
struct BatchItem {
    const void* w;
    int64_t rows, cols;
    uint8_t* packed;
    float* scale;
    void* aux;
};

// Queues a whole batch with as few launches as possible; returns false if an item does not fit the grouped
// kernel (unaligned pointers, group bytes not a power-of-two multiple of 16 up to 4 KiB, ragged lanes), in which
// case the caller runs that batch tensor by tensor.
template <typename Policy>
static bool batch_fits(int w_dtype, const BatchItem* items, int n, int64_t group_size, int* vpg_out) {
    const int eb = w_dtype == NNCF_AMD_F32 ? 4 : 2;
    int vpg = -1;
    for (int i = 0; i < n; ++i) {
        const int64_t gs = group_size == -1 ? items[i].cols : group_size;
        if (gs <= 0 || items[i].rows <= 0 || items[i].cols % gs != 0) return false;
        const int64_t gb = gs * eb;
        if (gb % 16 != 0) return false;
        const int64_t v = gb / 16;
        if (v > 256 || (v & (v - 1)) != 0) return false;
        if (vpg != -1 && vpg != (int)v) return false;
        vpg = (int)v;
        const int64_t nvec = items[i].rows * items[i].cols * eb / 16;
        if (nvec % kVPL != 0 || nvec > kMaxBatchVec) return false;
        if (!is16(items[i].w) || !items[i].packed || !is16(items[i].packed) || !items[i].scale) return false;
    }
    *vpg_out = vpg;
    return n > 0;
}

template <typename Policy>
static int run_batch(int w_dtype, const BatchItem* items, int n, int vpg, int mode, const char* what, hipStream_t s) {
    const int eb = w_dtype == NNCF_AMD_F32 ? 4 : 2;
    for (int first = 0; first < n;) {
        BatchArgs b = {};
        b.mode = mode;
        uint32_t acc = 0;
        int64_t vecs = 0;
        int j = 0;
        for (; j < kMaxBatch && first + j < n; ++j) {
            const BatchItem& it = items[first + j];
            const int64_t nvec = it.rows * it.cols * eb / 16;
            if (j > 0 && vecs + nvec > kMaxBatchVec) break;  // keep the launch's tile count in 32 bits
            vecs += nvec;
            acc = batch_set(b, j, acc, nvec, it.w, it.packed, it.scale, it.aux);
        }
        b.n = j;
        first += j;
        bool ok;
        switch (w_dtype) {
            case NNCF_AMD_F32: ok = dispatch_batch<Policy, NNCF_AMD_F32>(vpg, b, s); break;
            case NNCF_AMD_F16: ok = dispatch_batch<Policy, NNCF_AMD_F16>(vpg, b, s); break;
            case NNCF_AMD_BF16: ok = dispatch_batch<Policy, NNCF_AMD_BF16>(vpg, b, s); break;
            default:
                set_error("%s: unsupported weight dtype %d", what, w_dtype);
                return NNCF_AMD_ERR_DTYPE;
        }
        if (!ok) {
            set_error("%s: internal dispatch error (vpg %d)", what, vpg);
            return NNCF_AMD_ERR_UNSUPPORTED;
        }
        int st = launch_status(what);
        if (st != NNCF_AMD_OK) return st;
    }
    return NNCF_AMD_OK;
}

// ------------------------------------------------------------------------------------------------ row
template <typename Policy, int DT, int TEAM, int KEEP>
__global__ __launch_bounds__(256) void k_row(GroupArgs a, int64_t vpg) {
    constexpr int E = Elem<DT>::kPerVec;
    constexpr int TEAMS_PER_BLOCK = 256 / TEAM;
    __shared__ __align__(16) float lds[kPolicyLds];
    __shared__ float smem[16];
    Policy::init_shared(lds);
    __syncthreads();
    const Policy pol(a, lds);
    const Vec16* __restrict__ wv = (const Vec16*)a.w;
    const int t = threadIdx.x % TEAM;
    const int team = threadIdx.x / TEAM;
    for (int64_t g0 = (int64_t)blockIdx.x * TEAMS_PER_BLOCK; g0 < a.n_groups;
         g0 += (int64_t)gridDim.x * TEAMS_PER_BLOCK) {
        const int64_t g = g0 + team;
        const bool live_g = g < a.n_groups;  // uniform per team (a team is one or more whole waves)
        const int64_t vbase = g * vpg;
        Vec16 keep[KEEP];
        if (live_g) {
#pragma unroll
            for (int k = 0; k < KEEP; ++k) {
                const int64_t j = t + (int64_t)k * TEAM;
                if (j < vpg) keep[k] = wv[vbase + j];
            }
        }
        typename Policy::Par par;
        if (a.scale_in == nullptr) {
            typename Policy::Stat st = pol.stat_init();
            if (live_g) {
#pragma unroll
                for (int k = 0; k < KEEP; ++k) {
                    const int64_t j = t + (int64_t)k * TEAM;
                    if (j < vpg) {
                        float f[E];
                        unpack<DT>(keep[k], f);
                        pol.template stat_add_vec<E>(st, f);
                    }
                }
                for (int64_t j = t + (int64_t)KEEP * TEAM; j < vpg; j += TEAM) {
                    float f[E];
                    unpack<DT>(wv[vbase + j], f);
                    pol.template stat_add_vec<E>(st, f);
                }
            }
            if constexpr (TEAM == 64) {
                pol.template stat_lanes<64>(st);
            } else {
                pol.stat_block(st, smem);  // contains __syncthreads(): live_g is block-uniform for TEAM == 256
            }
            par = pol.compute(st);
        } else {
            par = pol.load(a, live_g ? g : 0);
        }
        if (!live_g) continue;
        if (t == 0) pol.template store<false>(a, g, par);
#pragma unroll
        for (int k = 0; k < KEEP; ++k) {
            const int64_t j = t + (int64_t)k * TEAM;
            if (j < vpg) {
                float f[E];
                unpack<DT>(keep[k], f);
                pol.template emit<E>(a, par, f, vbase + j);
            }
        }
        for (int64_t j = t + (int64_t)KEEP * TEAM; j < vpg; j += TEAM) {
            float f[E];
            unpack<DT>(wv[vbase + j], f);
            pol.template emit<E>(a, par, f, vbase + j);
        }
    }
}

// ------------------------------------------------------------------------------------------------ generic
// element k of group (o, rg, i) lives at ((o*R + rg*gs + k)*inner + i); consecutive threads take consecutive
// `inner` positions, so inner > 1 layouts (reduction along a non-last axis) stay coalesced.
template <typename Policy, int DT>
__global__ __launch_bounds__(256) void k_generic(GroupArgs a) {
    __shared__ __align__(16) float lds[kPolicyLds];
    Policy::init_shared(lds);
    __syncthreads();
    const Policy pol(a, lds);
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= a.n_groups) return;
    const int64_t i = g % a.inner;
    const int64_t org = g / a.inner;
    const int64_t first = org * a.gs * a.inner + i;
    typename Policy::Par par;
    if (a.scale_in == nullptr) {
        typename Policy::Stat st = pol.stat_init();
        for (int64_t k = 0; k < a.gs; ++k) pol.stat_add(st, Elem<DT>::load(a.w, first + k * a.inner));
        par = pol.compute(st);
    } else {
        par = pol.load(a, g);
    }
    pol.template store<false>(a, g, par);
    int prev_nibble = 0;
    for (int64_t k = 0; k < a.gs; ++k) {
        const int64_t idx = first + k * a.inner;
        int code;
        float val;
        pol.quant(Elem<DT>::load(a.w, idx), par, code, val);
        if (a.codes) ((uint8_t*)a.codes)[idx] = (uint8_t)code;
        if (a.values) a.values[idx] = val;
        if (a.dq) a.dq[idx] = pol.dq(code, val, par);
        if (a.packed) {  // host guarantees inner == 1 and an even group size here: pairs never straddle groups
            const int nib = pol.nibble(code) & 15;
            if (k & 1) a.packed[idx >> 1] = (uint8_t)(prev_nibble | (nib << 4));
            prev_nibble = nib;
        }
    }
}

// nibble-packs byte codes (used after the generic kernel): out[k] = ((c[2k]+bias)&15) | ((c[2k+1]+bias)&15) << 4
__global__ __launch_bounds__(256) static void k_pack_from_codes(const uint8_t* codes, uint8_t* packed, int64_t n_pairs,
                                                                int bias) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= n_pairs) return;
    const uint32_t lo = (uint32_t)(codes[2 * k] + bias) & 15u, hi = (uint32_t)(codes[2 * k + 1] + bias) & 15u;
    packed[k] = (uint8_t)(lo | (hi << 4));
}

// ------------------------------------------------------------------------------------------------ host side
static inline int clamp_grid(int64_t blocks) {
    const int64_t cap = 256 * 16;  // 256 CUs x 16 blocks: enough to fill the chip, rest is grid-strided
    return (int)(blocks < 1 ? 1 : (blocks > cap ? cap : blocks));
}

template <typename Policy, int DT, int VPG>
static void launch_grouped(const GroupArgs& a, hipStream_t s) {
    const int64_t nvec = a.n_groups * VPG;
    const int64_t tiles = (nvec + 64 * kVPL - 1) / (64 * kVPL);
    const int64_t blocks = (tiles + 3) / 4;
    const bool lean = a.packed && a.scale_out && !a.scale_in && !a.codes && !a.values && !a.dq && nvec <= kMaxBatchVec;
    if (lean) {
        BatchArgs b = {};
        b.n = 1;
        b.mode = a.mode;
        batch_set(b, 0, 0, nvec, a.w, a.packed, a.scale_out, a.zp_out ? (void*)a.zp_out : (void*)a.e8m0_out);
        launch_batch<Policy, DT, VPG>(b, s);
    } else if (a.scale_in) {
        hipLaunchKernelGGL((k_grouped<Policy, DT, VPG, true>), dim3((unsigned)blocks), dim3(256), 0, s, a);
    } else {
        hipLaunchKernelGGL((k_grouped<Policy, DT, VPG, false>), dim3((unsigned)blocks), dim3(256), 0, s, a);
    }
}

// pack_bias: added to byte codes before nibble packing on the generic path (8 for INT4_SYM)
template <typename Policy, int DT>
static int run_group_op(GroupArgs a, int pack_bias, const char* what, hipStream_t s) {
    constexpr int EB = Elem<DT>::kBytes;
    const bool aligned = is16(a.w) && (!a.codes || is16(a.codes)) && (!a.packed || is16(a.packed)) &&
                         (!a.dq || is16(a.dq)) && (!a.values || is16(a.values));
    const int64_t group_bytes = a.gs * EB;
    if (a.inner == 1 && aligned && group_bytes % 16 == 0) {
        const int64_t vpg = group_bytes / 16;
        const bool whole_lanes = (a.n_groups * vpg) % kVPL == 0;  // k_grouped: a lane is all live or all dead
        if (vpg <= 256 && (vpg & (vpg - 1)) == 0 && whole_lanes) {
            switch ((int)vpg) {
                case 1: launch_grouped<Policy, DT, 1>(a, s); break;
                case 2: launch_grouped<Policy, DT, 2>(a, s); break;
                case 4: launch_grouped<Policy, DT, 4>(a, s); break;
                case 8: launch_grouped<Policy, DT, 8>(a, s); break;
                case 16: launch_grouped<Policy, DT, 16>(a, s); break;
                case 32: launch_grouped<Policy, DT, 32>(a, s); break;
                case 64: launch_grouped<Policy, DT, 64>(a, s); break;
                case 128: launch_grouped<Policy, DT, 128>(a, s); break;
                default: launch_grouped<Policy, DT, 256>(a, s); break;
            }
            return launch_status(what);
        }
        if (vpg < kVPL) goto generic_path;
        if (vpg <= 512) {
            hipLaunchKernelGGL((k_row<Policy, DT, 64, 8>), dim3(clamp_grid((a.n_groups + 3) / 4)), dim3(256), 0, s, a, vpg);
        } else {
            hipLaunchKernelGGL((k_row<Policy, DT, 256, 8>), dim3(clamp_grid(a.n_groups)), dim3(256), 0, s, a, vpg);
        }
        return launch_status(what);
    }
generic_path:
    uint8_t* packed = a.packed;
    if (packed && a.inner == 1 && (a.gs & 1) == 0) packed = nullptr;  // packed inline by k_generic
    if (packed) a.packed = nullptr;
    if (packed && !a.codes) {
        set_error("%s: packed output for this layout (unaligned, odd group size or inner>1) needs codes_out too", what);
        return NNCF_AMD_ERR_UNSUPPORTED;
    }
    hipLaunchKernelGGL((k_generic<Policy, DT>), dim3((unsigned)((a.n_groups + 255) / 256)), dim3(256), 0, s, a);
    int st = launch_status(what);
    if (st != NNCF_AMD_OK) return st;
    if (packed) {
        const int64_t n_pairs = a.n_groups * a.gs / 2;
        hipLaunchKernelGGL(k_pack_from_codes, dim3((unsigned)((n_pairs + 255) / 256)), dim3(256), 0, s,
                           (const uint8_t*)a.codes, packed, n_pairs, pack_bias);
        st = launch_status("pack_from_codes");
    }
    return st;
}

template <typename Policy>
static int run_group_op_dtype(int w_dtype, const GroupArgs& a, int pack_bias, const char* what, hipStream_t s) {
    switch (w_dtype) {
        case NNCF_AMD_F32: return run_group_op<Policy, NNCF_AMD_F32>(a, pack_bias, what, s);
        case NNCF_AMD_F16: return run_group_op<Policy, NNCF_AMD_F16>(a, pack_bias, what, s);
        case NNCF_AMD_BF16: return run_group_op<Policy, NNCF_AMD_BF16>(a, pack_bias, what, s);
        default:
            set_error("%s: unsupported weight dtype %d", what, w_dtype);
            return NNCF_AMD_ERR_DTYPE;
    }
}

// shared argument validation; returns NNCF_AMD_OK, an error, or +1 for "empty tensor, nothing to do"
static inline int check_group_shape(const char* what, const void* weight, int64_t outer, int64_t r, int64_t inner,
                                    int64_t group_size, int64_t& gs) {
    if (outer < 0 || r < 0 || inner < 0 || (group_size < 1 && group_size != -1)) {
        set_error("%s: bad shape outer=%lld r=%lld inner=%lld group_size=%lld", what, (long long)outer, (long long)r,
                  (long long)inner, (long long)group_size);
        return NNCF_AMD_ERR_BAD_ARG;
    }
    gs = group_size == -1 ? r : group_size;
    if (outer * r * inner == 0) return 1;
    if (r % gs != 0) {
        set_error("Channel size %lld should be divisible by size of group %lld.", (long long)r, (long long)gs);
        return NNCF_AMD_ERR_GROUP_SIZE;
    }
    if (!weight) {
        set_error("%s: weight is NULL", what);
        return NNCF_AMD_ERR_BAD_ARG;
    }
    return NNCF_AMD_OK;
}

}  // namespace wc
