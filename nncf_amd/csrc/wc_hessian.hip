// wc_hessian.hip -- GPTQ Hessian accumulation H = beta*H + alpha * X^T X on the gfx950 matrix cores.
//
// Reference: GPTQ._calculate_hessian (gptq.py:237-240) does, per calibration item,
//     H *= n/(n+b);  n += b;  X' = float32(X) * sqrt(2/n);  H += X'^T X'
// with an fp32 matmul.  Here X stays in its own dtype in HBM, is upcast while it is staged into LDS and
// multiplied with the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32: a k-ordered fp32 fma chain, no reduced
// precision), alpha = 2/n is applied once in the epilogue.  X^T X is symmetric: only tile pairs (ti <= tj) are
// computed and only the upper triangle is accumulated; a mirror pass (k_mirror_upper) writes the lower one.
//
// Tiling: block = 128x128 of H, 4 waves in 2x2, each wave 64x64 = 2x2 MFMA tiles (64 accumulator VGPRs).
// K-step (tokens) = 16, double-buffered LDS image [16][128] per operand, k-major so that both MFMA operands
// (A[i][k] = X[t0+k][i0+i], B[k][j] = X[t0+k][j0+j]) are conflict-free ds_read_b32 of 32 consecutive floats.
// The token axis is split over gridDim.y when there are too few tile pairs to fill 256 CUs; partial products
// are combined with fp32 atomic adds (H is pre-scaled by beta in a separate pass), so the low-order bits of H
// can differ from run to run by summation order -- well inside the 1e-5 relative contract.
#include <stdlib.h>
#include <type_traits>
#include "wc_common.h"

namespace wc {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int HB = 128;  // tile edge
constexpr int HK = 16;   // tokens per step

__global__ __launch_bounds__(256) void k_scale_matrix(float* h, int64_t n, float beta) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) h[i] = beta == 0.0f ? 0.0f : h[i] * beta;
}

// Block -> (token split, tile pair).  Blocks go round-robin over the 8 XCDs, each with an L2 of its own, and the
// kernels are fed from L2 / Infinity Cache (a 256x256 tile needs 1 KiB of X per 131 kflop).  So the blocks that
// run together on one XCD should share column panels of X: XCD x takes the x-th contiguous eighth of a list in which
// the upper-triangular tile pairs are ordered band by band (4 tile rows), inside a band in column blocks of 8 --
// 32 concurrent blocks then read ~12 panels instead of ~40.
struct PairId {
    int ti, tj, split;
};
__device__ __forceinline__ PairId decode_block(int ntile) {
    const uint32_t pairs = gridDim.x, total = gridDim.x * gridDim.y;
    const uint32_t lin = blockIdx.y * gridDim.x + blockIdx.x;
    const uint32_t xcd = lin & 7, n = lin >> 3;
    const uint32_t logical = xcd * (total >> 3) + (xcd < (total & 7) ? xcd : (total & 7)) + n;
    PairId id;
    id.split = (int)(logical / pairs);
    int q = (int)(logical % pairs);
    // band of 4 tile rows
    int r0 = 0;
    for (;;) {
        const int rows = ntile - r0 < 4 ? ntile - r0 : 4;
        int in_band = 0;
        for (int r = 0; r < rows; ++r) in_band += ntile - (r0 + r);
        if (q < in_band) break;
        q -= in_band;
        r0 += 4;
    }
    const int rows = ntile - r0 < 4 ? ntile - r0 : 4;
    // column block of 8 inside the band (columns start at r0; row r holds columns >= r only)
    for (int c0 = r0;; c0 += 8) {
        const int c1 = c0 + 8 < ntile ? c0 + 8 : ntile;
        for (int r = r0; r < r0 + rows; ++r) {
            const int lo = c0 > r ? c0 : r;
            const int cnt = c1 - lo > 0 ? c1 - lo : 0;
            if (q < cnt) {
                id.ti = r;
                id.tj = lo + q;
                return id;
            }
            q -= cnt;
        }
    }
}

// stages X[t0 .. t0+HK) x [c0 .. c0+HB) into registers as floats (zero outside the matrix)
template <int DT, bool VEC>
struct TileLoader {
    static constexpr int E = VEC ? Elem<DT>::kPerVec : 1;
    static constexpr int PER_THREAD = HK * HB / E / 256;  // vectors (or elements) per thread
    float v[PER_THREAD][E];
    __device__ __forceinline__ void load(const void* x, int64_t t0, int64_t t_end, int64_t c0, int64_t K) {
#pragma unroll
        for (int u = 0; u < PER_THREAD; ++u) {
            const int idx = threadIdx.x + u * 256;
            const int row = idx / (HB / E), cv = idx % (HB / E);
            const int64_t t = t0 + row, c = c0 + (int64_t)cv * E;
            if (t < t_end && c < K) {
                if constexpr (VEC) {
                    unpack<DT>(*(const Vec16*)((const char*)x + (t * K + c) * Elem<DT>::kBytes), v[u]);
                } else {
                    v[u][0] = Elem<DT>::load(x, t * K + c);
                }
            } else {
#pragma unroll
                for (int i = 0; i < E; ++i) v[u][i] = 0.0f;
            }
        }
    }
    __device__ __forceinline__ void store(float (*lds)[HB]) const {
#pragma unroll
        for (int u = 0; u < PER_THREAD; ++u) {
            const int idx = threadIdx.x + u * 256;
            const int row = idx / (HB / E), cv = idx % (HB / E);
            if constexpr (E == 4) {
                *(float4*)&lds[row][cv * 4] = make_float4(v[u][0], v[u][1], v[u][2], v[u][3]);
            } else if constexpr (E == 8) {
                *(float4*)&lds[row][cv * 8] = make_float4(v[u][0], v[u][1], v[u][2], v[u][3]);
                *(float4*)&lds[row][cv * 8 + 4] = make_float4(v[u][4], v[u][5], v[u][6], v[u][7]);
            } else {
                lds[row][cv] = v[u][0];
            }
        }
    }
};

template <int DT, bool VEC>
__global__ __launch_bounds__(256) void k_syrk(float* __restrict__ h, const void* __restrict__ x, int64_t T, int64_t K,
                                              float alpha, int ntile, int64_t rows_per_split) {
    __shared__ __attribute__((aligned(16))) float sa[2][HK][HB];
    __shared__ __attribute__((aligned(16))) float sb[2][HK][HB];
    const PairId id = decode_block(ntile);
    const int ti = id.ti, tj = id.tj;
    const int64_t i0 = (int64_t)ti * HB, j0 = (int64_t)tj * HB;
    const int64_t t_begin = (int64_t)id.split * rows_per_split;
    const int64_t t_end = t_begin + rows_per_split < T ? t_begin + rows_per_split : T;
    if (t_begin >= t_end) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wi = wave >> 1, wj = wave & 1;  // this wave's 64x64 quadrant
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
    TileLoader<DT, VEC> la, lb;
    la.load(x, t_begin, t_end, i0, K);
    lb.load(x, t_begin, t_end, j0, K);
    la.store(sa[0]);
    lb.store(sb[0]);
    __syncthreads();
    int buf = 0;
    const int kl = lane >> 5, cl = lane & 31;
    for (int64_t t0 = t_begin; t0 < t_end; t0 += HK) {
        const bool more = t0 + HK < t_end;
        if (more) {
            la.load(x, t0 + HK, t_end, i0, K);
            lb.load(x, t0 + HK, t_end, j0, K);
        }
#pragma unroll
        for (int k = 0; k < HK; k += 2) {
            float a[2], b[2];
#pragma unroll
            for (int m = 0; m < 2; ++m) a[m] = sa[buf][k + kl][wi * 64 + m * 32 + cl];
#pragma unroll
            for (int n = 0; n < 2; ++n) b[n] = sb[buf][k + kl][wj * 64 + n * 32 + cl];
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], b[n], acc[m][n], 0, 0, 0);
        }
        if (more) {
            la.store(sa[buf ^ 1]);
            lb.store(sb[buf ^ 1]);
        }
        __syncthreads();
        buf ^= 1;
    }
    // epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t gi = i0 + wi * 64 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * kl;
                const int64_t gj = j0 + wj * 64 + n * 32 + cl;
                if (gi <= gj && gj < K) {  // upper triangle only: k_mirror_upper fills in the rest
                    const float val = alpha * acc[m][n][r];
                    atomicAdd(&h[gi * K + gj], val);
                }
            }
}

// ------------------------------------------------------------------------------------------------ 16-bit inputs
// bf16 / f16 activations: the products of two 16-bit floats are exact in fp32, so the 16-bit MFMA
// (v_mfma_f32_32x32x16_{bf16,f16}: 16x the rate of the fp32 one) gives the same sums as the reference's fp32
// matmul up to the order of the additions.  X stays token-major in LDS exactly as it is in memory ([token][col],
// 16-byte copies, row pitch 320 B so that four token rows fall on disjoint banks); the transpose the MFMA operand
// layout needs (a lane holds 8 consecutive k = tokens of one column) is done by the LDS read itself:
// ds_read_b64_tr_b16 hands lane L of a 16-lane group column L of a 4(token) x 16(col) tile when input lane
// q + 4m supplies the address of (token m, columns 4q..4q+3) -- semantics probed by tools/trprobe.hip.
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int HK16 = 32;                 // tokens per stage (two MFMA k-steps)
constexpr int HP16 = (HB * 2 + 64) / 2;  // row pitch in 16-bit elements

template <int DT>
__device__ __forceinline__ f32x16 mfma16(const s16x4& lo, const s16x4& hi, const f32x16& c) {
    struct Pair {
        s16x4 a, b;
    } p = {lo, hi};
    if constexpr (DT == NNCF_AMD_BF16)
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, p), __builtin_bit_cast(bf16x8, p), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, p), __builtin_bit_cast(f16x8, p), c, 0, 0, 0);
}
template <int DT>
__device__ __forceinline__ f32x16 mfma16(const s16x4& alo, const s16x4& ahi, const s16x4& blo, const s16x4& bhi,
                                         const f32x16& c) {
    struct Pair {
        s16x4 a, b;
    } pa = {alo, ahi}, pb = {blo, bhi};
    if constexpr (DT == NNCF_AMD_BF16)
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, pa), __builtin_bit_cast(bf16x8, pb), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, pa), __builtin_bit_cast(f16x8, pb), c, 0, 0, 0);
}

// raw 16-byte vectors of X[t0 .. t0+HK16) x [c0 .. c0+HB) (zero outside the matrix); K % 8 == 0, x 16-byte aligned
struct RawTile {
    Vec16 v[2];
    __device__ __forceinline__ void load(const void* x, int64_t t0, int64_t t_end, int64_t c0, int64_t K) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int idx = threadIdx.x + u * 256;
            const int row = idx >> 4, cv = idx & 15;
            const int64_t t = t0 + row, c = c0 + cv * 8;
            // branch-free: clamped address, then zero what lies outside (no control flow between the loads)
            const bool ok = t < t_end && c < K;
            const int64_t tc = t < t_end ? t : t_end - 1, cc = c < K ? c : K - 8;
            v[u] = *(const Vec16*)((const char*)x + (tc * K + cc) * 2);
#pragma unroll
            for (int q = 0; q < 4; ++q) v[u].w[q] = ok ? v[u].w[q] : 0u;
        }
    }
    __device__ __forceinline__ void store(uint16_t* lds) const {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int idx = threadIdx.x + u * 256;
            const int row = idx >> 4, cv = idx & 15;
            *(Vec16*)(lds + row * HP16 + cv * 8) = v[u];
        }
    }
};

template <int DT>
__global__ __launch_bounds__(256) void k_syrk16(float* __restrict__ h, const void* __restrict__ x, int64_t T, int64_t K,
                                                float alpha, int ntile, int64_t rows_per_split) {
    __shared__ __attribute__((aligned(16))) uint16_t sa[2][HK16 * HP16];
    __shared__ __attribute__((aligned(16))) uint16_t sb[2][HK16 * HP16];
    const PairId id = decode_block(ntile);
    const int ti = id.ti, tj = id.tj;
    const bool diag = ti == tj;
    const int64_t i0 = (int64_t)ti * HB, j0 = (int64_t)tj * HB;
    const int64_t t_begin = (int64_t)id.split * rows_per_split;
    const int64_t t_end = t_begin + rows_per_split < T ? t_begin + rows_per_split : T;
    if (t_begin >= t_end) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wi = wave >> 1, wj = wave & 1;
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
    RawTile la, lb;
    la.load(x, t_begin, t_end, i0, K);
    if (!diag) lb.load(x, t_begin, t_end, j0, K);
    la.store(sa[0]);
    if (!diag) lb.store(sb[0]);
    __syncthreads();
    int buf = 0;
    // transposing read: this lane's address inside a k-step tile (see the header comment)
    const int kh = lane >> 5, grp = (lane >> 4) & 1, i16 = lane & 15;
    const int lane_off = (8 * kh + (i16 >> 2)) * HP16 + grp * 16 + 4 * (i16 & 3);
    for (int64_t t0 = t_begin; t0 < t_end; t0 += HK16) {
        const bool more = t0 + HK16 < t_end;
        if (more) {
            la.load(x, t0 + HK16, t_end, i0, K);
            if (!diag) lb.load(x, t0 + HK16, t_end, j0, K);
        }
        const uint16_t* pa = sa[buf] + lane_off;
        const uint16_t* pb = (diag ? sa[buf] : sb[buf]) + lane_off;
#pragma unroll
        for (int ks = 0; ks < HK16 / 16; ++ks) {
            s16x4 a[2][2], b[2][2];
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    a[m][half] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4*)(pa + (16 * ks + 4 * half) * HP16 + wi * 64 + m * 32));
                    b[m][half] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4*)(pb + (16 * ks + 4 * half) * HP16 + wj * 64 + m * 32));
                }
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) acc[m][n] = mfma16<DT>(a[m][0], a[m][1], b[n][0], b[n][1], acc[m][n]);
        }
        if (more) {
            la.store(sa[buf ^ 1]);
            if (!diag) lb.store(sb[buf ^ 1]);
        }
        __syncthreads();
        buf ^= 1;
    }
    const int kl = lane >> 5, cl = lane & 31;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t gi = i0 + wi * 64 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * kl;
                const int64_t gj = j0 + wj * 64 + n * 32 + cl;
                if (gi <= gj && gj < K) {  // upper triangle only: k_mirror_upper fills in the rest
                    const float val = alpha * acc[m][n][r];
                    atomicAdd(&h[gi * K + gj], val);
                }
            }
}

// Large-K version: block = 256x256 of H, 4 waves in 2x2, each wave 128x128 = 4x4 MFMA tiles (256 accumulator
// registers, one wave per SIMD).  Per 16-token k-step a wave issues 16 transposing LDS reads for 16 MFMAs, so the
// LDS port is busy half of the MFMA time (the 64x64-per-wave kernel above needs it all the time), and a block
// needs 1 KiB of X per 131 kflop (128 flop/B: what the L2 / Infinity Cache can feed at ~1 PFLOP/s).
// X goes global -> LDS by buffer DMA (buffer_load_dwordx4 ... lds: no staging registers, no ds_write; rows past
// the end of X read as zero through the buffer bounds check; columns past K only feed H entries that are never
// written), 4 stages of 32 tokens in flight/in use (4 x 32 KiB of LDS, three stages of lookahead).
// LDS image of a stage and operand: [32 tokens][512 B]; the 16-byte chunk c of token row t sits at chunk position
// c ^ ((t & 3) << 2), so that the four token rows one transposing read touches fall on disjoint banks; the DMA
// writes LDS linearly, hence the permutation is applied to its global source addresses.
constexpr int GB = 256;     // tile edge
constexpr int GK = 32;      // tokens per stage
constexpr int GST = 4;      // stages (LDS buffers); 5 (all of the LDS, one more stage of lookahead) measured no faster
constexpr int GROW = 512;   // bytes per token row in LDS

// buffer -> LDS DMA of 16 bytes per lane, written out: through the builtin the compiler sees an LDS store of
// unknown address and puts s_waitcnt vmcnt(0) in front of every later LDS read, which would serialise the pipeline
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void buffer_dma16(const i32x4& rsrc, uint32_t voff, uint32_t soff, uint32_t lds_addr) {
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds"
                 :
                 : "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_addr)
                 : "memory", "m0");
}

// A calibration set arrives as many separately allocated items [tokens_i, K]: the kernel walks a LIST of them (no
// concatenation copy).  Stages never straddle items: item i owns ceil(tokens_i / GK) stages of the global stage
// numbering and its rows past the end read as zero through the bounds check of its own buffer descriptor.
constexpr int kSyrkItems = 128;
struct SyrkItems {
    int n;
    uint32_t stage_start[kSyrkItems + 1];  // prefix sums of the items' stage counts
    uint32_t bytes[kSyrkItems];            // tokens_i * K * 2
    const void* x[kSyrkItems];
};

template <int DT>
__global__ __launch_bounds__(256) void k_syrk16b(float* __restrict__ h, const SyrkItems items, int64_t K, float alpha,
                                                 int ntile, int stages_per_split) {
    extern __shared__ __attribute__((aligned(1024))) char smem16[];  // [GST][A, B][GK * GROW]
    const PairId id = decode_block(ntile);
    const int ti = id.ti, tj = id.tj;
    const bool diag = ti == tj;
    const int64_t i0 = (int64_t)ti * GB, j0 = (int64_t)tj * GB;
    const int s_all = (int)items.stage_start[items.n];
    const int s_begin = id.split * stages_per_split;
    const int s_end = s_begin + stages_per_split < s_all ? s_begin + stages_per_split : s_all;
    if (s_begin >= s_end) return;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wi = wave >> 1, wj = wave & 1;
    f32x16 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
    // the item of the stage being requested, as a buffer: offsets >= num_records (tokens past its end) read as zero
    int cur_item = 0;
    i32x4 rsrc;
    auto set_item = [&](int it) {
        const uintptr_t base = (uintptr_t)items.x[it];
        rsrc[0] = __builtin_amdgcn_readfirstlane((int)(uint32_t)base);
        rsrc[1] = __builtin_amdgcn_readfirstlane((int)(uint32_t)((base >> 32) & 0xffffu));  // stride 0
        rsrc[2] = __builtin_amdgcn_readfirstlane((int)items.bytes[it]);
        rsrc[3] = 0x00020000;
    };
    while ((uint32_t)s_begin >= items.stage_start[cur_item + 1]) ++cur_item;
    set_item(cur_item);
    // DMA: wave w brings token rows 8w .. 8w+7 of a stage, two rows per instruction
    uint32_t voff[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const uint32_t row = 8 * wave + 2 * u + (lane >> 5);
        voff[u] = row * (uint32_t)K * 2u + (uint32_t)((lane & 31) ^ ((row & 3) << 2)) * 16u;
    }
    const int nst = s_end - s_begin;
    const uint32_t lds_base = (uint32_t)(uintptr_t)smem16;
    // DMA of stage st (called with st = 0, 1, 2, ... in order): issue_prepare, then issue_one(0 .. n_dma - 1), which
    // the main loop spreads between its MFMAs
    const int n_dma = diag ? 4 : 8;
    uint32_t cur_start = items.stage_start[cur_item], next_start = items.stage_start[cur_item + 1];  // SGPRs: no loads per stage
    uint32_t dma_dst = 0, dma_so = 0;
    auto issue_prepare = [&](int st) {
        dma_dst = lds_base + (uint32_t)((st % GST) * (2 * GK * GROW) + 8 * wave * GROW);
        const uint32_t gst = (uint32_t)(s_begin + st);
        if (gst >= next_start) {  // wave-uniform
            do ++cur_item; while (gst >= items.stage_start[cur_item + 1]);
            cur_start = items.stage_start[cur_item];
            next_start = items.stage_start[cur_item + 1];
            set_item(cur_item);
        }
        dma_so = (gst - cur_start) * GK * (uint32_t)K * 2u;
    };
    auto issue_one = [&](int u) {
        if (u < 4) buffer_dma16(rsrc, voff[u], dma_so + (uint32_t)i0 * 2u, dma_dst + u * 2 * GROW);
        else buffer_dma16(rsrc, voff[u - 4], dma_so + (uint32_t)j0 * 2u, dma_dst + GK * GROW + (u - 4) * 2 * GROW);
    };
    auto issue = [&](int st) {
        issue_prepare(st);
        for (int u = 0; u < n_dma; ++u) issue_one(u);
    };
    // transposing reads: per-lane byte offsets inside a stage image for the 4 m-tiles of this wave's A and B panels
    const int kh = lane >> 5, grp = (lane >> 4) & 1, i16 = lane & 15, r3 = i16 >> 2;
    const uint32_t lane_const = (uint32_t)(8 * kh + r3) * GROW + (uint32_t)(grp * 2 + ((i16 & 3) >> 1)) * 16u + (uint32_t)(i16 & 1) * 8u;
    uint32_t offa[4], offb[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        offa[m] = lane_const + (uint32_t)(wi * 16 + ((m ^ r3) << 2)) * 16u;
        offb[m] = lane_const + (uint32_t)(wj * 16 + ((m ^ r3) << 2)) * 16u + (diag ? 0u : (uint32_t)(GK * GROW));
    }
    // One wave per SIMD: whenever the wave waits, its matrix pipe idles, so the order of issue is written out.  The
    // MFMAs are volatile asm (the compiler keeps volatile asm and the LDS reads in source order), and between them go,
    // one or two at a time, the transposing LDS reads of the NEXT k-step's operands (register double buffer: by the
    // time the next k-step starts they have been on their way for >= 8 MFMAs) and the DMA instructions of the stage
    // GST - 1 ahead.  Per stage: k-step 0 [MFMAs + reads of k-step 1 | wait for the next stage's DMAs + barrier |
    // MFMAs + DMA issue], k-step 1 [MFMAs + reads of the next stage's k-step 0].
    typedef int op_t __attribute__((ext_vector_type(4)));  // one MFMA operand: 8 16-bit values of one row / column
    auto read_op = [&](const char* img, uint32_t off, int ks) {
        struct Pair {
            s16x4 lo, hi;
        } p;
        p.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(img + off + (16 * ks) * GROW));
        p.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(img + off + (16 * ks + 4) * GROW));
        return __builtin_bit_cast(op_t, p);
    };
    auto mfma = [&](f32x16& c, const op_t& a, const op_t& b) {
        if constexpr (DT == NNCF_AMD_BF16)
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b) : "memory");
        else
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b) : "memory");
    };
    auto wait_dma = [&](int outstanding) {  // all but the `outstanding` most recent DMAs have landed (immediates)
        switch (outstanding) {
            case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
            case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
            case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
            case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
            case 24: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        }
    };
    static_assert(GK == 32, "two k-steps per stage");
    static_assert(GST >= 3, "the steady state keeps GST - 3 stages in flight behind the one it waits for");
    constexpr int kBarrierAfter = 7;  // k-step 0: wait + barrier after this MFMA
    // the DMA instructions of a stage are spread evenly over the gaps behind the barrier (the rest of k-step 0 and
    // all of k-step 1): bunched together, the four waves' requests queue up in front of the texture addresser and
    // every queued instruction holds its wave's issue
    constexpr int kDmaSlots = 16 - (kBarrierAfter + 1) + 16;
    op_t a0[4], b0[4], a1[4], b1[4];
    // 16 MFMAs on (ca, cb).  READS: operand j (0..3: A rows, 4..7: B columns) of the next k-step is read after MFMA j.
    // SYNC (k-step 0 of a stage that has a successor): after MFMA kBarrierAfter wait until all but `outstanding` of
    // this wave's DMAs have landed, then the barrier (the next stage is in LDS for everybody, the previous stage's
    // buffer is free).  DMA: NDMA DMA instructions of the stage GST - 1 ahead, one after each of the following MFMAs.
    auto kstep = [&](const op_t* ca, const op_t* cb, op_t* na, op_t* nb, const char* img, int ks, auto reads, auto sync,
                     auto ndma, auto slot0, auto steady, int outstanding) {
        constexpr bool READS = decltype(reads)::value, SYNC = decltype(sync)::value, STEADY = decltype(steady)::value;
        constexpr int NDMA = decltype(ndma)::value;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            mfma(acc[i >> 2][i & 3], ca[i >> 2], cb[i & 3]);
            if constexpr (READS) {
                if (i < 4) na[i] = read_op(img, offa[i], ks);
                else if (i < 8) nb[i - 4] = read_op(img, offb[i - 4], ks);
            }
            if constexpr (SYNC) {
                if (i == kBarrierAfter) {
                    if constexpr (STEADY) {
                        if constexpr (NDMA * (GST - 3) == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                        else if constexpr (NDMA * (GST - 3) == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                        else if constexpr (NDMA * (GST - 3) == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    } else {
                        wait_dma(outstanding);
                    }
                    __builtin_amdgcn_s_barrier();
                }
            }
            if constexpr (NDMA > 0) {  // DMA slot of this gap; piece p goes into slot p * kDmaSlots / NDMA
                constexpr int SLOT0 = decltype(slot0)::value;
                const int slot = SLOT0 + i;
                if (slot >= 0 && slot < kDmaSlots && slot % (kDmaSlots / NDMA) == 0) issue_one(slot / (kDmaSlots / NDMA));
            }
        }
    };
    using T = std::true_type;
    using F = std::false_type;
    for (int st = 0; st < GST - 1 && st < nst; ++st) issue(st);
    wait_dma((nst - 1 < GST - 2 ? nst - 1 : GST - 2) * n_dma);
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        a0[m] = read_op(smem16, offa[m], 0);
        b0[m] = read_op(smem16, offb[m], 0);
    }
    auto run = [&](auto ndma) {  // ndma: DMA instructions per stage and wave (4 on a diagonal block, else 8)
        int st = 0;
        for (; st + GST - 1 < nst; ++st) {  // steady state: stage st + GST - 1 is requested while stage st is multiplied
            const char* img = smem16 + (st % GST) * (2 * GK * GROW);
            issue_prepare(st + GST - 1);
            kstep(a0, b0, a1, b1, img, 1, T{}, T{}, ndma, std::integral_constant<int, -(kBarrierAfter + 1)>{}, T{}, 0);
            kstep(a1, b1, a0, b0, smem16 + ((st + 1) % GST) * (2 * GK * GROW), 0, T{}, F{}, ndma, std::integral_constant<int, 16 - (kBarrierAfter + 1)>{}, T{}, 0);
        }
        for (; st + 1 < nst; ++st) {  // the last GST - 2 stages with a successor: nothing left to request
            const char* img = smem16 + (st % GST) * (2 * GK * GROW);
            kstep(a0, b0, a1, b1, img, 1, T{}, T{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, F{}, (nst - 2 - st) * decltype(ndma)::value);
            kstep(a1, b1, a0, b0, smem16 + ((st + 1) % GST) * (2 * GK * GROW), 0, T{}, F{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, F{}, 0);
        }
        const char* img = smem16 + (st % GST) * (2 * GK * GROW);  // the last stage
        kstep(a0, b0, a1, b1, img, 1, T{}, F{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, F{}, 0);
        kstep(a1, b1, a0, b0, img, 0, F{}, F{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, F{}, 0);
    };
    if (diag) run(std::integral_constant<int, 4>{});
    else run(std::integral_constant<int, 8>{});
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");  // the last MFMAs leave the pipe before the read-out
    // epilogue with 32-bit offsets (K < 46341 on this path: K*K fits; checked by the host)
    const uint32_t kl = lane >> 5, cl = lane & 31, Ku = (uint32_t)K;
    const uint32_t bi = (uint32_t)i0 + wi * 128 + 4 * kl, bj = (uint32_t)j0 + wj * 128 + cl;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            __builtin_amdgcn_sched_barrier(0);  // one accumulator tile at a time: no 256-register read-out
            const uint32_t gj = bj + n * 32;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const uint32_t gi = bi + m * 32 + (r & 3) + 8 * (r >> 2);
                if (gi <= gj && gj < Ku) {
                    const float val = alpha * acc[m][n][r];
                    atomicAdd(h + (gi * Ku + gj), val);  // upper triangle only: k_mirror_upper fills in the rest
                }
            }
        }
}

// ---------------------------------------------------------------------------------------------------------------------
// C += P^T Q on the same tile engine (256x256 tiles, 4 waves of 128x128, 32-row stages by buffer DMA, transposing LDS
// reads, written-out issue order): the dense products of GPTQ's algebra -- the far trailing update of gptq.py:363 and
// the large products of the Hinv recursion behind :283-285, fp32 GEMMs in the reference -- with both operands given as
// bf16 matrices whose ROWS are the contraction index: P [T, M] (row pitch ldp), Q [T, N] (row pitch ldq).  The caller
// passes the six-fold three-way bf16 splits of the fp32 operands (k_split3_rows / k_split3_cols below; T = 6k), so the
// sum over T is the fp32 product to fp32-GEMM accuracy.  One fp32 atomic add per element of C and token split (with
// one split the result does not depend on any order).  Rows of P / Q past T read as zero through the buffer bounds
// check; columns past M / N feed entries that are never written.
struct Gemm16Args {
    float* c;
    const void* p;
    const void* q;
    uint32_t p_bytes, q_bytes;  // T * ldp * 2, T * ldq * 2  (< 2^32: checked by the host)
    uint32_t ldp, ldq, ldc;     // elements
    uint32_t M, N;
    int tm, tn, stages, splits;
    int tri;  // which stages / tiles can contribute (a triangular operand or result), see nncf_amd_gemm_bf16_tn_acc
    int xcd_lists;  // tile order of a triangular product: per-XCD lists (large tile counts) or alternating bands
    uint32_t wanted;  // tri 4 with lists: number of tiles tj <= ti
    float alpha;      // list form (Hessian): factor of the accumulated product
};

// LIST: P = Q = a LIST of separately allocated 16-bit items [tokens_i, K] (the calibration activations: H += alpha X^T X
// without a concatenation copy), tri = 4 with the tile transposed (the UPPER triangle is accumulated, k_mirror_upper
// fills in the rest); stages never straddle items, an item's rows past its end read as zero through its own descriptor.
template <int DT, bool LIST>
__device__ __forceinline__ void gemm16t_body(const Gemm16Args& g, const SyrkItems* items) {
    extern __shared__ __attribute__((aligned(1024))) char smem16[];  // [GST][A, B][GK * GROW]
    // block -> (tile, split).  Blocks go round-robin over the 8 XCDs (block b runs on XCD b % 8), each with an L2 of its
    // own, so the blocks of one XCD should share panels and every XCD should get the same work.
    //   dense (tri 0): XCD x takes the x-th contiguous eighth of the tile list, tiles band by band (4 tile rows),
    //     column by column inside a band;
    //   triangular operand (tri 1..3): the work of a tile changes along ONE axis by a factor of up to the tile count
    //     (48 .. 48 * tn stages).  XCD x owns the lines across that axis with index = x (mod 8) -- every XCD the same
    //     work -- and walks its tiles heaviest first (longest-processing-time order: the tail of the launch is made of
    //     the lightest tiles); ~7 lines x ~5 positions are in flight per XCD: 12 panels;
    //   lower triangle wanted (tri 4): tiles of equal work, tile rows dealt to the XCDs back and forth (x, 15 - x,
    //     16 + x, 31 - x, ...) so that the row lengths even out.
    // The `splits` parts of a tile are neighbours in an XCD's list.
    //   Few tiles (an XCD would own 1-3 lines: uneven shares): the contiguous-eighth list of the dense case with the
    //     bands ACROSS the axis along which the work changes (every band the same work), bands and positions inside a
    //     band taken alternately from both ends (0, last, 1, last - 1, ...): every run of the list mixes light and heavy.
    const uint32_t lin = blockIdx.x, xcd = lin & 7, nth = lin >> 3;
    int ti, tj, split;
    if (g.tri == 0 || !g.xcd_lists) {
        const uint32_t tiles = (uint32_t)(g.tm * g.tn), total = gridDim.x;
        const uint32_t logical = xcd * (total >> 3) + (xcd < (total & 7) ? xcd : (total & 7)) + nth;
        split = (int)(logical / tiles);
        const uint32_t qid = logical % tiles;
        const bool across_cols = g.tri == 3 || g.tri == 4;  // work changes with the row of C: bands of tile columns
        const uint32_t outer_n = (uint32_t)(across_cols ? g.tn : g.tm), inner_n = (uint32_t)(across_cols ? g.tm : g.tn);
        const uint32_t nb = (outer_n + 3) / 4, rs = outer_n - 4 * (nb - 1), full = 4 * inner_n, shrt = rs * inner_n;
        const bool alt = g.tri != 0;
        uint32_t seq, in_band;  // (alternating order: the short last band sits at position 1)
        if (!alt) seq = qid / full, in_band = qid % full;
        else if (nb == 1 || qid < full) seq = 0, in_band = qid;
        else if (qid < full + shrt) seq = 1, in_band = qid - full;
        else seq = 2 + (qid - full - shrt) / full, in_band = (qid - full - shrt) % full;
        const uint32_t band = !alt ? seq : ((seq & 1) ? nb - 1 - (seq >> 1) : (seq >> 1));
        const uint32_t lines = band == nb - 1 ? rs : 4;
        const uint32_t line = band * 4 + in_band % lines, qc = in_band / lines;
        const uint32_t inner = !alt ? qc : ((qc & 1) ? inner_n - 1 - (qc >> 1) : (qc >> 1));
        ti = (int)(across_cols ? inner : line), tj = (int)(across_cols ? line : inner);
    } else {
        split = (int)(nth % (uint32_t)g.splits);
        const uint32_t idx = nth / (uint32_t)g.splits;
        if (g.tri == 4) {
            // tiles tj <= ti only, all of equal work: XCD x takes the x-th contiguous eighth of the list, the list goes
            // band by band (4 tile rows), column by column inside a band: 4 x 8 tiles in flight per XCD = 12 panels
            const uint32_t total = g.wanted;
            const uint32_t logical = xcd * (total >> 3) + (xcd < (total & 7) ? xcd : (total & 7)) + idx;
            if (idx >= (total >> 3) + (xcd < (total & 7) ? 1u : 0u)) return;
            uint32_t q = logical;
            int r0 = 0, rows = 0;
            for (;; r0 += 4) {
                rows = g.tm - r0 < 4 ? g.tm - r0 : 4;
                uint32_t in_band = 0;
                for (int r = r0; r < r0 + rows; ++r) in_band += (uint32_t)((r < g.tn ? r : g.tn - 1) + 1);
                if (q < in_band) break;
                q -= in_band;
            }
            const uint32_t full_cols = (uint32_t)(r0 < g.tn ? r0 : g.tn);  // columns every row of the band has
            if (q < full_cols * (uint32_t)rows) {
                tj = (int)(q / (uint32_t)rows);
                ti = r0 + (int)(q % (uint32_t)rows);
            } else {
                q -= full_cols * (uint32_t)rows;
                int c = r0;
                for (;; ++c) {  // the band's own diagonal block: column c holds rows max(c, r0) .. r0 + rows - 1
                    const uint32_t cnt = (uint32_t)(r0 + rows - c);
                    if (q < cnt) break;
                    q -= cnt;
                }
                tj = c;
                ti = c + (int)q;
            }
        } else {
            const int span = g.tri == 3 ? g.tn : g.tm;      // the lines an XCD owns run along this many tiles ...
            const int along = g.tri == 3 ? g.tm : g.tn;     // ... and each is this long
            const int nlines = (int)xcd < span ? (span - (int)xcd + 7) / 8 : 0;
            if (idx >= (uint32_t)(nlines * along)) return;
            const int line = (int)xcd + 8 * (int)(idx % (uint32_t)nlines), pos = (int)(idx / (uint32_t)nlines);
            if (g.tri == 1) ti = line, tj = along - 1 - pos;       // heaviest: the last column
            else if (g.tri == 2) ti = line, tj = pos;              // heaviest: the first column
            else ti = along - 1 - pos, tj = line;                  // heaviest: the last row
        }
    }
    if constexpr (LIST) {  // (tj <= ti from the lower-triangle list) -> the upper tile
        const int sw = ti;
        ti = tj;
        tj = sw;
    }
    const uint32_t i0 = (uint32_t)ti * GB, j0 = (uint32_t)tj * GB;
    // the stages of this tile that can contribute: contraction index t = row / 6 of P and Q (slot-minor rows), so a
    // 256-wide tile of a triangular operand bounds the rows at a multiple of 6 * 256 = 48 stages
    constexpr int kStagesPerTile = 6 * GB / GK;
    int lo = 0, hi = g.stages;
    if (g.tri == 1) hi = (tj + 1) * kStagesPerTile;        // Q^T lower triangular: t <= column of C
    else if (g.tri == 2) lo = tj * kStagesPerTile;         // Q lower triangular:   t >= column of C
    else if (g.tri == 3) hi = (ti + 1) * kStagesPerTile;   // P^T lower triangular: t <= row of C
    else if (g.tri == 4 && !LIST && tj > ti) return;       // only the lower triangle of C is wanted
    if (hi > g.stages) hi = g.stages;
    if (lo >= hi) return;
    // every tile's own range in `splits` equal parts
    const int s_begin = lo + (int)((int64_t)(hi - lo) * split / g.splits);
    const int s_end = lo + (int)((int64_t)(hi - lo) * (split + 1) / g.splits);
    if (s_begin >= s_end) return;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wi = wave >> 1, wj = wave & 1;
    f32x16 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
    i32x4 rsrc_p, rsrc_q;
    int cur_item = 0;
    auto set_item = [&](int it) {  // LIST: both operands read the item of the stage being requested
        const uintptr_t base = (uintptr_t)items->x[it];
        rsrc_p[0] = __builtin_amdgcn_readfirstlane((int)(uint32_t)base);
        rsrc_p[1] = __builtin_amdgcn_readfirstlane((int)(uint32_t)((base >> 32) & 0xffffu));
        rsrc_p[2] = __builtin_amdgcn_readfirstlane((int)items->bytes[it]);
        rsrc_p[3] = 0x00020000;
        rsrc_q = rsrc_p;
    };
    if constexpr (LIST) {
        while ((uint32_t)s_begin >= items->stage_start[cur_item + 1]) ++cur_item;
        set_item(cur_item);
    } else {
        const uintptr_t bp = (uintptr_t)g.p, bq = (uintptr_t)g.q;
        rsrc_p[0] = __builtin_amdgcn_readfirstlane((int)(uint32_t)bp);
        rsrc_p[1] = __builtin_amdgcn_readfirstlane((int)(uint32_t)((bp >> 32) & 0xffffu));
        rsrc_p[2] = __builtin_amdgcn_readfirstlane((int)g.p_bytes);
        rsrc_p[3] = 0x00020000;
        rsrc_q[0] = __builtin_amdgcn_readfirstlane((int)(uint32_t)bq);
        rsrc_q[1] = __builtin_amdgcn_readfirstlane((int)(uint32_t)((bq >> 32) & 0xffffu));
        rsrc_q[2] = __builtin_amdgcn_readfirstlane((int)g.q_bytes);
        rsrc_q[3] = 0x00020000;
    }
    // DMA: wave w brings rows 8w .. 8w+7 of a stage, two rows per instruction (chunk permutation as in k_syrk16b)
    uint32_t voff_p[4], voff_q[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const uint32_t row = 8 * wave + 2 * u + (lane >> 5);
        const uint32_t chunk = (uint32_t)((lane & 31) ^ ((row & 3) << 2)) * 16u;
        voff_p[u] = row * g.ldp * 2u + chunk;
        voff_q[u] = row * g.ldq * 2u + chunk;
    }
    const int nst = s_end - s_begin;
    const uint32_t lds_base = (uint32_t)(uintptr_t)smem16;
    uint32_t dma_dst = 0, so_p = 0, so_q = 0;
    uint32_t cur_start = 0, next_start = 0xffffffffu;  // SGPRs: no loads per stage
    if constexpr (LIST) cur_start = items->stage_start[cur_item], next_start = items->stage_start[cur_item + 1];
    auto issue_prepare = [&](int st) {
        dma_dst = lds_base + (uint32_t)((st % GST) * (2 * GK * GROW) + 8 * wave * GROW);
        uint32_t gst = (uint32_t)(s_begin + st);
        if constexpr (LIST) {
            if (gst >= next_start) {  // wave-uniform
                do ++cur_item; while (gst >= items->stage_start[cur_item + 1]);
                cur_start = items->stage_start[cur_item];
                next_start = items->stage_start[cur_item + 1];
                set_item(cur_item);
            }
            gst -= cur_start;
        }
        so_p = gst * GK * g.ldp * 2u + i0 * 2u;
        so_q = gst * GK * g.ldq * 2u + j0 * 2u;
    };
    auto issue_one = [&](int u) {
        if (u < 4) buffer_dma16(rsrc_p, voff_p[u], so_p, dma_dst + u * 2 * GROW);
        else buffer_dma16(rsrc_q, voff_q[u - 4], so_q, dma_dst + GK * GROW + (u - 4) * 2 * GROW);
    };
    auto issue = [&](int st) {
        issue_prepare(st);
        for (int u = 0; u < 8; ++u) issue_one(u);
    };
    const int kh = lane >> 5, grp = (lane >> 4) & 1, i16 = lane & 15, r3 = i16 >> 2;
    const uint32_t lane_const = (uint32_t)(8 * kh + r3) * GROW + (uint32_t)(grp * 2 + ((i16 & 3) >> 1)) * 16u + (uint32_t)(i16 & 1) * 8u;
    uint32_t offa[4], offb[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        offa[m] = lane_const + (uint32_t)(wi * 16 + ((m ^ r3) << 2)) * 16u;
        offb[m] = lane_const + (uint32_t)(wj * 16 + ((m ^ r3) << 2)) * 16u + (uint32_t)(GK * GROW);
    }
    typedef int op_t __attribute__((ext_vector_type(4)));
    auto read_op = [&](const char* img, uint32_t off, int ks) {
        struct Pair {
            s16x4 lo, hi;
        } pr;
        pr.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(img + off + (16 * ks) * GROW));
        pr.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(img + off + (16 * ks + 4) * GROW));
        return __builtin_bit_cast(op_t, pr);
    };
    auto mfma = [&](f32x16& c, const op_t& a, const op_t& b) {
        if constexpr (DT == NNCF_AMD_BF16)
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b) : "memory");
        else
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b) : "memory");
    };
    auto wait_dma = [&](int outstanding) {
        switch (outstanding) {
            case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
            case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
            case 24: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        }
    };
    constexpr int kBarrierAfter = 7;
    constexpr int kDmaSlots = 16 - (kBarrierAfter + 1) + 16;
    constexpr int NDMA = 8;
    op_t a0[4], b0[4], a1[4], b1[4];
    // one k-step (16 rows) of 16 MFMAs; see k_syrk16b for the interleaving of reads, the barrier and the DMA issue
    auto kstep = [&](const op_t* ca, const op_t* cb, op_t* na, op_t* nb, const char* img, int ks, auto reads, auto sync,
                     auto dma, auto slot0, auto steady, int outstanding) {
        constexpr bool READS = decltype(reads)::value, SYNC = decltype(sync)::value, STEADY = decltype(steady)::value;
        constexpr bool DMA = decltype(dma)::value;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            mfma(acc[i >> 2][i & 3], ca[i >> 2], cb[i & 3]);
            if constexpr (READS) {
                if (i < 4) na[i] = read_op(img, offa[i], ks);
                else if (i < 8) nb[i - 4] = read_op(img, offb[i - 4], ks);
            }
            if constexpr (SYNC) {
                if (i == kBarrierAfter) {
                    if constexpr (STEADY) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // NDMA * (GST - 3)
                    else wait_dma(outstanding);
                    __builtin_amdgcn_s_barrier();
                }
            }
            if constexpr (DMA) {
                constexpr int SLOT0 = decltype(slot0)::value;
                const int slot = SLOT0 + i;
                if (slot >= 0 && slot < kDmaSlots && slot % (kDmaSlots / NDMA) == 0) issue_one(slot / (kDmaSlots / NDMA));
            }
        }
    };
    static_assert(GST == 4, "vmcnt(8) above = one stage of lookahead behind the one waited for");
    using T = std::true_type;
    using F = std::false_type;
    for (int st = 0; st < GST - 1 && st < nst; ++st) issue(st);
    wait_dma((nst - 1 < GST - 2 ? nst - 1 : GST - 2) * NDMA);
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        a0[m] = read_op(smem16, offa[m], 0);
        b0[m] = read_op(smem16, offb[m], 0);
    }
    {
        int st = 0;
        for (; st + GST - 1 < nst; ++st) {
            const char* img = smem16 + (st % GST) * (2 * GK * GROW);
            issue_prepare(st + GST - 1);
            kstep(a0, b0, a1, b1, img, 1, T{}, T{}, T{}, std::integral_constant<int, -(kBarrierAfter + 1)>{}, T{}, 0);
            kstep(a1, b1, a0, b0, smem16 + ((st + 1) % GST) * (2 * GK * GROW), 0, T{}, F{}, T{}, std::integral_constant<int, 16 - (kBarrierAfter + 1)>{}, T{}, 0);
        }
        for (; st + 1 < nst; ++st) {
            const char* img = smem16 + (st % GST) * (2 * GK * GROW);
            kstep(a0, b0, a1, b1, img, 1, T{}, T{}, F{}, std::integral_constant<int, 0>{}, F{}, (nst - 2 - st) * NDMA);
            kstep(a1, b1, a0, b0, smem16 + ((st + 1) % GST) * (2 * GK * GROW), 0, T{}, F{}, F{}, std::integral_constant<int, 0>{}, F{}, 0);
        }
        const char* img = smem16 + (st % GST) * (2 * GK * GROW);
        kstep(a0, b0, a1, b1, img, 1, T{}, F{}, F{}, std::integral_constant<int, 0>{}, F{}, 0);
        kstep(a1, b1, a0, b0, img, 0, F{}, F{}, F{}, std::integral_constant<int, 0>{}, F{}, 0);
    }
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    const uint32_t kl = lane >> 5, cl = lane & 31;
    const uint32_t bi = i0 + wi * 128 + 4 * kl, bj = j0 + wj * 128 + cl;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            __builtin_amdgcn_sched_barrier(0);
            const uint32_t gj = bj + n * 32;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const uint32_t gi = bi + m * 32 + (r & 3) + 8 * (r >> 2);
                if constexpr (LIST) {
                    if (gi <= gj && gj < g.N) atomicAdd(g.c + ((uint64_t)gi * g.ldc + gj), g.alpha * acc[m][n][r]);
                } else {
                    if (gi < g.M && gj < g.N) atomicAdd(g.c + ((uint64_t)gi * g.ldc + gj), acc[m][n][r]);
                }
            }
        }
}

__global__ __launch_bounds__(256) void k_gemm16t(const Gemm16Args g) { gemm16t_body<NNCF_AMD_BF16, false>(g, nullptr); }

template <int DT>
__global__ __launch_bounds__(256) void k_gemm16t_list(const Gemm16Args g, const SyrkItems items) {
    gemm16t_body<DT, true>(g, &items);
}

// The six-fold three-way bf16 split of an fp32 matrix view, rows of the result = contraction index (what k_gemm16t
// reads).  x = h + m + l exactly (8 + 8 + 8 significand bits, truncating); the six slots hold, for a left operand
// (order 0) h h h m m l, for a right operand (order 1) h m l h m h, so that the sum over the slots is
// hh + hm + hl + mh + mm + lh: everything of the fp32 product above 2^-24 |a||b|.
//   k_split3_rows: src [k, n] (contraction = rows)    -> dst [6k, n], row 6t + s = slot s of src row t
//   k_split3_cols: src [m, k] (contraction = columns) -> dst [6k, m], row 6t + s = slot s of src column t (transposed
//                  through LDS: reads and writes are both row-contiguous)
// (slot-minor rows: a bound on the contraction index t is a bound on the rows, i.e. on the stages a tile walks)
__device__ __forceinline__ void split3_of(float x, uint32_t& h, uint32_t& m, uint32_t& l) {
    const uint32_t hb = __builtin_bit_cast(uint32_t, x) & 0xffff0000u;
    const float r1 = x - __builtin_bit_cast(float, hb);
    const uint32_t mb = __builtin_bit_cast(uint32_t, r1) & 0xffff0000u;
    const float r2 = r1 - __builtin_bit_cast(float, mb);
    h = hb >> 16, m = mb >> 16, l = __builtin_bit_cast(uint32_t, r2) >> 16;
}
__device__ __forceinline__ int split3_slot_piece(int order, int s) {
    return order == 0 ? (s < 3 ? 0 : (s < 5 ? 1 : 2)) : (s < 3 ? s : (s < 5 ? s - 3 : 0));
}

__global__ __launch_bounds__(256) void k_split3_rows(const float* __restrict__ src, int64_t rows, int64_t cols, int64_t src_pitch,
                                                     int order, float sign, uint16_t* __restrict__ dst, int64_t dst_pitch) {
    const int64_t vpr = (cols + 3) / 4;
    const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (v >= rows * vpr) return;
    const int64_t r = v / vpr, c = (v % vpr) * 4;
    const int nc = cols - c < 4 ? (int)(cols - c) : 4;
    uint32_t piece[3][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float x = i < nc ? src[r * src_pitch + c + i] * sign : 0.0f;
        split3_of(x, piece[0][i], piece[1][i], piece[2][i]);
    }
    const bool vec = nc == 4 && (dst_pitch & 3) == 0 && (((uintptr_t)dst) & 7u) == 0;  // (c is a multiple of 4)
#pragma unroll
    for (int s = 0; s < 6; ++s) {
        const int p = split3_slot_piece(order, s);
        uint16_t* out = dst + (r * 6 + s) * dst_pitch + c;
        uint32_t v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = p == 0 ? piece[0][i] : (p == 1 ? piece[1][i] : piece[2][i]);
        if (vec) {
            *(uint2*)out = make_uint2(v[0] | (v[1] << 16), v[2] | (v[3] << 16));
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (i < nc) out[i] = (uint16_t)v[i];
        }
    }
}

__global__ __launch_bounds__(256) void k_split3_cols(const float* __restrict__ src, int64_t rows, int64_t cols, int64_t src_pitch,
                                                     int order, float sign, uint16_t* __restrict__ dst, int64_t dst_pitch) {
    __shared__ float t[64][65];
    const int64_t tiles_c = (cols + 63) / 64;
    const int64_t r0 = (int64_t)(blockIdx.x / tiles_c) * 64, c0 = (int64_t)(blockIdx.x % tiles_c) * 64;
    const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
    for (int rr = ly; rr < 64; rr += 4) {
        const int64_t gr = r0 + rr, gc = c0 + lx;
        t[rr][lx] = (gr < rows && gc < cols) ? src[gr * src_pitch + gc] * sign : 0.0f;
    }
    __syncthreads();
    // thread (lx, ly): source column c0 + cc (a row of dst), source row r0 + lx (a column of dst)
    for (int cc = ly; cc < 64; cc += 4) {
        const int64_t gc = c0 + cc, gr = r0 + lx;
        if (gc >= cols || gr >= rows) continue;
        uint32_t pc[3];
        split3_of(t[lx][cc], pc[0], pc[1], pc[2]);
#pragma unroll
        for (int s = 0; s < 6; ++s) dst[(gc * 6 + s) * dst_pitch + gr] = (uint16_t)pc[split3_slot_piece(order, s)];
    }
}

// H[j][i] = H[i][j] for i < j: the SYRK kernels accumulate the upper triangle only, so H comes out exactly
// symmetric whatever the order of the atomic partial sums.  64x64 tiles transposed through LDS: reads and writes
// are row-contiguous.
// `scale` != 1 (finalize of a stream of raw sums): the upper triangle is multiplied in the same pass.
__global__ __launch_bounds__(256) void k_mirror_upper(float* __restrict__ h, uint32_t K, float scale) {
    __shared__ float t[64][65];
    const uint32_t nt = (K + 63) / 64;
    uint32_t p = blockIdx.x, bi = 0;  // (bi <= bj) from blockIdx.x
    while (p >= nt - bi) {
        p -= nt - bi;
        ++bi;
    }
    const uint32_t bj = bi + p;
    const uint32_t i0 = bi * 64, j0 = bj * 64;
    const uint32_t c = threadIdx.x & 63, r4 = threadIdx.x >> 6;
    for (uint32_t r = r4; r < 64; r += 4) {
        const uint32_t gi = i0 + r, gj = j0 + c;
        float v = 0.0f;
        if (gi < K && gj < K) {
            v = h[(size_t)gi * K + gj];
            if (scale != 1.0f && gi <= gj) {
                v *= scale;
                h[(size_t)gi * K + gj] = v;
            }
        }
        t[r][c] = v;
    }
    __syncthreads();
    for (uint32_t r = r4; r < 64; r += 4) {
        const uint32_t row = j0 + r, col = i0 + c;  // element (row, col) of the lower triangle <- (col, row)
        if (row < K && col < row) h[(size_t)row * K + col] = t[c][r];
    }
}

static int mirror(float* h, int64_t k, hipStream_t s, float scale = 1.0f) {
    const uint32_t mt = (uint32_t)((k + 63) / 64);
    hipLaunchKernelGGL(k_mirror_upper, dim3(mt * (mt + 1) / 2), dim3(256), 0, s, h, (uint32_t)k, scale);
    return launch_status("hessian_mirror");
}

}  // namespace wc

static int hessian_accumulate_impl(float* h, const void* x, int x_dtype, int64_t tokens, int64_t k, float alpha,
                                   float beta, void* stream, bool do_mirror);

// upper-triangle accumulation of a list of 16-bit items through the 256x256-tile kernel (2048 <= k < 46341, k % 8 == 0)
static int launch_syrk16b_list(float* h, const wc::SyrkItems& items, int x_dtype, int64_t k, float alpha, hipStream_t s) {
    using namespace wc;
    const int nt = (int)((k + GB - 1) / GB);
    const int64_t prs = (int64_t)nt * (nt + 1) / 2;
    const int64_t stages = items.stage_start[items.n];
    if (stages == 0) return NNCF_AMD_OK;
    // one block per CU at a time -> >= 8 rounds of blocks keep the tail short; >= 32 stages (1024 tokens) per block
    int64_t sp = prs >= 2048 ? 1 : (2048 + prs - 1) / prs;
    const int64_t maxsp = (stages + 31) / 32;
    if (sp > maxsp) sp = maxsp;
    if (sp > 65535) sp = 65535;
    const int64_t sps = (stages + sp - 1) / sp;
    sp = (stages + sps - 1) / sps;
    constexpr size_t lds_bytes = (size_t)GST * 2 * GK * GROW;  // 131072
    // Many tile pairs: the two-operand kernel in its list form (k_gemm16t_list: no diagonal special case, 340 registers
    // against 512 + spills) holds 1.30 PFLOP/s on a 73728 x 28672 bf16 chunk (46.7 ms; the whole accumulate call with
    // its mirror pass 52.4 against 53.2 ms through k_syrk16b, a Llama-3-70B GPTQ layer 455 -> 452 ms); at K = 8192 (528
    // pairs: rounds of blocks that do not fill) k_syrk16b stays ahead (14.1 against 15.0 ms).
    static const int64_t list_min_tiles = [] { const char* e = getenv("NNCF_AMD_SYRK_LIST_TILES"); return e ? atoll(e) : 48ll; }();  // developer knob
    if (nt >= list_min_tiles) {
        static OncePerDevice attr_list;
        if (attr_list.first_on_device() != 0) {
            if (hipFuncSetAttribute((const void*)k_gemm16t_list<NNCF_AMD_BF16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess ||
                hipFuncSetAttribute((const void*)k_gemm16t_list<NNCF_AMD_F16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess) {
                attr_list.forget_current();
                return launch_status("hessian gemm16t_list attribute");
            }
        }
        Gemm16Args g = {h, nullptr, nullptr, 0u, 0u, (uint32_t)k, (uint32_t)k, (uint32_t)k, (uint32_t)k, (uint32_t)k, nt, nt,
                        (int)stages, (int)sp, 4, 1, (uint32_t)prs, alpha};
        const int64_t blocks = ((prs + 7) / 8) * sp * 8;
        if (x_dtype == NNCF_AMD_BF16)
            hipLaunchKernelGGL((k_gemm16t_list<NNCF_AMD_BF16>), dim3((unsigned)blocks), dim3(256), lds_bytes, s, g, items);
        else
            hipLaunchKernelGGL((k_gemm16t_list<NNCF_AMD_F16>), dim3((unsigned)blocks), dim3(256), lds_bytes, s, g, items);
        return launch_status("hessian_gemm16t_list");
    }
    static OncePerDevice attr_once;
    if (attr_once.first_on_device() != 0) {
        if (hipFuncSetAttribute((const void*)k_syrk16b<NNCF_AMD_BF16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess ||
            hipFuncSetAttribute((const void*)k_syrk16b<NNCF_AMD_F16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess) {
            attr_once.forget_current();
            return launch_status("hessian_syrk16b attribute");
        }
    }
    const dim3 grid((unsigned)prs, (unsigned)sp);
    if (x_dtype == NNCF_AMD_BF16)
        hipLaunchKernelGGL((k_syrk16b<NNCF_AMD_BF16>), grid, dim3(256), lds_bytes, s, h, items, k, alpha, nt, (int)sps);
    else
        hipLaunchKernelGGL((k_syrk16b<NNCF_AMD_F16>), grid, dim3(256), lds_bytes, s, h, items, k, alpha, nt, (int)sps);
    return launch_status("hessian_syrk16b");
}

extern "C" int nncf_amd_hessian_accumulate_list(float* h, int n, const void* const* xs_host, const int64_t* tokens_host,
                                                int x_dtype, int64_t k, float alpha, void* stream) {
    using namespace wc;
    if (!h || k <= 0 || n < 0 || (n > 0 && (!xs_host || !tokens_host))) {
        set_error("hessian_accumulate_list: bad argument (k=%lld n=%d)", (long long)k, n);
        return NNCF_AMD_ERR_BAD_ARG;
    }
    for (int i = 0; i < n; ++i)
        if (tokens_host[i] < 0 || (tokens_host[i] > 0 && !xs_host[i])) {
            set_error("hessian_accumulate_list: item %d: bad pointer or token count", i);
            return NNCF_AMD_ERR_BAD_ARG;
        }
    if (x_dtype != NNCF_AMD_F32 && x_dtype != NNCF_AMD_F16 && x_dtype != NNCF_AMD_BF16) {
        set_error("hessian_accumulate_list: unsupported dtype %d", x_dtype);
        return NNCF_AMD_ERR_DTYPE;
    }
    hipStream_t s = (hipStream_t)stream;
    bool list_ok = x_dtype != NNCF_AMD_F32 && k >= 2048 && k < 46341 && k % 8 == 0;
    for (int i = 0; i < n && list_ok; ++i)
        list_ok = (((uintptr_t)xs_host[i] & 15u) == 0) && tokens_host[i] * k * 2 <= 0xffffffffll;
    if (!list_ok) {  // shapes the list kernel does not take: item by item through the single-chunk entry
        for (int i = 0; i < n; ++i) {
            const int st = hessian_accumulate_impl(h, xs_host[i], x_dtype, tokens_host[i], k, alpha, 1.0f, stream, false);
            if (st != NNCF_AMD_OK) return st;
        }
        return NNCF_AMD_OK;
    }
    for (int first = 0; first < n;) {
        SyrkItems items = {};
        int j = 0;
        uint32_t acc = 0;
        for (; j < kSyrkItems && first + j < n; ++j) {
            items.x[j] = xs_host[first + j];
            items.bytes[j] = (uint32_t)(tokens_host[first + j] * k * 2);
            items.stage_start[j] = acc;
            acc += (uint32_t)((tokens_host[first + j] + GK - 1) / GK);
        }
        items.stage_start[j] = acc;
        items.n = j;
        first += j;
        const int st = launch_syrk16b_list(h, items, x_dtype, k, alpha, s);
        if (st != NNCF_AMD_OK) return st;
    }
    return NNCF_AMD_OK;
}

extern "C" int nncf_amd_hessian_accumulate(float* h, const void* x, int x_dtype, int64_t tokens, int64_t k, float alpha,
                                           float beta, void* stream) {
    return hessian_accumulate_impl(h, x, x_dtype, tokens, k, alpha, beta, stream, true);
}

extern "C" int nncf_amd_hessian_accumulate_upper(float* h, const void* x, int x_dtype, int64_t tokens, int64_t k,
                                                 float alpha, float beta, void* stream) {
    return hessian_accumulate_impl(h, x, x_dtype, tokens, k, alpha, beta, stream, false);
}

extern "C" int nncf_amd_hessian_finalize(float* h, int64_t k, float scale, void* stream) {
    using namespace wc;
    if (!h || k <= 0) {
        set_error("hessian_finalize: bad argument (k=%lld)", (long long)k);
        return NNCF_AMD_ERR_BAD_ARG;
    }
    return mirror(h, k, (hipStream_t)stream, scale);  // scales the upper triangle and writes the lower one in one pass
}

static int hessian_accumulate_impl(float* h, const void* x, int x_dtype, int64_t tokens, int64_t k, float alpha,
                                   float beta, void* stream, bool do_mirror) {
    using namespace wc;
    if (!h || k <= 0 || tokens < 0 || (tokens > 0 && !x)) {
        set_error("hessian_accumulate: bad argument (k=%lld tokens=%lld)", (long long)k, (long long)tokens);
        return NNCF_AMD_ERR_BAD_ARG;
    }
    hipStream_t s = (hipStream_t)stream;
    if (beta != 1.0f) {
        const int64_t n = k * k;
        int64_t blocks = (n + 255) / 256;
        if (blocks > 8192) blocks = 8192;
        hipLaunchKernelGGL(k_scale_matrix, dim3((unsigned)blocks), dim3(256), 0, s, h, n, beta);
        int st = launch_status("hessian_scale");
        if (st != NNCF_AMD_OK) return st;
    }
    if (tokens == 0) return NNCF_AMD_OK;
    const int ntile = (int)((k + HB - 1) / HB);
    const int64_t pairs = (int64_t)ntile * (ntile + 1) / 2;
    // split the token axis until ~1024 blocks are in flight; every split is a multiple of the k-step
    int64_t splits = pairs >= 1024 ? 1 : (1024 + pairs - 1) / pairs;
    const int64_t max_splits = (tokens + 4 * HK - 1) / (4 * HK);
    if (splits > max_splits) splits = max_splits;
    if (splits > 65535) splits = 65535;
    int64_t rows_per_split = (tokens + splits - 1) / splits;
    rows_per_split = (rows_per_split + HK - 1) / HK * HK;
    splits = (tokens + rows_per_split - 1) / rows_per_split;
    dim3 grid((unsigned)pairs, (unsigned)splits);
    const int ev = x_dtype == NNCF_AMD_F32 ? 4 : 8;
    const bool vec = (((uintptr_t)x & 15u) == 0) && (k % ev == 0);
    if (vec && x_dtype != NNCF_AMD_F32) {  // 16-bit MFMA path
        const bool big = k >= 2048 && k < 46341;  // 256x256 tiles once there are enough of them (32-bit H offsets)
        const int tile = big ? GB : HB, kstep = big ? GK : HK16;
        const int nt = (int)((k + tile - 1) / tile);
        const int64_t prs = (int64_t)nt * (nt + 1) / 2;
        const int64_t want = big ? 2048 : 768;  // big: one block per CU at a time -> >= 8 rounds keep the tail short
        int64_t sp = prs >= want ? 1 : (want + prs - 1) / prs;
        const int64_t maxsp = (tokens + (big ? 32 : 8) * kstep - 1) / ((big ? 32 : 8) * kstep);  // >= 1024 / 256 tokens per block
        if (sp > maxsp) sp = maxsp;
        if (sp > 65535) sp = 65535;
        int64_t rps = (tokens + sp - 1) / sp;
        rps = (rps + kstep - 1) / kstep * kstep;
        sp = (tokens + rps - 1) / rps;
        dim3 g16((unsigned)prs, (unsigned)sp);
        if (big) {
            if (tokens * k * 2 > 0xffffffffll) {  // one item must fit a 32-bit buffer: cut it into row ranges
                const int64_t rows = (0xffffffffll / (k * 2)) / GK * GK;
                for (int64_t t0 = 0; t0 < tokens; t0 += rows) {
                    const int64_t nrow = tokens - t0 < rows ? tokens - t0 : rows;
                    int stp = hessian_accumulate_impl(h, (const char*)x + t0 * k * 2, x_dtype, nrow, k, alpha, 1.0f, stream, false);
                    if (stp != NNCF_AMD_OK) return stp;
                }
                return do_mirror ? mirror(h, k, s) : NNCF_AMD_OK;
            }
            SyrkItems items = {};
            items.n = 1;
            items.x[0] = x;
            items.bytes[0] = (uint32_t)(tokens * k * 2);
            items.stage_start[1] = (uint32_t)((tokens + GK - 1) / GK);
            int st16 = launch_syrk16b_list(h, items, x_dtype, k, alpha, s);
            if (st16 != NNCF_AMD_OK) return st16;
            return do_mirror ? mirror(h, k, s) : NNCF_AMD_OK;
        }
        if (x_dtype == NNCF_AMD_BF16)
            hipLaunchKernelGGL((k_syrk16<NNCF_AMD_BF16>), g16, dim3(256), 0, s, h, x, tokens, k, alpha, nt, rps);
        else
            hipLaunchKernelGGL((k_syrk16<NNCF_AMD_F16>), g16, dim3(256), 0, s, h, x, tokens, k, alpha, nt, rps);
        int st16 = launch_status("hessian_syrk16");
        if (st16 != NNCF_AMD_OK) return st16;
        return do_mirror ? mirror(h, k, s) : NNCF_AMD_OK;
    }
#define LAUNCH(DT)                                                                                                      \
    if (vec)                                                                                                            \
        hipLaunchKernelGGL((k_syrk<DT, true>), grid, dim3(256), 0, s, h, x, tokens, k, alpha, ntile, rows_per_split);   \
    else                                                                                                                \
        hipLaunchKernelGGL((k_syrk<DT, false>), grid, dim3(256), 0, s, h, x, tokens, k, alpha, ntile, rows_per_split)
    switch (x_dtype) {
        case NNCF_AMD_F32: LAUNCH(NNCF_AMD_F32); break;
        case NNCF_AMD_F16: LAUNCH(NNCF_AMD_F16); break;
        case NNCF_AMD_BF16: LAUNCH(NNCF_AMD_BF16); break;
        default:
            set_error("hessian_accumulate: unsupported dtype %d", x_dtype);
            return NNCF_AMD_ERR_DTYPE;
    }
#undef LAUNCH
    int st32 = launch_status("hessian_syrk");
    if (st32 != NNCF_AMD_OK) return st32;
    return do_mirror ? mirror(h, k, s) : NNCF_AMD_OK;
}


// ---------------------------------------------------------------------------------------------------------------------
extern "C" int nncf_amd_split3_bf16(const float* src, int64_t rows, int64_t cols, int64_t src_pitch, int contract_cols,
                                    int order, int negate, void* dst, int64_t dst_pitch, void* stream) {
    using namespace wc;
    if (rows < 0 || cols < 0 || src_pitch < cols || (order != 0 && order != 1) || dst_pitch < (contract_cols ? rows : cols)) {
        set_error("split3_bf16: bad argument");
        return NNCF_AMD_ERR_BAD_ARG;
    }
    if (rows == 0 || cols == 0) return NNCF_AMD_OK;
    if (!src || !dst) {
        set_error("split3_bf16: NULL buffer");
        return NNCF_AMD_ERR_BAD_ARG;
    }
    hipStream_t s = (hipStream_t)stream;
    const float sign = negate ? -1.0f : 1.0f;
    if (contract_cols) {
        const int64_t blocks = ((rows + 63) / 64) * ((cols + 63) / 64);
        if (blocks > 0x7fffffffLL) {
            set_error("split3_bf16: matrix too large for one launch");
            return NNCF_AMD_ERR_UNSUPPORTED;
        }
        hipLaunchKernelGGL(k_split3_cols, dim3((unsigned)blocks), dim3(256), 0, s, src, rows, cols, src_pitch, order, sign,
                           (uint16_t*)dst, dst_pitch);
    } else {
        const int64_t vecs = rows * ((cols + 3) / 4);
        if ((vecs + 255) / 256 > 0x7fffffffLL) {
            set_error("split3_bf16: matrix too large for one launch");
            return NNCF_AMD_ERR_UNSUPPORTED;
        }
        hipLaunchKernelGGL(k_split3_rows, dim3((unsigned)((vecs + 255) / 256)), dim3(256), 0, s, src, rows, cols, src_pitch,
                           order, sign, (uint16_t*)dst, dst_pitch);
    }
    return launch_status("split3_bf16");
}

extern "C" int nncf_amd_gemm_bf16_tn_acc(float* c, int64_t ldc, const void* p, int64_t ldp, const void* q, int64_t ldq,
                                         int64_t m, int64_t n, int64_t t, int tri, void* stream) {
    using namespace wc;
    if (m < 0 || n < 0 || t < 0 || ldc < n || ldp < m || ldq < n || tri < 0 || tri > 4) {
        set_error("gemm_bf16_tn_acc: bad argument");
        return NNCF_AMD_ERR_BAD_ARG;
    }
    if (m == 0 || n == 0 || t == 0) return NNCF_AMD_OK;
    if (!c || !p || !q) {
        set_error("gemm_bf16_tn_acc: NULL buffer");
        return NNCF_AMD_ERR_BAD_ARG;
    }
    // the DMA reads 16-byte chunks at 32-bit offsets inside each operand
    if (ldp % 8 != 0 || ldq % 8 != 0 || ((uintptr_t)p & 15u) || ((uintptr_t)q & 15u) || (uint64_t)t * ldp * 2 >= (1ull << 32) ||
        (uint64_t)t * ldq * 2 >= (1ull << 32) || m >= (1ll << 31) || n >= (1ll << 31) || ldc >= (1ll << 32)) {
        set_error("gemm_bf16_tn_acc: operands must be 16-byte aligned with row pitches in multiples of 8 and below 4 GiB each");
        return NNCF_AMD_ERR_UNSUPPORTED;
    }
    const int tm = (int)((m + GB - 1) / GB), tn = (int)((n + GB - 1) / GB);
    const int64_t tiles = (int64_t)tm * tn;
    const int64_t stages = (t + GK - 1) / GK;
    // >= 2 rounds of blocks over the 256 CUs where the contraction is long enough to split (>= 64 stages per block on
    // average; a triangular product has about half the tile x stage work of the full one)
    const int64_t work_tiles = tri == 0 ? tiles : (tiles + 1) / 2;
    // (developer knobs; measured: 1024 / 2048 / 4096 blocks wanted or 128 stages at least are all slower on the GPTQ
    // layer's products than 512 and 64)
    static const int64_t want_blocks = [] { const char* e = getenv("NNCF_AMD_GEMM_BLOCKS"); return e ? atoll(e) : 512ll; }();
    static const int64_t min_stages = [] { const char* e = getenv("NNCF_AMD_GEMM_MINST"); return e ? atoll(e) : 64ll; }();
    if (want_blocks < 1 || min_stages < 1) {
        set_error("gemm_bf16_tn_acc: NNCF_AMD_GEMM_BLOCKS / NNCF_AMD_GEMM_MINST must be positive");
        return NNCF_AMD_ERR_BAD_ARG;
    }
    int64_t sp = work_tiles >= want_blocks ? 1 : (want_blocks + work_tiles - 1) / work_tiles;
    const int64_t maxsp = (stages + min_stages - 1) / min_stages;
    if (sp > maxsp) sp = maxsp;
    if (sp < 1) sp = 1;
    if (tiles * sp > 0x7fffffffLL) {
        set_error("gemm_bf16_tn_acc: too many tiles");
        return NNCF_AMD_ERR_UNSUPPORTED;
    }
    constexpr size_t lds_bytes = (size_t)GST * 2 * GK * GROW;
    static OncePerDevice attr_once;
    if (attr_once.first_on_device() != 0) {
        if (hipFuncSetAttribute((const void*)k_gemm16t, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess) {
            attr_once.forget_current();
            return launch_status("gemm_bf16_tn_acc attribute");
        }
    }
    Gemm16Args g = {c, p, q, (uint32_t)((uint64_t)t * ldp * 2), (uint32_t)((uint64_t)t * ldq * 2), (uint32_t)ldp, (uint32_t)ldq,
                    (uint32_t)ldc, (uint32_t)m, (uint32_t)n, tm, tn, (int)stages, (int)sp, tri, 0, 0, 1.0f};
    // tile order of a triangular product: per-XCD lists where every XCD gets (nearly) the same number of lines
    // (measured: tri 1-3 from 28 tiles across on, tri 4 at 56; 14 tiles across: the alternating bands are faster)
    g.xcd_lists = tri == 0 ? 0 : (tri == 4 ? tm >= 8 : (tri == 3 ? tn : tm) >= 24);
    // blocks: 8 XCD lists of equal length (the kernel's mapping; lists that run out return at once)
    int64_t per_xcd = 0;
    if (!g.xcd_lists) {
        per_xcd = (tiles * sp + 7) / 8;
    } else if (tri == 4) {
        int64_t wanted = 0;
        for (int r = 0; r < tm; ++r) wanted += (r < tn ? r : tn - 1) + 1;
        g.wanted = (uint32_t)wanted;
        per_xcd = ((wanted + 7) / 8) * sp;
    } else {
        const int span = tri == 3 ? tn : tm, along = tri == 3 ? tm : tn;
        per_xcd = (int64_t)((span + 7) / 8) * along * sp;
    }
    const int64_t blocks = !g.xcd_lists ? tiles * sp : per_xcd * 8;
    if (blocks > 0x7fffffffLL) {
        set_error("gemm_bf16_tn_acc: too many tiles");
        return NNCF_AMD_ERR_UNSUPPORTED;
    }
    hipLaunchKernelGGL(k_gemm16t, dim3((unsigned)blocks), dim3(256), lds_bytes, (hipStream_t)stream, g);
    return launch_status("gemm_bf16_tn_acc");
}
