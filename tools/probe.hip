// tools/probe.hip -- HBM streaming probes with the lowering kernels' traffic shape (16 B in, 4 B out per lane).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef uint32_t u4 __attribute__((ext_vector_type(4)));

template <int U, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void k_probe(const u4* __restrict__ in, uint32_t* __restrict__ out, int64_t nvec) {
    const int lane = threadIdx.x & 63;
    const int64_t wave_global = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int64_t step = (int64_t)gridDim.x * 256 * U;
    for (int64_t base = wave_global * 64 * U; base < nvec; base += step) {
        u4 r[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t v = base + u * 64 + lane;
            if (NTL) r[u] = __builtin_nontemporal_load(&in[v < nvec ? v : nvec - 1]);
            else r[u] = in[v < nvec ? v : nvec - 1];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t v = base + u * 64 + lane;
            const uint32_t x = r[u].x ^ r[u].y ^ r[u].z ^ r[u].w;
            if (v < nvec) { if (NTS) __builtin_nontemporal_store(x, &out[v]); else out[v] = x; }
        }
    }
}
// 2 : 1 mix (16 B in, 8 B out per lane): the traffic shape of the 8-bit lowering (bf16 in, one byte out)
typedef uint32_t u2 __attribute__((ext_vector_type(2)));
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_probe_8bit(const u4* __restrict__ in, u2* __restrict__ out, int64_t nvec) {
    const int lane = threadIdx.x & 63;
    const int64_t wave_global = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int64_t step = (int64_t)gridDim.x * 256 * U;
    for (int64_t base = wave_global * 64 * U; base < nvec; base += step) {
        u4 r[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t v = base + u * 64 + lane;
            if (NT) r[u] = __builtin_nontemporal_load(&in[v < nvec ? v : nvec - 1]);
            else r[u] = in[v < nvec ? v : nvec - 1];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t v = base + u * 64 + lane;
            u2 x;
            x.x = r[u].x ^ r[u].y;
            x.y = r[u].z ^ r[u].w;
            if (v < nvec) { if (NT) __builtin_nontemporal_store(x, &out[v]); else out[v] = x; }
        }
    }
}
// lane owns 64 contiguous bytes via LDS-DMA staging; 16-byte output store per lane
template <bool NT>
__global__ __launch_bounds__(256) void k_probe_dma(const u4* __restrict__ in, u4* __restrict__ out, int64_t nvec) {
    __shared__ u4 stage[4][256];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    u4* my = stage[wave];
    const int64_t ntiles = nvec / 256;
    const int64_t wg = (int64_t)blockIdx.x * 4 + wave, wc = (int64_t)gridDim.x * 4;
    auto dma = [&](int64_t t) {
        const char* tb = (const char*)in + t * 4096;
#pragma unroll
        for (int u = 0; u < 4; ++u)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(tb + (u * 64 + lane) * 16),
                                             (__attribute__((address_space(3))) void*)(my + u * 64), 16, 0, NT ? 2 : 0);
    };
    int64_t t = wg;
    if (t < ntiles) dma(t);
    for (; t < ntiles; t += wc) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        u4 r[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) r[k] = my[lane * 4 + (k ^ ((lane >> 2) & 3))];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (t + wc < ntiles) dma(t + wc);
        u4 o;
        o.x = r[0].x ^ r[0].y ^ r[0].z ^ r[0].w; o.y = r[1].x ^ r[1].y ^ r[1].z ^ r[1].w;
        o.z = r[2].x ^ r[2].y ^ r[2].z ^ r[2].w; o.w = r[3].x ^ r[3].y ^ r[3].z ^ r[3].w;
        if (NT) __builtin_nontemporal_store(o, &out[t * 64 + lane]); else out[t * 64 + lane] = o;
    }
}

int main(int argc, char** argv) {
    const int64_t n = 14336LL * 4096;  // bf16 elements per tensor
    const int NB = 16;
    std::vector<u4*> in(NB);
    std::vector<uint32_t*> out(NB);
    for (int i = 0; i < NB; ++i) { CK(hipMalloc(&in[i], n * 2)); CK(hipMemset(in[i], i + 1, n * 2)); CK(hipMalloc(&out[i], n)); }
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int64_t nvec = n / 8;
    auto timeit = [&](const char* name, auto launch) {
        for (int i = 0; i < NB; ++i) launch(i);
        CK(hipEventRecord(e0, st));
        for (int r = 0; r < 5; ++r) for (int i = 0; i < NB; ++i) launch(i);
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        double us = ms * 1e3 / (5 * NB);
        printf("%-40s %.2f us/tensor  %.1f GB/s (in+out)\n", name, us, n * 2.5 / us * 1e-3);
    };
    for (int grid : {2048, 4096, 8192, 16384}) {
        printf("grid %d\n", grid);
        timeit("U4 plain", [&](int i) { hipLaunchKernelGGL((k_probe<4, false, false>), dim3(grid), dim3(256), 0, st, in[i], out[i], nvec); });
        timeit("U4 nt-load", [&](int i) { hipLaunchKernelGGL((k_probe<4, true, false>), dim3(grid), dim3(256), 0, st, in[i], out[i], nvec); });
        timeit("U4 nt-store", [&](int i) { hipLaunchKernelGGL((k_probe<4, false, true>), dim3(grid), dim3(256), 0, st, in[i], out[i], nvec); });
        timeit("U4 nt-both", [&](int i) { hipLaunchKernelGGL((k_probe<4, true, true>), dim3(grid), dim3(256), 0, st, in[i], out[i], nvec); });
        timeit("U8 plain", [&](int i) { hipLaunchKernelGGL((k_probe<8, false, false>), dim3(grid), dim3(256), 0, st, in[i], out[i], nvec); });
        timeit("U8 nt-both", [&](int i) { hipLaunchKernelGGL((k_probe<8, true, true>), dim3(grid), dim3(256), 0, st, in[i], out[i], nvec); });
        timeit("U2 plain", [&](int i) { hipLaunchKernelGGL((k_probe<2, false, false>), dim3(grid), dim3(256), 0, st, in[i], out[i], nvec); });
        {
            auto timeit8 = [&](const char* name, auto launch) {
                for (int i = 0; i < NB; ++i) launch(i);
                CK(hipEventRecord(e0, st));
                for (int r = 0; r < 5; ++r) for (int i = 0; i < NB; ++i) launch(i);
                CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                double us = ms * 1e3 / (5 * NB);
                printf("%-40s %.2f us/tensor  %.1f GB/s (in+out, 2:1)\n", name, us, n * 3.0 / us * 1e-3);
            };
            timeit8("8-bit mix U4 plain", [&](int i) { hipLaunchKernelGGL((k_probe_8bit<4, false>), dim3(grid), dim3(256), 0, st, in[i], (u2*)out[i], nvec); });
            timeit8("8-bit mix U4 nt", [&](int i) { hipLaunchKernelGGL((k_probe_8bit<4, true>), dim3(grid), dim3(256), 0, st, in[i], (u2*)out[i], nvec); });
            timeit8("8-bit mix U8 nt", [&](int i) { hipLaunchKernelGGL((k_probe_8bit<8, true>), dim3(grid), dim3(256), 0, st, in[i], (u2*)out[i], nvec); });
        }
        timeit("DMA plain", [&](int i) { hipLaunchKernelGGL((k_probe_dma<false>), dim3(grid), dim3(256), 0, st, in[i], (u4*)out[i], nvec); });
        timeit("DMA nt", [&](int i) { hipLaunchKernelGGL((k_probe_dma<true>), dim3(grid), dim3(256), 0, st, in[i], (u4*)out[i], nvec); });
    }
    return 0;
}
