// tools/kbench.hip -- standalone (no Python) timing harness for the lowering kernels through the C ABI,
// plus a bandwidth probe with the same traffic shape (16 B in, 4 B out per lane).  Developer tool.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/kbench.hip -Lnncf_amd -lnncf_amd_wc -o /tmp/kbench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "../include/nncf_amd_wc.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int U>
__global__ __launch_bounds__(256) void k_probe(const uint4* __restrict__ in, uint32_t* __restrict__ out, int64_t nvec) {
    const int lane = threadIdx.x & 63;
    const int64_t wave_global = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int64_t step = (int64_t)gridDim.x * 256 * U;
    for (int64_t base = wave_global * 64 * U; base < nvec; base += step) {
        uint4 r[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t v = base + u * 64 + lane;
            r[u] = v < nvec ? in[v] : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t v = base + u * 64 + lane;
            if (v < nvec) out[v] = r[u].x ^ r[u].y ^ r[u].z ^ r[u].w;
        }
    }
}

int main(int argc, char** argv) {
    int64_t rows = argc > 1 ? atoll(argv[1]) : 14336, cols = argc > 2 ? atoll(argv[2]) : 4096;
    int mode = argc > 3 ? atoi(argv[3]) : NNCF_AMD_INT4_SYM;
    int64_t gs = argc > 4 ? atoll(argv[4]) : 128;
    int reps = 50;
    int64_t n = rows * cols;
    uint16_t* w;
    uint8_t* packed;
    float* scale;
    int32_t* zp;
    CK(hipMalloc(&w, n * 2));
    CK(hipMalloc(&packed, n));
    CK(hipMalloc(&scale, (n / (gs > 0 ? gs : cols)) * 4));
    CK(hipMalloc(&zp, (n / (gs > 0 ? gs : cols)) * 4));
    std::vector<uint16_t> h(n);
    uint32_t s = 12345;
    for (int64_t i = 0; i < n; ++i) {
        s = s * 1664525u + 1013904223u;
        float f = ((int)(s >> 8) % 20001 - 10000) * 4e-6f;
        uint32_t b;
        memcpy(&b, &f, 4);
        h[i] = (uint16_t)(b >> 16);
    }
    CK(hipMemcpy(w, h.data(), n * 2, hipMemcpyHostToDevice));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const bool four = mode >= NNCF_AMD_INT4_ASYM;
    const bool asym = mode == NNCF_AMD_INT4_ASYM || mode == NNCF_AMD_INT8_ASYM;
    auto run = [&]() {
        int rc;
        if (mode <= NNCF_AMD_INT4_SYM)
            rc = nncf_amd_int_quantize(w, NNCF_AMD_BF16, rows, cols, 1, gs, mode, nullptr, nullptr, 0, four ? nullptr : packed,
                                       four ? packed : nullptr, scale, asym ? zp : nullptr, nullptr, st);
        else
            rc = nncf_amd_float_quantize(w, NNCF_AMD_BF16, rows, cols, 1, gs, mode, nullptr, nullptr, nullptr, packed, scale,
                                         nullptr, nullptr, st);
        if (rc) { printf("error %d: %s\n", rc, nncf_amd_last_error()); exit(1); }
    };
    for (int i = 0; i < 5; ++i) run();
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i) run();
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    double us = ms * 1e3 / reps;
    double g = gs > 0 ? gs : cols;
    double alg = n * (2.0 + (four ? 0.5 : 1.0) + 4.0 / g * (asym ? 2 : 1));
    printf("kernel mode=%d gs=%lld %lldx%lld: %.2f us  in %.1f GB/s  algorithmic %.1f GB/s (%.1f%% of 8 TB/s)\n", mode,
           (long long)gs, (long long)rows, (long long)cols, us, n * 2.0 / us * 1e-3, alg / us * 1e-3, alg / us * 1e-3 / 80.0);
    {   // HBM-cold: NB distinct tensors (> 256 MiB Infinity Cache in total) through the batched entry point
        const int NB = argc > 5 ? atoi(argv[5]) : 16;
        std::vector<const void*> ws(NB);
        std::vector<uint8_t*> ps(NB);
        std::vector<float*> ss(NB);
        std::vector<int32_t*> zs(NB);
        std::vector<int64_t> rs(NB, rows), cs(NB, cols);
        for (int i = 0; i < NB; ++i) {
            uint16_t* wi;
            CK(hipMalloc(&wi, n * 2));
            CK(hipMemcpy(wi, w, n * 2, hipMemcpyDeviceToDevice));
            ws[i] = wi;
            CK(hipMalloc(&ps[i], n));
            CK(hipMalloc(&ss[i], (n / (gs > 0 ? gs : cols)) * 4));
            CK(hipMalloc(&zs[i], (n / (gs > 0 ? gs : cols)) * 4));
        }
        auto runb = [&]() {
            int rc;
            if (mode <= NNCF_AMD_INT4_SYM)
                rc = nncf_amd_int_quantize_batched(NB, ws.data(), NNCF_AMD_BF16, rs.data(), cs.data(), gs, mode,
                                                   four ? nullptr : (void* const*)ps.data(), four ? ps.data() : nullptr,
                                                   ss.data(), asym ? zs.data() : nullptr, st);
            else
                rc = nncf_amd_float_quantize_batched(NB, ws.data(), NNCF_AMD_BF16, rs.data(), cs.data(), gs, mode, ps.data(),
                                                     ss.data(), nullptr, st);
            if (rc) { printf("error %d: %s\n", rc, nncf_amd_last_error()); exit(1); }
        };
        for (int i = 0; i < 2; ++i) runb();
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < 10; ++i) runb();
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        double usb = ms * 1e3 / 10;
        printf("batched x%d (HBM-cold): %.2f us  in %.1f GB/s  algorithmic %.1f GB/s (%.1f%% of 8 TB/s)\n", NB, usb,
               NB * n * 2.0 / usb * 1e-3, NB * alg / usb * 1e-3, NB * alg / usb * 1e-3 / 80.0);
        // probe over the same distinct buffers
        int64_t nvecb = n / 8;
        for (int grid : {2048, 8192}) {
            CK(hipEventRecord(e0, st));
            for (int r = 0; r < 5; ++r)
                for (int i = 0; i < NB; ++i)
                    hipLaunchKernelGGL(k_probe<4>, dim3(grid), dim3(256), 0, st, (const uint4*)ws[i], (uint32_t*)ps[i], nvecb);
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            double u = ms * 1e3 / (5 * NB);
            printf("probe (HBM-cold) grid=%d: %.2f us per tensor  %.1f GB/s (in+out)\n", grid, u, n * 2.5 / u * 1e-3);
        }
    }
    // bandwidth probe: same traffic shape
    int64_t nvec = n / 8;
    for (int grid : {1024, 2048, 4096, 8192, 16384}) {
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_probe<4>, dim3(grid), dim3(256), 0, st, (const uint4*)w, (uint32_t*)packed, nvec);
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_probe<4>, dim3(grid), dim3(256), 0, st, (const uint4*)w, (uint32_t*)packed, nvec);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        us = ms * 1e3 / reps;
        printf("probe U=4 grid=%d: %.2f us  %.1f GB/s (in+out)\n", grid, us, n * 2.5 / us * 1e-3);
    }
    return 0;
}
