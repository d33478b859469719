// tools/mfmaprobe.hip -- raw MFMA issue rate on gfx950 (developer probe): bf16 32x32x16 and 16x16x32, 1..4 waves/SIMD
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int NACC>
__global__ __launch_bounds__(256) void k32(float* out, int iters) {
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    bf16x8 x, y;
    for (int i = 0; i < 8; ++i) { x[i] = (__bf16)(float)(threadIdx.x + i); y[i] = (__bf16)(float)(i + 1); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[a], 0, 0, 0);
    }
    float s = 0; for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k16(float* out, int iters) {
    f32x4 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 4; ++r) acc[a][r] = 0.f;
    bf16x8 x, y;
    for (int i = 0; i < 8; ++i) { x[i] = (__bf16)(float)(threadIdx.x + i); y[i] = (__bf16)(float)(i + 1); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, acc[a], 0, 0, 0);
    }
    float s = 0; for (int a = 0; a < NACC; ++a) for (int r = 0; r < 4; ++r) s += acc[a][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float* d; hipMalloc(&d, 4096 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int blocks : {256, 512, 1024}) {
        for (int which = 0; which < 4; ++which) {
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (which == 0) hipLaunchKernelGGL(k32<4>, dim3(blocks), dim3(256), 0, 0, d, iters);
                if (which == 1) hipLaunchKernelGGL(k32<8>, dim3(blocks), dim3(256), 0, 0, d, iters);
                if (which == 2) hipLaunchKernelGGL(k16<4>, dim3(blocks), dim3(256), 0, 0, d, iters);
                if (which == 3) hipLaunchKernelGGL(k16<8>, dim3(blocks), dim3(256), 0, 0, d, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const int nacc = (which & 1) ? 8 : 4;
            const double flops = (double)blocks * 4 * iters * nacc * 2.0 * 32 * 32 * 16;  // both shapes: 32768 flop
            const double fl = which < 2 ? flops : (double)blocks * 4 * iters * nacc * 2.0 * 16 * 16 * 32;
            printf("blocks %4d %s NACC=%d: %.3f ms  %.1f TFLOP/s  (%.1f ns per MFMA per wave)\n", blocks, which < 2 ? "32x32x16" : "16x16x32", nacc, ms,
                   fl / ms / 1e9, ms * 1e6 / ((double)iters * nacc));
        }
    }
    return 0;
}
