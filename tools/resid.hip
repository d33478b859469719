// tools/resid.hip -- how many 256-thread blocks are co-resident per CU for a given LDS / VGPR footprint?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
template <int VG>
__global__ __launch_bounds__(256) void k_resid(unsigned* arrived, unsigned* max_seen, float* sink, int spin_us) {
    extern __shared__ float dyn[];
    float r[VG];
#pragma unroll
    for (int i = 0; i < VG; ++i) r[i] = threadIdx.x * 0.5f + i;
    if (threadIdx.x == 0) atomicAdd(arrived, 1u);
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    unsigned seen = 0;
    while (__builtin_amdgcn_s_memrealtime() - t0 < (uint64_t)spin_us * 100) {  // 100 MHz
        seen = __hip_atomic_load(arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int i = 0; i < VG; ++i) r[i] = r[i] * 1.0001f + 0.5f;
    }
    float acc = 0;
#pragma unroll
    for (int i = 0; i < VG; ++i) acc += r[i];
    if (threadIdx.x == 0) { atomicMax(max_seen, seen); if (dyn) dyn[0] = acc; }
    if (acc == 12345.678f) sink[0] = acc + dyn[0];
}
template <int VG>
void run(int lds) {
    unsigned *arr, *mx; float* sink;
    CK(hipMalloc(&arr, 4)); CK(hipMalloc(&mx, 4)); CK(hipMalloc(&sink, 4));
    CK(hipFuncSetAttribute((const void*)k_resid<VG>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    int occ = -1;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_resid<VG>, 256, lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("VG=%3d lds=%6d occupancy API %d blocks/CU; kernel us for grid = k*256:", VG, lds, occ);
    for (int k = 1; k <= 10; ++k) {
        CK(hipMemset(arr, 0, 4)); CK(hipMemset(mx, 0, 4));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_resid<VG>, dim3(k * 256), dim3(256), lds, 0, arr, mx, sink, 100);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf(" %d:%.0f", k, ms * 1e3);
    }
    printf("\n");
}
int main() {
    for (int lds : {0, 16896, 33792, 65536}) run<8>(lds);
    run<40>(16896); run<70>(16896); run<90>(16896);
    return 0;
}
