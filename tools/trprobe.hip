// tools/trprobe.hip -- semantics of ds_read_b64_tr_b16 on gfx950 (developer probe)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void k(uint16_t* out, int stride_bytes) {
    __shared__ uint16_t lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const char* base = (const char*)lds + threadIdx.x * stride_bytes;
    s4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)base);
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (uint16_t)r[j];
}
int main() {
    uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
    for (int stride : {8, 64, 256}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, stride);
        uint16_t h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("lane address = lane * %d bytes (element index = lane * %d); value = element index in LDS\n", stride, stride / 2);
        for (int l = 0; l < 64; ++l) printf("  lane %2d: %5d %5d %5d %5d\n", l, h[4 * l], h[4 * l + 1], h[4 * l + 2], h[4 * l + 3]);
    }
    return 0;
}
