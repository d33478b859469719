// tools/fp4probe.hip -- semantics of v_cvt_scalef32_pk_fp4_f32 on gfx950 (developer probe)
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <vector>
__global__ void k(const float* in, unsigned* out, float scale, int n) {
    int i = threadIdx.x;
    if (i < n) out[i] = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(0xffffff00u, in[i], -in[i], scale, 0);
}
int main() {
    std::vector<float> v = {0.f, 0.1f, 0.25f, 0.2500001f, 0.5f, 0.75f, 0.7500001f, 0.7499999f, 1.0f, 1.25f, 1.2500001f, 1.5f, 1.75f, 2.0f, 2.5f, 2.5000002f,
                            3.0f, 3.5f, 4.0f, 5.0f, 5.0000005f, 6.0f, 7.0f, 100.f, INFINITY, NAN, 1e-30f, 1e-40f, -0.0f};
    float *din; unsigned* dout;
    hipMalloc(&din, v.size() * 4); hipMalloc(&dout, v.size() * 4);
    for (float scale : {1.0f, 2.0f, 0.5f, 3.0f, 1.0f / 1048576.0f}) {
        std::vector<float> x(v);
        for (auto& e : x) e *= (scale == 3.0f ? 2.0f : scale);
        hipMemcpy(din, x.data(), v.size() * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, din, dout, scale, (int)v.size());
        std::vector<unsigned> o(v.size());
        hipMemcpy(o.data(), dout, v.size() * 4, hipMemcpyDeviceToHost);
        printf("scale %g (inputs = listed value * %g):\n", scale, scale == 3.0f ? 2.0f : scale);
        for (size_t i = 0; i < v.size(); ++i) printf("  %-12g -> +:%x  -:%x   (word %08x)\n", v[i], o[i] & 15, (o[i] >> 4) & 15, o[i]);
    }
    return 0;
}
