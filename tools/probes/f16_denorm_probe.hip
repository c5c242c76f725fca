// Does v_mfma_f32_16x16x32_f16 honour f16 subnormal INPUTS on gfx950, and does v_cvt_f16_f32 produce them?
// (the two-way f16 split engine relies on both: the low part of a small element is a subnormal f16)
// build: hipcc --offload-arch=gfx950 -O2 tools/probes/f16_denorm_probe.hip -o tools/probes/f16_denorm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void probe(float a_val, float b_val, float *out) {
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)a_val; b[i] = (_Float16)b_val; }
    f4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) { out[0] = c[0]; out[1] = (float)(_Float16)a_val; }
}
int main() {
    float *d, h[2];
    hipMalloc(&d, 8);
    const float vals[3] = {1.0f, 6.1035e-5f /* 2^-14 min normal */, 9.5367e-7f /* 2^-20 subnormal */};
    for (float v : vals) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, v, 1024.f, d);
        hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
        printf("a=%g: cvt->f16->f32 = %g ; mfma sum_k a*1024 over 32 k = %g (expect %g)\n", v, h[1], h[0], 32.0 * v * 1024.0);
    }
    return 0;
}
