// probe: does v_mfma_f32_16x16x32_f16 honour fp16 SUBNORMAL inputs on gfx950?  (the f16x2 split idea needs the low term of a
// small value, which is subnormal in fp16, to survive)   hipcc --offload-arch=gfx950 -O2 tools/mfma_f16_denorm.hip -o tools/mfma_f16_denorm
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out, float aval, float bval) {
    h8 a, b;
    for (int i = 0; i < 8; ++i) a[i] = (_Float16)aval, b[i] = (_Float16)bval;
    f4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = c[0], out[1] = (float)a[0];
}
int main() {
    float* d;
    hipMalloc(&d, 16);
    const float vals[] = {1.0f, 1e-3f, 3e-5f, 2e-6f, 1.2e-7f, 6e-8f};
    for (float v : vals) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, v, 1.0f);
        float h[2];
        hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
        printf("a = %.3e (as fp16 %.6e): sum over 32 k of a*1 = %.6e  -> per term %.6e\n", v, h[1], h[0], h[0] / 32);
    }
    return 0;
}
