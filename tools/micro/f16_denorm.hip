// Does v_mfma_f32_32x32x16_f16 honour fp16 subnormal inputs?  A = 2^-20 (subnormal in fp16) everywhere, B = 1: every
// C element must be 16 * 2^-20 = 1.52587890625e-05 if subnormals are kept, 0 if they are flushed.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(float* out, float aval, float bval) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)aval; b[i] = (_Float16)bval; }
    f32x16 c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    out[threadIdx.x] = c[0];
}
int main() {
    float* d; hipMalloc(&d, 64 * 4);
    float h[64];
    const float vals[4][2] = {{9.5367431640625e-07f, 1.f}, {1.f, 9.5367431640625e-07f}, {9.5367431640625e-07f, 9.5367431640625e-07f * 1024.f}, {6.103515625e-05f, 1.f}};
    for (int t = 0; t < 4; ++t) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, vals[t][0], vals[t][1]);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("a=%g b=%g -> c=%.10g (expected %.10g)\n", vals[t][0], vals[t][1], h[0], 16.0 * vals[t][0] * vals[t][1]);
    }
    return 0;
}
