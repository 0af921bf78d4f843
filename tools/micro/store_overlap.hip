// Does a wave that issues global_store_dwordx4 stall for the ~270 clocks each one costs, or can it issue other work
// (VALU here, MFMA in the GEMM) underneath?  One wave per SIMD, 4 per CU.
// Build: hipcc -O2 --offload-arch=gfx950 store_overlap.hip -o store_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
#define VALU64 asm volatile("v_fma_f32 %0, %0, %4, %0\n v_fma_f32 %1, %1, %4, %1\n v_fma_f32 %2, %2, %4, %2\n v_fma_f32 %3, %3, %4, %3\n" \
                            "v_fma_f32 %0, %0, %4, %0\n v_fma_f32 %1, %1, %4, %1\n v_fma_f32 %2, %2, %4, %2\n v_fma_f32 %3, %3, %4, %3\n" \
                            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));
template <int MODE>   // 1: stores only, 2: VALU only, 3: both interleaved, 4: MFMA only, 5: MFMA + stores
__global__ void k(float* out, int iters) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b = 1.0001f;
    f4 v = {a0, a1, a2, a3};
    f16v acc = {0};
    bf8 fa, fb;
    for (int i = 0; i < 8; ++i) { fa[i] = (__bf16)(float)i; fb[i] = (__bf16)1.f; }
    f4* dst = reinterpret_cast<f4*>(out) + (size_t)blockIdx.x * blockDim.x * 64 + threadIdx.x;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 1 || MODE == 3 || MODE == 5) dst[(i & 63) * blockDim.x] = v;
        if (MODE == 2 || MODE == 3) { VALU64 VALU64 VALU64 VALU64 VALU64 VALU64 VALU64 VALU64 }   // 64 VALU ~ 290 clk
        if (MODE == 4 || MODE == 5) {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);   // 8 x 32 clk
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + acc[0] + v.x;
}
template <int MODE> static float run(float* out, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, out, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e6f / iters;   // ns per iteration
}
int main() {
    float* out; hipMalloc(&out, (size_t)256 * 256 * 64 * 16 + 1024);
    const int iters = 4000;
    printf("per iteration (one 1-KB store per wave and/or 64 VALU and/or 8 MFMA), ns:\n");
    printf("stores only %.0f   VALU only %.0f   stores+VALU %.0f   MFMA only %.0f   MFMA+stores %.0f\n", run<1>(out, iters), run<2>(out, iters),
           run<3>(out, iters), run<4>(out, iters), run<5>(out, iters));
    return 0;
}
