// Issue rate of the VALU instructions the bf16x6 producers are made of, one wave per SIMD (the producers' situation)
// and four waves per SIMD.  Build: hipcc -O2 --offload-arch=gfx950 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(X) X X X X X X X X
#define KERNEL(NAME, ASM)                                                                         \
    __global__ void NAME(float* out, int iters) {                                                 \
        float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5,  \
              a6 = a0 + 6, a7 = a0 + 7;                                                           \
        float b0 = 1.5f, b1 = 2.5f;                                                               \
        for (int i = 0; i < iters; ++i) {                                                         \
            REP8(asm volatile(ASM : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1));) \
        }                                                                                         \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;       \
    }
// 8 independent instructions per asm block
KERNEL(k_sub, "v_sub_f32 %0, %0, %8\n v_sub_f32 %1, %1, %8\n v_sub_f32 %2, %2, %8\n v_sub_f32 %3, %3, %8\n v_sub_f32 %4, %4, %8\n v_sub_f32 %5, %5, %8\n v_sub_f32 %6, %6, %8\n v_sub_f32 %7, %7, %8")
KERNEL(k_and, "v_and_b32 %0, %0, %8\n v_and_b32 %1, %1, %8\n v_and_b32 %2, %2, %8\n v_and_b32 %3, %3, %8\n v_and_b32 %4, %4, %8\n v_and_b32 %5, %5, %8\n v_and_b32 %6, %6, %8\n v_and_b32 %7, %7, %8")
KERNEL(k_cvt, "v_cvt_pk_bf16_f32 %0, %0, %8\n v_cvt_pk_bf16_f32 %1, %1, %8\n v_cvt_pk_bf16_f32 %2, %2, %8\n v_cvt_pk_bf16_f32 %3, %3, %8\n v_cvt_pk_bf16_f32 %4, %4, %8\n v_cvt_pk_bf16_f32 %5, %5, %8\n v_cvt_pk_bf16_f32 %6, %6, %8\n v_cvt_pk_bf16_f32 %7, %7, %8")
KERNEL(k_perm, "v_perm_b32 %0, %0, %8, %9\n v_perm_b32 %1, %1, %8, %9\n v_perm_b32 %2, %2, %8, %9\n v_perm_b32 %3, %3, %8, %9\n v_perm_b32 %4, %4, %8, %9\n v_perm_b32 %5, %5, %8, %9\n v_perm_b32 %6, %6, %8, %9\n v_perm_b32 %7, %7, %8, %9")
KERNEL(k_lshl, "v_lshlrev_b32 %0, 16, %0\n v_lshlrev_b32 %1, 16, %1\n v_lshlrev_b32 %2, 16, %2\n v_lshlrev_b32 %3, 16, %3\n v_lshlrev_b32 %4, 16, %4\n v_lshlrev_b32 %5, 16, %5\n v_lshlrev_b32 %6, 16, %6\n v_lshlrev_b32 %7, 16, %7")
KERNEL(k_cndmask, "v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc")
// dependent chain of 8
KERNEL(k_sub_dep, "v_sub_f32 %0, %0, %8\n v_sub_f32 %0, %0, %8\n v_sub_f32 %0, %0, %8\n v_sub_f32 %0, %0, %8\n v_sub_f32 %0, %0, %8\n v_sub_f32 %0, %0, %8\n v_sub_f32 %0, %0, %8\n v_sub_f32 %0, %0, %8")
KERNEL(k_cvt_dep, "v_cvt_pk_bf16_f32 %0, %0, %8\n v_cvt_pk_bf16_f32 %0, %0, %8\n v_cvt_pk_bf16_f32 %0, %0, %8\n v_cvt_pk_bf16_f32 %0, %0, %8\n v_cvt_pk_bf16_f32 %0, %0, %8\n v_cvt_pk_bf16_f32 %0, %0, %8\n v_cvt_pk_bf16_f32 %0, %0, %8\n v_cvt_pk_bf16_f32 %0, %0, %8")
// packed fp32 add on register pairs
__global__ void k_pkadd(float* out, int iters) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 a0 = {1.f * threadIdx.x, 2.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, b = {1.5f, 2.5f};
    for (int i = 0; i < iters; ++i) {
        REP8(asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));)
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0.x + a1.y + a2.x + a3.y;
}
template <typename K>
static void run(const char* name, K kern, int threads) {
    float* out;
    hipMalloc(&out, 256 * 1024 * 4);
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, out, 100);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double inst_per_wave = 64.0 * iters;   // 8 x 8 per iteration
    const double waves_per_simd = threads / 64 / 4.0;
    printf("%-10s %4d threads/CU: %.2f ns per instruction per wave (%.1f clk at 2.2 GHz); per SIMD one instruction every %.1f clk\n", name, threads,
           ms * 1e6 / inst_per_wave, ms * 1e6 / inst_per_wave * 2.2, ms * 1e6 / inst_per_wave * 2.2 / (waves_per_simd < 1 ? 1 : waves_per_simd));
    hipFree(out);
}
int main() {
    for (int threads : {256, 1024}) {
        run("v_sub_f32", k_sub, threads); run("v_and_b32", k_and, threads); run("cvt_pk_bf16", k_cvt, threads); run("v_perm_b32", k_perm, threads);
        run("v_lshlrev", k_lshl, threads); run("v_cndmask", k_cndmask, threads); run("v_pk_add_f32", k_pkadd, threads);
        run("sub (dep)", k_sub_dep, threads); run("cvt (dep)", k_cvt_dep, threads);
    }
    return 0;
}
