// Shared host/device helpers for the Interactron gfx950 kernel library.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define IX_OK 0
#define IX_ERR_ARG (-1)
#define IX_ERR_LAUNCH (-2)
#define IX_ERR_WORKSPACE (-3)

void ix_set_error(const char* fmt, ...);

#define IX_CHECK_ARG(cond, ...)                         \
    do {                                                \
        if (!(cond)) {                                  \
            ix_set_error(__VA_ARGS__);                  \
            return IX_ERR_ARG;                          \
        }                                               \
    } while (0)

// Launches are asynchronous; this only catches configuration errors (bad grid, missing code object...).
#define IX_CHECK_LAUNCH(name)                                                         \
    do {                                                                              \
        hipError_t e__ = hipGetLastError();                                           \
        if (e__ != hipSuccess) {                                                      \
            ix_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));      \
            return IX_ERR_LAUNCH;                                                     \
        }                                                                             \
    } while (0)

// launch statistics / per-launch HIP-event brackets shared by the contraction and attention kernels (gemm.hip)
void ix_prof_begin(hipStream_t stream, int kind, double flops, double mfma_flops, int tag);
void ix_prof_end(hipStream_t stream);

static inline int ix_div_up(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// Memory-bound elementwise launches: cap the grid at ~8 blocks per CU and grid-stride the rest.
static inline int ix_grid_1d(int64_t work_items, int block) {
    int64_t g = (work_items + block - 1) / block;
    if (g > 256 * 16) g = 256 * 16;
    if (g < 1) g = 1;
    return (int)g;
}

#ifdef __HIPCC__
__device__ __forceinline__ float ix_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float ix_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// Block-wide sum for blockDim.x == 256 (4 waves). `red` is a 4-float LDS scratch.
__device__ __forceinline__ float ix_block_sum_256(float v, float* red) {
    v = ix_wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}
__device__ __forceinline__ float ix_block_max_256(float v, float* red) {
    v = ix_wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
#endif
