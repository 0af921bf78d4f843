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

// ---- ordered multi-block reductions ---------------------------------------------------------------------------------
// Column / scalar reductions that span several workgroups write one partial per workgroup into caller scratch; the LAST
// workgroup to arrive (a ticket counter, CUDA's threadFenceReduction pattern) adds the partials in index order.  One
// launch, no zero-fill, and the same bits on every run (the fp32 atomics this replaces rounded in arrival order).
// Workspace layout: [IX_TICKET_BYTES of tickets, zero on entry, left zero][partials ...].
#define IX_TICKET_BYTES 65536
#define IX_MAX_TICKETS (IX_TICKET_BYTES / 4)
static inline bool ix_al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

#ifdef __HIPCC__
// call after this workgroup's partial stores; true in exactly one workgroup per ticket, after all `nblk` have arrived
__device__ __forceinline__ bool ix_last_block(unsigned int* ticket, unsigned int nblk) {
    __shared__ int ix_last_flag;
    __threadfence();   // this thread's partial stores are visible device-wide before the ticket is taken
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int t = atomicAdd(ticket, 1u);
        ix_last_flag = (t == nblk - 1);
        if (ix_last_flag) atomicExch(ticket, 0u);   // ready for the next launch (stream-ordered)
    }
    __syncthreads();
    const bool last = ix_last_flag != 0;
    if (last) __threadfence();   // acquire: the other workgroups' partials
    return last;
}
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
