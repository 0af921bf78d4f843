// Shared host/device helpers for the Interactron gfx950 kernel library.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>

#define IX_OK 0
#define IX_ERR_ARG (-1)
#define IX_ERR_LAUNCH (-2)
#define IX_ERR_WORKSPACE (-3)

void ix_set_error(const char* fmt, ...);

// ix_set_dropout_salt: optional device word every dropout kernel XORs into its seed (see api.cpp)
extern const uint64_t* ix_g_salt;

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
void ix_prof_begin_wp(hipStream_t stream, int M, int N, int K, int nbatch);

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
// Partials cross XCDs (one L2 each): they are written and read with agent-scope accesses (write-through / L2-bypassing), so
// no cache-wide fence is needed -- __threadfence() here costs an L2 write-back per workgroup (measured: LayerNorm backward
// 16 -> 210 us).
__device__ __forceinline__ void ix_store_agent(float* p, float v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float ix_load_agent(const float* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// call after this workgroup's partial stores (ix_store_agent); true in exactly one workgroup per ticket, after all `nblk`
// have arrived.  The partials of the others are then read with ix_load_agent.
__device__ __forceinline__ bool ix_last_block(unsigned int* ticket, unsigned int nblk) {
    __shared__ int ix_last_flag;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this thread's partial stores have completed ...
    __syncthreads();                                    // ... and so have everybody else's in the workgroup
    if (threadIdx.x == 0) {
        const unsigned int t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ix_last_flag = (t == nblk - 1);
        if (ix_last_flag) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
    }
    __syncthreads();
    return ix_last_flag != 0;
}
// Two-level ordered column reduction.  Workgroup `blk` of `nblk` has stored its partial vector (`width` floats) at
// part1[blk * width ..].  Cohorts of IX_COHORT consecutive workgroups: the last arriver of a cohort adds its cohort's partials
// in workgroup order (into part2[cohort * width ..], or straight to `store` when there is one cohort); the last cohort to
// finish adds part2 in cohort order and calls store(c, sum).  The cohort sums run in parallel on different CUs, so the
// serial tail is IX_COHORT + nblk / IX_COHORT loads deep, not nblk.  tickets: ceil(nblk / IX_COHORT) + 1 counters.
#define IX_COHORT 32
static inline int ix_cohorts(int64_t nblk) { return (int)((nblk + IX_COHORT - 1) / IX_COHORT); }
template <typename Store>
__device__ __forceinline__ void ix_ordered_colsum(const float* __restrict__ part1, float* __restrict__ part2,
                                                  unsigned int* tickets, int blk, int nblk, int width, Store store) {
    const int ncoh = (nblk + IX_COHORT - 1) / IX_COHORT, coh = blk / IX_COHORT;
    const int members = min(IX_COHORT, nblk - coh * IX_COHORT);
    if (!ix_last_block(tickets + coh, members)) return;
    const float* src = part1 + (int64_t)coh * IX_COHORT * width;
    for (int c = threadIdx.x; c < width; c += blockDim.x) {
        float s = 0.f;
        int j = 0;
        for (; j + 8 <= members; j += 8) {   // eight loads in flight, added in index order
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = ix_load_agent(src + (int64_t)(j + u) * width + c);
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; j < members; ++j) s += ix_load_agent(src + (int64_t)j * width + c);
        if (ncoh == 1)
            store(c, s);
        else
            ix_store_agent(part2 + (int64_t)coh * width + c, s);
    }
    if (ncoh == 1) return;
    if (!ix_last_block(tickets + ncoh, ncoh)) return;
    for (int c = threadIdx.x; c < width; c += blockDim.x) {
        float s = 0.f;
        int k = 0;
        for (; k + 8 <= ncoh; k += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = ix_load_agent(part2 + (int64_t)(k + u) * width + c);
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; k < ncoh; ++k) s += ix_load_agent(part2 + (int64_t)k * width + c);
        store(c, s);
    }
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
