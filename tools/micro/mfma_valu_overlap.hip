// Do matrix instructions of one wave and VALU instructions of ANOTHER wave of the same SIMD overlap on gfx950?  (Round 4: the
// dissection of the contraction kernels is additive -- conversion + matrix instructions + LDS traffic -- as if they did not.)
// One workgroup per CU; waves 0-3 (one per SIMD) issue back-to-back independent v_mfma_f32_32x32x16_f16, waves 4-7 / 4-11 (one
// or two more per SIMD) the producers' conversion instructions (v_fma_mixlo_f16 / v_max3_f32) or ds_read_b128 / ds_write_b64.
// Each role is timed alone and together.  Build: hipcc -O2 --offload-arch=gfx950 mfma_valu_overlap.hip -o mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define REP8(X) X X X X X X X X

// role bits: 1 = waves 0-3 MFMA, 2 = other waves VALU, 4 = other waves LDS reads, 8 = other waves LDS writes
__global__ __launch_bounds__(1024) void k(float* out, int iters, int roles, long long* clk) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long t0 = clock64(), w0 = wall_clock64();
    if (wave < 4) {
        if ((roles & 1) && (roles & 16)) {   // accumulators in AGPRs
            f16x8 a, b;
            for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(lane * 0.01f + i); b[i] = (_Float16)(i - lane * 0.02f); }
            f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
            for (int i = 0; i < iters; ++i) {
                REP8(asm volatile("v_mfma_f32_32x32x16_f16 %0, %4, %5, %0\n v_mfma_f32_32x32x16_f16 %1, %4, %5, %1\n"
                                  "v_mfma_f32_32x32x16_f16 %2, %4, %5, %2\n v_mfma_f32_32x32x16_f16 %3, %4, %5, %3"
                                  : "+a"(c0), "+a"(c1), "+a"(c2), "+a"(c3) : "v"(a), "v"(b));)
            }
            out[blockIdx.x * 1024 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
        } else if ((roles & 1) && (roles & 64)) {   // 16x16x32: 4 result registers per 16 clocks instead of 16 per 32
            f16x8 a, b;
            for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(lane * 0.01f + i); b[i] = (_Float16)(i - lane * 0.02f); }
            f32x4 c0 = {}, c1 = {}, c2 = {}, c3 = {}, c4 = {}, c5 = {}, c6 = {}, c7 = {};
            for (int i = 0; i < iters; ++i) {   // 64 x 16 clocks = the same matrix-pipe time per iteration as 32 x 32 clocks
                REP8(c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c1, 0, 0, 0);
                     c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c3, 0, 0, 0);
                     c4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c4, 0, 0, 0); c5 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c5, 0, 0, 0);
                     c6 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c6, 0, 0, 0); c7 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c7, 0, 0, 0);)
            }
            out[blockIdx.x * 1024 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + c4[0] + c5[1] + c6[2] + c7[3];
        } else if ((roles & 1) && (roles & 128)) {   // 32x32x16 with an idle gap of 8 clocks behind every instruction
            f16x8 a, b;
            for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(lane * 0.01f + i); b[i] = (_Float16)(i - lane * 0.02f); }
            f32x16 c0 = {}, c1 = {};
            for (int i = 0; i < iters; ++i) {
                REP8(asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n s_nop 7\n v_mfma_f32_32x32x16_f16 %1, %2, %3, %1\n s_nop 7\n"
                                  "v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n s_nop 7\n v_mfma_f32_32x32x16_f16 %1, %2, %3, %1\n s_nop 7"
                                  : "+v"(c0), "+v"(c1) : "v"(a), "v"(b));)
            }
            out[blockIdx.x * 1024 + threadIdx.x] = c0[0] + c1[1];
        } else if (roles & 1) {
            f16x8 a, b;
            for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(lane * 0.01f + i); b[i] = (_Float16)(i - lane * 0.02f); }
            f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
            for (int i = 0; i < iters; ++i) {
                REP8(c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
                     c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);)
            }
            out[blockIdx.x * 1024 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
        }
    } else if (roles & 2) {
        float a0 = lane, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7, sc = 0.5f;
        unsigned d0 = 0, d1 = 0, d2 = 0, d3 = 0;
        for (int i = 0; i < iters; ++i) {   // 32 instructions per iteration: the producers' mix
            REP8(asm volatile("v_fma_mixlo_f16 %0, %4, %8, 0 op_sel_hi:[0,0,0]\n v_fma_mixhi_f16 %0, %5, %8, 0 op_sel_hi:[0,0,0]\n"
                              "v_fma_mixlo_f16 %1, %6, %8, 0 op_sel_hi:[0,0,0]\n v_fma_mixhi_f16 %1, %7, %8, 0 op_sel_hi:[0,0,0]"
                              : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(sc));)
        }
        out[blockIdx.x * 1024 + threadIdx.x] = a4 + a5 + a6 + a7 + __uint_as_float(d0 ^ d1 ^ d2 ^ d3);
    } else if (roles & 4) {
        f32x4 acc = {};
        const unsigned char* p = lds + (wave - 4) * 4096 + lane * 16;
        for (int i = 0; i < iters; ++i) {   // 8 x 1 KB per iteration
            f32x4 v0, v1, v2, v3, v4, v5, v6, v7;
            asm volatile("ds_read_b128 %0, %8\n ds_read_b128 %1, %8 offset:1024\n ds_read_b128 %2, %8 offset:2048\n ds_read_b128 %3, %8 offset:3072\n"
                         "ds_read_b128 %4, %8\n ds_read_b128 %5, %8 offset:1024\n ds_read_b128 %6, %8 offset:2048\n ds_read_b128 %7, %8 offset:3072\n"
                         "s_waitcnt lgkmcnt(0)"
                         : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3), "=v"(v4), "=v"(v5), "=v"(v6), "=v"(v7) : "v"((unsigned)(size_t)p) : "memory");
            acc += v0 + v7;
        }
        out[blockIdx.x * 1024 + threadIdx.x] = acc.x;
    } else if (roles & 256) {   // global -> LDS directly (no VGPR return): 8 x 1 KB per iteration
        const float* g = out + 262144 + ((blockIdx.x * 16 + wave) * 16384) % 1048576 + lane * 4;
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)g, 0, 65536, 0x00020000);
        auto* l = (__attribute__((address_space(3))) unsigned char*)(lds + (wave - 4) * 4096);
        for (int i = 0; i < iters; ++i) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, l, 16, lane * 16, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, l, 16, lane * 16, 0, 1024, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, l, 16, lane * 16, 0, 2048, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, l, 16, lane * 16, 0, 3072, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, l, 16, lane * 16, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, l, 16, lane * 16, 0, 1024, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, l, 16, lane * 16, 0, 2048, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, l, 16, lane * 16, 0, 3072, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    } else if (roles & 32) {   // global loads (L2-resident 64 KB per wave): data returning into VGPRs
        f32x4 acc = {};
        const float* g = out + 262144 + ((blockIdx.x * 16 + wave) * 16384) % 1048576 + lane * 4;
        for (int i = 0; i < iters; ++i) {   // 8 x 1 KB per iteration
            f32x4 v0, v1, v2, v3, v4, v5, v6, v7;
            asm volatile("global_load_dwordx4 %0, %8, off\n global_load_dwordx4 %1, %8, off offset:1024\n global_load_dwordx4 %2, %8, off offset:2048\n"
                         "global_load_dwordx4 %3, %8, off offset:3072\n global_load_dwordx4 %4, %8, off\n global_load_dwordx4 %5, %8, off offset:1024\n"
                         "global_load_dwordx4 %6, %8, off offset:2048\n global_load_dwordx4 %7, %8, off offset:3072\n s_waitcnt vmcnt(0)"
                         : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3), "=v"(v4), "=v"(v5), "=v"(v6), "=v"(v7) : "v"(g) : "memory");
            acc += v0 + v7;
        }
        out[blockIdx.x * 1024 + threadIdx.x] = acc.x;
    } else if (roles & 512) {   // ds_write_b128: 8 x 1 KB per iteration
        unsigned a = (unsigned)(size_t)(lds + (wave - 4) * 4096 + lane * 16);
        f32x4 v = {1.f * lane, 2.f, 3.f, 4.f};
        for (int i = 0; i < iters; ++i) {
            asm volatile("ds_write_b128 %0, %1\n ds_write_b128 %0, %1 offset:1024\n ds_write_b128 %0, %1 offset:2048\n ds_write_b128 %0, %1 offset:3072\n"
                         "ds_write_b128 %0, %1\n ds_write_b128 %0, %1 offset:1024\n ds_write_b128 %0, %1 offset:2048\n ds_write_b128 %0, %1 offset:3072\n"
                         "s_waitcnt lgkmcnt(0)" ::"v"(a), "v"(v) : "memory");
        }
    } else if (roles & 8) {
        unsigned a = (unsigned)(size_t)(lds + (wave - 4) * 4096 + lane * 8);
        float v0 = lane, v1 = 1.f;
        for (int i = 0; i < iters; ++i) {   // 8 x 512 B per iteration
            asm volatile("ds_write_b64 %0, %1\n ds_write_b64 %0, %1 offset:512\n ds_write_b64 %0, %1 offset:1024\n ds_write_b64 %0, %1 offset:1536\n"
                         "ds_write_b64 %0, %1 offset:2048\n ds_write_b64 %0, %1 offset:2560\n ds_write_b64 %0, %1 offset:3072\n ds_write_b64 %0, %1 offset:3584\n"
                         "s_waitcnt lgkmcnt(0)" ::"v"(a), "v"((double)v0 + v1) : "memory");
        }
    }
    if (lane == 0 && blockIdx.x == 0) {
        clk[wave] = clock64() - t0;
        clk[16 + wave] = wall_clock64() - w0;   // constant 100 MHz
    }
}

static float* out = nullptr;
static long long* clk = nullptr;
static double run(int threads, int iters, int roles) {
    if (!out) { hipMalloc(&out, (256 * 1024 + 1048576 + 65536) * 4); hipMemset(out, 0, (256 * 1024 + 1048576 + 65536) * 4); hipMalloc(&clk, 32 * 8); }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, out, 10, roles, clk);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, out, iters, roles, clk);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3;
}

int main() {
    const int iters = 4000;
    for (int form : {0, 64}) {
        for (int other : {2, 4, 8, 512, 32, 256}) {
            for (int threads : {512, 768}) {
                const double tm = run(256, iters, 1 | form), to = run(threads, iters, other), tb = run(threads, iters, 1 | form | other);
                long long c[32];
                hipMemcpy(c, clk, 32 * 8, hipMemcpyDeviceToHost);   // (of the "together" run: the last launch)
                const char* nm = other == 2 ? "VALU (fma_mix)" : other == 4 ? "LDS read b128" : other == 8 ? "LDS write b64" : other == 512 ? "LDS write b128" : other == 32 ? "global load x4" : "global->LDS x4";
                const char* fm = form == 0 ? "32x32x16 acc VGPR" : form == 16 ? "32x32x16 acc AGPR" : form == 64 ? "16x16x32         " : "32x32x16 + s_nop 7";
                printf("%s | %-15s %d other wave(s)/SIMD: MFMA alone %7.1f us, other alone %7.1f us, together %7.1f us -> %3.0f %% hidden | in workgroup 0 the MFMA wave took %7.1f us, the other %7.1f us\n",
                       fm, nm, (threads / 64 - 4) / 4, tm, to, tb, 100.0 * (tm + to - tb) / (tm < to ? tm : to), c[16] * 0.01, c[20] * 0.01);
            }
        }
    }
    // the shader clock under an all-CU matrix load: clock64() ticks of wave 0 of workgroup 0 over the MFMA loop
    {
        const double tm = run(256, iters, 1);
        long long c[16];
        hipMemcpy(c, clk, 16 * 8, hipMemcpyDeviceToHost);
        printf("MFMA alone: %lld clock64 ticks for %d MFMAs = %.1f ticks / MFMA, %.1f us => %.2f GHz\n", c[0], iters * 32, (double)c[0] / (iters * 32.0), tm,
               (double)c[0] / tm * 1e-3);
    }
    return 0;
}
