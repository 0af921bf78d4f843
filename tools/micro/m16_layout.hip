// Layout facts the 16x16x32 attention kernels (csrc/flash16.hip) rest on, checked on the device:
//  (1) v_mfma_f32_16x16x32_f16: A[i][k] in lane (i = lane & 15, g = lane >> 4) elements k = 8 g + e; B[k][j] likewise with
//      j = lane & 15; C[i][j] in lane (j = lane & 15, g) register r <-> row i = 4 g + r.
//  (2) ds_read_b64_tr_b16 with PER-LANE addresses: lane l element e = the (l & 3)-th 16-bit value of the 8 bytes addressed
//      by lane 16 (l >> 4) + 4 e + ((l & 15) >> 2).
//  (3) v_permlane16_swap(x, x): [0] | [1] hold the values of lane ^ 16's row pair (xor-16 exchange).
// build: hipcc --offload-arch=gfx950 -O2 tools/micro/m16_layout.hip -o tools/micro/m16_layout
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void k_mfma(const float* A, const float* B, float* C) {   // A [16][32], B [32][16], C [16][16] row-major
    const int lane = threadIdx.x, n = lane & 15, g = lane >> 4;
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)A[n * 32 + 8 * g + e]; b[e] = (_Float16)B[(8 * g + e) * 16 + n]; }
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) C[(4 * g + r) * 16 + n] = c[r];
}

__global__ void k_tr(const unsigned short* in, const int* addr, unsigned short* out) {   // in: 2048 values; addr[lane]: element offset (multiple of 4)
    __shared__ __attribute__((aligned(16))) unsigned short lds[2048];
    for (int i = threadIdx.x; i < 2048; i += 64) lds[i] = in[i];
    __syncthreads();
    const int lane = threadIdx.x;
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lds + addr[lane]));
    for (int e = 0; e < 4; ++e) out[lane * 4 + e] = (unsigned short)v[e];
}

__global__ void k_swap(unsigned* out) {
    const unsigned x = threadIdx.x;
    auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false);
    auto s = __builtin_amdgcn_permlane32_swap(x, x, false, false);
    out[threadIdx.x * 4 + 0] = r[0]; out[threadIdx.x * 4 + 1] = r[1];
    out[threadIdx.x * 4 + 2] = s[0]; out[threadIdx.x * 4 + 3] = s[1];
}

int main() {
    int bad = 0;
    {   // (1)
        float hA[16 * 32], hB[32 * 16], hC[256], *dA, *dB, *dC;
        for (int i = 0; i < 512; ++i) { hA[i] = (float)((i * 7 + 3) % 13 - 6); hB[i] = (float)((i * 5 + 1) % 11 - 5); }
        hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dC, sizeof(hC));
        hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, 0, dA, dB, dC);
        hipMemcpy(hC, dC, sizeof(hC), hipMemcpyDeviceToHost);
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
            float s = 0; for (int k = 0; k < 32; ++k) s += hA[i * 32 + k] * hB[k * 16 + j];
            if (s != hC[i * 16 + j]) { if (bad < 5) printf("mfma mismatch at (%d,%d): %g vs %g\n", i, j, hC[i * 16 + j], s); ++bad; }
        }
        printf("(1) mfma_f32_16x16x32_f16 layout: %s\n", bad ? "FAIL" : "PASS");
    }
    {   // (2)
        unsigned short hin[2048], hout[256], *din, *dout; int haddr[64], *daddr; int b2 = 0;
        for (int i = 0; i < 2048; ++i) hin[i] = (unsigned short)i;
        for (int l = 0; l < 64; ++l) haddr[l] = 4 * ((l * 37 + 11) % 512);   // scrambled, 8-byte aligned
        hipMalloc(&din, sizeof(hin)); hipMalloc(&dout, sizeof(hout)); hipMalloc(&daddr, sizeof(haddr));
        hipMemcpy(din, hin, sizeof(hin), hipMemcpyHostToDevice); hipMemcpy(daddr, haddr, sizeof(haddr), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_tr, dim3(1), dim3(64), 0, 0, din, daddr, dout);
        hipMemcpy(hout, dout, sizeof(hout), hipMemcpyDeviceToHost);
        for (int l = 0; l < 64; ++l) for (int e = 0; e < 4; ++e) {
            const int src = 16 * (l >> 4) + 4 * e + ((l & 15) >> 2);
            const unsigned short want = (unsigned short)(haddr[src] + (l & 3));
            if (hout[l * 4 + e] != want) { if (b2 < 8) printf("tr mismatch lane %d elem %d: %u vs %u\n", l, e, hout[l * 4 + e], want); ++b2; }
        }
        printf("(2) ds_read_b64_tr_b16 per-lane addressing: %s\n", b2 ? "FAIL" : "PASS");
        bad += b2;
    }
    {   // (3)
        unsigned h[256], *d; int b3 = 0;
        hipMalloc(&d, sizeof(h));
        hipLaunchKernelGGL(k_swap, dim3(1), dim3(64), 0, 0, d);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        for (int l = 0; l < 64; ++l) {
            const unsigned a = h[l * 4], b = h[l * 4 + 1], c = h[l * 4 + 2], e = h[l * 4 + 3];
            if (!((a == (unsigned)l && b == (unsigned)(l ^ 16)) || (b == (unsigned)l && a == (unsigned)(l ^ 16)))) { if (b3 < 4) printf("permlane16_swap lane %d: %u %u\n", l, a, b); ++b3; }
            if (!((c == (unsigned)l && e == (unsigned)(l ^ 32)) || (e == (unsigned)l && c == (unsigned)(l ^ 32)))) { if (b3 < 8) printf("permlane32_swap lane %d: %u %u\n", l, c, e); ++b3; }
        }
        printf("(3) permlane16/32_swap(x, x) = {own, xor-partner}: %s\n", b3 ? "FAIL" : "PASS");
        bad += b3;
    }
    return bad ? 1 : 0;
}
