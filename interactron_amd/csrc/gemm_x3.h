// Internal interface of the pre-split fp16x3 contraction path (gemm_x3.hip), used by ix_gemm_f32_ws (gemm.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

struct X3Call {
    const float* A;      // A(m, k): a_kc ? A[m * lda + k] : A[k * lda + m]
    const float* B;      // B(k, n): b_kc ? B[n * ldb + k] : B[k * ldb + n]
    float* C;            // row-major, ldc
    const float* bias;   // [N] per batch slice (sBias elements apart; 0 = shared) or null
    int M, N, K, a_kc, b_kc, nbatch;
    int64_t lda, ldb, ldc, sA, sB, sC, sBias;   // batch strides in elements (sA / sB 0: operand shared by all slices)
    float alpha;
};

size_t ix_x3_workspace_bytes(int M, int N, int K, int nbA, int nbB);
int ix_x3_gemm(const X3Call& c, void* workspace, hipStream_t stream);
