// bf16 GEMM for the 16-bit activation mode (MODEL.COMPUTE_DTYPE: bf16 -- BASELINE.json configs[1] "multi_frame_baseline ... bf16"):
//
//   C[b](M x N) = act( (alpha * A[b](M x K) * B[b](K x N) + bias[n]) * scale[n] + shift[n] + residual[m][n] )
//
// A, B live in HBM as bf16 -- activations written as bf16 by their producers, weights cast once per version -- so nothing is
// converted on the way to the matrix cores: both operand tiles go HBM -> LDS by LDS-DMA (buffer_load ... lds, 16 bytes per lane, no
// registers, no VALU), one v_mfma_f32_16x16x32_bf16 per k-slice, fp32 accumulation, bf16 (or fp32) store.  This is the kernel form the
// fp32-on-16-bit kernels of gemm.hip cannot be (their producers split every fp32 element into fp16 planes on the SIMDs that issue the
// matrix instructions: DESIGN.md 4.1c).  Reference call sites: every nn.Linear / 1x1 convolution of models/detr_models/transformer.py:
// 148-232, models/gpt.py:39-78, models/detr_models/detr.py:37-40,299-311, models/transformer.py:47-66 and the two gradients autograd
// derives from each of them.
//
// Operand layouts (the same four the fp32 entry point takes):  A(m, k) = a_kc ? A[m * lda + k] : A[k * lda + m],
// B(k, n) = b_kc ? B[n * ldb + k] : B[k * ldb + n];  C[m * ldc + n].  Rows must start on 16 bytes (ld, offsets, strides % 8 == 0).
//
// Tile 128 x 128 x 64, four waves of 64 x 64 outputs, ONE 32 KB LDS stage, <= 128 registers: four workgroups per CU cover each
// other's DMA latency and barriers (the structure of gemm_wp.hip, measured there against a two-stage / two-workgroup predecessor).
//   * k-contiguous operand: LDS image [128 rows][64 k] (128-byte rows), 16-byte chunks XOR-swizzled with (row >> 1) & 7 ON THE
//     SOURCE ADDRESS (the DMA writes lane-linear); fragments by ds_read_b128, conflict-free.
//   * m / n-contiguous operand (the transposed ones of the gradients: dW = dY^T X has BOTH): LDS image [64 k][128 m] (256-byte
//     rows), 16-byte units XOR-swizzled with 2 ((k & 3) + 4 ((k >> 3) & 1)); fragments by ds_read_b64_tr_b16, the gfx950 transposing
//     read (lane (i, g) receives k = kbase + 0..3 of column i: tools/micro/m16_layout.hip fact 2), conflict-free.
//   * the matrix instruction computes C^T blocks (its A operand is the N-side fragment, its B operand the M-side one): a lane then
//     holds FOUR CONSECUTIVE n of one row m -- 8-byte bf16 / 16-byte fp32 stores, and bias / scale / shift as one float4 per block.
//   * K tails and rows beyond the operand are requested out of range (the buffer load returns zeros); M / N tails clamp the row.
//   * split-K (small outputs with long K: the weight gradients): fp32 partial planes in the caller's workspace, added in order by
//     gemm16_reduce_kernel together with the epilogue -- deterministic, like the fp32 kernels' split-K.
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int G16_BM = 128, G16_BN = 128, G16_BK = 64;
constexpr int G16_IMG = 128 * 64 * 2;   // one operand tile in LDS: 16 KB in either orientation

void ix_prof_begin_b16(hipStream_t stream, int M, int N, int K, int nbatch, double bytes);

// Division by an invariant positive integer d:  n / d = (mulhi(n, m) + n) >> s  for 0 <= n < 2^31 (the FastDiv of gemm.hip)
struct G16Div { unsigned m; int s; };
__device__ __forceinline__ int g16_div(int n, G16Div f) { return (int)((__umulhi((unsigned)n, f.m) + (unsigned)n) >> f.s); }
static inline G16Div g16_make_div(unsigned d) {
    G16Div f;
    f.s = 0;
    while ((1ull << f.s) < d) ++f.s;
    f.m = (unsigned)((((1ull << 32) * ((1ull << f.s) - d)) / d) + 1);
    return f;
}
// Implicit-GEMM convolution operands (no patch matrix in HBM; the scheme of ConvGather in gemm.hip, here as per-lane SOURCE
// addresses of the LDS-DMA).  A "pixel row" index decomposes over a grid, row = (img * gH + gy) * gW + gx, and tap (ky, kx) of that
// pixel reads the NHWC source [img][sH][sW][sC] at  sy = (gy * a + b + ky * d) >> qs,  sx likewise (valid iff divisible by 1 << qs
// and inside; invalid taps are requested out of range: zeros).
//   mode 1: A rows are pixels, k = (tap, c): a K step of 64 lies inside one tap (sC % 64 == 0)            forward, data gradient
//           + btap: B(k, n) with k = (tap, co) is W[co][tap][n]: rows co of a matrix with ld = ldb at column offset tap * bcol   (data gradient)
//   mode 2: B rows k are pixels, n = (tap, c): an N tile of 128 lies inside one tap (sC % 128 == 0)       weight gradient
struct G16Conv {
    int mode, btap, bcol;
    int gH, gW, sH, sW, sC;
    int a, b, d, qs, KW;
    G16Div dW, dHW, dC;   // divisions by gW, gH * gW, sC (mode 1: k -> tap; with btap: by the channels per tap of B's k index)
};

struct G16Args {
    G16Conv cg;
    const unsigned short* A;
    const unsigned short* B;
    void* C;
    const float* bias;
    const float *scale, *shift;
    const void* res;
    float* planes;          // split-K partial planes [split][batch][M][N] (fp32) or null
    float* rowsum;          // optional side output (m-contiguous A: the weight gradient's dY): rowsum[batch][m] = sum_k A(m, k) -- the bias
    float* rs_planes;       // gradient riding on the weight-gradient contraction; with split-K per-split partials [split][batch][M]
    int64_t sRowsum;
    int64_t lda, ldb, ldc, sAo, sAi, sBo, sBi, sCo, sCi, sBias;
    unsigned extA, extB;    // bytes addressable from a batch slice's base (buffer bounds)
    int M, N, K, nk, kps, split, tiles_m, tiles_n, batch_inner;
    float alpha;
    int act;                // 0 none, 1 ReLU, 2 GELU (tanh-free erf form, as F.gelu)
    int cst;                // bf16 C (and residual) rows start on 16 bytes: the epilogue stores whole row pieces through LDS
};

__device__ __forceinline__ int g16_xcd_swizzle(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

__device__ __forceinline__ void g16_dma16(__amdgpu_buffer_rsrc_t r, int voff, int soff, void* lds) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 0);
}
__device__ __forceinline__ u32x2 g16_tr(const unsigned char* p) {
    return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p)));
}
__device__ __forceinline__ unsigned g16_pack2(float a, float b) {   // two fp32 -> packed bf16 pair, round to nearest even
    unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
    ua = (ua + 0x7fffu + ((ua >> 16) & 1u)) >> 16;
    ub = (ub + 0x7fffu + ((ub >> 16) & 1u)) & 0xffff0000u;
    return ua | ub;
}
__device__ __forceinline__ float g16_bf(unsigned short v) { return __uint_as_float((unsigned)v << 16); }
__device__ __forceinline__ float g16_act(float v, int act) {
    if (act == 1) return fmaxf(v, 0.f);
    if (act == 2) return 0.5f * v * (1.f + erff(v * 0.70710678118654752f));
    return v;
}

#define G16_OOB 0x7ffffff0
// Diagnostic builds (`make diag16`, tools/gemm16_bench.py under IX_LIB_PATH; never loaded by the package): what one launch costs
// without one of its three parts.  1: no LDS reads / matrix instructions;  2: no DMA requests;  3: no epilogue stores
#ifndef G16_DIAG
#define G16_DIAG 0
#endif

// A_KC / B_KC: the operand's contracted index is contiguous in HBM.  F32OUT: C (and the residual) are fp32.
// CONV: 0 plain operands; 1 / 2: the gathering operand of an implicit-GEMM convolution (G16Conv.mode)
// STAGES: 1 = one 32 KB LDS stage, four workgroups per CU cover each other's DMA latency; 2 = two stages, the next K step's DMA in
// flight under this one's matrix instructions (counted vmcnt, raw barriers), two workgroups per CU
// RS: also rowsum[m] = sum_k A(m, k) (four more matrix instructions per k-slice against a fragment of ones, in the workgroups of the
// first N tile only)
template <bool A_KC, bool B_KC, bool F32OUT, int CONV = 0, int STAGES = 1, bool RS = false>
__global__ __launch_bounds__(256, STAGES == 1 ? (RS ? 3 : 4) : 2) void gemm16_kernel(G16Args p) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[STAGES * 2 * G16_IMG];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int idx = lane & 15, g = lane >> 4;
    const int nwg = p.tiles_m * p.tiles_n;
    const int tile = g16_xcd_swizzle(blockIdx.x, nwg);
    constexpr int GROUP_M = 8;
    const int group_size = GROUP_M * p.tiles_n;
    const int first_m = (tile / group_size) * GROUP_M;
    const int gm = min(p.tiles_m - first_m, GROUP_M);
    const int tm = first_m + (tile % group_size) % gm, tn = (tile % group_size) / gm;
    const int m0 = tm * G16_BM, n0 = tn * G16_BN;
    const int zb = blockIdx.y, bo = zb / p.batch_inner, bi = zb - bo * p.batch_inner;
    const int ks = blockIdx.z;
    const int kt0 = ks * p.kps, kt1 = min(p.nk, kt0 + p.kps);
    const unsigned short* A = p.A + bo * p.sAo + bi * p.sAi;
    const unsigned short* B = p.B + bo * p.sBo + bi * p.sBi;
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, p.extA, 0x00020000);
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, p.extB, 0x00020000);
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;

    // ---- DMA: pieces q = 4 wave .. 4 wave + 3 of each image; lane l of piece q fills the 16-byte slot q * 64 + l ----------------
    // per-lane byte offset of the slot's source at k step 0, the k index (0..63) it holds, and the per-k-step advance (scalar)
    int va[4], vb[4], ka[4], kb[4];
    int gpy[4], gpx[4], gbase[4];   // CONV: per slot, the pixel row's source coordinates at tap (0, 0) and its image's first pixel
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int slot = (wave * 4 + q) * 64 + lane;
        gpy[q] = gpx[q] = gbase[q] = 0;
        if (A_KC) {
            const int row = slot >> 3, c = (slot & 7) ^ ((row >> 1) & 7);
            va[q] = (min(m0 + row, p.M - 1) * (int)p.lda + c * 8) * 2;
            ka[q] = c * 8;
            if (CONV == 1) {
                const int r = min(m0 + row, p.M - 1);
                const int img = g16_div(r, p.cg.dHW), rem = r - img * (p.cg.gH * p.cg.gW);
                const int gy = g16_div(rem, p.cg.dW), gx = rem - gy * p.cg.gW;
                gpy[q] = gy * p.cg.a + p.cg.b;
                gpx[q] = gx * p.cg.a + p.cg.b;
                gbase[q] = img * (p.cg.sH * p.cg.sW);
                va[q] = c * 16;
            }
        } else {
            const int kr = slot >> 4, u = (slot & 15) ^ (2 * ((kr & 3) + 4 * ((kr >> 3) & 1)));
            va[q] = (kr * (int)p.lda + m0 + u * 8) * 2;
            ka[q] = kr;
        }
        if (B_KC) {
            const int row = slot >> 3, c = (slot & 7) ^ ((row >> 1) & 7);
            vb[q] = (min(n0 + row, p.N - 1) * (int)p.ldb + c * 8) * 2;
            kb[q] = c * 8;
        } else {
            const int kr = slot >> 4, u = (slot & 15) ^ (2 * ((kr & 3) + 4 * ((kr >> 3) & 1)));
            vb[q] = (kr * (int)p.ldb + n0 + u * 8) * 2;
            kb[q] = kr;
            if (CONV == 2) vb[q] = u * 16;   // (the column inside the tap's channels; the pixel part is computed per k step)
        }
    }
    const int stepA = A_KC ? G16_BK * 2 : G16_BK * (int)p.lda * 2;
    const int stepB = B_KC ? G16_BK * 2 : G16_BK * (int)p.ldb * 2;
    // CONV == 2: this N tile's tap and first channel (the tile lies inside one tap)
    int wt_ky = 0, wt_kx = 0, wt_c0 = 0;
    if (CONV == 2) {
        const int t = g16_div(n0, p.cg.dC);
        wt_ky = t / p.cg.KW; wt_kx = t - wt_ky * p.cg.KW; wt_c0 = n0 - t * p.cg.sC;
    }

    // ---- fragment addresses (bytes inside an image) -------------------------------------------------------------------------------
    // k-contiguous image: row (w + 16 rb + idx) * 128 + ((4 s + g) ^ swz) * 16, swz = (idx >> 1) & 7 (w and 16 rb are multiples of 16)
    // m-contiguous image: k = 32 s + 8 g + 4 h + (idx >> 2); unit (w / 8 + 2 rb + ((idx & 3) >> 1)) ^ f, f = 2 ((idx >> 2) + 4 (g & 1))
    const int swz = (idx >> 1) & 7;
    const int kcA = (wm + idx) * 128, kcB = (wn + idx) * 128;
    const int kc0 = ((g ^ swz) << 4), kc1 = (((4 + g) ^ swz) << 4);
    const int trk = (8 * g + (idx >> 2)) * 256 + 8 * (idx & 1);
    const int trf = 2 * ((idx >> 2) + 4 * (g & 1));
    const int truA = (wm >> 3) + ((idx & 3) >> 1), truB = (wn >> 3) + ((idx & 3) >> 1);

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 accr[4];   // RS: (ones x A^T) blocks -- every row of a block is the row sum of its 16 m
#pragma unroll
    for (int i = 0; i < 4; ++i) accr[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool rs_here = RS && tn == 0 && wn == 0;

    // the DMA requests of K step kt into the images at imgA / imgB
    auto issue = [&](int kt, unsigned char* imgA, unsigned char* imgB) {
        unsigned char* dA = imgA + wave * 4096;
        unsigned char* dB = imgB + wave * 4096;
        const int klim = p.K - kt * G16_BK;   // k indices >= klim of this stage do not exist: out-of-range request -> zeros
        int sa = kt * stepA, sb = kt * stepB;
        if (CONV == 1) {
            // this K step's tap (scalar) and, per slot, the source pixel of the slot's row under that tap
            const int k0 = kt * G16_BK, t = g16_div(k0, p.cg.dC);
            const int ky = t / p.cg.KW, kx = t - ky * p.cg.KW;
            sa = (k0 - t * p.cg.sC) * 2;
            const int qm = (1 << p.cg.qs) - 1;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                int y = gpy[q] + ky * p.cg.d, x = gpx[q] + kx * p.cg.d;
                bool ok = ((y | x) & qm) == 0;
                y >>= p.cg.qs; x >>= p.cg.qs;
                ok = ok && (unsigned)y < (unsigned)p.cg.sH && (unsigned)x < (unsigned)p.cg.sW;
                const int off = ((gbase[q] + y * p.cg.sW + x) * p.cg.sC) * 2 + va[q];
                g16_dma16(rA, ok ? off : G16_OOB, sa, dA + q * 1024);
            }
            if (p.cg.btap) {   // B rows k = (tap, co) live at W[co][tap][n]: row co of a [.., ldb] matrix at column offset tap * bcol
                const int tb = k0 / p.cg.btap;
                sb = ((k0 - tb * p.cg.btap) * (int)p.ldb + tb * p.cg.bcol) * 2;
            }
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) g16_dma16(rA, ka[q] < klim ? va[q] : G16_OOB, sa, dA + q * 1024);
        }
        if (CONV == 2) {
            // B rows are pixels: row kr of this stage is pixel kt * 64 + kr of the output grid, read at this tile's tap
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int pix = kt * G16_BK + kb[q];
                const int img = g16_div(pix, p.cg.dHW), rem = pix - img * (p.cg.gH * p.cg.gW);
                const int gy = g16_div(rem, p.cg.dW), gx = rem - gy * p.cg.gW;
                const int y = gy * p.cg.a + p.cg.b + wt_ky * p.cg.d, x = gx * p.cg.a + p.cg.b + wt_kx * p.cg.d;
                const bool ok = pix < p.K && (unsigned)y < (unsigned)p.cg.sH && (unsigned)x < (unsigned)p.cg.sW;
                const int off = (((img * p.cg.sH + y) * p.cg.sW + x) * p.cg.sC + wt_c0) * 2 + vb[q];
                g16_dma16(rB, ok ? off : G16_OOB, 0, dB + q * 1024);
            }
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) g16_dma16(rB, kb[q] < klim ? vb[q] : G16_OOB, sb, dB + q * 1024);
        }
    };
    if (STAGES == 2 && G16_DIAG != 2) issue(kt0, lds, lds + G16_IMG);
    for (int kt = kt0; kt < kt1; ++kt) {
        unsigned char* const imgA = lds + (STAGES == 2 ? ((kt - kt0) & 1) * 2 * G16_IMG : 0);
        unsigned char* const imgB = imgA + G16_IMG;
        if (STAGES == 2) {
            if (kt + 1 < kt1) {
                unsigned char* const nA = lds + (((kt - kt0) & 1) ^ 1) * 2 * G16_IMG;
                if (G16_DIAG != 2) issue(kt + 1, nA, nA + G16_IMG);
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // this wave's eight pieces of THIS stage have landed; the next stage's fly
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
        } else {
            if (G16_DIAG != 2) issue(kt, imgA, imgB);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of the stage have landed
            __syncthreads();                                    // everybody's have
        }
#pragma unroll
        for (int s = 0; s < (G16_DIAG == 1 ? 0 : 2); ++s) {
            u32x4 fm[4], fn[4];
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) {
                if (A_KC) {
                    fm[rb] = *reinterpret_cast<const u32x4*>(imgA + kcA + rb * 2048 + (s ? kc1 : kc0));
                } else {
                    const unsigned char* b0 = imgA + trk + s * 8192 + (((truA + 2 * rb) ^ trf) << 4);
                    const u32x2 lo = g16_tr(b0), hi = g16_tr(b0 + 4 * 256);
                    fm[rb] = u32x4{lo.x, lo.y, hi.x, hi.y};
                }
                if (B_KC) {
                    fn[rb] = *reinterpret_cast<const u32x4*>(imgB + kcB + rb * 2048 + (s ? kc1 : kc0));
                } else {
                    const unsigned char* b0 = imgB + trk + s * 8192 + (((truB + 2 * rb) ^ trf) << 4);
                    const u32x2 lo = g16_tr(b0), hi = g16_tr(b0 + 4 * 256);
                    fn[rb] = u32x4{lo.x, lo.y, hi.x, hi.y};
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fn[j]), __builtin_bit_cast(bf16x8, fm[i]),
                                                                        acc[i][j], 0, 0, 0);
            if (RS && rs_here) {   // (wave-uniform)
                const u32x4 ones = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    accr[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ones), __builtin_bit_cast(bf16x8, fm[i]), accr[i], 0, 0, 0);
            }
        }
        if (STAGES == 2) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();   // everybody has read this stage: the DMA of K step kt + 2 may overwrite it
        } else {
            __syncthreads();   // the stage may be overwritten
        }
    }

    if (RS && rs_here && g == 0) {   // lane (idx, 0), register 0 of block i: the sum over this split's k of row m = wm + 16 i + idx
        float* R = p.rs_planes ? p.rs_planes + ((int64_t)ks * gridDim.y + zb) * p.M : p.rowsum + bo * p.sRowsum;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + wm + 16 * i + idx;
            if (m < p.M) R[m] = accr[i].x;
        }
    }
    // ---- epilogue: lane (idx, g) of block (i, j) holds row m = wm + 16 i + idx, columns n = wn + 16 j + 4 g .. + 3 -----------------
    if (G16_DIAG == 3 && acc[0][0].x + acc[1][1].y + acc[2][2].z + acc[3][3].w != 12345.678f) return;
    if (p.planes) {   // split-K: the raw partial sums of this split, fp32, dense [M][N]
        float* P = p.planes + ((int64_t)ks * gridDim.y + zb) * (int64_t)p.M * p.N;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + wm + 16 * i + idx;
            if (m >= p.M) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = n0 + wn + 16 * j + 4 * g;
                if (n >= p.N) continue;
                *reinterpret_cast<f32x4*>(P + (int64_t)m * p.N + n) = acc[i][j];
            }
        }
        return;
    }
    const float* bias = p.bias ? p.bias + bo * p.sBias : nullptr;
    const int64_t cbase = bo * p.sCo + bi * p.sCi;
    if (!F32OUT && p.cst) {
        // bf16 result, rows of C on 16 bytes: through the (now free) LDS stage, so that HBM sees whole 256-byte row pieces instead of the
        // matrix instruction's layout (a lane holds 4 consecutive n: 8-byte stores, a wave instruction = sixteen 32-byte pieces of sixteen
        // rows -- measured round 6, `make diag16`: the stores were 6 of the 14.5 ms of the step's Linear shapes)
        unsigned short* const Cb = (unsigned short*)p.C + cbase;
        if (!p.res) {
            // the finished tile as bf16 [128 rows][256 B], 16-byte chunks XOR-swizzled with the row (& 15: = idx for every block)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = n0 + wn + 16 * j + 4 * g;
                if (n >= p.N) continue;
                f32x4 bv = {0.f, 0.f, 0.f, 0.f}, sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
                if (bias) bv = *reinterpret_cast<const f32x4*>(bias + n);
                if (p.scale) { sc = *reinterpret_cast<const f32x4*>(p.scale + n); sh = *reinterpret_cast<const f32x4*>(p.shift + n); }
                const int chunk = (((wn >> 3) + 2 * j + (g >> 1)) ^ idx) << 4;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    f32x4 v = (acc[i][j] * p.alpha + bv) * sc + sh;
                    v.x = g16_act(v.x, p.act); v.y = g16_act(v.y, p.act); v.z = g16_act(v.z, p.act); v.w = g16_act(v.w, p.act);
                    *reinterpret_cast<u32x2*>(lds + (wm + 16 * i + idx) * 256 + chunk + (g & 1) * 8) = u32x2{g16_pack2(v.x, v.y), g16_pack2(v.z, v.w)};
                }
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int row = q * 16 + (tid >> 4), c = tid & 15;
                const int m = m0 + row, n = n0 + c * 8;
                if (m >= p.M || n >= p.N) continue;
                const u32x4 v = *reinterpret_cast<const u32x4*>(lds + row * 256 + ((c ^ (row & 15)) << 4));
                unsigned short* dst = Cb + (int64_t)m * p.ldc + n;
                if (n + 8 <= p.N) *reinterpret_cast<u32x4*>(dst) = v;
                else *reinterpret_cast<u32x2*>(dst) = u32x2{v.x, v.y};   // (N % 4 == 0)
            }
        } else {
            // with a residual: the pre-residual values stay fp32 (one rounding, at the end) -- two halves of 64 rows x 512 B
            const unsigned short* const Rb = (const unsigned short*)p.res + cbase;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if ((wave >> 1) == h) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int n = n0 + wn + 16 * j + 4 * g;
                        if (n >= p.N) continue;
                        f32x4 bv = {0.f, 0.f, 0.f, 0.f}, sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
                        if (bias) bv = *reinterpret_cast<const f32x4*>(bias + n);
                        if (p.scale) { sc = *reinterpret_cast<const f32x4*>(p.scale + n); sh = *reinterpret_cast<const f32x4*>(p.shift + n); }
                        const int chunk = (((wn >> 2) + 4 * j + g) ^ idx) << 4;
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            *reinterpret_cast<f32x4*>(lds + (16 * i + idx) * 512 + chunk) = (acc[i][j] * p.alpha + bv) * sc + sh;
                    }
                }
                __syncthreads();
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int row = q * 16 + (tid >> 4), c = tid & 15;
                    const int m = m0 + 64 * h + row, n = n0 + c * 8;
                    if (m < p.M && n < p.N) {
                        const unsigned char* src = lds + row * 512;
                        f32x4 a = *reinterpret_cast<const f32x4*>(src + (((2 * c) ^ (row & 15)) << 4));
                        f32x4 b = *reinterpret_cast<const f32x4*>(src + (((2 * c + 1) ^ (row & 15)) << 4));
                        const int64_t o = (int64_t)m * p.ldc + n;
                        const bool full = n + 8 <= p.N;
                        u32x4 r = {0u, 0u, 0u, 0u};
                        if (full) r = *reinterpret_cast<const u32x4*>(Rb + o);
                        else { const u32x2 r2 = *reinterpret_cast<const u32x2*>(Rb + o); r.x = r2.x; r.y = r2.y; }
                        a.x += __uint_as_float(r.x << 16); a.y += __uint_as_float(r.x & 0xffff0000u);
                        a.z += __uint_as_float(r.y << 16); a.w += __uint_as_float(r.y & 0xffff0000u);
                        b.x += __uint_as_float(r.z << 16); b.y += __uint_as_float(r.z & 0xffff0000u);
                        b.z += __uint_as_float(r.w << 16); b.w += __uint_as_float(r.w & 0xffff0000u);
                        a.x = g16_act(a.x, p.act); a.y = g16_act(a.y, p.act); a.z = g16_act(a.z, p.act); a.w = g16_act(a.w, p.act);
                        b.x = g16_act(b.x, p.act); b.y = g16_act(b.y, p.act); b.z = g16_act(b.z, p.act); b.w = g16_act(b.w, p.act);
                        if (full) *reinterpret_cast<u32x4*>(Cb + o) = u32x4{g16_pack2(a.x, a.y), g16_pack2(a.z, a.w), g16_pack2(b.x, b.y), g16_pack2(b.z, b.w)};
                        else *reinterpret_cast<u32x2*>(Cb + o) = u32x2{g16_pack2(a.x, a.y), g16_pack2(a.z, a.w)};
                    }
                }
                if (h == 0) __syncthreads();
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = n0 + wn + 16 * j + 4 * g;
        if (n >= p.N) continue;
        f32x4 bv = {0.f, 0.f, 0.f, 0.f}, sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
        if (bias) bv = *reinterpret_cast<const f32x4*>(bias + n);
        if (p.scale) { sc = *reinterpret_cast<const f32x4*>(p.scale + n); sh = *reinterpret_cast<const f32x4*>(p.shift + n); }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + wm + 16 * i + idx;
            if (m >= p.M) continue;
            const int64_t o = cbase + (int64_t)m * p.ldc + n;
            f32x4 v = (acc[i][j] * p.alpha + bv) * sc + sh;
            if (p.res) {
                if (F32OUT) v += *reinterpret_cast<const f32x4*>((const float*)p.res + o);
                else {
                    const u32x2 r = *reinterpret_cast<const u32x2*>((const unsigned short*)p.res + o);
                    v.x += __uint_as_float(r.x << 16); v.y += __uint_as_float(r.x & 0xffff0000u);
                    v.z += __uint_as_float(r.y << 16); v.w += __uint_as_float(r.y & 0xffff0000u);
                }
            }
            v.x = g16_act(v.x, p.act); v.y = g16_act(v.y, p.act); v.z = g16_act(v.z, p.act); v.w = g16_act(v.w, p.act);
            if (F32OUT) *reinterpret_cast<f32x4*>((float*)p.C + o) = v;
            else *reinterpret_cast<u32x2*>((unsigned short*)p.C + o) = u32x2{g16_pack2(v.x, v.y), g16_pack2(v.z, v.w)};
        }
    }
}

// ---- the 256 x 256 tile form (round 6) ----------------------------------------------------------------------------------------------
// `make diag16` on the 128 x 128 kernel: operand traffic L2 -> LDS (2 bytes x (1/128 + 1/128) per multiply-add: 100 GB per step's Linear
// shapes, ~19 TB/s), the matrix instructions and the stores each cost about a third of a launch and barely overlap.  This form halves
// the operand traffic (1/256 + 1/256), runs ONE workgroup of eight waves per CU -- each wave 128 x 64 outputs, 128 accumulator
// registers -- and overlaps the DMA of K step k + 1 with the matrix instructions of K step k inside the workgroup: two 64 KB LDS stages
// (A rows 0..127, A rows 128..255, B rows 0..127, B rows 128..255: four images in the layouts of the 128 x 128 kernel, each filled by
// two waves), counted vmcnt, raw barriers.  Taken for plain operands where it measured faster (g16_use_big).
constexpr int GB_T = 256;
constexpr int GB_STAGE = 4 * G16_IMG;

template <bool A_KC, bool B_KC, bool F32OUT, bool RS>
__global__ __launch_bounds__(512, 1) void gemm16_big_kernel(G16Args p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // 2 x GB_STAGE
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int idx = lane & 15, g = lane >> 4;
    const int nwg = p.tiles_m * p.tiles_n;
    const int tile = g16_xcd_swizzle(blockIdx.x, nwg);
    constexpr int GROUP_M = 8;
    const int group_size = GROUP_M * p.tiles_n;
    const int first_m = (tile / group_size) * GROUP_M;
    const int gm = min(p.tiles_m - first_m, GROUP_M);
    const int tm = first_m + (tile % group_size) % gm, tn = (tile % group_size) / gm;
    const int m0 = tm * GB_T, n0 = tn * GB_T;
    const int zb = blockIdx.y, bo = zb / p.batch_inner, bi = zb - bo * p.batch_inner;
    const int ks = blockIdx.z;
    const int kt0 = ks * p.kps, kt1 = min(p.nk, kt0 + p.kps);
    const unsigned short* A = p.A + bo * p.sAo + bi * p.sAi;
    const unsigned short* B = p.B + bo * p.sBo + bi * p.sBi;
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, p.extA, 0x00020000);
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, p.extB, 0x00020000);
    const int wm = (wave >> 2) * 128, wn = (wave & 3) * 64, wnl = wn & 127;

    // ---- DMA: waves 2 h, 2 h + 1 fill image h (0, 1: A rows 0..127 / 128..255; 2, 3: B likewise), eight pieces of 64 slots each --------
    const int img = wave >> 1;
    const bool isA = img < 2;   // (wave-uniform)
    const int rbase = (isA ? m0 : n0) + 128 * (img & 1);
    int vo[8], kk[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int slot = ((wave & 1) * 8 + q) * 64 + lane;
        const bool kc = isA ? A_KC : B_KC;
        const int ld = isA ? (int)p.lda : (int)p.ldb, lim = isA ? p.M : p.N;
        if (kc) {
            const int row = slot >> 3, c = (slot & 7) ^ ((row >> 1) & 7);
            vo[q] = (min(rbase + row, lim - 1) * ld + c * 8) * 2;
            kk[q] = c * 8;
        } else {
            const int kr = slot >> 4, u = (slot & 15) ^ (2 * ((kr & 3) + 4 * ((kr >> 3) & 1)));
            vo[q] = (kr * ld + rbase + u * 8) * 2;
            kk[q] = kr;
        }
    }
    const int stepA = A_KC ? G16_BK * 2 : G16_BK * (int)p.lda * 2;
    const int stepB = B_KC ? G16_BK * 2 : G16_BK * (int)p.ldb * 2;
    const int step = isA ? stepA : stepB;

    const int swz = (idx >> 1) & 7;
    const int kc0 = ((g ^ swz) << 4), kc1 = (((4 + g) ^ swz) << 4);
    const int trk = (8 * g + (idx >> 2)) * 256 + 8 * (idx & 1);
    const int trf = 2 * ((idx >> 2) + 4 * (g & 1));
    const int tru0 = (idx & 3) >> 1;

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 accr[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) accr[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool rs_here = RS && tn == 0 && wn == 0;

    auto issue = [&](int kt, unsigned char* stage) {
        unsigned char* d = stage + img * G16_IMG + (wave & 1) * 8192;
        const int klim = p.K - kt * G16_BK;
        const int so = kt * step;
        if (isA) {
#pragma unroll
            for (int q = 0; q < 8; ++q) g16_dma16(rA, kk[q] < klim ? vo[q] : G16_OOB, so, d + q * 1024);
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) g16_dma16(rB, kk[q] < klim ? vo[q] : G16_OOB, so, d + q * 1024);
        }
    };
    issue(kt0, lds);
    for (int kt = kt0; kt < kt1; ++kt) {
        unsigned char* const stage = lds + ((kt - kt0) & 1) * GB_STAGE;
        if (kt + 1 < kt1) {
            issue(kt + 1, lds + (((kt - kt0) & 1) ^ 1) * GB_STAGE);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // this wave's eight pieces of THIS stage have landed; the next stage's fly
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        const unsigned char* const imgA = stage + (wm >> 7) * G16_IMG;
        const unsigned char* const imgB = stage + (2 + (wn >> 7)) * G16_IMG;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            u32x4 fn[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (B_KC) {
                    fn[j] = *reinterpret_cast<const u32x4*>(imgB + (wnl + idx) * 128 + j * 2048 + (s ? kc1 : kc0));
                } else {
                    const unsigned char* b0 = imgB + trk + s * 8192 + ((((wnl >> 3) + tru0 + 2 * j) ^ trf) << 4);
                    const u32x2 lo = g16_tr(b0), hi = g16_tr(b0 + 4 * 256);
                    fn[j] = u32x4{lo.x, lo.y, hi.x, hi.y};
                }
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                u32x4 fm;
                if (A_KC) {
                    fm = *reinterpret_cast<const u32x4*>(imgA + idx * 128 + i * 2048 + (s ? kc1 : kc0));
                } else {
                    const unsigned char* b0 = imgA + trk + s * 8192 + (((tru0 + 2 * i) ^ trf) << 4);
                    const u32x2 lo = g16_tr(b0), hi = g16_tr(b0 + 4 * 256);
                    fm = u32x4{lo.x, lo.y, hi.x, hi.y};
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fn[j]), __builtin_bit_cast(bf16x8, fm), acc[i][j], 0, 0, 0);
                if (RS && rs_here) {   // (wave-uniform)
                    const u32x4 ones = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
                    accr[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ones), __builtin_bit_cast(bf16x8, fm), accr[i], 0, 0, 0);
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // everybody has read this stage: the DMA of K step kt + 2 may overwrite it
    }

    if (RS && rs_here && g == 0) {
        float* R = p.rs_planes ? p.rs_planes + ((int64_t)ks * gridDim.y + zb) * p.M : p.rowsum + bo * p.sRowsum;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int m = m0 + wm + 16 * i + idx;
            if (m < p.M) R[m] = accr[i].x;
        }
    }
    // ---- epilogue: lane (idx, g) of block (i, j) holds row m = wm + 16 i + idx, columns n = wn + 16 j + 4 g .. + 3 -----------------
    if (p.planes) {
        float* P = p.planes + ((int64_t)ks * gridDim.y + zb) * (int64_t)p.M * p.N;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int m = m0 + wm + 16 * i + idx;
            if (m >= p.M) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = n0 + wn + 16 * j + 4 * g;
                if (n >= p.N) continue;
                *reinterpret_cast<f32x4*>(P + (int64_t)m * p.N + n) = acc[i][j];
            }
        }
        return;
    }
    const float* bias = p.bias ? p.bias + bo * p.sBias : nullptr;
    const int64_t cbase = bo * p.sCo + bi * p.sCi;
    if (!F32OUT && p.cst) {
        unsigned short* const Cb = (unsigned short*)p.C + cbase;
        if (!p.res) {
            // the finished tile as bf16 [256 rows][512 B] over both stages, 16-byte chunks XOR-swizzled with the row (& 15)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = n0 + wn + 16 * j + 4 * g;
                if (n >= p.N) continue;
                f32x4 bv = {0.f, 0.f, 0.f, 0.f}, sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
                if (bias) bv = *reinterpret_cast<const f32x4*>(bias + n);
                if (p.scale) { sc = *reinterpret_cast<const f32x4*>(p.scale + n); sh = *reinterpret_cast<const f32x4*>(p.shift + n); }
                const int chunk = (((wn >> 3) + 2 * j + (g >> 1)) ^ idx) << 4;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    f32x4 v = (acc[i][j] * p.alpha + bv) * sc + sh;
                    v.x = g16_act(v.x, p.act); v.y = g16_act(v.y, p.act); v.z = g16_act(v.z, p.act); v.w = g16_act(v.w, p.act);
                    *reinterpret_cast<u32x2*>(lds + (wm + 16 * i + idx) * 512 + chunk + (g & 1) * 8) = u32x2{g16_pack2(v.x, v.y), g16_pack2(v.z, v.w)};
                }
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int row = q * 16 + (tid >> 5), c = tid & 31;
                const int m = m0 + row, n = n0 + c * 8;
                if (m >= p.M || n >= p.N) continue;
                const u32x4 v = *reinterpret_cast<const u32x4*>(lds + row * 512 + ((c ^ (row & 15)) << 4));
                unsigned short* dst = Cb + (int64_t)m * p.ldc + n;
                if (n + 8 <= p.N) *reinterpret_cast<u32x4*>(dst) = v;
                else *reinterpret_cast<u32x2*>(dst) = u32x2{v.x, v.y};
            }
        } else {
            // with a residual: fp32 pre-residual values, two halves of 128 rows x 1024 B
            const unsigned short* const Rb = (const unsigned short*)p.res + cbase;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if ((wave >> 2) == h) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int n = n0 + wn + 16 * j + 4 * g;
                        if (n >= p.N) continue;
                        f32x4 bv = {0.f, 0.f, 0.f, 0.f}, sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
                        if (bias) bv = *reinterpret_cast<const f32x4*>(bias + n);
                        if (p.scale) { sc = *reinterpret_cast<const f32x4*>(p.scale + n); sh = *reinterpret_cast<const f32x4*>(p.shift + n); }
                        const int chunk = (((wn >> 2) + 4 * j + g) ^ idx) << 4;
#pragma unroll
                        for (int i = 0; i < 8; ++i)
                            *reinterpret_cast<f32x4*>(lds + (16 * i + idx) * 1024 + chunk) = (acc[i][j] * p.alpha + bv) * sc + sh;
                    }
                }
                __syncthreads();
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int row = q * 16 + (tid >> 5), c = tid & 31;
                    const int m = m0 + 128 * h + row, n = n0 + c * 8;
                    if (m < p.M && n < p.N) {
                        const unsigned char* src = lds + row * 1024;
                        f32x4 a = *reinterpret_cast<const f32x4*>(src + (((2 * c) ^ (row & 15)) << 4));
                        f32x4 b = *reinterpret_cast<const f32x4*>(src + (((2 * c + 1) ^ (row & 15)) << 4));
                        const int64_t o = (int64_t)m * p.ldc + n;
                        const bool full = n + 8 <= p.N;
                        u32x4 r = {0u, 0u, 0u, 0u};
                        if (full) r = *reinterpret_cast<const u32x4*>(Rb + o);
                        else { const u32x2 r2 = *reinterpret_cast<const u32x2*>(Rb + o); r.x = r2.x; r.y = r2.y; }
                        a.x += __uint_as_float(r.x << 16); a.y += __uint_as_float(r.x & 0xffff0000u);
                        a.z += __uint_as_float(r.y << 16); a.w += __uint_as_float(r.y & 0xffff0000u);
                        b.x += __uint_as_float(r.z << 16); b.y += __uint_as_float(r.z & 0xffff0000u);
                        b.z += __uint_as_float(r.w << 16); b.w += __uint_as_float(r.w & 0xffff0000u);
                        a.x = g16_act(a.x, p.act); a.y = g16_act(a.y, p.act); a.z = g16_act(a.z, p.act); a.w = g16_act(a.w, p.act);
                        b.x = g16_act(b.x, p.act); b.y = g16_act(b.y, p.act); b.z = g16_act(b.z, p.act); b.w = g16_act(b.w, p.act);
                        if (full) *reinterpret_cast<u32x4*>(Cb + o) = u32x4{g16_pack2(a.x, a.y), g16_pack2(a.z, a.w), g16_pack2(b.x, b.y), g16_pack2(b.z, b.w)};
                        else *reinterpret_cast<u32x2*>(Cb + o) = u32x2{g16_pack2(a.x, a.y), g16_pack2(a.z, a.w)};
                    }
                }
                if (h == 0) __syncthreads();
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = n0 + wn + 16 * j + 4 * g;
        if (n >= p.N) continue;
        f32x4 bv = {0.f, 0.f, 0.f, 0.f}, sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
        if (bias) bv = *reinterpret_cast<const f32x4*>(bias + n);
        if (p.scale) { sc = *reinterpret_cast<const f32x4*>(p.scale + n); sh = *reinterpret_cast<const f32x4*>(p.shift + n); }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int m = m0 + wm + 16 * i + idx;
            if (m >= p.M) continue;
            const int64_t o = cbase + (int64_t)m * p.ldc + n;
            f32x4 v = (acc[i][j] * p.alpha + bv) * sc + sh;
            if (p.res) {
                if (F32OUT) v += *reinterpret_cast<const f32x4*>((const float*)p.res + o);
                else {
                    const u32x2 r = *reinterpret_cast<const u32x2*>((const unsigned short*)p.res + o);
                    v.x += __uint_as_float(r.x << 16); v.y += __uint_as_float(r.x & 0xffff0000u);
                    v.z += __uint_as_float(r.y << 16); v.w += __uint_as_float(r.y & 0xffff0000u);
                }
            }
            v.x = g16_act(v.x, p.act); v.y = g16_act(v.y, p.act); v.z = g16_act(v.z, p.act); v.w = g16_act(v.w, p.act);
            if (F32OUT) *reinterpret_cast<f32x4*>((float*)p.C + o) = v;
            else *reinterpret_cast<u32x2*>((unsigned short*)p.C + o) = u32x2{g16_pack2(v.x, v.y), g16_pack2(v.z, v.w)};
        }
    }
}

// split-K tail: C = epilogue(sum over splits, in order); one thread per four consecutive n
template <bool F32OUT>
__global__ __launch_bounds__(256) void gemm16_reduce_kernel(G16Args p, int nbatch) {
    const int64_t per = (int64_t)p.M * (p.N / 4);
    const int64_t total = per * nbatch;
    const int64_t plane = (int64_t)nbatch * p.M * p.N;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int zb = (int)(t / per);
        const int64_t r = t - (int64_t)zb * per;
        const int m = (int)(r / (p.N / 4)), n = (int)(r - (int64_t)m * (p.N / 4)) * 4;
        const int bo = zb / p.batch_inner, bi = zb - bo * p.batch_inner;
        const float* src = p.planes + ((int64_t)zb * p.M + m) * p.N + n;
        f32x4 a = *reinterpret_cast<const f32x4*>(src);
        for (int s = 1; s < p.split; ++s) a += *reinterpret_cast<const f32x4*>(src + (int64_t)s * plane);
        f32x4 bv = {0.f, 0.f, 0.f, 0.f}, sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
        if (p.bias) bv = *reinterpret_cast<const f32x4*>(p.bias + bo * p.sBias + n);
        if (p.scale) { sc = *reinterpret_cast<const f32x4*>(p.scale + n); sh = *reinterpret_cast<const f32x4*>(p.shift + n); }
        const int64_t o = bo * p.sCo + bi * p.sCi + (int64_t)m * p.ldc + n;
        f32x4 v = (a * p.alpha + bv) * sc + sh;
        if (p.res) {
            if (F32OUT) v += *reinterpret_cast<const f32x4*>((const float*)p.res + o);
            else {
                const u32x2 q = *reinterpret_cast<const u32x2*>((const unsigned short*)p.res + o);
                v.x += __uint_as_float(q.x << 16); v.y += __uint_as_float(q.x & 0xffff0000u);
                v.z += __uint_as_float(q.y << 16); v.w += __uint_as_float(q.y & 0xffff0000u);
            }
        }
        v.x = g16_act(v.x, p.act); v.y = g16_act(v.y, p.act); v.z = g16_act(v.z, p.act); v.w = g16_act(v.w, p.act);
        if (F32OUT) *reinterpret_cast<f32x4*>((float*)p.C + o) = v;
        else *reinterpret_cast<u32x2*>((unsigned short*)p.C + o) = u32x2{g16_pack2(v.x, v.y), g16_pack2(v.z, v.w)};
    }
    if (p.rs_planes)   // the per-split partial row sums, in split order
        for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < (int64_t)nbatch * p.M; t += (int64_t)gridDim.x * 256) {
            float a = p.rs_planes[t];
            for (int s = 1; s < p.split; ++s) a += p.rs_planes[(int64_t)s * nbatch * p.M + t];
            const int zb = (int)(t / p.M);
            p.rowsum[(int64_t)(zb / p.batch_inner) * p.sRowsum + (t - (int64_t)zb * p.M)] = a;
        }
}

// ---- plan: how many splits of K ------------------------------------------------------------------------------------------------
static void g16_plan(int M, int N, int K, int nbatch, int* split, int* kps) {
    const int nk = ix_div_up(K, G16_BK);
    const int64_t tiles = (int64_t)ix_div_up(M, G16_BM) * ix_div_up(N, G16_BN) * nbatch;
    int s = 1;
    // 256 CUs x 4 resident workgroups: a launch with fewer than ~512 tiles and a long K is cut along K (at least 4 k steps a split)
    if (tiles < 512 && nk >= 8) {
        s = (int)((1024 + tiles - 1) / tiles);
        if (s > nk / 4) s = nk / 4;
        if (s > 64) s = 64;
        // ... and the planes are written and read once each: no more than ~24 MB of them (a 2048 x 512 output cut 16 ways moved 128 MB
        // for a 70 us contraction)
        const int64_t plane = (int64_t)M * N * nbatch * 4;
        const int cap = (int)((24ll << 20) / (plane > 0 ? plane : 1));
        if (s > cap) s = cap;
        if (s < 1) s = 1;
    }
    const int per = ix_div_up(nk, s);
    *kps = per;
    *split = ix_div_up(nk, per);
}

// the 256 x 256 form: one workgroup per CU -- a launch of fewer than ~256 tiles with a long K is cut along K
static void g16_plan_big(int M, int N, int K, int nbatch, int* split, int* kps) {
    const int nk = ix_div_up(K, G16_BK);
    const int64_t tiles = (int64_t)ix_div_up(M, GB_T) * ix_div_up(N, GB_T) * nbatch;
    int s = 1;
    if (tiles < 192 && nk >= 8) {
        s = (int)((256 + tiles - 1) / tiles);
        if (s > nk / 4) s = nk / 4;
        if (s > 64) s = 64;
        const int64_t plane = (int64_t)M * N * nbatch * 4;
        const int cap = (int)((24ll << 20) / (plane > 0 ? plane : 1));
        if (s > cap) s = cap;
        if (s < 1) s = 1;
    }
    const int per = ix_div_up(nk, s);
    *kps = per;
    *split = ix_div_up(nk, per);
}
// 1 (default): where g16_use_big says | 0: never | 2: for every plain contraction (tests) -- ix_gemm_b16_set_big / IX_GEMM16_BIG
static int g_g16_big = -1;
static int g16_big() {
    if (g_g16_big < 0) {
        const char* e = getenv("IX_GEMM16_BIG");
        g_g16_big = (e && e[0] == '0') ? 0 : (e && e[0] == '2') ? 2 : 1;
    }
    return g_g16_big;
}
extern "C" int ix_gemm_b16_set_big(int on) {
    const int old = g16_big();
    if (on >= 0) g_g16_big = on > 2 ? 1 : on;
    return old;
}
// plain operands whose launch in 256 x 256 tiles still gives (nearly) every CU a workgroup
static bool g16_use_big(int M, int N, int K, int nbatch, bool conv) {
    if (!g16_big() || conv) return false;
    if (g16_big() == 2) return true;
    if (M < 192 || N < 192 || K < 128) return false;
    // measured (r6o, the step's Linear shapes): the form wins 5-15 % where ITS launch is one round of workgroups cut along K (weight
    // gradients, N = 256 layers with K >= 1024) and loses 20-40 % where it takes several rounds -- one workgroup per CU leaves a tile's
    // stores and the next tile's first loads uncovered, which four 128 x 128 workgroups per CU overlap
    int split, kps;
    g16_plan_big(M, N, K, nbatch, &split, &kps);
    const int64_t tiles = (int64_t)ix_div_up(M, GB_T) * ix_div_up(N, GB_T) * nbatch;
    return tiles < 192 && tiles * split >= 200;
}

extern "C" int ix_workspace_bytes_gemm_b16(int M, int N, int K, int nbatch, size_t* out) {
    IX_CHECK_ARG(out && M >= 0 && N >= 0 && K >= 0 && nbatch >= 0, "ix_workspace_bytes_gemm_b16: bad args");
    int split = 1, kps = 1;
    if (M > 0 && N > 0 && K > 0 && nbatch > 0) {
        g16_plan(M, N, K, nbatch, &split, &kps);
        int sb = 1, kb = 1;
        g16_plan_big(M, N, K, nbatch, &sb, &kb);   // (either form may run: ix_gemm_b16_set_big)
        if (sb > split) split = sb;
    }
    // (+ M floats per plane: the partial row sums of ix_gemm_rowsum_b16 ride in the same scratch)
    *out = split > 1 ? IX_TICKET_BYTES + (size_t)split * (size_t)nbatch * ((size_t)M * (size_t)N + (size_t)M) * sizeof(float) : 0;
    return IX_OK;
}

// 1 when ix_gemm_b16 takes these operands (16-byte aligned rows everywhere, 4-element aligned C rows), else 0: the caller then
// converts and uses the fp32 entry point.
extern "C" int ix_gemm_b16_supported(const void* A, const void* B, const void* C, int M, int N, int K, int a_kcontig, int b_kcontig,
                                     int64_t lda, int64_t ldb, int64_t ldc, int64_t sAo, int64_t sAi, int64_t sBo, int64_t sBi,
                                     int64_t sCo, int64_t sCi) {
    const bool al = ix_al16(A) && ix_al16(B) && (reinterpret_cast<uintptr_t>(C) & 15) == 0 && lda % 8 == 0 && ldb % 8 == 0 &&
                    sAo % 8 == 0 && sAi % 8 == 0 && sBo % 8 == 0 && sBi % 8 == 0 && ldc % 4 == 0 && sCo % 4 == 0 && sCi % 4 == 0 && N % 4 == 0;
    if (!al || M <= 0 || N <= 0 || K <= 0) return 0;
    if ((a_kcontig || b_kcontig) && K % 8 != 0) return 0;   // a 16-byte chunk of a k-contiguous row must not straddle the K edge
    const int64_t extA = a_kcontig ? ((int64_t)(M - 1) * lda + K) : ((int64_t)(K - 1) * lda + M);
    const int64_t extB = b_kcontig ? ((int64_t)(N - 1) * ldb + K) : ((int64_t)(K - 1) * ldb + N);
    if (extA * 2 >= 0x7fffff00ll || extB * 2 >= 0x7fffff00ll) return 0;   // 32-bit DMA offsets
    return 1;
}

// 1 (default until measured otherwise) | 2 LDS stages per workgroup: ix_gemm_b16_set_stages / IX_GEMM16_STAGES (A/B runs, tests)
static int g_g16_stages = -1;
static int g16_stages() {
    if (g_g16_stages < 0) {
        const char* e = getenv("IX_GEMM16_STAGES");
        g_g16_stages = (e && e[0] == '2') ? 2 : 1;
    }
    return g_g16_stages;
}
extern "C" int ix_gemm_b16_set_stages(int stages) {
    const int old = g16_stages();
    g_g16_stages = stages == 2 ? 2 : 1;
    return old;
}

// IX_GEMM16_CST=0: the epilogue stores from the matrix instruction's layout, as before round 6's LDS pass (A/B runs)
static int g16_coalesced_stores() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("IX_GEMM16_CST"); v = (e && e[0] == '0') ? 0 : 1; }
    return v;
}

// plan + launch (+ the split-K tail) of a prepared argument block: the plain contraction and the three convolution kinds share it
static int g16_run(G16Args& a, bool a_kc, bool b_kc, bool f32, int nbatch, void* workspace, size_t workspace_bytes, hipStream_t stream,
                   const char* name) {
    const int M = a.M, N = a.N, K = a.K;
    a.nk = ix_div_up(K, G16_BK);
    const bool big = g16_use_big(M, N, K, nbatch, a.cg.mode != 0) && g16_stages() == 1;
    if (big) {
        g16_plan_big(M, N, K, nbatch, &a.split, &a.kps);
        a.tiles_m = ix_div_up(M, GB_T); a.tiles_n = ix_div_up(N, GB_T);
    } else {
        g16_plan(M, N, K, nbatch, &a.split, &a.kps);
        a.tiles_m = ix_div_up(M, G16_BM); a.tiles_n = ix_div_up(N, G16_BN);
    }
    a.planes = nullptr;
    a.rs_planes = nullptr;
    if (a.split > 1) {
        const size_t need = (size_t)a.split * (size_t)nbatch * ((size_t)M * (size_t)N + (a.rowsum ? (size_t)M : 0)) * sizeof(float);
        if (!workspace || workspace_bytes < need + IX_TICKET_BYTES) { a.split = 1; a.kps = a.nk; }   // no scratch: one pass over K
        else {
            a.planes = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(workspace) + IX_TICKET_BYTES);
            if (a.rowsum) a.rs_planes = a.planes + (size_t)a.split * (size_t)nbatch * (size_t)M * (size_t)N;
        }
    }
    IX_CHECK_ARG(nbatch <= 65535 && a.split <= 65535, "%s: too many batch slices", name);
    a.cst = (!f32 && g16_coalesced_stores() && a.ldc % 8 == 0 && a.sCo % 8 == 0 && a.sCi % 8 == 0 && ix_al16(a.C) && (!a.res || ix_al16(a.res))) ? 1 : 0;
    const dim3 grid(a.tiles_m * a.tiles_n, nbatch, a.split);
    const double bytes = 2.0 * ((double)M * K + (double)K * N) * nbatch + (f32 ? 4.0 : 2.0) * (double)M * N * nbatch;
    ix_prof_begin_b16(stream, M, N, K, nbatch, bytes);
#define G16_LAUNCH(AK, BK_, F, CV)                                                                                       \
    do {                                                                                                                 \
        if (g16_stages() == 2) hipLaunchKernelGGL((gemm16_kernel<AK, BK_, F, CV, 2>), grid, dim3(256), 0, stream, a);    \
        else hipLaunchKernelGGL((gemm16_kernel<AK, BK_, F, CV, 1>), grid, dim3(256), 0, stream, a);                      \
    } while (0)
#define G16_BIG(AK, BK_, F, R)                                                                                                     \
    do {                                                                                                                           \
        static bool attr = false;   /* (128 KB of dynamic LDS: allowed once per kernel) */                                         \
        if (!attr) {                                                                                                               \
            const hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm16_big_kernel<AK, BK_, F, R>),             \
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, 2 * GB_STAGE);                    \
            IX_CHECK_ARG(e_ == hipSuccess, "%s: 128 KB of LDS per workgroup refused: %s", name, hipGetErrorString(e_));             \
            attr = true;                                                                                                           \
        }                                                                                                                          \
        hipLaunchKernelGGL((gemm16_big_kernel<AK, BK_, F, R>), grid, dim3(512), 2 * GB_STAGE, stream, a);                           \
    } while (0)
    if (big) {
        if (a.rowsum) G16_BIG(false, false, true, true);
        else if (a_kc && b_kc) { if (f32) G16_BIG(true, true, true, false); else G16_BIG(true, true, false, false); }
        else if (a_kc && !b_kc) { if (f32) G16_BIG(true, false, true, false); else G16_BIG(true, false, false, false); }
        else if (!a_kc && b_kc) { if (f32) G16_BIG(false, true, true, false); else G16_BIG(false, true, false, false); }
        else { if (f32) G16_BIG(false, false, true, false); else G16_BIG(false, false, false, false); }
    } else if (a.rowsum) {   // (the weight gradient of a Linear with its bias gradient: dY m-contiguous, x n-contiguous, fp32 result)
        hipLaunchKernelGGL((gemm16_kernel<false, false, true, 0, 1, true>), grid, dim3(256), 0, stream, a);
    } else if (a.cg.mode == 1) {
        if (b_kc) { if (f32) G16_LAUNCH(true, true, true, 1); else G16_LAUNCH(true, true, false, 1); }
        else { if (f32) G16_LAUNCH(true, false, true, 1); else G16_LAUNCH(true, false, false, 1); }
    } else if (a.cg.mode == 2) {
        if (f32) G16_LAUNCH(false, false, true, 2); else G16_LAUNCH(false, false, false, 2);
    } else if (a_kc && b_kc) { if (f32) G16_LAUNCH(true, true, true, 0); else G16_LAUNCH(true, true, false, 0); }
    else if (a_kc && !b_kc) { if (f32) G16_LAUNCH(true, false, true, 0); else G16_LAUNCH(true, false, false, 0); }
    else if (!a_kc && b_kc) { if (f32) G16_LAUNCH(false, true, true, 0); else G16_LAUNCH(false, true, false, 0); }
    else { if (f32) G16_LAUNCH(false, false, true, 0); else G16_LAUNCH(false, false, false, 0); }
#undef G16_LAUNCH
#undef G16_BIG
    if (a.planes) {
        const int64_t work = (int64_t)nbatch * M * (N / 4);
        const int g = ix_grid_1d(work, 256);
        if (f32) hipLaunchKernelGGL(gemm16_reduce_kernel<true>, dim3(g), dim3(256), 0, stream, a, nbatch);
        else hipLaunchKernelGGL(gemm16_reduce_kernel<false>, dim3(g), dim3(256), 0, stream, a, nbatch);
    }
    ix_prof_end(stream);
    IX_CHECK_LAUNCH(name);
    return IX_OK;
}

extern "C" int ix_gemm_b16(const void* A, const void* B, void* C, const float* bias, int M, int N, int K, int a_kcontig, int b_kcontig,
                           int64_t lda, int64_t ldb, int64_t ldc, int batch_outer, int batch_inner, int64_t sAo, int64_t sAi,
                           int64_t sBo, int64_t sBi, int64_t sCo, int64_t sCi, int64_t bias_stride, float alpha, int c_f32,
                           const float* scale, const float* shift, const void* residual, int act, void* workspace,
                           size_t workspace_bytes, hipStream_t stream) {
    if (M <= 0 || N <= 0 || batch_outer <= 0 || batch_inner <= 0) return IX_OK;
    IX_CHECK_ARG(A && B && C && K > 0, "ix_gemm_b16: null operand or K <= 0");
    IX_CHECK_ARG(ix_gemm_b16_supported(A, B, C, M, N, K, a_kcontig, b_kcontig, lda, ldb, ldc, sAo, sAi, sBo, sBi, sCo, sCi),
                 "ix_gemm_b16: operands must have 16-byte aligned rows (ld, offsets, strides multiples of 8 elements; K %% 8 == 0 for "
                 "k-contiguous operands; N, ldc multiples of 4) and fit 2 GB per batch slice");
    IX_CHECK_ARG((scale == nullptr) == (shift == nullptr), "ix_gemm_b16: scale and shift come together");
    IX_CHECK_ARG((!bias || ix_al16(bias)) && (!scale || (ix_al16(scale) && ix_al16(shift))) && bias_stride % 4 == 0,
                 "ix_gemm_b16: bias / scale / shift must be 16-byte aligned");
    IX_CHECK_ARG(act >= 0 && act <= 2, "ix_gemm_b16: act must be 0 (none), 1 (ReLU) or 2 (GELU)");
    const int nbatch = batch_outer * batch_inner;
    G16Args a;
    a.A = (const unsigned short*)A; a.B = (const unsigned short*)B; a.C = C; a.bias = bias; a.scale = scale; a.shift = shift; a.res = residual;
    a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.sAo = sAo; a.sAi = sAi; a.sBo = sBo; a.sBi = sBi; a.sCo = sCo; a.sCi = sCi; a.sBias = bias_stride;
    a.extA = (unsigned)(2 * (a_kcontig ? ((int64_t)(M - 1) * lda + K) : ((int64_t)(K - 1) * lda + M)));
    a.extB = (unsigned)(2 * (b_kcontig ? ((int64_t)(N - 1) * ldb + K) : ((int64_t)(K - 1) * ldb + N)));
    a.M = M; a.N = N; a.K = K; a.batch_inner = batch_inner;
    a.alpha = alpha; a.act = act;
    a.cg.mode = 0; a.cg.btap = 0;
    a.rowsum = nullptr; a.sRowsum = 0;
    return g16_run(a, a_kcontig != 0, b_kcontig != 0, c_f32 != 0, nbatch, workspace, workspace_bytes, stream, "ix_gemm_b16");
}

// C[b] = alpha A[b] B[b] (fp32) AND rowsum[b][m] = sum_k A(m, k) for an m-contiguous A and an n-contiguous B -- the weight gradient of a
// Linear layer, dW = dY^T x, with its bias gradient colsum(dY) riding on it (the fp32 twin: ix_gemm_rowsum_f32)
extern "C" int ix_gemm_rowsum_b16(const void* A, const void* B, float* C, float* rowsum, int M, int N, int K, int64_t lda, int64_t ldb,
                                  int64_t ldc, int batch_outer, int64_t sAo, int64_t sBo, int64_t sCo, int64_t rowsum_stride, float alpha,
                                  void* workspace, size_t workspace_bytes, hipStream_t stream) {
    if (M <= 0 || N <= 0 || batch_outer <= 0) return IX_OK;
    IX_CHECK_ARG(A && B && C && rowsum && K > 0, "ix_gemm_rowsum_b16: null operand or K <= 0");
    IX_CHECK_ARG(ix_gemm_b16_supported(A, B, C, M, N, K, 0, 0, lda, ldb, ldc, sAo, 0, sBo, 0, sCo, 0), "ix_gemm_rowsum_b16: operands must have 16-byte aligned rows");
    G16Args a;
    a.A = (const unsigned short*)A; a.B = (const unsigned short*)B; a.C = C; a.bias = nullptr; a.scale = nullptr; a.shift = nullptr; a.res = nullptr;
    a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.sAo = sAo; a.sAi = 0; a.sBo = sBo; a.sBi = 0; a.sCo = sCo; a.sCi = 0; a.sBias = 0;
    a.extA = (unsigned)(2 * ((int64_t)(K - 1) * lda + M));
    a.extB = (unsigned)(2 * ((int64_t)(K - 1) * ldb + N));
    a.M = M; a.N = N; a.K = K; a.batch_inner = 1; a.alpha = alpha; a.act = 0;
    a.cg.mode = 0; a.cg.btap = 0;
    a.rowsum = rowsum; a.sRowsum = rowsum_stride;
    return g16_run(a, false, false, true, batch_outer, workspace, workspace_bytes, stream, "ix_gemm_rowsum_b16");
}

// ---- implicit-GEMM convolution on bf16 NHWC activations (the three kinds of ix_conv_gemm_f32; csrc/gemm.hip has the fp32 twin) ------
//   kind 0  y[g][img][oy][ox][co]  = sum_{ky,kx,c} x[g][img][oy*s-p+ky*d][ox*s-p+kx*d][c] * w[g][co][ky][kx][c]     (+ epilogue)
//   kind 1  dx[g][img][y][x][c]    = sum_{ky,kx,co} dy[g][img][(y+p-ky*d)/s][(x+p-kx*d)/s][co] * w[g][co][ky][kx][c]
//   kind 2  dw[g][co][ky][kx][c]   = sum_{img,oy,ox} dy[g][img][oy][ox][co] * x[g][img][oy*s-p+ky*d][ox*s-p+kx*d][c]
// `groups` = episodes with their own (fast) weights; shared weights: groups = 1 and imgs = all images.
static bool g16_conv_ok(int kind, int groups, int imgs, int H, int W, int Cin, int OH, int OW, int Cout, int KH, int KW, int stride,
                        int pad, int dil) {
    if (groups <= 0 || imgs <= 0 || H <= 0 || W <= 0 || KH <= 0 || KW <= 0 || stride <= 0 || dil <= 0 || pad < 0) return false;
    if (stride != 1 && stride != 2 && stride != 4) return false;
    if (Cin % 8 || Cout % 8) return false;
    if (kind == 0 && Cin % 64) return false;          // a K step inside one tap
    if (kind == 1 && Cout % 64) return false;
    if (kind == 2 && Cin % 128) return false;         // an N tile inside one tap
    const int64_t px_in = (int64_t)imgs * H * W, px_out = (int64_t)imgs * OH * OW;
    if (px_in * Cin * 2 >= 0x7fffff00ll || px_out * Cout * 2 >= 0x7fffff00ll || (int64_t)Cout * KH * KW * Cin * 2 >= 0x7fffff00ll) return false;
    if (px_in >= (1ll << 30) || px_out >= (1ll << 30)) return false;
    return true;
}
extern "C" int ix_conv_gemm_b16_supported(int kind, int groups, int imgs, int H, int W, int Cin, int OH, int OW, int Cout, int KH, int KW,
                                          int stride, int pad, int dil) {
    return (kind >= 0 && kind <= 2 && g16_conv_ok(kind, groups, imgs, H, W, Cin, OH, OW, Cout, KH, KW, stride, pad, dil)) ? 1 : 0;
}
static void g16_conv_dims(int kind, int imgs, int H, int W, int Cin, int OH, int OW, int Cout, int T, int* M, int* N, int* K) {
    if (kind == 0) { *M = imgs * OH * OW; *N = Cout; *K = T * Cin; }
    else if (kind == 1) { *M = imgs * H * W; *N = Cin; *K = T * Cout; }
    else { *M = Cout; *N = T * Cin; *K = imgs * OH * OW; }
}
extern "C" int ix_workspace_bytes_conv_gemm_b16(int kind, int groups, int imgs, int H, int W, int Cin, int OH, int OW, int Cout, int KH,
                                                int KW, size_t* out) {
    IX_CHECK_ARG(out && kind >= 0 && kind <= 2, "ix_workspace_bytes_conv_gemm_b16: bad args");
    int M, N, K;
    g16_conv_dims(kind, imgs, H, W, Cin, OH, OW, Cout, KH * KW, &M, &N, &K);
    return ix_workspace_bytes_gemm_b16(M, N, K, groups, out);
}
extern "C" int ix_conv_gemm_b16(int kind, const void* src, const void* other, void* out, int groups, int imgs, int H, int W, int Cin,
                                int OH, int OW, int Cout, int KH, int KW, int stride, int pad, int dil, int c_f32, const float* scale,
                                const float* shift, const void* residual, int relu, void* workspace, size_t workspace_bytes,
                                hipStream_t stream) {
    IX_CHECK_ARG(kind >= 0 && kind <= 2 && src && other && out, "ix_conv_gemm_b16: bad kind or null pointer");
    IX_CHECK_ARG(g16_conv_ok(kind, groups, imgs, H, W, Cin, OH, OW, Cout, KH, KW, stride, pad, dil),
                 "ix_conv_gemm_b16: geometry not supported (ix_conv_gemm_b16_supported)");
    IX_CHECK_ARG(ix_al16(src) && ix_al16(other) && ix_al16(out), "ix_conv_gemm_b16: 16-byte aligned tensors needed");
    IX_CHECK_ARG((scale == nullptr) == (shift == nullptr) && (kind == 0 || (!scale && !residual && !relu)),
                 "ix_conv_gemm_b16: the affine / residual / ReLU epilogue belongs to the forward kind");
    const int T = KH * KW;
    G16Args a;
    int M, N, K;
    g16_conv_dims(kind, imgs, H, W, Cin, OH, OW, Cout, T, &M, &N, &K);
    a.C = out; a.bias = nullptr; a.scale = scale; a.shift = shift; a.res = residual;
    a.M = M; a.N = N; a.K = K; a.batch_inner = 1; a.alpha = 1.f; a.act = relu ? 1 : 0; a.sBias = 0;
    a.sAi = a.sBi = a.sCi = 0;
    a.rowsum = nullptr; a.sRowsum = 0;
    a.ldc = N; a.sCo = (int64_t)M * N;
    G16Conv& g = a.cg;
    g.btap = 0; g.bcol = 0; g.KW = KW;
    int qs = 0;
    while ((1 << qs) < stride) ++qs;
    const int64_t x_slice = (int64_t)imgs * H * W * Cin, y_slice = (int64_t)imgs * OH * OW * Cout, w_slice = (int64_t)Cout * T * Cin;
    bool a_kc, b_kc;
    if (kind == 0) {          // A = x gathered over output pixels, B = w [Cout][T * Cin]
        a.A = (const unsigned short*)src; a.B = (const unsigned short*)other;
        a.lda = Cin; a.ldb = (int64_t)T * Cin; a.sAo = x_slice; a.sBo = w_slice;
        a.extA = (unsigned)(2 * x_slice); a.extB = (unsigned)(2 * w_slice);
        g.mode = 1; g.gH = OH; g.gW = OW; g.sH = H; g.sW = W; g.sC = Cin; g.a = stride; g.b = -pad; g.d = dil; g.qs = 0;
        a_kc = true; b_kc = true;
    } else if (kind == 1) {   // A = dy gathered over input pixels (flipped taps), B(k = (tap, co), n = c) = w[co][tap][c]
        a.A = (const unsigned short*)src; a.B = (const unsigned short*)other;
        a.lda = Cout; a.ldb = (int64_t)T * Cin; a.sAo = y_slice; a.sBo = w_slice;
        a.extA = (unsigned)(2 * y_slice); a.extB = (unsigned)(2 * w_slice);
        g.mode = 1; g.gH = H; g.gW = W; g.sH = OH; g.sW = OW; g.sC = Cout; g.a = 1; g.b = pad; g.d = -dil; g.qs = qs;
        g.btap = Cout; g.bcol = Cin;
        a_kc = true; b_kc = false;
    } else {                  // A = dy^T (m = co contiguous), B = x gathered over output pixels, n = (tap, c)
        a.A = (const unsigned short*)src; a.B = (const unsigned short*)other;
        a.lda = Cout; a.ldb = Cin; a.sAo = y_slice; a.sBo = x_slice;
        a.extA = (unsigned)(2 * y_slice); a.extB = (unsigned)(2 * x_slice);
        g.mode = 2; g.gH = OH; g.gW = OW; g.sH = H; g.sW = W; g.sC = Cin; g.a = stride; g.b = -pad; g.d = dil; g.qs = 0;
        a_kc = false; b_kc = false;
    }
    g.dW = g16_make_div((unsigned)g.gW); g.dHW = g16_make_div((unsigned)(g.gH * g.gW)); g.dC = g16_make_div((unsigned)g.sC);
    return g16_run(a, a_kc, b_kc, c_f32 != 0, groups, workspace, workspace_bytes, stream, "ix_conv_gemm_b16");
}

// ---- dtype conversion passes (HBM-bound: 6 bytes per element) -----------------------------------------------------------------
__global__ __launch_bounds__(256) void cast_f32_b16_kernel(const float* __restrict__ x, unsigned short* __restrict__ y, int64_t n8, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        const f32x4 a = reinterpret_cast<const f32x4*>(x)[2 * i], b = reinterpret_cast<const f32x4*>(x)[2 * i + 1];
        reinterpret_cast<u32x4*>(y)[i] = u32x4{g16_pack2(a.x, a.y), g16_pack2(a.z, a.w), g16_pack2(b.x, b.y), g16_pack2(b.z, b.w)};
    }
    if (blockIdx.x == 0)
        for (int64_t i = n8 * 8 + threadIdx.x; i < n; i += 256) y[i] = (unsigned short)(g16_pack2(x[i], 0.f) & 0xffffu);
}
__global__ __launch_bounds__(256) void cast_b16_f32_kernel(const unsigned short* __restrict__ x, float* __restrict__ y, int64_t n8, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        const u32x4 v = reinterpret_cast<const u32x4*>(x)[i];
        reinterpret_cast<f32x4*>(y)[2 * i] = f32x4{__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u),
                                                  __uint_as_float(v.y << 16), __uint_as_float(v.y & 0xffff0000u)};
        reinterpret_cast<f32x4*>(y)[2 * i + 1] = f32x4{__uint_as_float(v.z << 16), __uint_as_float(v.z & 0xffff0000u),
                                                      __uint_as_float(v.w << 16), __uint_as_float(v.w & 0xffff0000u)};
    }
    if (blockIdx.x == 0)
        for (int64_t i = n8 * 8 + threadIdx.x; i < n; i += 256) y[i] = g16_bf(x[i]);
}

extern "C" int ix_cast_f32_b16(const float* x, void* y, int64_t n, hipStream_t stream) {
    if (n <= 0) return IX_OK;
    IX_CHECK_ARG(x && y && ix_al16(x) && ix_al16(y), "ix_cast_f32_b16: null or unaligned pointer");
    hipLaunchKernelGGL(cast_f32_b16_kernel, dim3(ix_grid_1d(n / 8 + 1, 256)), dim3(256), 0, stream, x, (unsigned short*)y, n / 8, n);
    IX_CHECK_LAUNCH("ix_cast_f32_b16");
    return IX_OK;
}
extern "C" int ix_cast_b16_f32(const void* x, float* y, int64_t n, hipStream_t stream) {
    if (n <= 0) return IX_OK;
    IX_CHECK_ARG(x && y && ix_al16(x) && ix_al16(y), "ix_cast_b16_f32: null or unaligned pointer");
    hipLaunchKernelGGL(cast_b16_f32_kernel, dim3(ix_grid_1d(n / 8 + 1, 256)), dim3(256), 0, stream, (const unsigned short*)x, y, n / 8, n);
    IX_CHECK_LAUNCH("ix_cast_b16_f32");
    return IX_OK;
}
