// Batched strided fp32 GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32: f32 in, f32 accumulate,
// bit-exact k-ordered fmaf chain), LDS-staged and double-buffered.
//
//   C[b](M x N, row-major, ldc) = alpha * A[b](M x K) * B[b](K x N) (+ bias[n])
//
// Every contraction on the Interactron hot path is routed here: the Linear layers of the DETR
// encoder/decoder and of the GPT fusion (reference models/detr_models/transformer.py:148-232,
// models/gpt.py:39-78), the ResNet-50 convolutions (reference models/detr_models/backbone.py:88-90) as implicit
// GEMMs (ConvGather / ix_conv_gemm_f32 below: the producer waves gather the NHWC taps), and all of their first-
// and second-order derivatives, which are again contractions of this form with the operand layouts flipped.
// (The attention products live in flash.hip.)  File layout: the exact-fp32 MFMA kernel first (this header describes
// it), then the bf16x6 kernel family (the dominant kernel: "fp32 GEMM on the bf16 matrix cores" further down), the
// host-side choice between them (gemm_impl), the ride-alongs (row sums, convolution gathers) and the profiling hooks.
//
// Operand layouts (so that no transposed copy is ever materialised):
//   A_KC : A(m,k) = A[m*lda + k]   else A(m,k) = A[k*lda + m]
//   B_KC : B(k,n) = B[n*ldb + k]   else B(k,n) = B[k*ldb + n]
// Two-level batch index b = bo*batch_inner + bi with independent (outer, inner) element strides per operand:
// that is how a [n, T, heads*hd] activation is consumed per (frame, head) without a permute.
//
// Tiling: 256 threads = 4 waves (2 x 2); block tile BM x BN x BK in {128x128x32, 64x64x64}; each wave owns a
// (BM/2 x BN/2) sub-tile as TM x TN accumulators of 32x32.  LDS holds the tiles k-major ([k][m], [k][n]) so an
// MFMA operand fetch is one conflict-free ds_read_b32 per lane (lane l reads row l>>5, column l&31).  One K step
// is >= 2048 MFMA cycles per wave, i.e. longer than an HBM round trip, so the register-staged prefetch of the next
// K tile (issued before the MFMA block, written to the other LDS buffer after it) is fully hidden; two workgroups
// per CU overlap each other's barriers.
// Global loads are 16-byte vectors along the contiguous dimension when alignment allows, scalar otherwise.
// Split-K (atomic f32 accumulation into a zeroed C) fills the chip for the weight-gradient shapes
// (small M x N, K = tokens).
#include "common.h"
#include "gemm_x3.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

// Division by an invariant positive integer d:  n / d = (mulhi(n, m) + n) >> s  for 0 <= n < 2^31 (host: make_fastdiv).
struct FastDiv {
    unsigned m;
    int s;
};
__device__ __forceinline__ int fd_div(int n, FastDiv f) { return (int)((__umulhi((unsigned)n, f.m) + (unsigned)n) >> f.s); }
static inline FastDiv make_fastdiv(unsigned d) {
    FastDiv f;
    f.s = 0;
    while ((1ull << f.s) < d) ++f.s;
    f.m = (unsigned)((((1ull << 32) * ((1ull << f.s) - d)) / d) + 1);
    return f;
}

// Implicit-GEMM convolution on the bf16x6 kernel: instead of a patch matrix in HBM (im2col) the producer waves compute,
// per 16-byte load, where the element lives in the NHWC tensor.  A "pixel row" index decomposes over a grid
//     row = (img * gH + gy) * gW + gx
// and tap (ky, kx) of that pixel reads the source tensor [img][sH][sW][sC] at
//     sy = (gy * a + b + ky * d) >> qs,  sx = (gx * a + bx + kx * d) >> qs     (valid iff both divisible by 1 << qs and inside)
// -- forward / weight gradient: grid = output pixels, a = stride, b = -pad, d = dilation, qs = 0;  data gradient: grid =
// input pixels, source = dY, a = 1, b = pad, d = -dilation, qs = log2(stride).  Invalid taps are requested out of range
// (the buffer load returns zeros without touching memory).
//   mode_a 1: A rows are pixels, k = (tap, c) -- a K tile of 32 lies inside one tap (sC % 32 == 0)
//   mode_b 2: B rows k are pixels, n = (tap, c) -- an N tile lies inside one tap (sC % BN == 0)        (weight gradient)
//   mode_b 3: B rows k = (tap, co) live at W[co][tap][c]: offset (k % bmod) * ldb + (k / bmod) * btap     (data gradient)
struct ConvGather {
    int mode_a, mode_b;
    int gH, gW, sH, sW, sC;
    int a, b, bx, d, qs, KW;   // (b: row offset, bx: column offset -- equal except in the parity classes of conv_bwd_data_s2)
    int bmod, btap;
    // C row map of a parity-class launch of conv_bwd_data_s2 (cmap != 0; forward-kind kernel, fp16x3 form, no split-K): output
    // row m = (img, i, j) of the class grid gH x gW is pixel (2 i + cpy, 2 j + cpx) of the cH x cW image -- the class writes its
    // pixels of dx in place
    int cmap, cH, cW, cpy, cpx;
    FastDiv dW, dHW, dC, dKW, dBmod;   // divisions by gW, gH * gW, sC, KW, bmod
};

struct GemmArgs {
    const float* A;
    const float* B;
    float* C;
    const float* bias;
    int M, N, K;
    int64_t lda, ldb, ldc;
    int64_t sAo, sAi, sBo, sBi, sCo, sCi;
    int64_t sBias;
    int64_t extA, extB;   // elements addressable from one batch slice's base pointer (buffer bounds, bf16x6 kernel)
    int batch_inner;
    int split_k;
    int k_per_split;
    int atomic;        // split-K partial sums: 1 = fp32 atomics onto a zero-filled C (legacy, order-dependent rounding), 0 = every
    int64_t sSplit;    // split writes its own plane C + ks * sSplit (a workspace) and splitk_reduce_kernel sums them in order
    float alpha;
    int a_vec, b_vec, c_vec;
    int tiles_m, tiles_n;
    struct ConvGather cg;   // implicit-GEMM convolution operands (ix_conv_gemm_f32); mode 0 for plain contractions
    // affine epilogue of an UNSPLIT launch in the store of the fp16x3 kernel's EPI instances (frozen BN + identity + ReLU riding on a
    // forward convolution): C = [relu](acc * scale[n] + shift[n] (+ res[same place as C]))
    const float *epi_scale, *epi_shift, *epi_res;
    int epi_relu;
    float* rowsum;          // optional side output (bf16x6 kernel, A stored m-contiguous): rowsum[bo][m] = sum_k A(m, k)
    int64_t sRowsum;
    int64_t sSplitRowsum;   // plane stride of the per-split partial row sums (0 with atomics)
    // item decoding of the persistent kernels (x6_item) without integer divisions: by tiles_m * tiles_n * split_k, by
    // tiles_m * tiles_n, by 8 * tiles_n, by batch_inner (set_item_divs, host)
    FastDiv fd_per_batch, fd_nt, fd_group, fd_bi;
};
static void set_item_divs(GemmArgs& a) {
    const unsigned nt = (unsigned)a.tiles_m * (unsigned)a.tiles_n;
    a.fd_per_batch = make_fastdiv(nt * (unsigned)a.split_k);
    a.fd_nt = make_fastdiv(nt);
    a.fd_group = make_fastdiv(8u * (unsigned)a.tiles_n);
    a.fd_bi = make_fastdiv((unsigned)(a.batch_inner > 0 ? a.batch_inner : 1));
}

// XCD-aware, bijective remap: consecutive logical tiles land on the same XCD (same L2).
__device__ __forceinline__ int xcd_swizzle(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// LDS row pitch (floats) of a k-major [BK][BT] operand image.  Operands whose k is contiguous in HBM are written
// transposed with ds_write_b32: an odd pitch keeps that at <= 2-way bank conflicts (free for writes); operands whose
// m/n is contiguous are written as float4 and need a 16-byte aligned pitch.
template <int BT, bool KC>
struct Pitch {
    static constexpr int value = KC ? BT + 1 : BT + 4;
};

template <int BT, int BK, bool KC>
struct TileLoader {
    // BT x BK tile of an operand; KC = k is the contiguous dimension in global memory.
    static constexpr int NV = BT * BK / 4 / 256;  // float4 per thread
    static constexpr int P = Pitch<BT, KC>::value;
    static constexpr int CPR = BK / 4;            // KC: float4 chunks per tile row
    static constexpr int Q = BT / 4;              // !KC: float4 chunks per k-row
    float4 v[NV];

    __device__ __forceinline__ void coords(int i, int& tr, int& kr) const {
        const int tid = threadIdx.x;
        if (KC) {
            tr = tid / CPR + (256 / CPR) * i;
            kr = (tid % CPR) * 4;
        } else {
            tr = (tid % Q) * 4;
            kr = tid / Q + (256 / Q) * i;
        }
    }

    __device__ __forceinline__ void load(const float* __restrict__ base, int64_t ld, int t0, int k0, int tmax, int kmax,
                                         bool vec) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            int tr, kr;  // tile row (m or n) / k of the first of 4 contiguous elements
            coords(i, tr, kr);
            const int gt = t0 + tr, gk = k0 + kr;
            float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
            if (KC) {
                if (gt < tmax) {
                    const float* p = base + (int64_t)gt * ld + gk;
                    if (vec && gk + 3 < kmax) {
                        x = *reinterpret_cast<const float4*>(p);
                    } else {
                        if (gk + 0 < kmax) x.x = p[0];
                        if (gk + 1 < kmax) x.y = p[1];
                        if (gk + 2 < kmax) x.z = p[2];
                        if (gk + 3 < kmax) x.w = p[3];
                    }
                }
            } else {
                if (gk < kmax) {
                    const float* p = base + (int64_t)gk * ld + gt;
                    if (vec && gt + 3 < tmax) {
                        x = *reinterpret_cast<const float4*>(p);
                    } else {
                        if (gt + 0 < tmax) x.x = p[0];
                        if (gt + 1 < tmax) x.y = p[1];
                        if (gt + 2 < tmax) x.z = p[2];
                        if (gt + 3 < tmax) x.w = p[3];
                    }
                }
            }
            v[i] = x;
        }
    }

    // LDS image is k-major: s[k][t], row pitch P floats.
    __device__ __forceinline__ void store(float* __restrict__ s) const {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            int tr, kr;
            coords(i, tr, kr);
            if (KC) {
                s[(kr + 0) * P + tr] = v[i].x;
                s[(kr + 1) * P + tr] = v[i].y;
                s[(kr + 2) * P + tr] = v[i].z;
                s[(kr + 3) * P + tr] = v[i].w;
            } else {
                // two 8-byte stores, not one ds_write_b128: a 16-byte store wants an aligned VGPR quad, and hipcc then
                // recycles half of the in-flight load destinations as address temporaries, which drags the
                // s_waitcnt vmcnt(0) of the prefetch in FRONT of the MFMA block (measured: -20 % on [K][N] operands)
                // (volatile: hipcc would fuse the pair back into one ds_write_b128)
                typedef float f32x2 __attribute__((ext_vector_type(2)));
                f32x2 lo, hi;
                lo.x = v[i].x; lo.y = v[i].y; hi.x = v[i].z; hi.y = v[i].w;
                typedef __attribute__((address_space(3))) volatile f32x2 lds_f32x2;
                *(lds_f32x2*)(&s[kr * P + tr]) = lo;
                *(lds_f32x2*)(&s[kr * P + tr + 2]) = hi;
            }
        }
    }
};

// 2 blocks per CU (launch bound 2 waves / SIMD): the second block's MFMAs cover this block's barrier + LDS refill.
template <int BM, int BN, int BK, bool A_KC, bool B_KC>
__global__ __launch_bounds__(256, 2) void gemm_f32_mfma_kernel(GemmArgs p) {
    constexpr int PA = Pitch<BM, A_KC>::value, PB = Pitch<BN, B_KC>::value;
    constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
    __shared__ __attribute__((aligned(16))) float As[2][BK * PA];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK * PB];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nwg = p.tiles_m * p.tiles_n;
    const int tile = xcd_swizzle(blockIdx.x, nwg);
    // Grouped tile order: the ~64 workgroups resident on one XCD cover an (up to) 8 x 8 patch of output tiles instead
    // of a 1 x 64 strip, so per K step they pull 16 operand tiles through that XCD's L2 instead of 65.
    constexpr int GROUP_M = 8;
    const int group_size = GROUP_M * p.tiles_n;
    const int first_m = (tile / group_size) * GROUP_M;
    const int gm = min(p.tiles_m - first_m, GROUP_M);
    const int m0 = (first_m + (tile % group_size) % gm) * BM, n0 = ((tile % group_size) / gm) * BN;
    const int zb = blockIdx.y;            // batch
    const int ks = blockIdx.z;            // k split
    const int bo = zb / p.batch_inner, bi = zb % p.batch_inner;
    const float* A = p.A + bo * p.sAo + bi * p.sAi;
    const float* B = p.B + bo * p.sBo + bi * p.sBi;
    float* C = p.C + bo * p.sCo + bi * p.sCi + ks * p.sSplit;
    const float* bias = p.bias ? p.bias + bo * p.sBias : nullptr;
    const int kbeg = ks * p.k_per_split;
    const int kend = min(p.K, kbeg + p.k_per_split);
    const int nk = (kend - kbeg + BK - 1) / BK;

    TileLoader<BM, BK, A_KC> la;
    TileLoader<BN, BK, B_KC> lb;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int wm = (wave >> 1) * WM, wn = (wave & 1) * WN;
    const int lrow = lane >> 5, lcol = lane & 31;

    if (nk > 0) {
        la.load(A, p.lda, m0, kbeg, p.M, kend, p.a_vec);
        lb.load(B, p.ldb, n0, kbeg, p.N, kend, p.b_vec);
        la.store(As[0]);
        lb.store(Bs[0]);
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) {
            la.load(A, p.lda, m0, kbeg + (kt + 1) * BK, p.M, kend, p.a_vec);
            lb.load(B, p.ldb, n0, kbeg + (kt + 1) * BK, p.N, kend, p.b_vec);
        }
        const float* as = As[cur];
        const float* bs = Bs[cur];
        // operand fragments are fetched one k-pair ahead of the MFMAs that consume them (separate registers), so the
        // ds_read latency hides behind the 64-cycle matrix instructions instead of stalling every 4 of them
        float a[2][TM], b[2][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[0][i] = as[lrow * PA + wm + i * 32 + lcol];
#pragma unroll
        for (int j = 0; j < TN; ++j) b[0][j] = bs[lrow * PB + wn + j * 32 + lcol];
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            const int c = kk & 1, n = c ^ 1;
            if (kk + 1 < BK / 2) {
                const int kr = (kk + 1) * 2 + lrow;
#pragma unroll
                for (int i = 0; i < TM; ++i) a[n][i] = as[kr * PA + wm + i * 32 + lcol];
#pragma unroll
                for (int j = 0; j < TN; ++j) b[n][j] = bs[kr * PB + wn + j * 32 + lcol];
            }
            __builtin_amdgcn_sched_barrier(0);   // keep the prefetch ahead of the MFMAs (hipcc sinks it otherwise)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c][i], b[c][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (kt + 1 < nk) {
            la.store(As[cur ^ 1]);
            lb.store(Bs[cur ^ 1]);
        }
        __syncthreads();
    }

    const bool add_bias = bias != nullptr && ks == 0;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn + j * 32 + lcol;
            if (col >= p.N) continue;
            const float bv = add_bias ? bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lrow;
                if (row < p.M) {
                    const float v = p.alpha * acc[i][j][r] + bv;
                    float* dst = C + (int64_t)row * p.ldc + col;
                    if (p.atomic)
                        unsafeAtomicAdd(dst, v);
                    else
                        *dst = v;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// fp32 GEMM on the bf16 matrix cores ("bf16x6"): every fp32 operand element is split EXACTLY into three bf16 values
//     x = h + m + l,   h = bf16(x), m = bf16(x - h), l = bf16(x - h - m)        (3 x 8 = 24 significant bits)
// on its way from the staging registers into LDS (three bf16 planes per operand), and a k-slice of 16 is the six
// products  h.h + (h.m + m.h) + (h.l + m.m + l.h)  on v_mfma_f32_32x32x16_bf16 with fp32 accumulation; the dropped
// terms are below 2^-24 relative.  That is fp32-grade accuracy (fewer roundings than the k-ordered fmaf chain of
// v_mfma_f32_32x32x2_f32: one per 16 products instead of one per product) at 6/16 of the fp32-MFMA cycle cost.
// LDS image: [plane][row][k] with 80-byte rows (32 bf16 + 16 B pad), so an operand fragment is one conflict-free
// ds_read_b128 per lane (lane l: row l&31, k = 8*(l>>5) .. +8).
// ------------------------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));

constexpr int X6_BT = 128, X6_BK = 32, X6_ROWB = 80, X6_PLANE = X6_BT * X6_ROWB;

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {   // v_cvt_pk_bf16_f32 (RNE), a in the low half
    f32x2v v;
    v.x = a;
    v.y = b;
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

// (x0, x1) -> packed bf16 pairs of the three planes
__device__ __forceinline__ void split3(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
#ifdef X6_DIAG_NOCONV   // diagnostic build: no conversion arithmetic (wrong numbers), isolates the producers' VALU cost
    h = __float_as_uint(x0);
    m = __float_as_uint(x1);
    l = h ^ m;
    return;
#endif
    h = pack_bf16(x0, x1);
    const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);
    m = pack_bf16(r0, r1);
    const float s0 = r0 - __uint_as_float(m << 16), s1 = r1 - __uint_as_float(m & 0xffff0000u);
    l = pack_bf16(s0, s1);
}

// KC operands: which of the 32 tile rows of a pass thread group g = tid >> 3 handles.  Consecutive groups sit 4 rows
// apart (row = 4*(g&1) + ((g>>1)&3) + (g & 24)): a 16-lane ds_write_b64 group then covers rows r and r+4, whose 64-byte
// chunks fall into disjoint halves of the 128-byte bank window (80-byte rows: 4*80 mod 128 = 64) instead of
// overlapping by 16 bytes (2-way conflict on every store).  Global loads are unaffected (8 rows x 128 B either way).
__device__ __forceinline__ int x6_kc_row(int tid) {
    const int g = tid >> 3;
    return ((g & 1) << 2) | ((g >> 1) & 3) | (g & 24);
}

// Wait until the oldest ring stage (registers a[0..3], b[0..NB-1]) has landed: at most `younger` * L of this wave's
// (inline-asm) buffer loads may stay outstanding, where L = loads per stage and `younger` = how many younger stages
// were actually requested.  The stage's registers are in/out operands of the wait, so that every use of the data
// depends on it -- otherwise hipcc is free to hoist the conversion arithmetic above the wait (it does not know the
// registers are still being written by the memory system).
typedef float x6_f32x4 __attribute__((ext_vector_type(4)));
// (macros, not a function taking pointers: the stage registers must never have their address taken, or they are
// demoted to scratch memory and copied there right behind the asm load, i.e. before the data has arrived)
#define X6_WAIT_ASM(SA, SB, CNT)                                                                                      \
    if (NB_ == 4)                                                                                                     \
        asm volatile("s_waitcnt vmcnt(" #CNT ")" : "+v"(SA.v[0]), "+v"(SA.v[1]), "+v"(SA.v[2]), "+v"(SA.v[3]),        \
                     "+v"(SB.v[0]), "+v"(SB.v[NB_ > 1 ? 1 : 0]), "+v"(SB.v[NB_ > 2 ? 2 : 0]), "+v"(SB.v[NB_ > 3 ? 3 : 0])::"memory"); \
    else if (NB_ == 2)                                                                                                \
        asm volatile("s_waitcnt vmcnt(" #CNT ")" : "+v"(SA.v[0]), "+v"(SA.v[1]), "+v"(SA.v[2]), "+v"(SA.v[3]),        \
                     "+v"(SB.v[0]), "+v"(SB.v[NB_ > 1 ? 1 : 0])::"memory");                                           \
    else                                                                                                              \
        asm volatile("s_waitcnt vmcnt(" #CNT ")" : "+v"(SA.v[0]), "+v"(SA.v[1]), "+v"(SA.v[2]), "+v"(SA.v[3]),        \
                     "+v"(SB.v[0])::"memory");
#define X6_WAIT_STAGE(SA, SB, YOUNGER)                                                         \
    {                                                                                          \
        const int y_ = (YOUNGER);                                                              \
        if (y_ >= 2) {                                                                         \
            if (NB_ == 4) { X6_WAIT_ASM(SA, SB, 16) } else if (NB_ == 2) { X6_WAIT_ASM(SA, SB, 12) } else { X6_WAIT_ASM(SA, SB, 10) } \
        } else if (y_ == 1) {                                                                  \
            if (NB_ == 4) { X6_WAIT_ASM(SA, SB, 8) } else if (NB_ == 2) { X6_WAIT_ASM(SA, SB, 6) } else { X6_WAIT_ASM(SA, SB, 5) } \
        } else {                                                                               \
            X6_WAIT_ASM(SA, SB, 0)                                                             \
        }                                                                                      \
    }

// SWZ: compact LDS image -- 64-byte rows (no padding) whose four 16-byte chunks are XOR-swizzled with bits 2-3 of the
// row, which keeps the consumers' ds_read_b128 fragment reads conflict-free (a 16-lane read group covers rows with all
// 16 combinations of row & 3 and (row >> 2) & 3) and frees 24 KB of LDS for full-size C strips (deferred C stores).
// X3 (fp16x3 form of the kernel, see "fp16x3" below): the tile goes to LDS as TWO fp16 planes of x * 2^-E with one
// exponent E per 32-row x 32-k sub-block, and a producer WAVE owns whole sub-blocks (so that E is a wave reduction):
// m-contiguous operands already are laid out that way (wave w holds rows 32 w .. 32 w + 31 of all 32 k lines); for
// k-contiguous operands the rows are dealt per wave instead of per pass (row_of).
// ONE (with X3): the single-pass 16-bit form -- only the h plane is written (no residual), the consumers issue one MFMA per
// k-slice (MODEL.COMPUTE_DTYPE: bf16 / fp16, never the parity path).
// Running sub-block exponent of a producer wave within an item (fp16x3 form, SplitLoader::store_x3) with the two values
// derived from it -- the overflow threshold 2^(15 + e) and the scale 2^-e -- kept beside it: they only change when e does
// (first tile of an item, or a sub-block maximum that grew), not once per K step.  All three are wave-uniform (scalar registers).
struct X3Expo {
    int e;
    float lim, sc;
    __device__ __forceinline__ void reset() { e = -1000; lim = 0.f; sc = 1.f; }
    __device__ __forceinline__ void set(int v) {
        e = v;
        lim = v <= -1000 ? 0.f : __uint_as_float((unsigned)(142 + v) << 23);   // 2^(15 + e)
        sc = v <= -1000 ? 1.f : __uint_as_float((unsigned)(127 - v) << 23);    // 2^-e
    }
};

template <int BT, bool KC, bool SWZ = false, bool X3 = false, bool ONE = false>
struct SplitLoader {   // BT x 32 fp32 operand tile (BT = 128, 64 or 32 rows) -> registers -> three bf16 planes in LDS
    static constexpr int NI = KC ? BT / 32 : 4;       // float4 per thread
    static constexpr int ROWB = SWZ ? 64 : X6_ROWB;
    static constexpr int PLANE = BT * ROWB;
    static_assert(!X3 || (BT == 128 && !SWZ), "the fp16x3 form exists for 128-row tiles");
    x6_f32x4 v[NI];
    // KC operands: tile row of thread group g = tid >> 3 within a 32-row pass.  Padded image: rows 4 apart per 16-lane
    // store group (see x6_kc_row); compact image: consecutive rows (different 64-byte segments of the bank window).
    static __device__ __forceinline__ int kc_row(int tid) { return SWZ ? (tid >> 3) : x6_kc_row(tid); }
    // KC operands: tile row of this thread's i-th 16-byte load
    static __device__ __forceinline__ int row_of(int tid, int i) {
        if (X3) {   // wave tid >> 6 owns rows 32 w .. 32 w + 31; a 16-lane store group covers rows r and r + 4 as in x6_kc_row
            const int g = (tid >> 3) & 7;
            return 32 * (tid >> 6) + 8 * i + (((g & 1) << 2) | ((g >> 1) & 3));
        }
        return kc_row(tid) + 32 * i;
    }

    // Branch-free: raw buffer loads (out-of-range bytes read as 0, so tiles past the K range or past the last row are
    // safe to request), tile rows clamped / surplus rows left as don't-care (they only feed C rows/cols that are never
    // stored), k >= kmax zeroed by selects at conversion time.
        // The loads are inline asm: hipcc's own waitcnt insertion would drain the whole ring (vmcnt(0)) at the loop header
    // once per three K steps; hidden from it, the ring is waited for by hand with a counted s_waitcnt vmcnt(2 * loads
    // per stage) in front of every conversion (x6_wait_stage), so two younger stages always stay in flight.
    // `valid` false (a tile past the K range): the request is sent with an out-of-range offset -- it returns zeros without
    // touching memory, so every ring stage always has the same number of loads in flight and the waits stay constant.
    __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t rsrc, int ld, int t0, int k0, int tmax, int kmax, int tid,
                                         bool valid) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            int off;
            if (KC)     // 4 consecutive k of tile row tr: a wave instruction reads 8 rows x 128 B
                off = min(t0 + row_of(tid, i), tmax - 1) * ld + k0 + (tid & 7) * 4;
            else        // 4 consecutive rows at k = 4*(tid&7)+i: a wave instruction reads 8 k-rows x 128 B
                off = min(k0 + (tid & 7) * 4 + i, kmax - 1) * ld + t0 + (tid >> 3) * 4;
            off = valid ? off * 4 : 0x7ffffff0;
#ifdef X3_DIAG_NOLOAD   // diagnostic build: every request goes out of range (returns zeros without touching memory)
            off = 0x7ffffff0;
#endif
            asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(v[i]) : "v"(off), "s"(rsrc));
        }
    }

    // Plain loads with the per-item part of the address hoisted out of the K loop (round 4: the producers are the kernel's
    // critical path, csrc DESIGN 4.1c; load() spends 4 x (add, min, quarter-rate mul, add) per K step on values that only
    // change with the item).  rb[]: KC -- element offset of this thread's i-th (clamped) tile row plus its k chunk; !KC --
    // element offset of k line (tid & 7) * 4 + i plus the column chunk.  `valid` false is expressed by a zero-length
    // buffer descriptor (every offset then returns zeros), not by a per-load select.
    struct PlainRows {
        int rb[NI];   // BYTE offsets
        __device__ __forceinline__ void setup(int ld, int t0, int tmax, int tid) {
#pragma unroll
            for (int i = 0; i < NI; ++i)
                rb[i] = (KC ? min(t0 + row_of(tid, i), tmax - 1) * ld + (tid & 7) * 4 : ((tid & 7) * 4 + i) * ld + t0 + (tid >> 3) * 4) * 4;
        }
    };
    // The K-tile term of the address is wave-uniform: it rides in the load's SCALAR offset (no vector instruction per load and
    // K step).  Only where the whole tile lies inside the tensor: the buffer bounds check is only relied upon for the vector
    // offset, and a K-tail tile (or, for an m-contiguous operand, a tile overhanging the last row / column) would reach past
    // the allocation from its last line -- those tiles keep the vector form, whose out-of-range bytes read as zeros.
    __device__ __forceinline__ void load_plain(__amdgpu_buffer_rsrc_t rsrc, int ld, const PlainRows& pr, int t0, int tmax, int k0, int kmax,
                                               int tid) {
#ifndef X3_DIAG_NOLOAD
        if (k0 + X6_BK <= kmax && (KC || t0 + BT <= tmax)) {   // (wave-uniform)
            const int soff = (KC ? k0 : k0 * ld) * 4;
#pragma unroll
            for (int i = 0; i < NI; ++i)
                asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(v[i]) : "v"(pr.rb[i]), "s"(rsrc), "s"(soff));
            return;
        }
#endif
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            int off = KC ? pr.rb[i] + k0 * 4 : (min(k0 + (tid & 7) * 4 + i, kmax - 1) * ld + t0 + (tid >> 3) * 4) * 4;
#ifdef X3_DIAG_NOLOAD
            off = 0x7ffffff0;
#endif
            asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(v[i]) : "v"(off), "s"(rsrc));
        }
    }
    // mode 3 through the same hoisted rows (PlainRows of a !KC operand: pr.rb[i] = ((tid & 7) * 4 + i) * ld + t0 + (tid >> 3) * 4):
    // a K tile of 32 lies inside one tap (bmod % 32 == 0), so line k of the tile sits at (k - tap * bmod) * ld + tap * btap --
    // a wave-uniform term (scalar arithmetic) on top of the per-item rows.  K = taps * bmod has no tail.
    __device__ __forceinline__ void load_kremap_plain(__amdgpu_buffer_rsrc_t rsrc, const ConvGather& g, int ld, const PlainRows& pr, int t0,
                                                      int tmax, int k0) {
        const int tap = fd_div(k0, g.dBmod);   // (only ever called for a !KC operand: see x6q_produce's static_assert)
        const int soff = ((k0 - tap * g.bmod) * ld + tap * g.btap) * 4;   // (K = taps * bmod: every tile lies inside the K range)
        const bool inside = t0 + BT <= tmax;   // (wave-uniform) else: vector offsets, bounds-checked (see load_plain)
#pragma unroll
        for (int i = 0; i < NI; ++i) {
#ifdef X3_DIAG_NOLOAD
            asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(v[i]) : "v"(0x7ffffff0), "s"(rsrc));
#else
            if (inside)
                asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(v[i]) : "v"(pr.rb[i]), "s"(rsrc), "s"(soff));
            else
                asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(v[i]) : "v"(pr.rb[i] + soff), "s"(rsrc));
#endif
        }
    }

    // ---- implicit-GEMM gathers (ConvGather) ----
    // mode 1 (KC): this thread's NI pixel rows, decomposed once per item: image base (pixels), tap-0 source coordinates
    struct PixRows {
        int base[NI], ys[NI], xs[NI];
        __device__ __forceinline__ void setup(const ConvGather& g, int t0, int tmax, int tid) {
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int r = min(t0 + row_of(tid, i), tmax - 1);
                const int img = fd_div(r, g.dHW), rem = r - img * (g.gH * g.gW);
                const int gy = fd_div(rem, g.dW), gx = rem - gy * g.gW;
                base[i] = img * (g.sH * g.sW);
                ys[i] = gy * g.a + g.b;
                xs[i] = gx * g.a + g.bx;
            }
        }
    };
    // mode 1 with the per-TAP part of the address hoisted out of the K loop (round 4: the per-step form spent ~ 45 vector
    // instructions per K step on values that change once per tap, i.e. every sC / 32 K steps): byte offset of this thread's
    // 16-byte chunk at channel 0 of tap `tap` for each of its NI pixel rows, or a sentinel that stays out of range when the
    // channel offset (< 8 KB) is added (ix_conv_gemm_supported keeps every tensor below 2^31 - 16 640 bytes).
    struct PixTap {
        int off[NI];
        int tap, c0;   // (wave-uniform) the load cursor's tap and channel offset within it
        __device__ __forceinline__ void setup(const ConvGather& g, const PixRows& pr, int tid) {
            const int ky = fd_div(tap, g.dKW), kx = tap - ky * g.KW;
            const int ty = ky * g.d, tx = kx * g.d, qm = (1 << g.qs) - 1;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                int sy = pr.ys[i] + ty, sx = pr.xs[i] + tx;
                bool ok = ((sy | sx) & qm) == 0;
                sy >>= g.qs;
                sx >>= g.qs;
                ok = ok && (unsigned)sy < (unsigned)g.sH && (unsigned)sx < (unsigned)g.sW;
                off[i] = ok ? ((pr.base[i] + sy * g.sW + sx) * g.sC + (tid & 7) * 4) * 4 : 0x7fffc000;
            }
        }
        __device__ __forceinline__ void start(const ConvGather& g, const PixRows& pr, int k0, int tid) {
            tap = fd_div(k0, g.dC);
            c0 = k0 - tap * g.sC;
            setup(g, pr, tid);
        }
        // after a K tile: the next one lies 32 channels on, or at channel 0 of the next tap
        __device__ __forceinline__ void advance(const ConvGather& g, const PixRows& pr, int tid) {
            c0 += X6_BK;
            if (c0 >= g.sC) {   // (wave-uniform; every sC / 32 K steps)
                c0 = 0;
                ++tap;
                setup(g, pr, tid);
            }
        }
    };
    __device__ __forceinline__ void load_taps(__amdgpu_buffer_rsrc_t rsrc, const PixTap& pt_) {
        static_assert(KC || NI == 4, "");
        const int cb = pt_.c0 * 4;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            int off = pt_.off[i] + cb;
#ifdef X3_DIAG_NOLOAD
            off = 0x7ffffff0;
#endif
            asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(v[i]) : "v"(off), "s"(rsrc));
        }
    }
    // mode 2 (!KC): this thread's four k rows are pixels (decomposed per tile), the tile's n range lies inside tap (ty, tx)
    __device__ __forceinline__ void load_pixk(__amdgpu_buffer_rsrc_t rsrc, const ConvGather& g, int t0, int k0, int kmax, int tid,
                                              bool valid) {
        const int tap = fd_div(t0, g.dC), c0 = t0 - tap * g.sC;
        const int ky = fd_div(tap, g.dKW), kx = tap - ky * g.KW;
        const int ty = ky * g.d + g.b, tx = kx * g.d + g.bx, qm = (1 << g.qs) - 1;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int k = min(k0 + (tid & 7) * 4 + i, kmax - 1);
            const int img = fd_div(k, g.dHW), rem = k - img * (g.gH * g.gW);
            const int gy = fd_div(rem, g.dW), gx = rem - gy * g.gW;
            int sy = gy * g.a + ty, sx = gx * g.a + tx;
            bool ok = valid && ((sy | sx) & qm) == 0;
            sy >>= g.qs;
            sx >>= g.qs;
            ok = ok && (unsigned)sy < (unsigned)g.sH && (unsigned)sx < (unsigned)g.sW;
            int off = (((img * g.sH + sy) * g.sW + sx) * g.sC + c0 + (tid >> 3) * 4) * 4;
            off = ok ? off : 0x7ffffff0;
            asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(v[i]) : "v"(off), "s"(rsrc));
        }
    }
    // k0 / kmax of the tile held in v[]: elements with k >= kmax are zeroed here (selects), not at load time
    __device__ __forceinline__ void store(unsigned char* __restrict__ planes, int tid, int k0, int kmax) const {
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        typedef __attribute__((address_space(3))) u32x2 lds_u2;
        if (!KC && (tid >> 3) * 4 >= BT) return;   // narrow tiles: surplus row chunks were loaded as don't-care
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            float e0, e1, e2, e3;
            int row;
            if (KC) {
                row = row_of(tid, i);
                e0 = v[i].x; e1 = v[i].y; e2 = v[i].z; e3 = v[i].w;
            } else {    // row i of this thread's 4x4 (k x row) register block
                row = (tid >> 3) * 4 + i;
                e0 = i == 0 ? v[0].x : i == 1 ? v[0].y : i == 2 ? v[0].z : v[0].w;
                e1 = i == 0 ? v[1].x : i == 1 ? v[1].y : i == 2 ? v[1].z : v[1].w;
                e2 = i == 0 ? v[2].x : i == 1 ? v[2].y : i == 2 ? v[2].z : v[2].w;
                e3 = i == 0 ? v[3].x : i == 1 ? v[3].y : i == 2 ? v[3].z : v[3].w;
            }
            if (k0 + X6_BK > kmax) {   // (wave-uniform) only the last K tile of an item can reach past the K range
                const int gk = k0 + (tid & 7) * 4;
                e0 = gk + 0 < kmax ? e0 : 0.f;
                e1 = gk + 1 < kmax ? e1 : 0.f;
                e2 = gk + 2 < kmax ? e2 : 0.f;
                e3 = gk + 3 < kmax ? e3 : 0.f;
            }
            unsigned h0, m0, l0, h1, m1, l1;
            split3(e0, e1, h0, m0, l0);
            split3(e2, e3, h1, m1, l1);
            unsigned char* dst = SWZ ? planes + row * 64 + ((((tid & 7) >> 1) ^ ((row >> 2) & 3)) << 4) + ((tid & 1) << 3)
                                     : planes + row * X6_ROWB + (tid & 7) * 8;
            u32x2 ph, pm, pl;
            ph.x = h0; ph.y = h1; pm.x = m0; pm.y = m1; pl.x = l0; pl.y = l1;
            *(lds_u2*)(dst) = ph;
            *(lds_u2*)(dst + PLANE) = pm;
            *(lds_u2*)(dst + 2 * PLANE) = pl;
        }
    }

    // fp16x3 form: the wave's 32 x 32 sub-block as two fp16 planes of x * 2^-E.  E follows the sub-block maximum (scaled
    // into [2^14, 2^15)) but never DEcreases along the K tiles of an item (`erun`): the consumers then only ever have to
    // scale their accumulators DOWN, exactly, when E grows -- a later tile with smaller values keeps the larger E and is
    // resolved to 2^-25 of the running maximum, which is what its products are worth next to the earlier ones.
    // `expo` = the LDS word of this wave's sub-block in the image being written.
    __device__ __forceinline__ void store_x3(unsigned char* __restrict__ planes, int tid, int k0, int kmax, X3Expo& ex,
                                             int* __restrict__ expo) const {
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        typedef __attribute__((address_space(3))) u32x2 lds_u2;
#ifdef X3_DIAG_NOCONV   // diagnostic build: raw bits into the planes, no maximum / exponent / conversion work (wrong numbers)
        if ((tid & 63) == 0) *expo = 0;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int row = KC ? row_of(tid, i) : (tid >> 3) * 4 + i;
            unsigned char* dst = planes + row * X6_ROWB + (tid & 7) * 8;
            u32x2 ph, pl;
            ph.x = __float_as_uint(v[i].x) & 0x3bff3bffu; ph.y = __float_as_uint(v[i].y) & 0x3bff3bffu;
            pl.x = __float_as_uint(v[i].z) & 0x3bff3bffu; pl.y = __float_as_uint(v[i].w) & 0x3bff3bffu;
            *(lds_u2*)(dst) = ph;
            *(lds_u2*)(dst + PLANE) = pl;
        }
        (void)k0; (void)kmax; (void)ex;
        return;
#endif
        float e[NI][4];
        float mx;
        const bool tail = k0 + X6_BK > kmax;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            if (KC) {
                e[i][0] = v[i].x; e[i][1] = v[i].y; e[i][2] = v[i].z; e[i][3] = v[i].w;
            } else {    // row i of this thread's 4x4 (k x row) register block
                e[i][0] = i == 0 ? v[0].x : i == 1 ? v[0].y : i == 2 ? v[0].z : v[0].w;
                e[i][1] = i == 0 ? v[1].x : i == 1 ? v[1].y : i == 2 ? v[1].z : v[1].w;
                e[i][2] = i == 0 ? v[2].x : i == 1 ? v[2].y : i == 2 ? v[2].z : v[2].w;
                e[i][3] = i == 0 ? v[3].x : i == 1 ? v[3].y : i == 2 ? v[3].z : v[3].w;
            }
        }
        if (tail) {   // (wave-uniform, last K tile of an item only) zero the elements past the K range
            int km = kmax;
            asm volatile("" : "+s"(km));   // keeps the mask arithmetic INSIDE the branch (hoisted, it cost 8 instructions per K step)
            const int gk = k0 + (tid & 7) * 4;
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int c = 0; c < 4; ++c) e[i][c] = gk + c < km ? e[i][c] : 0.f;
        }
        // |.| as source modifiers of v_max3_f32: 8 instructions for the 16 elements (the compiler's own lowering of
        // fmaxf(fabsf) spends 28: it canonicalises every |x| with a v_max of its own)
        // (seeded by the first three elements: no zero to load; 1 + 2 (NI - 1) + 1 = 8 instructions for NI = 4)
        asm("v_max3_f32 %0, |%1|, |%2|, |%3|" : "=v"(mx) : "v"(e[0][0]), "v"(e[0][1]), "v"(e[0][2]));
        float carry = e[0][3];
#pragma unroll
        for (int i = 1; i < NI; ++i) {
            asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(mx) : "v"(carry), "v"(e[i][0]));
            asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(mx) : "v"(e[i][1]), "v"(e[i][2]));
            carry = e[i][3];
        }
        asm("v_max_f32 %0, %0, |%1|" : "+v"(mx) : "v"(carry));
        // wave maximum without the LDS crossbar (six ds_bpermute shuffles cost the producers 8 % of the kernel): the bit
        // pattern of a non-negative float orders like an integer; DPP row shifts, then row broadcasts; lane 63 has it
        // E only has to change when some value would leave [0, 2^15) under the running scale -- one compare and a vote;
        // the reduction itself (six dependent DPP steps + a readlane) then runs once per item and whenever the data grows
        if (__ballot(mx >= ex.lim) != 0) {
            const int erun = ex.e;
            int mi = (int)__float_as_uint(mx);
            mi = max(mi, __builtin_amdgcn_update_dpp(mi, mi, 0x111, 0xf, 0xf, false));   // row_shr:1
            mi = max(mi, __builtin_amdgcn_update_dpp(mi, mi, 0x112, 0xf, 0xf, false));   // row_shr:2
            mi = max(mi, __builtin_amdgcn_update_dpp(mi, mi, 0x114, 0xf, 0xf, false));   // row_shr:4
            mi = max(mi, __builtin_amdgcn_update_dpp(mi, mi, 0x118, 0xf, 0xf, false));   // row_shr:8  (lane 15 of a row: row maximum)
            mi = max(mi, __builtin_amdgcn_update_dpp(mi, mi, 0x142, 0xa, 0xf, false));   // row_bcast:15 into rows 1, 3
            mi = max(mi, __builtin_amdgcn_update_dpp(mi, mi, 0x143, 0xc, 0xf, false));   // row_bcast:31 into rows 2, 3
            const int eb = (__builtin_amdgcn_readlane(mi, 63) >> 23) & 0xff;
            // zero / denormal / inf sub-block: keep the running exponent (its values convert to 0 / inf whatever the scale)
            const int E = (eb < 16 || eb > 250) ? erun : eb - 141;   // x * 2^-E has its maximum in [2^14, 2^15)
            ex.set(max(erun, E));
        }
        const float sc = ex.sc;   // (a scalar register: the conversion instructions take it as their one scalar operand)
        if ((tid & 63) == 0) *expo = ex.e;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int row = KC ? row_of(tid, i) : (tid >> 3) * 4 + i;
            // two instructions per element (round 4, as in gemm_wp.hip): h = f16(x * sc) by v_fma_mixlo/hi_f16 (the product with a
            // power of two is exact: one rounding, the same bits as v_mul + v_cvt), l = f16(x * sc - h) by the same instruction
            // reading h's half directly (the fp32 residual is exact, so rounding the fused result once gives the same bits as
            // v_fma_mix_f32 + v_cvt) -- 2 instead of 3 instructions per element, and no separate pack
            unsigned hb0, hb1;
            asm("v_fma_mixlo_f16 %0, %1, %3, 0 op_sel_hi:[0,0,0]\n\t"
                "v_fma_mixhi_f16 %0, %2, %3, 0 op_sel_hi:[0,0,0]"
                : "=&v"(hb0) : "v"(e[i][0]), "v"(e[i][1]), "s"(sc));
            asm("v_fma_mixlo_f16 %0, %1, %3, 0 op_sel_hi:[0,0,0]\n\t"
                "v_fma_mixhi_f16 %0, %2, %3, 0 op_sel_hi:[0,0,0]"
                : "=&v"(hb1) : "v"(e[i][2]), "v"(e[i][3]), "s"(sc));
            if (ONE) {   // single-pass form: the rounded value is the operand
                u32x2 ph1;
                ph1.x = hb0; ph1.y = hb1;
                *(lds_u2*)(planes + row * X6_ROWB + (tid & 7) * 8) = ph1;
                continue;
            }
            unsigned lb0, lb1;
            asm("v_fma_mixlo_f16 %0, %1, %3, -%4 op_sel_hi:[0,0,1]\n\t"
                "v_fma_mixhi_f16 %0, %2, %3, -%4 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
                : "=&v"(lb0) : "v"(e[i][0]), "v"(e[i][1]), "s"(sc), "v"(hb0));
            asm("v_fma_mixlo_f16 %0, %1, %3, -%4 op_sel_hi:[0,0,1]\n\t"
                "v_fma_mixhi_f16 %0, %2, %3, -%4 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
                : "=&v"(lb1) : "v"(e[i][2]), "v"(e[i][3]), "s"(sc), "v"(hb1));
            unsigned char* dst = planes + row * X6_ROWB + (tid & 7) * 8;
            u32x2 ph, pl;
            ph.x = hb0; ph.y = hb1;
            pl.x = lb0; pl.y = lb1;
            *(lds_u2*)(dst) = ph;
            *(lds_u2*)(dst + PLANE) = pl;
        }
    }
};

// Cooperative write-out of a row-major C tile staged in LDS (pitch BN + 4 floats) by all 512 threads: one float4 per
// lane, consecutive lanes along a row.
template <int BN, int NT = 512>
__device__ __forceinline__ void x6_store_tile(const float* __restrict__ ct, float* __restrict__ C, int64_t ldc, int m0,
                                              int n0, int M, int N, int tid) {
    constexpr int CP = BN + 4, CPR = BN / 4;   // float4 chunks per tile row
#pragma unroll
    for (int q = 0; q < X6_BT * CPR / NT; ++q) {
        const int c = tid + NT * q;
        const int row = c / CPR, col = (c % CPR) * 4;
        const int gr = m0 + row, gc = n0 + col;
        if (gr >= M || gc >= N) continue;
        const float4 v = *reinterpret_cast<const float4*>(&ct[row * CP + col]);
        float* dst = C + (int64_t)gr * ldc + gc;
        if (gc + 3 < N) {
            *reinterpret_cast<float4*>(dst) = v;
        } else {
            dst[0] = v.x;
            if (gc + 1 < N) dst[1] = v.y;
            if (gc + 2 < N) dst[2] = v.z;
        }
    }
}

// Wave-specialised workgroup of 8 waves (one workgroup per CU, two waves per SIMD): waves 4-7 are PRODUCERS -- they
// stream the fp32 operand tiles from HBM through a 3-deep register ring (loads issued three K steps ahead), split them
// into bf16 planes on the VALU and write the LDS image of tile t+1 -- while waves 0-3 are CONSUMERS that only issue
// ds_read_b128 + MFMA on tile t.  Each SIMD therefore runs one VALU-bound and one MFMA-bound wave side by side (the
// two pipes are independent), the LDS image is double-buffered (2 x 60 KB) and there is ONE barrier per K step.
// Producer side of the bf16x6 kernel: K tile j lives in ring stage j % 3; per K step the oldest stage is converted and
// stored into LDS buffer (j & 1) and refilled with the tile three steps ahead; one barrier per K step (nk + 1 in total,
// matching the consumers).  Tiles past the K range are requested out of range (free, zeros) and not converted.
template <int BM, int BN, bool A_KC, bool B_KC>
__device__ __forceinline__ void x6_produce(__amdgpu_buffer_rsrc_t rA, __amdgpu_buffer_rsrc_t rB, int lda, int ldb, int m0,
                                           int n0, int kbeg, int kend, int nk, int M, int N, unsigned char* lds, int buf_bytes,
                                           int pt) {
    constexpr int BK = X6_BK, PLANE_A = BM * X6_ROWB;
    static_assert(SplitLoader<BM, A_KC>::NI == 4, "the A stage is four 16-byte loads per thread");
    constexpr int NB_ = SplitLoader<BN, B_KC>::NI;
    static_assert(NB_ == 4 || NB_ == 2 || NB_ == 1, "unexpected ring stage size");
    SplitLoader<BM, A_KC> a0, a1, a2;
    SplitLoader<BN, B_KC> b0, b1, b2;
    // sched_barriers pin the order [wait, convert + store the oldest stage] -> [refill it].  The loads are never inside
    // a branch: an asm output defined on one path only becomes a phi, i.e. a register copy right behind the load,
    // before the data has arrived.
#define X6_LD(SA, SB, T)                                                      \
    __builtin_amdgcn_sched_barrier(0);                                        \
    SA.load(rA, lda, m0, kbeg + (T) * BK, M, kend, pt, (T) < nk);             \
    SB.load(rB, ldb, n0, kbeg + (T) * BK, N, kend, pt, (T) < nk);             \
    __builtin_amdgcn_sched_barrier(0);
#define X6_ST(SA, SB, T)                                                                              \
    X6_WAIT_STAGE(SA, SB, 2)                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                                \
    if ((T) < nk) {                                                                                   \
        SA.store(lds + ((T) & 1) * buf_bytes, pt, kbeg + (T) * BK, kend);                             \
        SB.store(lds + ((T) & 1) * buf_bytes + 3 * PLANE_A, pt, kbeg + (T) * BK, kend);               \
    }                                                                                                 \
    __builtin_amdgcn_sched_barrier(0);
    X6_LD(a0, b0, 0)
    X6_LD(a1, b1, 1)
    X6_LD(a2, b2, 2)
    X6_ST(a0, b0, 0)
    X6_LD(a0, b0, 3)
    __syncthreads();   // tile 0 is visible
    for (int kt = 0; kt < nk; kt += 3) {
        X6_ST(a1, b1, kt + 1)
        X6_LD(a1, b1, kt + 4)
        __syncthreads();
        if (kt + 1 >= nk) break;
        X6_ST(a2, b2, kt + 2)
        X6_LD(a2, b2, kt + 5)
        __syncthreads();
        if (kt + 2 >= nk) break;
        X6_ST(a0, b0, kt + 3)
        X6_LD(a0, b0, kt + 6)
        __syncthreads();
    }
#undef X6_LD
#undef X6_ST
}

// BN = 128, 64 or 32 output columns per workgroup (attention's P.V products have N = head dim = 64 / 32).
template <int BN, bool A_KC, bool B_KC>
__global__ __launch_bounds__(512, 1) void gemm_f32_bf16x6_kernel(GemmArgs p) {
    constexpr int BM = X6_BT, BK = X6_BK;
    constexpr int PLANE_A = BM * X6_ROWB, PLANE_B = BN * X6_ROWB, BUF = 3 * (PLANE_A + PLANE_B);
    // consumer wave grid: 2 x 2 waves of (64 x BN/2) for BN >= 64, 4 x 1 waves of (32 x 32) for BN = 32
    constexpr int WM = BN >= 64 ? 64 : 32, WN = BN >= 64 ? BN / 2 : 32, TM = WM / 32, TN = WN / 32;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2][BUF];   // per buffer: A planes h,m,l, B planes h,m,l

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nwg = p.tiles_m * p.tiles_n;
    const int tile = xcd_swizzle(blockIdx.x, nwg);
    constexpr int GROUP_M = 8;
    const int group_size = GROUP_M * p.tiles_n;
    const int first_m = (tile / group_size) * GROUP_M;
    const int gm = min(p.tiles_m - first_m, GROUP_M);
    const int m0 = (first_m + (tile % group_size) % gm) * BM, n0 = ((tile % group_size) / gm) * BN;
    const int zb = blockIdx.y, ks = blockIdx.z;
    const int bo = zb / p.batch_inner, bi = zb % p.batch_inner;
    const float* A = p.A + bo * p.sAo + bi * p.sAi;
    const float* B = p.B + bo * p.sBo + bi * p.sBi;
    const int kbeg = ks * p.k_per_split;
    const int kend = min(p.K, kbeg + p.k_per_split);
    const int nk = (kend - kbeg + BK - 1) / BK;

    if (wave >= 4) {
        const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (int)(p.extA * 4), 0x00020000);
        const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, (int)(p.extB * 4), 0x00020000);
        const int lda = (int)p.lda, ldb = (int)p.ldb;
        // ------------------------------------------------ producers ------------------------------------------------
        const int pt = tid - 256;
        x6_produce<BM, BN, A_KC, B_KC>(rA, rB, lda, ldb, m0, n0, kbeg, kend, nk, p.M, p.N, &lds[0][0], BUF, pt);
        // nothing may still be in flight when this wave ends (its registers go to another wave; the compiler does not
        // know about the inline-asm loads)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (p.atomic || !p.c_vec) return;   // (consumers store straight from registers in those cases)
        __syncthreads();                          // C tile staged in LDS by the consumers: help writing it out
        x6_store_tile<BN>(reinterpret_cast<const float*>(&lds[0][0]), p.C + bo * p.sCo + bi * p.sCi + ks * p.sSplit, p.ldc, m0, n0, p.M, p.N, tid);
        return;
    }

    // ---------------------------------------------------- consumers ----------------------------------------------------
    float* C = p.C + bo * p.sCo + bi * p.sCi + ks * p.sSplit;
    const float* bias = p.bias ? p.bias + bo * p.sBias : nullptr;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int wm = BN >= 64 ? (wave >> 1) * WM : wave * WM, wn = BN >= 64 ? (wave & 1) * WN : 0;
    const int lrow = lane >> 5, lcol = lane & 31;

    // Fragment reads are software-pipelined around the MFMAs and across the K-step barrier: while the 24 MFMAs of one
    // k-slice issue, the ds_read_b128s of the next slice (of this tile, or -- after the barrier -- of the next tile)
    // are already in flight.  sched_barriers pin that order (hipcc otherwise hoists all reads and serialises).
    bf16x8 fa0[TM][3], fb0[TN][3], fa1[TM][3], fb1[TN][3];
#define X6_READ(FA, FB, BASE, S)                                                                                    \
    {                                                                                                               \
        const unsigned char* rA_ = (BASE);                                                                          \
        const unsigned char* rB_ = rA_ + 3 * PLANE_A;                                                               \
        const int koff_ = ((S) * 2 + lrow) * 16;                                                                    \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int pl = 0; pl < 3; ++pl) FA[i][pl] =  \
            *reinterpret_cast<const bf16x8*>(rA_ + pl * PLANE_A + (wm + i * 32 + lcol) * X6_ROWB + koff_);           \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) _Pragma("unroll") for (int pl = 0; pl < 3; ++pl) FB[j][pl] =  \
            *reinterpret_cast<const bf16x8*>(rB_ + pl * PLANE_B + (wn + j * 32 + lcol) * X6_ROWB + koff_);           \
    }
    // smallest terms first; consecutive MFMAs go to different accumulators
#define X6_TERM(FA, FB, PA_, PB_)                                                                                   \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j) acc[i][j] =        \
        __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[i][PA_], FB[j][PB_], acc[i][j], 0, 0, 0);
#define X6_MMA(FA, FB)  \
    X6_TERM(FA, FB, 2, 0) X6_TERM(FA, FB, 1, 1) X6_TERM(FA, FB, 0, 2) X6_TERM(FA, FB, 1, 0) X6_TERM(FA, FB, 0, 1) X6_TERM(FA, FB, 0, 0)

    __syncthreads();   // tile 0 is visible
    if (nk > 0) X6_READ(fa0, fb0, lds[0], 0)
    for (int kt = 0; kt < nk; ++kt) {
        X6_READ(fa1, fb1, lds[kt & 1], 1)
        __builtin_amdgcn_sched_barrier(0);
        X6_MMA(fa0, fb0)
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();   // this tile's reads are done (producers may refill it); the next tile is complete
        if (kt + 1 < nk) X6_READ(fa0, fb0, lds[(kt + 1) & 1], 0)
        __builtin_amdgcn_sched_barrier(0);
        X6_MMA(fa1, fb1)
        __builtin_amdgcn_sched_barrier(0);
    }
#undef X6_READ
#undef X6_TERM
#undef X6_MMA

    const bool add_bias = bias != nullptr && ks == 0;
    if (!p.atomic && p.c_vec) {
        // Epilogue through LDS: the MFMA C layout gives every lane 16 scattered dwords per accumulator (one 4-byte store
        // each, store-issue bound); staged as a row-major tile the whole workgroup (producers included) writes it out
        // as 16-byte stores, 512 contiguous bytes per row.  The operand buffers are dead after the last K-step barrier.
        float* ct = reinterpret_cast<float*>(&lds[0][0]);
        constexpr int CP = BN + 4;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int cl = wn + j * 32 + lcol;
                const float bv = (add_bias && n0 + cl < p.N) ? bias[n0 + cl] : 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    ct[(wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lrow) * CP + cl] = p.alpha * acc[i][j][r] + bv;
            }
        __syncthreads();
        x6_store_tile<BN>(ct, C, p.ldc, m0, n0, p.M, p.N, tid);
        return;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn + j * 32 + lcol;
            if (col >= p.N) continue;
            const float bv = add_bias ? bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lrow;
                if (row < p.M) {
                    const float v = p.alpha * acc[i][j][r] + bv;
                    float* dst = C + (int64_t)row * p.ldc + col;
                    if (p.atomic)
                        unsafeAtomicAdd(dst, v);
                    else
                        *dst = v;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// Persistent form of the bf16x6 kernel: one workgroup per CU walks a list of output tiles (tile, split, batch) instead
// of ending after one.  The producers request the first three K tiles of the NEXT output tile before the consumers
// start the epilogue of the current one, so the HBM round trip of every tile's prologue and the epilogue's stores
// overlap; with one workgroup per CU nothing else would hide them (K = 256 / 64 shapes spend a third of their time
// there).  Work items are dealt per XCD (workgroup g serves XCD g & 7 = blockIdx % 8, the observed placement): each XCD
// walks one contiguous eighth of the grouped tile order, which keeps neighbouring tiles in one L2.
// Barriers, both roles: one per flat K tile of the workgroup's item stream.  The epilogue needs none: every consumer wave
// transposes its own accumulators through a PRIVATE LDS strip (outside the operand buffers) into 16-byte row-major
// stores, so the producers convert the next item's first K tile into the operand buffers while the consumers are
// still writing the current C tile -- short-K items (K = 64 / 32 attention products) are mostly epilogue otherwise.
// ------------------------------------------------------------------------------------------------------------
struct X6Item {
    const float* A;
    const float* B;
    float* C;
    const float* bias;
    float* rowsum;   // non-null for the items of the first tile column when GemmArgs.rowsum is set
    int m0, n0, kbeg, kend, nk, ks;
};

template <int BN, int BM = X6_BT>
__device__ __forceinline__ X6Item x6_item(const GemmArgs& p, int w) {
    X6Item it;
    // (divisions by launch constants as multiply-high + shift: this runs once per item on every wave, and on the producer
    //  waves -- the kernel's critical path -- a generic 32-bit division is ~ 20 scalar instructions, five of them per item)
    const int nt = p.tiles_m * p.tiles_n, per_batch = nt * p.split_k;
    const int zb = fd_div(w, p.fd_per_batch), rem = w - zb * per_batch;
    const int ks = fd_div(rem, p.fd_nt), tile = rem - ks * nt;
    constexpr int GROUP_M = 8;
    const int group_size = GROUP_M * p.tiles_n;
    const int grp = fd_div(tile, p.fd_group), first_m = grp * GROUP_M, r = tile - grp * group_size;
    const int gm = min(p.tiles_m - first_m, GROUP_M);
    int rm, rn;
    if (gm == GROUP_M) { rm = r & (GROUP_M - 1); rn = r >> 3; } else { rn = r / gm; rm = r - rn * gm; }   // (last, partial group only)
    it.m0 = (first_m + rm) * BM;
    it.n0 = rn * BN;
    const int bo = fd_div(zb, p.fd_bi), bi = zb - bo * p.batch_inner;
    it.A = p.A + bo * p.sAo + bi * p.sAi;
    it.B = p.B + bo * p.sBo + bi * p.sBi;
    it.C = p.C + bo * p.sCo + bi * p.sCi + ks * p.sSplit;
    it.bias = p.bias ? p.bias + bo * p.sBias : nullptr;
    it.rowsum = (p.rowsum && it.n0 == 0) ? p.rowsum + bo * p.sRowsum + ks * p.sSplitRowsum : nullptr;
    it.ks = ks;
    it.kbeg = ks * p.k_per_split;
    it.kend = min(p.K, it.kbeg + p.k_per_split);
    it.nk = (it.kend - it.kbeg + X6_BK - 1) / X6_BK;
    return it;
}

#ifdef X6_DIAG_TIMING   // diagnostic build (tools/gemm_diag.py): clock stamps of workgroup 0's first producer / consumer wave
__device__ long long x6_dbg[2][1024];
__device__ long long x6_dbg_wall[2][512];   // s_memrealtime (constant 100 MHz) next to every s_memtime stamp
#define X6_STAMP(ROLE, TAG)                                                           \
    if (blockIdx.x == 0 && lane == 0 && wave == ((ROLE) ? 4 : 0) && dbgn < 511) {     \
        x6_dbg[ROLE][2 * dbgn] = (TAG);                                               \
        x6_dbg[ROLE][2 * dbgn + 1] = clock64();                                       \
        x6_dbg_wall[ROLE][dbgn] = wall_clock64();                                     \
        ++dbgn;                                                                       \
    }
extern "C" int ix_gemm_dbg_read(long long* dst) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(x6_dbg), sizeof(x6_dbg));
}
extern "C" int ix_gemm_dbg_read_wall(long long* dst) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(x6_dbg_wall), sizeof(x6_dbg_wall));
}
// write-pattern probe: 256 persistent workgroups store bm x bn tiles of a [batch, M, ldc] tensor (no compute), tiles dealt
// like the GEMM's (order 0: per-XCD contiguous chunks of the m-fastest grouped order; 1: plain round robin, n fastest)
__global__ __launch_bounds__(1024) void diag_tile_fill_kernel(float* C, int M, int N, int64_t ldc, int batch, int bm, int bn, int order) {
    const int tm = (M + bm - 1) / bm, tn = (N + bn - 1) / bn, nt = tm * tn, total = nt * batch;
    const int per_xcd = (total + 7) >> 3, xcd = blockIdx.x & 7, stride = gridDim.x >> 3;
    const bool grouped = (order & 1) == 0;   // order bit 0: tile order, bit 1: non-temporal stores
    const int last = grouped ? min(total, (xcd + 1) * per_xcd) : total;
    int w = grouped ? xcd * per_xcd + (blockIdx.x >> 3) : blockIdx.x;
    const int step = grouped ? stride : gridDim.x;
    const int cpr = bn / 4;
    for (; w < last; w += step) {
        const int zb = w / nt, tile = w % nt;
        int m0, n0;
        if (grouped) {
            const int gs = 8 * tn, fm = (tile / gs) * 8, gm = min(tm - fm, 8);
            m0 = (fm + (tile % gs) % gm) * bm;
            n0 = ((tile % gs) / gm) * bn;
        } else {
            m0 = (tile / tn) * bm;
            n0 = (tile % tn) * bn;
        }
        float* base = C + (int64_t)zb * M * ldc;
        for (int c = threadIdx.x; c < bm * cpr; c += blockDim.x) {
            const int gr = m0 + c / cpr, gc = n0 + (c % cpr) * 4;
            if (gr < M && gc + 3 < N) {
                typedef float f4 __attribute__((ext_vector_type(4)));
                f4 v = {1.f, 2.f, 3.f, 4.f};
                f4* dst = reinterpret_cast<f4*>(base + (int64_t)gr * ldc + gc);
                if (order & 2)
                    __builtin_nontemporal_store(v, dst);
                else
                    *dst = v;
            }
        }
    }
}
extern "C" int ix_diag_tile_fill(float* C, int M, int N, int64_t ldc, int batch, int bm, int bn, int order, int grid, hipStream_t stream) {
    // order bits 8..: threads per workgroup / 64 (0 = 4 waves)
    const int waves = (order >> 8) ? (order >> 8) : 4;
    order &= 255;
    hipLaunchKernelGGL(diag_tile_fill_kernel, dim3(grid), dim3(64 * waves), 0, stream, C, M, N, ldc, batch, bm, bn, order);
    return (int)hipGetLastError();
}
#else
#define X6_STAMP(ROLE, TAG)
#endif

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the wave's global-memory counter
// (vmcnt(0)): in the persistent kernel that would make the consumers sit out the HBM acknowledgement of the C tile
// they have just stored before they may join the next item's first barrier.  Global memory is never used to
// communicate inside the workgroup, so only the LDS operations have to have completed.
#ifdef X3_DIAG_HALFBAR   // diagnostic build: every second barrier of a wave is skipped (RACES, wrong numbers): what the
__device__ __forceinline__ void x6_lds_barrier(int& bc) {   // synchronisation of a K step costs (tools/gemm_x3_diag.py)
    if ((bc++ & 1) == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    }
}
#define x6_lds_barrier() x6_lds_barrier(ix_bc_)
#define X6_BC_DECL int ix_bc_ = 0;
#else
__device__ __forceinline__ void x6_lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
#define X6_BC_DECL
#endif

template <int BN, bool A_KC, bool B_KC>
__global__ __launch_bounds__(512, 1) void gemm_f32_bf16x6_persistent_kernel(GemmArgs p, int total_items) {
    X6_BC_DECL
    constexpr int BM = X6_BT, BK = X6_BK;
    constexpr int PLANE_A = BM * X6_ROWB, PLANE_B = BN * X6_ROWB, BUF = 3 * (PLANE_A + PLANE_B);
    constexpr int WM = BN >= 64 ? 64 : 32, WN = BN >= 64 ? BN / 2 : 32, TM = WM / 32, TN = WN / 32;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2][BUF];
    constexpr int CP = WN + 4;                                           // pitch of a consumer wave's C strip (floats)
    __shared__ __attribute__((aligned(16))) float cstrip[4][32 * CP];    // 32 rows x WN columns per consumer wave

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int dbgn = 0;
    (void)dbgn;
    // this workgroup's items: w = first, first + stride, ... < last   (per-XCD contiguous chunks)
    const int per_xcd = (total_items + 7) >> 3, xcd = blockIdx.x & 7;
    const int stride = gridDim.x >> 3, last = min(total_items, (xcd + 1) * per_xcd);
    int w = xcd * per_xcd + (blockIdx.x >> 3);
    if (w >= last) return;
    const bool staged = !p.atomic && p.c_vec;

    if (wave >= 4) {
        // ------------------------------------------------ producers ------------------------------------------------
        // One flat stream of K tiles over all of this workgroup's items: flat tile g lives in ring stage g % 3 and LDS
        // buffer g & 1, and is requested three flat tiles ahead -- for K = 64 that is one and a half items ahead, so
        // the HBM round trip of an item's first tile is hidden behind the previous items, not exposed once per item.
        // Two cursors walk the (item, K tile) sequence: L (next tile to request) runs three tiles ahead of S (next
        // tile to convert and store).
        const int pt = tid - 256, lda = (int)p.lda, ldb = (int)p.ldb;
        static_assert(SplitLoader<BM, A_KC>::NI == 4, "the A stage is four 16-byte loads per thread");
        constexpr int NB_ = SplitLoader<BN, B_KC>::NI;
        static_assert(NB_ == 4 || NB_ == 2 || NB_ == 1, "unexpected ring stage size");
        SplitLoader<BM, A_KC> a0, a1, a2;
        SplitLoader<BN, B_KC> b0, b1, b2;
        X6Item itL = x6_item<BN>(p, w), itS = itL;
        int wL = w, tL = 0, wS = w, tS = 0, buf = 0;
        bool moreL = true, moreS = true;
        __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)itL.A, 0, (int)(p.extA * 4), 0x00020000);
        __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)itL.B, 0, (int)(p.extB * 4), 0x00020000);
        // (no next tile: same instructions with out-of-range offsets -- the loads must never sit in a branch)
#define X6_LD(SA, SB)                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                          \
    SA.load(rA, lda, itL.m0, itL.kbeg + tL * BK, p.M, itL.kend, pt, moreL);                     \
    SB.load(rB, ldb, itL.n0, itL.kbeg + tL * BK, p.N, itL.kend, pt, moreL);                     \
    __builtin_amdgcn_sched_barrier(0);                                                          \
    if (++tL >= itL.nk) {                                                                       \
        tL = 0;                                                                                 \
        wL += stride;                                                                           \
        moreL = wL < last;                                                                      \
        itL = x6_item<BN>(p, moreL ? wL : last - 1);                                            \
        rA = __builtin_amdgcn_make_buffer_rsrc((void*)itL.A, 0, (int)(p.extA * 4), 0x00020000); \
        rB = __builtin_amdgcn_make_buffer_rsrc((void*)itL.B, 0, (int)(p.extB * 4), 0x00020000); \
    }
#define X6_STEP(SA, SB)                                                                         \
    X6_STAMP(1, 0)                                                                              \
    X6_WAIT_STAGE(SA, SB, 2)                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                          \
    X6_STAMP(1, 1)                                                                              \
    SA.store(lds[buf], pt, itS.kbeg + tS * BK, itS.kend);                                       \
    SB.store(lds[buf] + 3 * PLANE_A, pt, itS.kbeg + tS * BK, itS.kend);                         \
    __builtin_amdgcn_sched_barrier(0);                                                          \
    X6_STAMP(1, 2)                                                                              \
    buf ^= 1;                                                                                   \
    if (++tS >= itS.nk) {                                                                       \
        tS = 0;                                                                                 \
        wS += stride;                                                                           \
        moreS = wS < last;                                                                      \
        if (moreS) itS = x6_item<BN>(p, wS);                                                    \
    }                                                                                           \
    X6_LD(SA, SB)                                                                               \
    X6_STAMP(1, 3)                                                                              \
    x6_lds_barrier();   /* flat tile g is visible; the consumers are done reading tile g - 1 */ \
    X6_STAMP(1, 4)
        X6_LD(a0, b0)
        X6_LD(a1, b1)
        X6_LD(a2, b2)
        for (;;) {
            X6_STEP(a0, b0)
            if (!moreS) break;
            X6_STEP(a1, b1)
            if (!moreS) break;
            X6_STEP(a2, b2)
            if (!moreS) break;
        }
#undef X6_LD
#undef X6_STEP
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // nothing may be in flight when the wave ends
        return;
    }

    // ---------------------------------------------------- consumers ----------------------------------------------------
    const int wm = BN >= 64 ? (wave >> 1) * WM : wave * WM, wn = BN >= 64 ? (wave & 1) * WN : 0;
    const int lrow = lane >> 5, lcol = lane & 31;
    bf16x8 fa0[TM][3], fb0[TN][3], fa1[TM][3], fb1[TN][3];
#define X6_READ(FA, FB, BASE, S)                                                                                    \
    {                                                                                                               \
        const unsigned char* rA_ = (BASE);                                                                          \
        const unsigned char* rB_ = rA_ + 3 * PLANE_A;                                                               \
        const int koff_ = ((S) * 2 + lrow) * 16;                                                                    \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int pl = 0; pl < 3; ++pl) FA[i][pl] =  \
            *reinterpret_cast<const bf16x8*>(rA_ + pl * PLANE_A + (wm + i * 32 + lcol) * X6_ROWB + koff_);           \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) _Pragma("unroll") for (int pl = 0; pl < 3; ++pl) FB[j][pl] =  \
            *reinterpret_cast<const bf16x8*>(rB_ + pl * PLANE_B + (wn + j * 32 + lcol) * X6_ROWB + koff_);           \
    }
#define X6_TERM(FA, FB, PA_, PB_)                                                                                   \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j) acc[i][j] =        \
        __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[i][PA_], FB[j][PB_], acc[i][j], 0, 0, 0);
#ifdef X6_DIAG_NOMMA    // diagnostic build: one MFMA term instead of six (wrong numbers), isolates the consumers' MFMA cost
#define X6_MMA(FA, FB) X6_TERM(FA, FB, 0, 0)
#else
#define X6_MMA(FA, FB)  \
    X6_TERM(FA, FB, 2, 0) X6_TERM(FA, FB, 1, 1) X6_TERM(FA, FB, 0, 2) X6_TERM(FA, FB, 1, 0) X6_TERM(FA, FB, 0, 1) X6_TERM(FA, FB, 0, 0)
#endif
    int buf = 0;
    x6_lds_barrier();   // flat tile 0 is visible
    X6_READ(fa0, fb0, lds[0], 0)
    for (; w < last; w += stride) {
        X6_STAMP(0, 10)
        const X6Item it = x6_item<BN>(p, w);
        f32x16 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        X6_STAMP(0, 11)
        for (int kt = 0; kt < it.nk; ++kt) {
            X6_READ(fa1, fb1, lds[buf], 1)
            __builtin_amdgcn_sched_barrier(0);
            X6_MMA(fa0, fb0)
            __builtin_amdgcn_sched_barrier(0);
            // the next flat tile (of this item, or the first one of the next item: its fragments then stay in
            // registers across the epilogue) is complete behind this barrier; none after the very last tile
            if (kt + 1 < it.nk || w + stride < last) {
                X6_STAMP(0, 12)
                x6_lds_barrier();
                X6_STAMP(0, 13)
                X6_READ(fa0, fb0, lds[buf ^ 1], 0)
            }
            __builtin_amdgcn_sched_barrier(0);
            X6_MMA(fa1, fb1)
            __builtin_amdgcn_sched_barrier(0);
            buf ^= 1;
        }
        X6_STAMP(0, 14)
        const bool add_bias = it.bias != nullptr && it.ks == 0;
        if (staged) {
            float* ct = cstrip[wave];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int cl = wn + j * 32 + lcol;
                    const float bv = (add_bias && it.n0 + cl < p.N) ? it.bias[it.n0 + cl] : 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        ct[((r & 3) + 8 * (r >> 2) + 4 * lrow) * CP + j * 32 + lcol] = p.alpha * acc[i][j][r] + bv;
                }
                // same wave wrote and reads the strip: LDS operations of one wave execute in order (the wave barrier only
                // keeps the compiler from moving the reads above the writes)
                __builtin_amdgcn_wave_barrier();
                constexpr int CPR = WN / 4, NQ = 32 * CPR / 64;   // float4 chunks per strip row / per lane
                const int r0 = it.m0 + wm + i * 32, c0 = it.n0 + wn;
                if (r0 + 32 <= p.M && c0 + WN <= p.N) {
                    // interior strip (wave-uniform test): all reads first, then all stores, no per-lane branches
                    float4 v[NQ];
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        const int c = lane + 64 * q;
                        v[q] = *reinterpret_cast<const float4*>(&ct[(c / CPR) * CP + (c % CPR) * 4]);
                    }
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        const int c = lane + 64 * q;
#ifdef X6_DIAG_NOSTORE   // diagnostic build: C is not written (isolates the store stream's cost)
                        asm volatile("" ::"v"(v[q].x), "v"(v[q].y), "v"(v[q].z), "v"(v[q].w));
#else
                        *reinterpret_cast<float4*>(it.C + (int64_t)(r0 + c / CPR) * p.ldc + c0 + (c % CPR) * 4) = v[q];
#endif
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        const int c = lane + 64 * q;
                        const int row = c / CPR, col = (c % CPR) * 4;
                        const int gr = r0 + row, gc = c0 + col;
                        if (gr >= p.M || gc >= p.N) continue;
                        const float* src = &ct[row * CP + col];
                        float* dst = it.C + (int64_t)gr * p.ldc + gc;
                        dst[0] = src[0];
                        if (gc + 1 < p.N) dst[1] = src[1];
                        if (gc + 2 < p.N) dst[2] = src[2];
                        if (gc + 3 < p.N) dst[3] = src[3];
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        } else {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int col = it.n0 + wn + j * 32 + lcol;
                    if (col >= p.N) continue;
                    const float bv = add_bias ? it.bias[col] : 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = it.m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lrow;
                        if (row < p.M) {
                            const float v = p.alpha * acc[i][j][r] + bv;
                            float* dst = it.C + (int64_t)row * p.ldc + col;
                            if (p.atomic)
                                unsafeAtomicAdd(dst, v);
                            else
                                *dst = v;
                        }
                    }
                }
            }
        }
    }
#undef X6_READ
#undef X6_TERM
#undef X6_MMA
}

// ------------------------------------------------------------------------------------------------------------
// Twelve-wave form of the persistent kernel (768 threads, three waves per SIMD): waves 0-3 consume as before, waves
// 4-7 produce the A operand's planes and waves 8-11 the B operand's.  Measured on gfx950 (tools/micro/valu_rate): one
// wave issues a VALU instruction every ~4.5 clocks while the SIMD retires one every ~2.2, so a single producer wave
// per SIMD leaves half of the conversion throughput unused and the K step was bound by it (consumers waited ~2000 of
// every ~4000 clocks).  Two producer waves per SIMD halve the per-wave conversion work.  Three waves per SIMD leave 168
// registers per wave, so the consumers keep ONE set of operand fragments and refill each plane as soon as its last
// MFMA of the slice has issued (only plane 0, which is used last and needed first, is double-buffered):
//     order per k-slice   l.h  h.l  m.m  m.h  h.m  h.h      (A plane, B plane)
//     refill after        A.l  B.l   -   A.m  B.m   -        A.h / B.h of the next slice load at the slice start
// ------------------------------------------------------------------------------------------------------------
#define X6Q_WAIT1(S, CNT)                                                                                              \
    if (NI_ == 4)                                                                                                      \
        asm volatile("s_waitcnt vmcnt(" #CNT ")" : "+v"(S.v[0]), "+v"(S.v[NI_ > 1 ? 1 : 0]), "+v"(S.v[NI_ > 2 ? 2 : 0]), \
                     "+v"(S.v[NI_ > 3 ? 3 : 0])::"memory");                                                            \
    else if (NI_ == 2)                                                                                                 \
        asm volatile("s_waitcnt vmcnt(" #CNT ")" : "+v"(S.v[0]), "+v"(S.v[NI_ > 1 ? 1 : 0])::"memory");                \
    else                                                                                                               \
        asm volatile("s_waitcnt vmcnt(" #CNT ")" : "+v"(S.v[0])::"memory");
// the oldest ring stage has landed: two younger stages (2 * NI_ loads) may stay in flight
#define X6Q_WAIT_STAGE(S)                                                                      \
    if (NI_ == 4) { X6Q_WAIT1(S, 8) } else if (NI_ == 2) { X6Q_WAIT1(S, 4) } else { X6Q_WAIT1(S, 2) }

// One operand's producer waves (256 threads): flat stream of K tiles over the workgroup's items, as in the 8-wave
// kernel, for the A operand (IS_B false: BT = 128 rows of M) or the B operand (BT = BN rows of N).
// PIPE3 (fp16x3 kernel, -DX3_PIPE3): THREE LDS images and delayed visibility -- a producer does not wait for its plane
// writes before the barrier of the step that issued them (LDS writes run at ~ 80 B / clock / CU, tools/micro/
// mfma_valu_overlap.hip: the 32 KB of a K step drain for 400+ clocks) but before its NEXT writes; barrier #j therefore
// publishes tile j - 1, the consumers run one barrier behind, and one extra barrier at the end publishes the last tile.
template <int BN, int BT, bool KC, bool IS_B, bool SWZ, int G = 0, bool X3 = false, bool ONE = false, bool PIPE3 = false>
__device__ __forceinline__ void x6q_produce(const GemmArgs& p, int w, int stride, int last, unsigned char* lds0, int buf_bytes,
                                            int plane_off, int pt, int* expo0 = nullptr) {
    constexpr int BK = X6_BK;
    constexpr int NI_ = SplitLoader<BT, KC, SWZ, X3, ONE>::NI;
    static_assert(NI_ == 4 || NI_ == 2 || NI_ == 1, "unexpected ring stage size");
    static_assert(G == 0 || (G == 1 && KC && !IS_B) || ((G == 2 || G == 3) && !KC && IS_B), "gather mode vs operand layout");
    SplitLoader<BT, KC, SWZ, X3, ONE> s0, s1, s2;
    X6_BC_DECL
#ifndef X3_NO_SETPRIO
    // the producers are the critical path of a K step (tools/gemm_kstep_summary.py: the consumers wait ~ 900 of ~ 2 600 clocks
    // at the barrier): their instructions go first whenever they can issue, the matrix wave fills the rest
    if (X3) __builtin_amdgcn_s_setprio(3);
#endif
#ifdef X6_DIAG_TIMING
    const int lane = pt & 63, wave = IS_B ? 100 : 4 + (pt >> 6);   // (stamps: the first A-producer wave)
    int dbgn = 0;
#endif
    typename SplitLoader<BT, KC, SWZ, X3, ONE>::PixRows pr;   // (mode 1 only; dead otherwise)
    typename SplitLoader<BT, KC, SWZ, X3, ONE>::PlainRows plr;   // (plain operands only)
    typename SplitLoader<BT, KC, SWZ, X3, ONE>::PixTap ptap;     // (mode 1 only)
    X3Expo erun;                                         // fp16x3 form: running sub-block exponent of the item (store_x3)
    erun.reset();
    int* const expo = expo0 + (IS_B ? 4 : 0) + (pt >> 6);   // this wave's word in image 0 (image 1: + 8)
    const int ld = (int)(IS_B ? p.ldb : p.lda), tmax = IS_B ? p.N : p.M;
    const int ext = (int)((IS_B ? p.extB : p.extA) * 4);
    X6Item itL = x6_item<BN>(p, w), itS = itL;
    int wL = w, tL = 0, wS = w, tS = 0, buf = 0;
    bool moreL = true, moreS = true;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(IS_B ? itL.B : itL.A), 0, ext, 0x00020000);
    if (G == 1) { pr.setup(p.cg, itL.m0, tmax, pt); ptap.start(p.cg, pr, itL.kbeg, pt); }
    if (G == 0 || G == 3) plr.setup(ld, IS_B ? itL.n0 : itL.m0, tmax, pt);
    // Row sums of an m-contiguous A operand (the bias gradient riding on the weight-gradient contraction): this thread
    // holds the same four rows (pt >> 3) * 4 .. + 3 in every K tile, so it keeps four running sums over its k lines; at
    // the end of an item the eight threads of a row group are combined and one of them adds the result to rowsum[m].
    constexpr bool RSUM = !KC && !IS_B && G == 0;
    x6_f32x4 rsum = {0.f, 0.f, 0.f, 0.f};
#define X6Q_RSUM_ACC(S)                                                                                     \
    if (RSUM && itS.rowsum) {                                                                               \
        const int gk = itS.kbeg + tS * BK + (pt & 7) * 4;                                                   \
        _Pragma("unroll") for (int i = 0; i < NI_; ++i) if (gk + i < itS.kend) rsum += S.v[i];              \
    }
#define X6Q_RSUM_FLUSH                                                                                      \
    if (RSUM && itS.rowsum) {                                                                               \
        _Pragma("unroll") for (int o = 1; o < 8; o <<= 1) {                                                 \
            rsum.x += __shfl_xor(rsum.x, o); rsum.y += __shfl_xor(rsum.y, o);                               \
            rsum.z += __shfl_xor(rsum.z, o); rsum.w += __shfl_xor(rsum.w, o);                               \
        }                                                                                                   \
        if ((pt & 7) == 0) {                                                                                \
            const int m = itS.m0 + (pt >> 3) * 4;                                                           \
            float* dst = itS.rowsum + m;                                                                    \
            if (p.atomic) {                                                                                 \
                if (m + 0 < p.M) unsafeAtomicAdd(dst + 0, rsum.x);                                          \
                if (m + 1 < p.M) unsafeAtomicAdd(dst + 1, rsum.y);                                          \
                if (m + 2 < p.M) unsafeAtomicAdd(dst + 2, rsum.z);                                          \
                if (m + 3 < p.M) unsafeAtomicAdd(dst + 3, rsum.w);                                          \
            } else {                                                                                        \
                if (m + 0 < p.M) dst[0] = rsum.x;                                                           \
                if (m + 1 < p.M) dst[1] = rsum.y;                                                           \
                if (m + 2 < p.M) dst[2] = rsum.z;                                                           \
                if (m + 3 < p.M) dst[3] = rsum.w;                                                           \
            }                                                                                               \
        }                                                                                                   \
        rsum = x6_f32x4{0.f, 0.f, 0.f, 0.f};                                                                \
    }
#define X6Q_LD(S)                                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    if (G == 1) {                                                                                           \
        S.load_taps(rs, ptap);                                                                              \
        if (tL + 1 < itL.nk) ptap.advance(p.cg, pr, pt);                                                    \
    } else if (G == 2)                                                                                        \
        S.load_pixk(rs, p.cg, itL.n0, itL.kbeg + tL * BK, itL.kend, pt, moreL);                             \
    else if (G == 3)                                                                                        \
        S.load_kremap_plain(rs, p.cg, ld, plr, itL.n0, tmax, itL.kbeg + tL * BK);                           \
    else                                                                                                    \
        S.load_plain(rs, ld, plr, IS_B ? itL.n0 : itL.m0, tmax, itL.kbeg + tL * BK, itL.kend, pt);          \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    if (++tL >= itL.nk) {                                                                                   \
        tL = 0;                                                                                             \
        wL += stride;                                                                                       \
        moreL = wL < last;                                                                                  \
        itL = x6_item<BN>(p, moreL ? wL : last - 1);                                                        \
        /* past the last item: a zero-length descriptor (plain loads return zeros without a per-load select) */ \
        rs = __builtin_amdgcn_make_buffer_rsrc((void*)(IS_B ? itL.B : itL.A), 0, (G != 2 && !moreL) ? 0 : ext, 0x00020000); \
        if (G == 1) { pr.setup(p.cg, itL.m0, tmax, pt); ptap.start(p.cg, pr, itL.kbeg, pt); }                \
        if (G == 0 || G == 3) plr.setup(ld, IS_B ? itL.n0 : itL.m0, tmax, pt);                              \
    }
#define X6Q_STEP(S)                                                                                         \
    X6_STAMP(1, 0)                                                                                          \
    X6Q_WAIT_STAGE(S)                                                                                       \
    if (PIPE3) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   /* the previous tile's plane writes have landed */ \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    X6_STAMP(1, 1)                                                                                          \
    if (X3)                                                                                                 \
        S.store_x3(lds0 + buf * buf_bytes + plane_off, pt, itS.kbeg + tS * BK, itS.kend, erun, expo + buf * 8); \
    else                                                                                                    \
        S.store(lds0 + buf * buf_bytes + plane_off, pt, itS.kbeg + tS * BK, itS.kend);                      \
    X6Q_RSUM_ACC(S)                                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    X6_STAMP(1, 2)                                                                                          \
    if (PIPE3) buf = buf == 2 ? 0 : buf + 1; else buf ^= 1;                                                 \
    if (++tS >= itS.nk) {                                                                                   \
        X6Q_RSUM_FLUSH                                                                                      \
        erun.reset();                                                                                       \
        tS = 0;                                                                                             \
        wS += stride;                                                                                       \
        moreS = wS < last;                                                                                  \
        /* the load cursor runs three K tiles ahead: it normally sits in the item the store cursor enters */  \
        if (moreS) { if (wS == wL) itS = itL; else itS = x6_item<BN>(p, wS); }                              \
    }                                                                                                       \
    X6Q_LD(S)                                                                                               \
    X6_STAMP(1, 3)                                                                                          \
    if (PIPE3) {        /* flat tile g - 1 is visible (waited for above); tile g's writes stay in flight */  \
        asm volatile("" ::: "memory");                                                                      \
        __builtin_amdgcn_s_barrier();                                                                       \
        asm volatile("" ::: "memory");                                                                      \
    } else                                                                                                  \
        x6_lds_barrier();   /* flat tile g is visible; the consumers are done reading tile g - 1 */       \
    X6_STAMP(1, 4)
    X6Q_LD(s0)
    X6Q_LD(s1)
    X6Q_LD(s2)
    for (;;) {
        X6Q_STEP(s0)
        if (!moreS) break;
        X6Q_STEP(s1)
        if (!moreS) break;
        X6Q_STEP(s2)
        if (!moreS) break;
    }
    if (PIPE3) x6_lds_barrier();   // the last tile's writes have landed: the extra barrier publishes it
#undef X6Q_LD
#undef X6Q_STEP
#undef X6Q_RSUM_ACC
#undef X6Q_RSUM_FLUSH
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // nothing may be in flight when the wave ends
}

// DEFER (mode 4): compact swizzled operand image (SplitLoader SWZ) + one full-size private C strip per consumer wave, and
// the C tile of item i leaves the strip as 16-byte stores interleaved into the MFMA stream of item i + 1.  Measured
// (tools/micro/store_overlap, tools/tile_fill_nt.py): a wave completes one 1-KB store per ~300 clocks however it is
// issued, but only stalls when the NEXT store comes sooner -- MFMAs issue underneath.  Back to back the 16 stores of a
// sub-tile stall a consumer ~4600 clocks per item (a whole K = 64 item is ~8000 clocks of MFMA).
// NC = 8 (mode 5, experiment): eight consumer waves of (64 x 32) / (32 x 32) sub-tiles beside the eight producers
// (four waves per SIMD, 128 registers each): two MFMA-issuing waves per SIMD.
template <int BN, bool A_KC, bool B_KC, bool DEFER, int NC = 4, int GA = 0, int GB = 0>
__global__ __launch_bounds__((NC + 8) * 64, 1) void gemm_f32_bf16x6_p12_kernel(GemmArgs p, int total_items) {
    X6_BC_DECL
    static_assert(NC == 4 || (NC == 8 && !DEFER && BN >= 64), "consumer wave count");
    constexpr int BM = X6_BT;
    constexpr int ROWB = DEFER ? 64 : X6_ROWB;
    constexpr int PLANE_A = BM * ROWB, PLANE_B = BN * ROWB, BUF = 3 * (PLANE_A + PLANE_B);
    constexpr int WN = NC == 8 ? 32 : (BN >= 64 ? BN / 2 : 32), WM = BM * BN / (NC * WN), TM = WM / 32, TN = WN / 32;
    constexpr int NWN = BN / WN;   // consumer waves along n
    __shared__ __attribute__((aligned(16))) unsigned char lds[2][BUF];
    constexpr int CP = DEFER ? WN : WN + 4, SROWS = DEFER ? WM : 32;
    __shared__ __attribute__((aligned(16))) float cstrip[NC][SROWS * CP];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int dbgn = 0;
    (void)dbgn;
    const int per_xcd = (total_items + 7) >> 3, xcd = blockIdx.x & 7;
    const int stride = gridDim.x >> 3, last = min(total_items, (xcd + 1) * per_xcd);
    int w = xcd * per_xcd + (blockIdx.x >> 3);
    if (w >= last) return;
    const bool staged = !p.atomic && p.c_vec;

    if (wave >= NC + 4) {
        x6q_produce<BN, BN, B_KC, true, DEFER, GB>(p, w, stride, last, &lds[0][0], BUF, 3 * PLANE_A, tid - (NC + 4) * 64);
        return;
    }
    if (wave >= NC) {
        x6q_produce<BN, BM, A_KC, false, DEFER, GA>(p, w, stride, last, &lds[0][0], BUF, 0, tid - NC * 64);
        return;
    }

    // ---------------------------------------------------- consumers ----------------------------------------------------
    const int wm = (wave / NWN) * WM, wn = (wave % NWN) * WN;
    const int lrow = lane >> 5, lcol = lane & 31;
    // byte offset of this lane's 16-byte fragment chunk inside its row, for k-slice 0 / 1 (compact image: swizzled with
    // bits 2-3 of the row; the fragment row is wm|wn + 32 i + lcol with wm, wn multiples of 32)
    const int swz = DEFER ? (lcol >> 2) & 3 : 0;
    const int ko0 = (lrow ^ swz) * 16, ko1 = ((2 + lrow) ^ swz) * 16;
    bf16x8 a0x[TM], a0y[TM], a1[TM], a2[TM], b0x[TN], b0y[TN], b1[TN], b2[TN];
#define X6Q_LDA(DST, PL, BASE, S)                                                                                    \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) DST[i] = *reinterpret_cast<const bf16x8*>(                        \
        (BASE) + (PL) * PLANE_A + (wm + i * 32 + lcol) * ROWB + ((S) ? ko1 : ko0));
#define X6Q_LDB(DST, PL, BASE, S)                                                                                    \
    _Pragma("unroll") for (int j = 0; j < TN; ++j) DST[j] = *reinterpret_cast<const bf16x8*>(                        \
        (BASE) + 3 * PLANE_A + (PL) * PLANE_B + (wn + j * 32 + lcol) * ROWB + ((S) ? ko1 : ko0));
#define X6Q_MM(FA, FB)                                                                                               \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j) acc[i][j] =         \
        __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[i], FB[j], acc[i][j], 0, 0, 0);
#define X6Q_SB __builtin_amdgcn_sched_barrier(0);
    // one k-slice on fragments (A0C, a1, a2, B0C, b1, b2); meanwhile the fragments of the NEXT slice (LDS buffer NBASE,
    // slice NS) are fetched: plane 0 into (A0N, B0N), the others in place as soon as their last MFMA has issued
    // Deferred C stores of the previous item.  Per hook: store the strip chunk held in vq (unit dq), read the next one.
    // The hooks are straight-line code (a branch inside the MFMA stream costs ~8 %: the waitcnt pass drains the
    // rolling fragment loads at every block boundary), so K steps come in two copies -- with hooks while units are
    // pending, plain otherwise -- and a hook past the last unit re-stores the last unit (idempotent, <= 5 per item).
    // Hooks are >= ~300 clocks apart (three per slice on 4-MFMA groups, one per slice on narrower tiles): a wave
    // completes one 1-KB store per ~300 clocks and only stalls when the next store comes sooner.
    float* const ct = cstrip[wave];
    constexpr int CPR = WN / 4, RPU = 64 / CPR;            // float4 chunks per strip row; strip rows per 64-lane unit
    const int sl = (lane / CPR) * CP + (lane % CPR) * 4;   // this lane's float offset inside a unit
    float* dC = nullptr;                                   // this lane's address in unit 0 of the pending tile
    int64_t dstep = 0;                                     // floats between consecutive units (RPU rows of C)
    int dn = 0, dq = 0;                                    // units of the pending tile / units stored so far
    float4 vq = make_float4(0.f, 0.f, 0.f, 0.f);           // strip chunk of unit min(dq, dn - 1)
#define X6Q_HOOK                                                                                                      \
    {                                                                                                                 \
        *reinterpret_cast<float4*>(dC + min(dq, dn - 1) * dstep) = vq;                                                \
        ++dq;                                                                                                         \
        vq = *reinterpret_cast<const float4*>(&ct[sl + min(dq, dn - 1) * (RPU * CP)]);                                \
        X6Q_SB                                                                                                        \
    }
#define X6Q_HOOK3 if (TN == 2) X6Q_HOOK
#define X6Q_NOHOOK
#ifdef X6_DIAG_HALFMMA   // diagnostic build (wrong numbers): three of the six products, everything else unchanged
#define X6Q_SLICE_(A0C, B0C, A0N, B0N, NBASE, NS, H3, H1)                                                              \
    X6Q_LDA(A0N, 0, NBASE, NS) X6Q_LDB(B0N, 0, NBASE, NS) X6Q_SB                                                      \
    X6Q_MM(a2, B0C) X6Q_SB X6Q_LDA(a2, 2, NBASE, NS) X6Q_SB                                                           \
    X6Q_LDB(b2, 2, NBASE, NS) X6Q_SB H3                                                                               \
    X6Q_MM(a1, b1) X6Q_SB                                                                                             \
    X6Q_LDA(a1, 1, NBASE, NS) X6Q_SB H3                                                                               \
    X6Q_LDB(b1, 1, NBASE, NS) X6Q_SB                                                                                  \
    X6Q_MM(A0C, B0C) X6Q_SB H1
#else
#define X6Q_SLICE_(A0C, B0C, A0N, B0N, NBASE, NS, H3, H1)                                                              \
    X6Q_LDA(A0N, 0, NBASE, NS) X6Q_LDB(B0N, 0, NBASE, NS) X6Q_SB                                                      \
    X6Q_MM(a2, B0C) X6Q_SB X6Q_LDA(a2, 2, NBASE, NS) X6Q_SB                                                           \
    X6Q_MM(A0C, b2) X6Q_SB X6Q_LDB(b2, 2, NBASE, NS) X6Q_SB H3                                                        \
    X6Q_MM(a1, b1) X6Q_SB                                                                                             \
    X6Q_MM(a1, B0C) X6Q_SB X6Q_LDA(a1, 1, NBASE, NS) X6Q_SB H3                                                        \
    X6Q_MM(A0C, b1) X6Q_SB X6Q_LDB(b1, 1, NBASE, NS) X6Q_SB                                                           \
    X6Q_MM(A0C, B0C) X6Q_SB H1
#endif
#define X6Q_SLICE(A0C, B0C, A0N, B0N, NBASE, NS) X6Q_SLICE_(A0C, B0C, A0N, B0N, NBASE, NS, X6Q_NOHOOK, X6Q_NOHOOK)
#define X6Q_SLICE_H(A0C, B0C, A0N, B0N, NBASE, NS) X6Q_SLICE_(A0C, B0C, A0N, B0N, NBASE, NS, X6Q_HOOK3, X6Q_HOOK)
    // one K step = two k-slices around the flat-tile barrier
#define X6Q_KSTEP(SL)                                                                                                 \
    {                                                                                                                 \
        X6Q_SB                                                                                                        \
        SL(a0x, b0x, a0y, b0y, lds[buf], 1)                                                                           \
        /* every read of this tile has been issued; the next flat tile (of this item or the next one) is complete  */ \
        /* behind the barrier -- none after the very last tile (its prefetch then re-reads this buffer, unused)    */ \
        const bool more = kt + 1 < it.nk || w + stride < last;                                                        \
        if (more) x6_lds_barrier();                                                                                   \
        const unsigned char* nb = lds[more ? buf ^ 1 : buf];                                                          \
        X6Q_SB                                                                                                        \
        SL(a0y, b0y, a0x, b0x, nb, 0)                                                                                 \
        buf ^= 1;                                                                                                     \
    }
    int buf = 0;
    x6_lds_barrier();   // flat tile 0 is visible
    X6Q_LDA(a0x, 0, lds[0], 0) X6Q_LDA(a1, 1, lds[0], 0) X6Q_LDA(a2, 2, lds[0], 0)
    X6Q_LDB(b0x, 0, lds[0], 0) X6Q_LDB(b1, 1, lds[0], 0) X6Q_LDB(b2, 2, lds[0], 0)
    for (; w < last; w += stride) {
        X6_STAMP(0, 10)
        const X6Item it = x6_item<BN>(p, w);
        f32x16 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        X6_STAMP(0, 11)
        for (int kt = 0; kt < it.nk; ++kt) {
            if (DEFER && dq < dn)
                X6Q_KSTEP(X6Q_SLICE_H)
            else
                X6Q_KSTEP(X6Q_SLICE)
        }
        X6_STAMP(0, 14)
        const bool add_bias = it.bias != nullptr && it.ks == 0;
        if (staged && DEFER) {
            // whatever is still pending of the previous tile goes out now (short K loops), then the strip is refilled
            while (dq < dn) X6Q_HOOK
            dn = dq = 0;
            X6_STAMP(0, 15)
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int cl = wn + j * 32 + lcol;
                    const float bv = (add_bias && it.n0 + cl < p.N) ? it.bias[it.n0 + cl] : 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        ct[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lrow) * CP + j * 32 + lcol] = p.alpha * acc[i][j][r] + bv;
                }
            __builtin_amdgcn_wave_barrier();
            X6_STAMP(0, 16)
            const int r0 = it.m0 + wm, c0 = it.n0 + wn;
            if (r0 + WM <= p.M && c0 + WN <= p.N) {   // interior sub-tile (wave-uniform): deferred, branch-free stores
                dC = it.C + (int64_t)(r0 + lane / CPR) * p.ldc + c0 + (lane % CPR) * 4;
                dstep = (int64_t)RPU * p.ldc;
                dn = WM * CPR / 64;
                dq = 0;
                vq = *reinterpret_cast<const float4*>(&ct[sl]);
            } else {
#pragma unroll 1
                for (int q = 0; q < WM * CPR / 64; ++q) {
                    const int c = lane + 64 * q;
                    const int row = c / CPR, col = (c % CPR) * 4;
                    const int gr = r0 + row, gc = c0 + col;
                    if (gr >= p.M || gc >= p.N) continue;
                    const float* src = &ct[row * CP + col];
                    float* dst = it.C + (int64_t)gr * p.ldc + gc;
                    dst[0] = src[0];
                    if (gc + 1 < p.N) dst[1] = src[1];
                    if (gc + 2 < p.N) dst[2] = src[2];
                    if (gc + 3 < p.N) dst[3] = src[3];
                }
                __builtin_amdgcn_wave_barrier();
            }
        } else if (staged) {
            constexpr int CPR = WN / 4, NQ = 32 * CPR / 64;   // float4 chunks per strip row / per lane
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int cl = wn + j * 32 + lcol;
                    const float bv = (add_bias && it.n0 + cl < p.N) ? it.bias[it.n0 + cl] : 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        ct[((r & 3) + 8 * (r >> 2) + 4 * lrow) * CP + j * 32 + lcol] = p.alpha * acc[i][j][r] + bv;
                }
                __builtin_amdgcn_wave_barrier();
                const int r0 = it.m0 + wm + i * 32, c0 = it.n0 + wn;
                if (r0 + 32 <= p.M && c0 + WN <= p.N) {
                    float4 v[NQ];
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        const int c = lane + 64 * q;
                        v[q] = *reinterpret_cast<const float4*>(&ct[(c / CPR) * CP + (c % CPR) * 4]);
                    }
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        const int c = lane + 64 * q;
                        *reinterpret_cast<float4*>(it.C + (int64_t)(r0 + c / CPR) * p.ldc + c0 + (c % CPR) * 4) = v[q];
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        const int c = lane + 64 * q;
                        const int row = c / CPR, col = (c % CPR) * 4;
                        const int gr = r0 + row, gc = c0 + col;
                        if (gr >= p.M || gc >= p.N) continue;
                        const float* src = &ct[row * CP + col];
                        float* dst = it.C + (int64_t)gr * p.ldc + gc;
                        dst[0] = src[0];
                        if (gc + 1 < p.N) dst[1] = src[1];
                        if (gc + 2 < p.N) dst[2] = src[2];
                        if (gc + 3 < p.N) dst[3] = src[3];
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        } else {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int col = it.n0 + wn + j * 32 + lcol;
                    if (col >= p.N) continue;
                    const float bv = add_bias ? it.bias[col] : 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = it.m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lrow;
                        if (row < p.M) {
                            const float v = p.alpha * acc[i][j][r] + bv;
                            float* dst = it.C + (int64_t)row * p.ldc + col;
                            if (p.atomic)
                                unsafeAtomicAdd(dst, v);
                            else
                                *dst = v;
                        }
                    }
                }
            }
        }
    }
    if (DEFER) {   // the last tile's pending stores
        while (dq < dn) X6Q_HOOK
    }
#undef X6Q_HOOK
#undef X6Q_HOOK3
#undef X6Q_NOHOOK
#undef X6Q_SLICE_
#undef X6Q_SLICE_H
#undef X6Q_KSTEP
#undef X6Q_LDA
#undef X6Q_LDB
#undef X6Q_MM
#undef X6Q_SB
#undef X6Q_SLICE
}

// ------------------------------------------------------------------------------------------------------------
// fp16x3 form of the 12-wave kernel (128 x 128 x 32 tiles).  Same workgroup, same producers' load ring, same item
// stream -- but an operand tile goes to LDS as TWO fp16 planes of x * 2^-E (h + l = 22 significant bits) with one
// exponent E per 32 x 32 sub-block (SplitLoader::store_x3), and a k-slice of 16 is the three products
// l.h + h.l + h.h on v_mfma_f32_32x32x16_f16: half the matrix instructions and two thirds of the LDS traffic of the
// bf16x6 form (whose diagnostic two-plane / three-product build bounded the gain at -27 % of the contraction time).
// A consumer wave keeps, per 32 x 32 accumulator block, the exponent U = E_A + E_B its sums are expressed in; the
// producers never lower E inside an item, so when a tile arrives with a larger exponent the block is scaled down by
// the exact power of two (v_ldexp_f32) before the tile is added -- a wave-uniform branch that data of one magnitude
// never takes.  The epilogue multiplies by 2^U.  Accuracy class of an fp32 dot product
// (tests/test_ops_gpu.py::test_f16x3_kernel_*): dropped l.l terms 2^-22, each element resolved to 2^-25 of its
// sub-block's running maximum.
// ------------------------------------------------------------------------------------------------------------
typedef _Float16 x3_f16x8 __attribute__((ext_vector_type(8)));

template <bool A_KC, bool B_KC, int GA = 0, int GB = 0, bool ONE = false, bool CMAP = false, bool EPI = false>
__global__ __launch_bounds__(768, 1) void gemm_f32_f16x3_p12_kernel(GemmArgs p, int total_items) {
    constexpr int BN = 128, BM = X6_BT, ROWB = X6_ROWB, NC = 4;
    constexpr int PLANE_A = BM * ROWB, PLANE_B = BN * ROWB, BUF = 2 * (PLANE_A + PLANE_B);
    constexpr int WN = 64, WM = 64, TM = 2, TN = 2, NWN = 2;
#ifdef X3_PIPE3
    constexpr bool PIPE3 = true;
#else
    constexpr bool PIPE3 = false;
#endif
    constexpr int NIMG = PIPE3 ? 3 : 2;
    __shared__ __attribute__((aligned(16))) unsigned char lds[NIMG][BUF];
    constexpr int CP = WN + 4;
    __shared__ __attribute__((aligned(16))) float cstrip[NC][32 * CP];
    __shared__ int expo[NIMG][8];   // [image][A sub-blocks 0-3, B sub-blocks 4-7]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per_xcd = (total_items + 7) >> 3, xcd = blockIdx.x & 7;
    const int stride = gridDim.x >> 3, last = min(total_items, (xcd + 1) * per_xcd);
    int w = xcd * per_xcd + (blockIdx.x >> 3);
    if (w >= last) return;
    const bool staged = !p.atomic && p.c_vec;
    // (CMAP, a separate instance of the forward-kind convolution kernel: the row of C a tile row goes to when the launch is a parity
    //  class of conv_bwd_data_s2 writing its pixels of dx in place, ConvGather::cmap.  Separate because the same code behind a
    //  run-time flag cost the plain forward convolutions 10 %.)
    static_assert(!CMAP || (GA == 1 && GB == 0), "row map: forward-kind convolution only");
    auto crow = [&](int m) -> int {
        if (!CMAP) return m;
        const int img = fd_div(m, p.cg.dHW), rem = m - img * (p.cg.gH * p.cg.gW);
        const int ci = fd_div(rem, p.cg.dW), cj = rem - ci * p.cg.gW;
        return (img * p.cg.cH + 2 * ci + p.cg.cpy) * p.cg.cW + 2 * cj + p.cg.cpx;
    };
#define X3_CROW(M_) crow(M_)

    if (wave >= NC + 4) {
        x6q_produce<BN, BN, B_KC, true, false, GB, true, ONE, PIPE3>(p, w, stride, last, &lds[0][0], BUF, 2 * PLANE_A, tid - (NC + 4) * 64,
                                                                     &expo[0][0]);
        return;
    }
    if (wave >= NC) {
        x6q_produce<BN, BM, A_KC, false, false, GA, true, ONE, PIPE3>(p, w, stride, last, &lds[0][0], BUF, 0, tid - NC * 64, &expo[0][0]);
        return;
    }

    // ---------------------------------------------------- consumers ----------------------------------------------------
#ifdef X3_M16
    // Round-4 experiment (-DX3_M16; NOT the product build): v_mfma_f32_16x16x32_f16 instead of 32x32x16.
    // tools/micro/mfma_valu_overlap.hip: a back-to-back stream of 32x32x16 instructions (16 result registers per 32 clocks)
    // starves the LDS reads and global loads of the SIMD's other waves (<= 10 % of their time hidden: they run after the matrix
    // wave), under 16x16x32 (4 result registers per 16 clocks, the same matrix-pipe time per product) 81 % of the LDS read time
    // and 47 % of the global load time is hidden.  In THIS kernel it changes nothing (tools/w256_bench.py, 20 shapes: 2 640 us
    // either way): the consumers are not its critical path -- tools/gemm_kstep_summary.py: they wait ~ 900 of every ~ 2 600
    // clocks at the barrier for the producers.  Kept for the record; passes the same accuracy tests (different rounding order
    // inside a K tile, so not bit-identical to the 32x32x16 form).  The wave's 64 x 64 outputs are 4 x 4 blocks of
    // 16 x 16; one instruction spans the whole K tile; the tile is worked off in four quarters (A half x B half: 12
    // instructions each) so that only 64 fragment registers are live: q1 (A0 B0) while A1, B1 of the tile are fetched, q2
    // (A0 B1), barrier, q3 (A1 B0) while the next tile's A0 arrives, q4 (A1 B1) while its B0 arrives.
    const int wm = (wave / NWN) * WM, wn = (wave % NWN) * WN;
    const int l16 = lane & 15, lg = lane >> 4;
    const int lrow = lg, lcol = l16;
    (void)lrow; (void)lcol;
    x3_f16x8 fa[2][2][2], fb[2][2][2];   // [half of the wave's rows][plane h, l][16-row block]
    X6_BC_DECL
#define M16_LDA(H, BASE)                                                                                              \
    _Pragma("unroll") for (int pl = 0; pl < (ONE ? 1 : 2); ++pl) _Pragma("unroll") for (int b = 0; b < 2; ++b)          \
        fa[H][pl][b] = *reinterpret_cast<const x3_f16x8*>((BASE) + pl * PLANE_A + (wm + (H) * 32 + b * 16 + l16) * ROWB + lg * 16);
#define M16_LDB(H, BASE)                                                                                              \
    _Pragma("unroll") for (int pl = 0; pl < (ONE ? 1 : 2); ++pl) _Pragma("unroll") for (int b = 0; b < 2; ++b)          \
        fb[H][pl][b] = *reinterpret_cast<const x3_f16x8*>((BASE) + 2 * PLANE_A + pl * PLANE_B + (wn + (H) * 32 + b * 16 + l16) * ROWB + lg * 16);
#define M16_T(HA, HB, PA_, PB_)                                                                                       \
    _Pragma("unroll") for (int ba = 0; ba < 2; ++ba) _Pragma("unroll") for (int bb = 0; bb < 2; ++bb)                   \
        acc[2 * (HA) + ba][2 * (HB) + bb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[HA][PA_][ba], fb[HB][PB_][bb], acc[2 * (HA) + ba][2 * (HB) + bb], 0, 0, 0);
#define X3Q_SB __builtin_amdgcn_sched_barrier(0);
#ifdef X3_DIAG_NOMMA   // diagnostic build: one of the three terms (wrong numbers)
#define M16_Q(HA, HB) M16_T(HA, HB, 0, 0) X3Q_SB
#else
#define M16_Q(HA, HB)                                                                                                 \
    if (ONE) { M16_T(HA, HB, 0, 0) X3Q_SB } else { M16_T(HA, HB, 1, 0) M16_T(HA, HB, 0, 1) M16_T(HA, HB, 0, 0) X3Q_SB }
#endif
#define X3Q_EXPO(B_)                                                                                                 \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) eA[i] = __builtin_amdgcn_readfirstlane(expo[B_][wm / 32 + i]);    \
    _Pragma("unroll") for (int j = 0; j < TN; ++j) eB[j] = __builtin_amdgcn_readfirstlane(expo[B_][4 + wn / 32 + j]);
    int buf = 0;
    int eA[TM], eB[TN];
    x6_lds_barrier();   // flat tile 0 is visible
    if (PIPE3) x6_lds_barrier();   // (... one barrier later: see x6q_produce)
    X3Q_EXPO(0)
    M16_LDA(0, lds[0]) M16_LDB(0, lds[0])
    int dbgn = 0;
    (void)dbgn;
    for (; w < last; w += stride) {
        X6_STAMP(0, 10)
        const X6Item it = x6_item<BN>(p, w);
        x6_f32x4 acc[4][4];
        int U[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) U[i][j] = -1000;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = x6_f32x4{0.f, 0.f, 0.f, 0.f};
        for (int kt = 0; kt < it.nk; ++kt) {
            // the tile about to be added is expressed in 2^(eA + eB): bring the sums there first (never upwards)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int E = eA[i] + eB[j];
                    if (E != U[i][j]) {   // wave-uniform; rare: first tile of an item, or a sub-block maximum that grew
                        const int d = max(U[i][j] - E, -400);
#pragma unroll
                        for (int ba = 0; ba < 2; ++ba)
#pragma unroll
                            for (int bb = 0; bb < 2; ++bb)
#pragma unroll
                                for (int r = 0; r < 4; ++r)
                                    acc[2 * i + ba][2 * j + bb][r] = __builtin_amdgcn_ldexpf(acc[2 * i + ba][2 * j + bb][r], d);
                        U[i][j] = E;
                    }
                }
            X3Q_SB
            M16_LDB(1, lds[buf]) M16_LDA(1, lds[buf]) X3Q_SB
            M16_Q(0, 0)
            M16_Q(0, 1)
            const bool more = kt + 1 < it.nk || w + stride < last;
            X6_STAMP(0, 12)
            if (more) x6_lds_barrier();   // (every read of this tile has returned: the producers may overwrite it)
            X6_STAMP(0, 13)
            const int nxt = PIPE3 ? (buf == 2 ? 0 : buf + 1) : (buf ^ 1);
            const int nbuf = more ? nxt : buf;
            const unsigned char* nb = lds[nbuf];
            X3Q_EXPO(nbuf)
            X3Q_SB
            M16_LDA(0, nb) X3Q_SB
            M16_Q(1, 0)
            M16_LDB(0, nb) X3Q_SB
            M16_Q(1, 1)
            X6_STAMP(0, 15)
            buf = nxt;
        }
        X6_STAMP(0, 14)
        const bool add_bias = it.bias != nullptr && it.ks == 0;
        if (staged) {
            float* ct = cstrip[wave];
            constexpr int CPR = WN / 4, NQ = 32 * CPR / 64;   // float4 chunks per strip row / per lane
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int jb = 0; jb < 4; ++jb) {
                    const int cl = wn + jb * 16 + l16;
                    const float bv = (add_bias && it.n0 + cl < p.N) ? it.bias[it.n0 + cl] : 0.f;
                    const int u = max(U[i][jb >> 1], -400);
#pragma unroll
                    for (int ba = 0; ba < 2; ++ba)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            ct[(ba * 16 + 4 * lg + r) * CP + jb * 16 + l16] = p.alpha * __builtin_amdgcn_ldexpf(acc[2 * i + ba][jb][r], u) + bv;
                }
                __builtin_amdgcn_wave_barrier();
                const int r0 = it.m0 + wm + i * 32, c0 = it.n0 + wn;
                if (r0 + 32 <= p.M && c0 + WN <= p.N) {
                    float4 v[NQ];
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        const int c = lane + 64 * q;
                        v[q] = *reinterpret_cast<const float4*>(&ct[(c / CPR) * CP + (c % CPR) * 4]);
                    }
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        const int c = lane + 64 * q;
                        *reinterpret_cast<float4*>(it.C + (int64_t)(r0 + c / CPR) * p.ldc + c0 + (c % CPR) * 4) = v[q];
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        const int c = lane + 64 * q;
                        const int row = c / CPR, col = (c % CPR) * 4;
                        const int gr = r0 + row, gc = c0 + col;
                        if (gr >= p.M || gc >= p.N) continue;
                        const float* src = &ct[row * CP + col];
                        float* dst = it.C + (int64_t)gr * p.ldc + gc;
                        dst[0] = src[0];
                        if (gc + 1 < p.N) dst[1] = src[1];
                        if (gc + 2 < p.N) dst[2] = src[2];
                        if (gc + 3 < p.N) dst[3] = src[3];
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        } else {
#pragma unroll
            for (int ib = 0; ib < 4; ++ib) {
#pragma unroll
                for (int jb = 0; jb < 4; ++jb) {
                    const int col = it.n0 + wn + jb * 16 + l16;
                    if (col >= p.N) continue;
                    const float bv = add_bias ? it.bias[col] : 0.f;
                    const int u = max(U[ib >> 1][jb >> 1], -400);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = it.m0 + wm + ib * 16 + 4 * lg + r;
                        if (row < p.M) {
                            const float v = p.alpha * __builtin_amdgcn_ldexpf(acc[ib][jb][r], u) + bv;
                            float* dst = it.C + (int64_t)row * p.ldc + col;
                            if (p.atomic)
                                unsafeAtomicAdd(dst, v);
                            else
                                *dst = v;
                        }
                    }
                }
            }
        }
    }
#undef M16_LDA
#undef M16_LDB
#undef M16_T
#undef M16_Q
#undef X3Q_SB
#undef X3Q_EXPO
#else   // the product build: 32x32x16 consumers (round 3)
    const int wm = (wave / NWN) * WM, wn = (wave % NWN) * WN;
    const int lrow = lane >> 5, lcol = lane & 31;
    const int ko0 = lrow * 16, ko1 = (2 + lrow) * 16;
    x3_f16x8 ahx[TM], ahy[TM], al[TM], bhx[TN], bhy[TN], bl[TN];
    X6_BC_DECL
#define X3Q_LDA(DST, PL, BASE, S)                                                                                    \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) DST[i] = *reinterpret_cast<const x3_f16x8*>(                      \
        (BASE) + (PL) * PLANE_A + (wm + i * 32 + lcol) * ROWB + ((S) ? ko1 : ko0));
#define X3Q_LDB(DST, PL, BASE, S)                                                                                    \
    _Pragma("unroll") for (int j = 0; j < TN; ++j) DST[j] = *reinterpret_cast<const x3_f16x8*>(                      \
        (BASE) + 2 * PLANE_A + (PL) * PLANE_B + (wn + j * 32 + lcol) * ROWB + ((S) ? ko1 : ko0));
#define X3Q_MM(FA, FB)                                                                                               \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j) acc[i][j] =         \
        __builtin_amdgcn_mfma_f32_32x32x16_f16(FA[i], FB[j], acc[i][j], 0, 0, 0);
#define X3Q_SB __builtin_amdgcn_sched_barrier(0);
    // one k-slice on (A0C, al, B0C, bl); meanwhile the next slice's fragments are fetched: the h planes into (A0N, B0N),
    // the l planes in place as soon as their only product of the slice has issued
#ifdef X3_DIAG_NOMMA   // diagnostic build (tools/gemm_x3_diag.py): one of the three terms (wrong numbers): the consumers' MFMA cost
#define X3Q_SLICE(A0C, B0C, A0N, B0N, NBASE, NS)                                                                     \
    X3Q_LDA(A0N, 0, NBASE, NS) X3Q_LDB(B0N, 0, NBASE, NS) X3Q_SB                                                     \
    X3Q_LDA(al, 1, NBASE, NS) X3Q_SB X3Q_LDB(bl, 1, NBASE, NS) X3Q_SB                                                \
    X3Q_MM(A0C, B0C) X3Q_SB
#else
#define X3Q_SLICE(A0C, B0C, A0N, B0N, NBASE, NS)                                                                     \
    if (ONE) {   /* single-pass form: h planes only, one product per k-slice */                                       \
        X3Q_LDA(A0N, 0, NBASE, NS) X3Q_LDB(B0N, 0, NBASE, NS) X3Q_SB                                                 \
        X3Q_MM(A0C, B0C) X3Q_SB                                                                                      \
    } else {                                                                                                         \
        X3Q_LDA(A0N, 0, NBASE, NS) X3Q_LDB(B0N, 0, NBASE, NS) X3Q_SB                                                 \
        X3Q_MM(al, B0C) X3Q_SB X3Q_LDA(al, 1, NBASE, NS) X3Q_SB                                                      \
        X3Q_MM(A0C, bl) X3Q_SB X3Q_LDB(bl, 1, NBASE, NS) X3Q_SB                                                      \
        X3Q_MM(A0C, B0C) X3Q_SB                                                                                      \
    }
#endif
    // exponents of the image in LDS buffer B_ (wave-uniform values -> scalar registers)
#define X3Q_EXPO(B_)                                                                                                 \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) eA[i] = __builtin_amdgcn_readfirstlane(expo[B_][wm / 32 + i]);    \
    _Pragma("unroll") for (int j = 0; j < TN; ++j) eB[j] = __builtin_amdgcn_readfirstlane(expo[B_][4 + wn / 32 + j]);
    int buf = 0;
    int eA[TM], eB[TN];
    x6_lds_barrier();   // flat tile 0 is visible
    if (PIPE3) x6_lds_barrier();   // (... one barrier later: see x6q_produce)
    X3Q_EXPO(0)
    X3Q_LDA(ahx, 0, lds[0], 0) X3Q_LDB(bhx, 0, lds[0], 0)
    if (!ONE) { X3Q_LDA(al, 1, lds[0], 0) X3Q_LDB(bl, 1, lds[0], 0) }
    int dbgn = 0;
    (void)dbgn;
    for (; w < last; w += stride) {
        X6_STAMP(0, 10)
        const X6Item it = x6_item<BN>(p, w);
        f32x16 acc[TM][TN];
        int U[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                U[i][j] = -1000;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
            }
        for (int kt = 0; kt < it.nk; ++kt) {
            // the tile about to be added is expressed in 2^(eA + eB): bring the sums there first (never upwards)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int E = eA[i] + eB[j];
                    if (E != U[i][j]) {   // wave-uniform; rare: first tile of an item, or a sub-block maximum that grew
                        const int d = max(U[i][j] - E, -400);
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[i][j][r] = __builtin_amdgcn_ldexpf(acc[i][j][r], d);
                        U[i][j] = E;
                    }
                }
            X3Q_SB
            X3Q_SLICE(ahx, bhx, ahy, bhy, lds[buf], 1)
            const bool more = kt + 1 < it.nk || w + stride < last;
            X6_STAMP(0, 12)
            if (more) x6_lds_barrier();
            X6_STAMP(0, 13)
            const int nxt = PIPE3 ? (buf == 2 ? 0 : buf + 1) : (buf ^ 1);
            const int nbuf = more ? nxt : buf;
            const unsigned char* nb = lds[nbuf];
            X3Q_EXPO(nbuf)
            X3Q_SB
            X3Q_SLICE(ahy, bhy, ahx, bhx, nb, 0)
            X6_STAMP(0, 15)
            buf = nxt;
        }
        X6_STAMP(0, 14)
        const bool add_bias = it.bias != nullptr && it.ks == 0;
        if (staged) {
            float* ct = cstrip[wave];
            constexpr int CPR = WN / 4, NQ = 32 * CPR / 64;   // float4 chunks per strip row / per lane
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int cl = wn + j * 32 + lcol;
                    const float bv = (add_bias && it.n0 + cl < p.N) ? it.bias[it.n0 + cl] : 0.f;
                    const int u = max(U[i][j], -400);
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        ct[((r & 3) + 8 * (r >> 2) + 4 * lrow) * CP + j * 32 + lcol] =
                            p.alpha * __builtin_amdgcn_ldexpf(acc[i][j][r], u) + bv;
                }
                __builtin_amdgcn_wave_barrier();
                const int r0 = it.m0 + wm + i * 32, c0 = it.n0 + wn;
                if (EPI) {
                    // affine epilogue (N % 4 == 0, ldc % 4 == 0: whole float4 chunks are inside or outside): the residual chunks are
                    // requested first, the strip is read while they fly
                    const float* rbase = p.epi_res ? p.epi_res + (it.C - p.C) : nullptr;
                    const int gc = c0 + (lane % CPR) * 4;
                    const bool cin = gc < p.N;
                    float4 sc = make_float4(0.f, 0.f, 0.f, 0.f), sh = sc, rr[NQ];
                    if (cin) { sc = *reinterpret_cast<const float4*>(p.epi_scale + gc); sh = *reinterpret_cast<const float4*>(p.epi_shift + gc); }
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        const int gr = r0 + (lane + 64 * q) / CPR;
                        rr[q] = (rbase && cin && gr < p.M) ? *reinterpret_cast<const float4*>(rbase + (int64_t)gr * p.ldc + gc) : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        const int c = lane + 64 * q, gr = r0 + c / CPR;
                        float4 v = *reinterpret_cast<const float4*>(&ct[(c / CPR) * CP + (c % CPR) * 4]);
                        v.x = v.x * sc.x + sh.x + rr[q].x; v.y = v.y * sc.y + sh.y + rr[q].y;
                        v.z = v.z * sc.z + sh.z + rr[q].z; v.w = v.w * sc.w + sh.w + rr[q].w;
                        if (p.epi_relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                        if (cin && gr < p.M) *reinterpret_cast<float4*>(it.C + (int64_t)gr * p.ldc + gc) = v;
                    }
                } else
                if (r0 + 32 <= p.M && c0 + WN <= p.N) {
                    float4 v[NQ];
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        const int c = lane + 64 * q;
                        v[q] = *reinterpret_cast<const float4*>(&ct[(c / CPR) * CP + (c % CPR) * 4]);
                    }
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        const int c = lane + 64 * q;
                        *reinterpret_cast<float4*>(it.C + (int64_t)X3_CROW(r0 + c / CPR) * p.ldc + c0 + (c % CPR) * 4) = v[q];
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        const int c = lane + 64 * q;
                        const int row = c / CPR, col = (c % CPR) * 4;
                        const int gr = r0 + row, gc = c0 + col;
                        if (gr >= p.M || gc >= p.N) continue;
                        const float* src = &ct[row * CP + col];
                        float* dst = it.C + (int64_t)X3_CROW(gr) * p.ldc + gc;
                        dst[0] = src[0];
                        if (gc + 1 < p.N) dst[1] = src[1];
                        if (gc + 2 < p.N) dst[2] = src[2];
                        if (gc + 3 < p.N) dst[3] = src[3];
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        } else {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int col = it.n0 + wn + j * 32 + lcol;
                    if (col >= p.N) continue;
                    const float bv = add_bias ? it.bias[col] : 0.f;
                    const int u = max(U[i][j], -400);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = it.m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lrow;
                        if (row < p.M) {
                            const float v = p.alpha * __builtin_amdgcn_ldexpf(acc[i][j][r], u) + bv;
                            float* dst = it.C + (int64_t)row * p.ldc + col;
                            if (p.atomic)
                                unsafeAtomicAdd(dst, v);
                            else
                                *dst = v;
                        }
                    }
                }
            }
        }
    }
#undef X3Q_LDA
#undef X3Q_LDB
#undef X3Q_MM
#undef X3Q_SB
#undef X3Q_SLICE
#undef X3Q_EXPO
#endif
#undef X3_CROW
}

// ------------------------------------------------------------------------------------------------------------
// 256 x 128 x 32 tiles of the fp16x3 form (round 4): EIGHT waves, two per SIMD -- four consumers of 128 x 64 outputs
// each (TM = 4, TN = 2: 48 matrix instructions per K step and wave) and four producers that each own three 32 x 32
// sub-blocks per K step (two of A, one of B).  Round 4's measurements (profiles/README.md, "What bounds a 128 x 128
// contraction tile") say the 128 x 128 kernel runs out of bytes in flight, not of LDS bandwidth or matrix cycles: this
// tile needs 48 KB of operands per K step for TWICE the products (0.75x the bytes per matrix instruction), keeps three
// register stages x 48 KB in flight per CU, reads 0.75x the LDS bytes per product, and pays one workgroup barrier per
// 96 matrix instructions per SIMD instead of one per 24.  The operand image of a stage is three independent 128-row
// images (A rows 0-127, A rows 128-255, B) of two fp16 planes each, so SplitLoader<128, ., ., X3> serves all three with
// the producer thread index it already expects.  Same arithmetic, same exponents, same epilogue as the 12-wave kernel:
// bit-identical results (tests/test_ops_gpu.py::test_w256_kernel_*).  Plain contractions only (no gathers).
// ------------------------------------------------------------------------------------------------------------
#define W2_WAIT_ASM(S, CNT)                                                                                              \
    asm volatile("s_waitcnt vmcnt(" #CNT ")"                                                                             \
                 : "+v"(S##a0.v[0]), "+v"(S##a0.v[1]), "+v"(S##a0.v[2]), "+v"(S##a0.v[3]), "+v"(S##a1.v[0]), "+v"(S##a1.v[1]), \
                   "+v"(S##a1.v[2]), "+v"(S##a1.v[3]), "+v"(S##b.v[0]), "+v"(S##b.v[1]), "+v"(S##b.v[2]), "+v"(S##b.v[3])::"memory");

template <bool A_KC, bool B_KC>
__device__ __forceinline__ void w2_produce(const GemmArgs& p, int w, int stride, int last, unsigned char* lds0, int pt, int* expo0) {
    constexpr int BK = X6_BK, IMG = 2 * X6_PLANE, BUF = 3 * IMG;
    typedef SplitLoader<128, A_KC, false, true, false> LA;
    typedef SplitLoader<128, B_KC, false, true, false> LB;
    static_assert(LA::NI == 4 && LB::NI == 4, "ring stage = 12 loads");
    LA s0a0, s0a1, s1a0, s1a1, s2a0, s2a1;
    LB s0b, s1b, s2b;
    X6_BC_DECL
#ifndef X3_NO_SETPRIO
    __builtin_amdgcn_s_setprio(3);
#endif
    X3Expo eA0, eA1, eB;                           // running sub-block exponents of the item (store_x3)
    eA0.reset(); eA1.reset(); eB.reset();
    int* const expo = expo0 + (pt >> 6);           // image words: A rows 0-127: 0-3, A rows 128-255: 4-7, B: 8-11
    const int lda = (int)p.lda, ldb = (int)p.ldb;
    const int extA = (int)(p.extA * 4), extB = (int)(p.extB * 4);
    X6Item itL = x6_item<128, 256>(p, w), itS = itL;
    int wL = w, tL = 0, wS = w, tS = 0, buf = 0;
    bool moreL = true, moreS = true;
    __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)itL.A, 0, extA, 0x00020000);
    __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)itL.B, 0, extB, 0x00020000);
    typename LA::PlainRows ra0, ra1;   // per-item row bases of the loads (SplitLoader::load_plain)
    typename LB::PlainRows rb0;
    ra0.setup(lda, itL.m0, p.M, pt);
    ra1.setup(lda, itL.m0 + 128, p.M, pt);
    rb0.setup(ldb, itL.n0, p.N, pt);
    constexpr bool RSUM = !A_KC;   // row sums of an m-contiguous A (see x6q_produce)
    x6_f32x4 rsum0 = {0.f, 0.f, 0.f, 0.f}, rsum1 = {0.f, 0.f, 0.f, 0.f};
#define W2_RSUM_ACC(S)                                                                                      \
    if (RSUM && itS.rowsum) {                                                                               \
        const int gk = itS.kbeg + tS * BK + (pt & 7) * 4;                                                   \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) if (gk + i < itS.kend) { rsum0 += S##a0.v[i]; rsum1 += S##a1.v[i]; } \
    }
#define W2_RSUM_FLUSH1(RS, MOFF)                                                                            \
    {                                                                                                       \
        _Pragma("unroll") for (int o = 1; o < 8; o <<= 1) {                                                 \
            RS.x += __shfl_xor(RS.x, o); RS.y += __shfl_xor(RS.y, o);                                       \
            RS.z += __shfl_xor(RS.z, o); RS.w += __shfl_xor(RS.w, o);                                       \
        }                                                                                                   \
        if ((pt & 7) == 0) {                                                                                \
            const int m = itS.m0 + (MOFF) + (pt >> 3) * 4;                                                  \
            float* dst = itS.rowsum + m;                                                                    \
            if (p.atomic) {                                                                                 \
                if (m + 0 < p.M) unsafeAtomicAdd(dst + 0, RS.x);                                            \
                if (m + 1 < p.M) unsafeAtomicAdd(dst + 1, RS.y);                                            \
                if (m + 2 < p.M) unsafeAtomicAdd(dst + 2, RS.z);                                            \
                if (m + 3 < p.M) unsafeAtomicAdd(dst + 3, RS.w);                                            \
            } else {                                                                                        \
                if (m + 0 < p.M) dst[0] = RS.x;                                                             \
                if (m + 1 < p.M) dst[1] = RS.y;                                                             \
                if (m + 2 < p.M) dst[2] = RS.z;                                                             \
                if (m + 3 < p.M) dst[3] = RS.w;                                                             \
            }                                                                                               \
        }                                                                                                   \
        RS = x6_f32x4{0.f, 0.f, 0.f, 0.f};                                                                  \
    }
#define W2_LD(S)                                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    S##a0.load_plain(rsA, lda, ra0, itL.m0, p.M, itL.kbeg + tL * BK, itL.kend, pt);                         \
    S##a1.load_plain(rsA, lda, ra1, itL.m0 + 128, p.M, itL.kbeg + tL * BK, itL.kend, pt);                   \
    S##b.load_plain(rsB, ldb, rb0, itL.n0, p.N, itL.kbeg + tL * BK, itL.kend, pt);                          \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    if (++tL >= itL.nk) {                                                                                   \
        tL = 0;                                                                                             \
        wL += stride;                                                                                       \
        moreL = wL < last;                                                                                  \
        itL = x6_item<128, 256>(p, moreL ? wL : last - 1);                                                  \
        rsA = __builtin_amdgcn_make_buffer_rsrc((void*)itL.A, 0, moreL ? extA : 0, 0x00020000);             \
        rsB = __builtin_amdgcn_make_buffer_rsrc((void*)itL.B, 0, moreL ? extB : 0, 0x00020000);             \
        ra0.setup(lda, itL.m0, p.M, pt);                                                                    \
        ra1.setup(lda, itL.m0 + 128, p.M, pt);                                                              \
        rb0.setup(ldb, itL.n0, p.N, pt);                                                                    \
    }
#define W2_STEP(S)                                                                                          \
    W2_WAIT_ASM(S, 24)   /* the oldest ring stage has landed: two younger stages (24 loads) stay in flight */ \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    S##a0.store_x3(lds0 + buf * BUF, pt, itS.kbeg + tS * BK, itS.kend, eA0, expo + buf * 12);               \
    S##a1.store_x3(lds0 + buf * BUF + IMG, pt, itS.kbeg + tS * BK, itS.kend, eA1, expo + buf * 12 + 4);     \
    S##b.store_x3(lds0 + buf * BUF + 2 * IMG, pt, itS.kbeg + tS * BK, itS.kend, eB, expo + buf * 12 + 8);   \
    W2_RSUM_ACC(S)                                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    buf ^= 1;                                                                                               \
    if (++tS >= itS.nk) {                                                                                   \
        if (RSUM && itS.rowsum) { W2_RSUM_FLUSH1(rsum0, 0) W2_RSUM_FLUSH1(rsum1, 128) }                     \
        eA0.reset(); eA1.reset(); eB.reset();                                                               \
        tS = 0;                                                                                             \
        wS += stride;                                                                                       \
        moreS = wS < last;                                                                                  \
        if (moreS) { if (wS == wL) itS = itL; else itS = x6_item<128, 256>(p, wS); }                        \
    }                                                                                                       \
    W2_LD(S)                                                                                                \
    x6_lds_barrier();   /* flat tile g is visible; the consumers are done reading tile g - 1 */
    W2_LD(s0)
    W2_LD(s1)
    W2_LD(s2)
    for (;;) {
        W2_STEP(s0)
        if (!moreS) break;
        W2_STEP(s1)
        if (!moreS) break;
        W2_STEP(s2)
        if (!moreS) break;
    }
#undef W2_LD
#undef W2_STEP
#undef W2_RSUM_ACC
#undef W2_RSUM_FLUSH1
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // nothing may be in flight when the wave ends
}

template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(512, 1) void gemm_f32_f16x3_w256_kernel(GemmArgs p, int total_items) {
    constexpr int ROWB = X6_ROWB, PLANE = X6_PLANE, IMG = 2 * PLANE, BUF = 3 * IMG;
    constexpr int TM = 4, TN = 2, WN = 64, CP = WN + 4;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2][BUF];
    __shared__ __attribute__((aligned(16))) float cstrip[4][32 * CP];
    __shared__ int expo[2][12];   // [image][A rows 0-127: 0-3, A rows 128-255: 4-7, B: 8-11]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per_xcd = (total_items + 7) >> 3, xcd = blockIdx.x & 7;
    const int stride = gridDim.x >> 3, last = min(total_items, (xcd + 1) * per_xcd);
    int w = xcd * per_xcd + (blockIdx.x >> 3);
    if (w >= last) return;
    const bool staged = !p.atomic && p.c_vec;

    if (wave >= 4) {
        w2_produce<A_KC, B_KC>(p, w, stride, last, &lds[0][0], tid - 256, &expo[0][0]);
        return;
    }

    // ---------------------------------------------------- consumers ----------------------------------------------------
    const int img = wave >> 1, wn = (wave & 1) * WN;   // A image (rows 128 img .. + 127 of the tile), B rows wn .. wn + 63
    const int lrow = lane >> 5, lcol = lane & 31;
    const int ko0 = lrow * 16, ko1 = (2 + lrow) * 16;
    x3_f16x8 ahx[TM], ahy[TM], al[TM], bhx[TN], bhy[TN], bl[TN];
    X6_BC_DECL
#define W2_LDA(DST, PL, BASE, S)                                                                                     \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) DST[i] = *reinterpret_cast<const x3_f16x8*>(                      \
        (BASE) + img * IMG + (PL) * PLANE + (i * 32 + lcol) * ROWB + ((S) ? ko1 : ko0));
#define W2_LDB(DST, PL, BASE, S)                                                                                     \
    _Pragma("unroll") for (int j = 0; j < TN; ++j) DST[j] = *reinterpret_cast<const x3_f16x8*>(                      \
        (BASE) + 2 * IMG + (PL) * PLANE + (wn + j * 32 + lcol) * ROWB + ((S) ? ko1 : ko0));
#define W2_MM(FA, FB)                                                                                                \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j) acc[i][j] =         \
        __builtin_amdgcn_mfma_f32_32x32x16_f16(FA[i], FB[j], acc[i][j], 0, 0, 0);
#define W2_SB __builtin_amdgcn_sched_barrier(0);
    // one k-slice on (A0C, al, B0C, bl); meanwhile the next slice's fragments are fetched (as in the 12-wave kernel)
#ifdef X3_DIAG_NOMMA   // diagnostic build (tools/w256_diag.py): one of the three terms (wrong numbers)
#define W2_SLICE(A0C, B0C, A0N, B0N, NBASE, NS)                                                                      \
    W2_LDA(A0N, 0, NBASE, NS) W2_LDB(B0N, 0, NBASE, NS) W2_SB                                                        \
    W2_LDA(al, 1, NBASE, NS) W2_SB W2_LDB(bl, 1, NBASE, NS) W2_SB                                                    \
    W2_MM(A0C, B0C) W2_SB
#else
#define W2_SLICE(A0C, B0C, A0N, B0N, NBASE, NS)                                                                      \
    W2_LDA(A0N, 0, NBASE, NS) W2_LDB(B0N, 0, NBASE, NS) W2_SB                                                        \
    W2_MM(al, B0C) W2_SB W2_LDA(al, 1, NBASE, NS) W2_SB                                                              \
    W2_MM(A0C, bl) W2_SB W2_LDB(bl, 1, NBASE, NS) W2_SB                                                              \
    W2_MM(A0C, B0C) W2_SB
#endif
#define W2_EXPO(B_)                                                                                                  \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) eA[i] = __builtin_amdgcn_readfirstlane(expo[B_][4 * img + i]);    \
    _Pragma("unroll") for (int j = 0; j < TN; ++j) eB[j] = __builtin_amdgcn_readfirstlane(expo[B_][8 + wn / 32 + j]);
    int buf = 0;
    int eA[TM], eB[TN];
    x6_lds_barrier();   // flat tile 0 is visible
    W2_EXPO(0)
    W2_LDA(ahx, 0, lds[0], 0) W2_LDB(bhx, 0, lds[0], 0)
    W2_LDA(al, 1, lds[0], 0) W2_LDB(bl, 1, lds[0], 0)
    for (; w < last; w += stride) {
        const X6Item it = x6_item<128, 256>(p, w);
        f32x16 acc[TM][TN];
        int U[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                U[i][j] = -1000;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
            }
        for (int kt = 0; kt < it.nk; ++kt) {
            // the tile about to be added is expressed in 2^(eA + eB): bring the sums there first (never upwards)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int E = eA[i] + eB[j];
                    if (E != U[i][j]) {   // wave-uniform; rare: first tile of an item, or a sub-block maximum that grew
                        const int d = max(U[i][j] - E, -400);
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[i][j][r] = __builtin_amdgcn_ldexpf(acc[i][j][r], d);
                        U[i][j] = E;
                    }
                }
            W2_SB
            W2_SLICE(ahx, bhx, ahy, bhy, lds[buf], 1)
            const bool more = kt + 1 < it.nk || w + stride < last;
            if (more) x6_lds_barrier();
            const int nbuf = more ? buf ^ 1 : buf;
            const unsigned char* nb = lds[nbuf];
            W2_EXPO(nbuf)
            W2_SB
            W2_SLICE(ahy, bhy, ahx, bhx, nb, 0)
            buf ^= 1;
        }
        const bool add_bias = it.bias != nullptr && it.ks == 0;
        if (staged) {
            float* ct = cstrip[wave];
            constexpr int CPR = WN / 4, NQ = 32 * CPR / 64;   // float4 chunks per strip row / per lane
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int cl = wn + j * 32 + lcol;
                    const float bv = (add_bias && it.n0 + cl < p.N) ? it.bias[it.n0 + cl] : 0.f;
                    const int u = max(U[i][j], -400);
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        ct[((r & 3) + 8 * (r >> 2) + 4 * lrow) * CP + j * 32 + lcol] =
                            p.alpha * __builtin_amdgcn_ldexpf(acc[i][j][r], u) + bv;
                }
                __builtin_amdgcn_wave_barrier();
                const int r0 = it.m0 + img * 128 + i * 32, c0 = it.n0 + wn;
                if (r0 + 32 <= p.M && c0 + WN <= p.N) {
                    float4 v[NQ];
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        const int c = lane + 64 * q;
                        v[q] = *reinterpret_cast<const float4*>(&ct[(c / CPR) * CP + (c % CPR) * 4]);
                    }
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        const int c = lane + 64 * q;
                        *reinterpret_cast<float4*>(it.C + (int64_t)(r0 + c / CPR) * p.ldc + c0 + (c % CPR) * 4) = v[q];
                    }
                } else if (r0 < p.M && c0 < p.N) {
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        const int c = lane + 64 * q;
                        const int row = c / CPR, col = (c % CPR) * 4;
                        const int gr = r0 + row, gc = c0 + col;
                        if (gr >= p.M || gc >= p.N) continue;
                        const float* src = &ct[row * CP + col];
                        float* dst = it.C + (int64_t)gr * p.ldc + gc;
                        dst[0] = src[0];
                        if (gc + 1 < p.N) dst[1] = src[1];
                        if (gc + 2 < p.N) dst[2] = src[2];
                        if (gc + 3 < p.N) dst[3] = src[3];
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        } else {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int col = it.n0 + wn + j * 32 + lcol;
                    if (col >= p.N) continue;
                    const float bv = add_bias ? it.bias[col] : 0.f;
                    const int u = max(U[i][j], -400);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = it.m0 + img * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lrow;
                        if (row < p.M) {
                            const float v = p.alpha * __builtin_amdgcn_ldexpf(acc[i][j][r], u) + bv;
                            float* dst = it.C + (int64_t)row * p.ldc + col;
                            if (p.atomic)
                                unsafeAtomicAdd(dst, v);
                            else
                                *dst = v;
                        }
                    }
                }
            }
        }
    }
#undef W2_LDA
#undef W2_LDB
#undef W2_MM
#undef W2_SB
#undef W2_SLICE
#undef W2_EXPO
}

// 0: never; 1 (default): where the cost model prefers it; 2: every eligible contraction (tests).  IX_GEMM_W256 / ix_gemm_set_w256.
static int g_w256 = -1;
static int w256_mode() {
    if (g_w256 < 0) {
        const char* e = getenv("IX_GEMM_W256");
        g_w256 = e ? atoi(e) : 1;
        if (g_w256 < 0 || g_w256 > 2) g_w256 = 1;
    }
    return g_w256;
}
extern "C" int ix_gemm_set_w256(int mode) {
    const int old = w256_mode();
    g_w256 = mode < 0 ? 0 : (mode > 2 ? 2 : mode);
    return old;
}
// cost of a 256 x 128 x 32 step in units of the 12-wave kernel's 128 x 128 x 32 step (IX_W256_RATIO overrides: tuning runs)
static double w256_step_ratio() {
    static double r = -1.0;
    if (r < 0) {
        const char* e = getenv("IX_W256_RATIO");
        r = e ? atof(e) : 0.0;   // 0: the fitted per-layout ratios of gemm_impl
        if (r < 0.5 || r > 4.0) r = 0.0;
    }
    return r;
}
static int64_t g_w256_launches = 0;
extern "C" int ix_gemm_w256_launches(int64_t* out) {
    if (out) *out = g_w256_launches;
    return IX_OK;
}

static void launch_w256(const GemmArgs& a, int a_kc, int b_kc, int items, hipStream_t stream) {
    int g = (items + 7) / 8 * 8;
    if (g > 256) g = 256;
    const dim3 grid(g);
    g_w256_launches += 1;
    if (a_kc && b_kc)
        hipLaunchKernelGGL((gemm_f32_f16x3_w256_kernel<true, true>), grid, dim3(512), 0, stream, a, items);
    else if (a_kc && !b_kc)
        hipLaunchKernelGGL((gemm_f32_f16x3_w256_kernel<true, false>), grid, dim3(512), 0, stream, a, items);
    else if (!a_kc && b_kc)
        hipLaunchKernelGGL((gemm_f32_f16x3_w256_kernel<false, true>), grid, dim3(512), 0, stream, a, items);
    else
        hipLaunchKernelGGL((gemm_f32_f16x3_w256_kernel<false, false>), grid, dim3(512), 0, stream, a, items);
}

// Opt-in (IX_GEMM_KERNEL=x3, or ix_gemm_set_x3 from the tests).  Measured on the step's 24 heaviest shapes: 93.4 vs 108.4 ms
// (-13.8 %; the diagnostic bound without any conversion / exponent work was -27 %), whole step 325-327 vs 338 ms (246 vs 237
// frames/s) with the convolutions still on the bf16x6 form.  Not the default in this round: under its own peak (2500 / 3
// TFLOP/s) its roofline fraction reads 0.23 where the bf16x6 form reads 0.39-0.40 of 2500 / 6, and the bf16x6 form's parity
// record is two rounds old.
static int g_x3k = -1;
static bool x3k_enabled() {
    if (g_x3k < 0) {
        const char* e = getenv("IX_GEMM_KERNEL");
        g_x3k = (e && e[0] == 'x' && e[1] == '6') ? 0 : 1;   // default since round 3: the fp16x3 form; IX_GEMM_KERNEL=x6 = bf16x6 everywhere
    }
    return g_x3k != 0;
}

extern "C" int ix_gemm_set_x3(int on) {
    const int old = x3k_enabled() ? 1 : 0;
    g_x3k = on ? 1 : 0;
    return old;
}

// Single-pass 16-bit mode (MODEL.COMPUTE_DTYPE: bf16 / fp16; hipops.set_compute_dtype): the fp16x3 form's h plane alone -- one
// fp16 value of x * 2^-E per element (11 significant bits, block exponent per 32 x 32 sub-block: at least bf16's accuracy
// with fp32's range), ONE v_mfma_f32_32x32x16_f16 per k-slice, fp32 accumulation.  Process-global like ix_gemm_set_x3; it
// is NOT fp32-grade and never the parity path or the headline.  Returns the previous setting.
static int g_single_pass = 0;
extern "C" int ix_gemm_set_single_pass(int on) {
    const int old = g_single_pass;
    g_single_pass = on ? 1 : 0;
    return old;
}

template <bool ONE>
static void launch_x3q_(const GemmArgs& a, int a_kc, int b_kc, int items, hipStream_t stream) {
    int g = (items + 7) / 8 * 8;
    if (g > 256) g = 256;
    const dim3 grid(g);
    if (a_kc && b_kc)
        hipLaunchKernelGGL((gemm_f32_f16x3_p12_kernel<true, true, 0, 0, ONE>), grid, dim3(768), 0, stream, a, items);
    else if (a_kc && !b_kc)
        hipLaunchKernelGGL((gemm_f32_f16x3_p12_kernel<true, false, 0, 0, ONE>), grid, dim3(768), 0, stream, a, items);
    else if (!a_kc && b_kc)
        hipLaunchKernelGGL((gemm_f32_f16x3_p12_kernel<false, true, 0, 0, ONE>), grid, dim3(768), 0, stream, a, items);
    else
        hipLaunchKernelGGL((gemm_f32_f16x3_p12_kernel<false, false, 0, 0, ONE>), grid, dim3(768), 0, stream, a, items);
}

// forward contraction (both operands k-contiguous) with the affine epilogue in its store
static void launch_x3q_epi(const GemmArgs& a, int items, hipStream_t stream) {
    int g = (items + 7) / 8 * 8;
    if (g > 256) g = 256;
    if (g_single_pass) hipLaunchKernelGGL((gemm_f32_f16x3_p12_kernel<true, true, 0, 0, true, false, true>), dim3(g), dim3(768), 0, stream, a, items);
    else hipLaunchKernelGGL((gemm_f32_f16x3_p12_kernel<true, true, 0, 0, false, false, true>), dim3(g), dim3(768), 0, stream, a, items);
}
static bool persistent_ok_pre(const GemmArgs& a, int nbatch, int split) {
    return (int64_t)a.tiles_m * a.tiles_n * nbatch * split < (1 << 30);
}

static void launch_x3q(const GemmArgs& a, int a_kc, int b_kc, int items, hipStream_t stream) {
    if (g_single_pass) { launch_x3q_<true>(a, a_kc, b_kc, items, stream); return; }
    int g = (items + 7) / 8 * 8;
    if (g > 256) g = 256;
    const dim3 grid(g);
    if (a_kc && b_kc)
        hipLaunchKernelGGL((gemm_f32_f16x3_p12_kernel<true, true>), grid, dim3(768), 0, stream, a, items);
    else if (a_kc && !b_kc)
        hipLaunchKernelGGL((gemm_f32_f16x3_p12_kernel<true, false>), grid, dim3(768), 0, stream, a, items);
    else if (!a_kc && b_kc)
        hipLaunchKernelGGL((gemm_f32_f16x3_p12_kernel<false, true>), grid, dim3(768), 0, stream, a, items);
    else
        hipLaunchKernelGGL((gemm_f32_f16x3_p12_kernel<false, false>), grid, dim3(768), 0, stream, a, items);
}

template <int BN, bool DEFER>
static void launch_x6q_bn(const GemmArgs& a, int a_kc, int b_kc, int items, hipStream_t stream) {
    int g = (items + 7) / 8 * 8;
    if (g > 256) g = 256;
    const dim3 grid(g);
    if (a_kc && b_kc)
        hipLaunchKernelGGL((gemm_f32_bf16x6_p12_kernel<BN, true, true, DEFER>), grid, dim3(768), 0, stream, a, items);
    else if (a_kc && !b_kc)
        hipLaunchKernelGGL((gemm_f32_bf16x6_p12_kernel<BN, true, false, DEFER>), grid, dim3(768), 0, stream, a, items);
    else if (!a_kc && b_kc)
        hipLaunchKernelGGL((gemm_f32_bf16x6_p12_kernel<BN, false, true, DEFER>), grid, dim3(768), 0, stream, a, items);
    else
        hipLaunchKernelGGL((gemm_f32_bf16x6_p12_kernel<BN, false, false, DEFER>), grid, dim3(768), 0, stream, a, items);
}

#ifdef X6_DIAG_MODES   // experiment: eight consumer waves (mode 5)
template <int BN>
static void launch_x6q8_bn(const GemmArgs& a, int a_kc, int b_kc, int items, hipStream_t stream) {
    int g = (items + 7) / 8 * 8;
    if (g > 256) g = 256;
    const dim3 grid(g);
    if (a_kc && b_kc)
        hipLaunchKernelGGL((gemm_f32_bf16x6_p12_kernel<BN, true, true, false, 8>), grid, dim3(1024), 0, stream, a, items);
    else if (a_kc && !b_kc)
        hipLaunchKernelGGL((gemm_f32_bf16x6_p12_kernel<BN, true, false, false, 8>), grid, dim3(1024), 0, stream, a, items);
    else if (!a_kc && b_kc)
        hipLaunchKernelGGL((gemm_f32_bf16x6_p12_kernel<BN, false, true, false, 8>), grid, dim3(1024), 0, stream, a, items);
    else
        hipLaunchKernelGGL((gemm_f32_bf16x6_p12_kernel<BN, false, false, false, 8>), grid, dim3(1024), 0, stream, a, items);
}

#endif

template <bool DEFER>
static void launch_x6q(const GemmArgs& a, int bn, int a_kc, int b_kc, int items, hipStream_t stream) {
    if (bn == 128)
        launch_x6q_bn<128, DEFER>(a, a_kc, b_kc, items, stream);
    else if (bn == 64)
        launch_x6q_bn<64, DEFER>(a, a_kc, b_kc, items, stream);
    else
        launch_x6q_bn<32, DEFER>(a, a_kc, b_kc, items, stream);
}

#ifdef X6_DIAG_MODES   // superseded forms: 8-wave persistent kernel (mode 2), one tile per workgroup (mode 1)
template <int BN>
static void launch_x6p_bn(const GemmArgs& a, int a_kc, int b_kc, int items, hipStream_t stream) {
    int g = (items + 7) / 8 * 8;
    if (g > 256) g = 256;
    const dim3 grid(g);
    if (a_kc && b_kc)
        hipLaunchKernelGGL((gemm_f32_bf16x6_persistent_kernel<BN, true, true>), grid, dim3(512), 0, stream, a, items);
    else if (a_kc && !b_kc)
        hipLaunchKernelGGL((gemm_f32_bf16x6_persistent_kernel<BN, true, false>), grid, dim3(512), 0, stream, a, items);
    else if (!a_kc && b_kc)
        hipLaunchKernelGGL((gemm_f32_bf16x6_persistent_kernel<BN, false, true>), grid, dim3(512), 0, stream, a, items);
    else
        hipLaunchKernelGGL((gemm_f32_bf16x6_persistent_kernel<BN, false, false>), grid, dim3(512), 0, stream, a, items);
}

static void launch_x6p(const GemmArgs& a, int bn, int a_kc, int b_kc, int items, hipStream_t stream) {
    if (bn == 128)
        launch_x6p_bn<128>(a, a_kc, b_kc, items, stream);
    else if (bn == 64)
        launch_x6p_bn<64>(a, a_kc, b_kc, items, stream);
    else
        launch_x6p_bn<32>(a, a_kc, b_kc, items, stream);
}

template <int BN>
static void launch_x6_bn(const GemmArgs& a, int a_kc, int b_kc, dim3 grid, hipStream_t stream) {
    if (a_kc && b_kc)
        hipLaunchKernelGGL((gemm_f32_bf16x6_kernel<BN, true, true>), grid, dim3(512), 0, stream, a);
    else if (a_kc && !b_kc)
        hipLaunchKernelGGL((gemm_f32_bf16x6_kernel<BN, true, false>), grid, dim3(512), 0, stream, a);
    else if (!a_kc && b_kc)
        hipLaunchKernelGGL((gemm_f32_bf16x6_kernel<BN, false, true>), grid, dim3(512), 0, stream, a);
    else
        hipLaunchKernelGGL((gemm_f32_bf16x6_kernel<BN, false, false>), grid, dim3(512), 0, stream, a);
}

static void launch_x6(const GemmArgs& a, int bn, int a_kc, int b_kc, dim3 grid, hipStream_t stream) {
    if (bn == 128)
        launch_x6_bn<128>(a, a_kc, b_kc, grid, stream);
    else if (bn == 64)
        launch_x6_bn<64>(a, a_kc, b_kc, grid, stream);
    else
        launch_x6_bn<32>(a, a_kc, b_kc, grid, stream);
}


#endif

__global__ void zero_strided_kernel(float* C, int M, int N, int64_t ldc, int64_t sCo, int64_t sCi, int batch_inner) {
    const int zb = blockIdx.y;
    float* c = C + (zb / batch_inner) * sCo + (zb % batch_inner) * sCi;
    const int64_t total = (int64_t)M * N;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
        c[(i / N) * ldc + (i % N)] = 0.f;
}

template <int BM, int BN, int BK>
static void launch_cfg(const GemmArgs& a, int a_kc, int b_kc, dim3 grid, hipStream_t stream) {
    if (a_kc && b_kc)
        hipLaunchKernelGGL((gemm_f32_mfma_kernel<BM, BN, BK, true, true>), grid, dim3(256), 0, stream, a);
    else if (a_kc && !b_kc)
        hipLaunchKernelGGL((gemm_f32_mfma_kernel<BM, BN, BK, true, false>), grid, dim3(256), 0, stream, a);
    else if (!a_kc && b_kc)
        hipLaunchKernelGGL((gemm_f32_mfma_kernel<BM, BN, BK, false, true>), grid, dim3(256), 0, stream, a);
    else
        hipLaunchKernelGGL((gemm_f32_mfma_kernel<BM, BN, BK, false, false>), grid, dim3(256), 0, stream, a);
}

extern "C" int ix_colsum_f32(const float* x, float* out, int64_t rows, int C, int groups, void* workspace, size_t workspace_bytes,
                             hipStream_t stream);
extern "C" int ix_workspace_bytes_colsum_f32(int64_t rows, int C, int groups, size_t* out);
static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// ---- launch statistics (bench.py's roofline object) ----------------------------------------------------------
// Always on: executed FLOPs (2*M*N*K*batch) and launch count of the contraction kernel.  Optional (ix_gemm_prof_enable):
// a hipEvent pair around every contraction launch on its own stream, summed by ix_gemm_prof_read after a sync --
// that is the "average launch duration measured with HIP events" the roofline fraction is computed from.
#include <vector>
static int g_x6 = 3;   // 3 (default): 12-wave persistent bf16x6 kernel; 2: 8-wave persistent; 1: one tile per workgroup; 0: fp32 MFMA only
static double g_flops = 0.0;
static int64_t g_launches = 0;
static bool g_prof_on = false;
static std::vector<hipEvent_t> g_ev;
static size_t g_ev_used = 0;
struct ProfRec { int M, N, K, nbatch, a_kc, b_kc, bm, split; double flops = 0.0, mfma_flops = 0.0, bytes = 0.0; };   // bm 2002: a flash attention launch
static std::vector<ProfRec> g_rec;
static double g_flash_flops = 0.0;
static int64_t g_flash_launches = 0;

// Measurement aid of bench.py's roofline object: the dense fp16 matrix rate this GPU SUSTAINS -- one wave per SIMD of every CU
// issuing back-to-back v_mfma_f32_32x32x16_f16 on four independent accumulators (32.0 shader clocks per instruction:
// tools/micro/mfma_valu_overlap.hip), timed with events on `stream`.  The data-sheet peak (2.5 PFLOP/s) assumes 2.4 GHz; under
// this load the part clocks ~ 1.8 GHz.  Not part of the compute surface.
__global__ __launch_bounds__(256) void mfma_rate_kernel(float* out, int iters) {
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    const int lane = threadIdx.x & 63;
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(lane * 0.01f + i); b[i] = (_Float16)(i - lane * 0.02f); }
    f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
        }
    }
    if (c0[0] + c1[1] + c2[2] + c3[3] == 12345.678f) out[0] = 1.f;   // (keeps the loop; never true)
}
extern "C" int ix_diag_mfma_rate_f16(double* tflops, void* scratch4, hipStream_t stream) {
    IX_CHECK_ARG(tflops && scratch4, "ix_diag_mfma_rate_f16: null pointer");
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) {
        ix_set_error("ix_diag_mfma_rate_f16: no device");
        return IX_ERR_LAUNCH;
    }
    const int iters = 3000;
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return IX_ERR_LAUNCH;
    hipLaunchKernelGGL(mfma_rate_kernel, dim3(cus), dim3(256), 0, stream, (float*)scratch4, 200);   // warm-up: clocks settle
    hipLaunchKernelGGL(mfma_rate_kernel, dim3(cus), dim3(256), 0, stream, (float*)scratch4, iters);
    hipEventRecord(e0, stream);
    hipLaunchKernelGGL(mfma_rate_kernel, dim3(cus), dim3(256), 0, stream, (float*)scratch4, iters);
    hipEventRecord(e1, stream);
    float ms = 0.f;
    const bool ok = hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(&ms, e0, e1) == hipSuccess && ms > 0.f;
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    if (!ok) { ix_set_error("ix_diag_mfma_rate_f16: timing failed"); return IX_ERR_LAUNCH; }
    *tflops = (double)cus * 4.0 * iters * 32.0 * (2.0 * 32 * 32 * 16) / (ms * 1e-3) / 1e12;
    return IX_OK;
}

extern "C" int ix_prof_kinds3(double* ms3, double* flops3, int64_t* launches3);
extern "C" int ix_gemm_stats(double* flops, int64_t* launches, int reset) {
    if (flops) *flops = g_flops;
    if (launches) *launches = g_launches;
    if (reset) { g_flops = 0.0; g_launches = 0; }
    return IX_OK;
}

// Test hook: 0 = every contraction on the exact-fp32 kernel (v_mfma_f32_32x32x2_f32), 3 (default) = eligible contractions
// on the 12-wave persistent bf16x6 kernel.  Process-global and NOT part of the re-entrant compute surface: the product
// never calls it; tests use it to cross-check the two kernels.  (The superseded kernel forms 1 / 2 / 4 / 5 exist only in
// the diagnostic build, `make diag`, -DX6_DIAG_MODES.)  Returns the previous mode.
extern "C" int ix_gemm_set_mode(int mode) {
    const int old = g_x6;
#ifdef X6_DIAG_MODES
    g_x6 = mode < 0 ? 0 : (mode > 5 ? 5 : mode);
#else
    g_x6 = mode == 0 ? 0 : 3;
#endif
    return old;
}

extern "C" int ix_gemm_prof_enable(int on) {
    g_prof_on = on != 0;
    g_ev_used = 0;
    g_rec.clear();
    return IX_OK;
}

// Writes one CSV line per profiled launch (shape, tile, split, milliseconds) -- tuning aid, host path.
extern "C" int ix_gemm_prof_dump(const char* path) {
    FILE* f = fopen(path, "w");
    IX_CHECK_ARG(f != nullptr, "ix_gemm_prof_dump: cannot open %s", path);
    fprintf(f, "M,N,K,batch,a_kc,b_kc,tile,split,ms\n");
    for (size_t i = 0; i + 1 < g_ev_used && i / 2 < g_rec.size(); i += 2) {
        hipEventSynchronize(g_ev[i + 1]);
        float t = 0.f;
        hipEventElapsedTime(&t, g_ev[i], g_ev[i + 1]);
        const ProfRec& r = g_rec[i / 2];
        fprintf(f, "%d,%d,%d,%d,%d,%d,%d,%d,%.6f\n", r.M, r.N, r.K, r.nbatch, r.a_kc, r.b_kc, r.bm, r.split, t);   // (tile 2002: flash attention launch, K = its kind tag)
    }
    fclose(f);
    return IX_OK;
}

// Per kernel kind (index 0: fp32-MFMA kernel, 1: bf16x6 kernel): summed event time (ms), executed algorithmic FLOPs and
// launch count of the profiled launches.  Host arrays of 2.  Call before ix_gemm_prof_read (which clears the records).
extern "C" int ix_gemm_prof_kinds(double* ms2, double* flops2, int64_t* launches2) {
    double ms[3], fl[3];
    int64_t n[3];
    const int rc = ix_prof_kinds3(ms, fl, n);
    for (int k = 0; k < 2; ++k) {
        if (ms2) ms2[k] = ms[k];
        if (flops2) flops2[k] = fl[k];
        if (launches2) launches2[k] = n[k];
    }
    return rc;
}

// As ix_gemm_prof_kinds with a third slot: [0] fp32-MFMA contraction kernel, [1] bf16x6 contraction kernel, [2] flash
// attention kernels.  Host arrays of 3.
extern "C" int ix_prof_kinds3(double* ms3, double* flops3, int64_t* launches3) {
    double ms[3] = {0, 0, 0}, fl[3] = {0, 0, 0};
    int64_t n[3] = {0, 0, 0};
    for (size_t i = 0; i + 1 < g_ev_used && i / 2 < g_rec.size(); i += 2) {
        hipEventSynchronize(g_ev[i + 1]);
        float t = 0.f;
        hipEventElapsedTime(&t, g_ev[i], g_ev[i + 1]);
        const ProfRec& r = g_rec[i / 2];
        if (r.bm == 3128 || r.bm == 1616) continue;   // pre-split fp16x3 kernel: reported by ix_prof_x3; bf16 GEMM: ix_prof_b16
        const int k = r.bm == 2002 ? 2 : ((r.bm == 1128 || r.bm == 1129 || r.bm == 1131) ? 1 : 0);
        ms[k] += t;
        fl[k] += r.flops;
        n[k] += 1;
    }
    for (int k = 0; k < 3; ++k) {
        if (ms3) ms3[k] = ms[k];
        if (flops3) flops3[k] = fl[k];
        if (launches3) launches3[k] = n[k];
    }
    return IX_OK;
}

// Profiled contraction launches by kernel form: [0] exact-fp32 MFMA kernel, [1] bf16x6 form of the 12-wave kernel, [2] its
// fp16x3 form; summed event time (ms), algorithmic FLOPs, FLOPs of the matrix instructions actually issued (1 / 6 / 3 per
// fp32 multiply-add), launches.  Host arrays of 3.  Call before ix_gemm_prof_read (which clears the records).
extern "C" int ix_prof_contractions(double* ms3, double* flops3, double* mfma_flops3, int64_t* launches3) {
    double ms[3] = {0, 0, 0}, fl[3] = {0, 0, 0}, mf[3] = {0, 0, 0};
    int64_t n[3] = {0, 0, 0};
    for (size_t i = 0; i + 1 < g_ev_used && i / 2 < g_rec.size(); i += 2) {
        const ProfRec& r = g_rec[i / 2];
        if (r.bm == 2002 || r.bm == 3128 || r.bm == 1616) continue;
        hipEventSynchronize(g_ev[i + 1]);
        float t = 0.f;
        hipEventElapsedTime(&t, g_ev[i], g_ev[i + 1]);
        const int k = (r.bm == 1129 || r.bm == 1131) ? 2 : (r.bm == 1128 ? 1 : 0);
        ms[k] += t; fl[k] += r.flops; mf[k] += r.mfma_flops > 0 ? r.mfma_flops : r.flops; n[k] += 1;
    }
    for (int k = 0; k < 3; ++k) {
        if (ms3) ms3[k] = ms[k];
        if (flops3) flops3[k] = fl[k];
        if (mfma_flops3) mfma_flops3[k] = mf[k];
        if (launches3) launches3[k] = n[k];
    }
    return IX_OK;
}

// Algorithmic HBM bytes of the profiled contraction launches by form (see the header).  Call before ix_gemm_prof_read.
extern "C" int ix_prof_contraction_bytes(double* bytes3) {
    double by[3] = {0, 0, 0};
    for (size_t i = 0; i / 2 < g_rec.size() && i + 1 < g_ev_used; i += 2) {
        const ProfRec& r = g_rec[i / 2];
        if (r.bm == 2002 || r.bm == 3128 || r.bm == 1616) continue;
        const int k = (r.bm == 1129 || r.bm == 1131) ? 2 : (r.bm == 1128 ? 1 : 0);
        by[k] += 4.0 * ((double)r.M * r.K + (double)r.K * r.N + (double)r.M * r.N) * (double)(r.nbatch > 0 ? r.nbatch : 1);
    }
    for (int k = 0; k < 3; ++k)
        if (bytes3) bytes3[k] = by[k];
    return IX_OK;
}

// Profiled flash attention launches by kernel tag (1 forward, 2 backward-q, 3 backward-kv, 4 statistics, 5 second-order q,
// 6 second-order kv): summed event time (ms), algorithmic FLOPs, FLOPs of the matrix instructions actually issued
// (3 fp16 terms per product over the head dim, 6 bf16 terms per product over tokens), launches.  Host arrays of 7
// (index 0 = all tags).  Call before ix_gemm_prof_read (which clears the records).
extern "C" int ix_prof_flash(double* ms7, double* flops7, double* mfma_flops7, int64_t* launches7) {
    double ms[7] = {0}, fl[7] = {0}, mf[7] = {0};
    int64_t n[7] = {0};
    for (size_t i = 0; i + 1 < g_ev_used && i / 2 < g_rec.size(); i += 2) {
        const ProfRec& r = g_rec[i / 2];
        if (r.bm != 2002 || r.K < 1 || r.K > 6) continue;
        hipEventSynchronize(g_ev[i + 1]);
        float t = 0.f;
        hipEventElapsedTime(&t, g_ev[i], g_ev[i + 1]);
        for (int k = 0; k < 2; ++k) {
            const int j = k ? r.K : 0;
            ms[j] += t; fl[j] += r.flops; mf[j] += r.mfma_flops; n[j] += 1;
        }
    }
    for (int k = 0; k < 7; ++k) {
        if (ms7) ms7[k] = ms[k];
        if (flops7) flops7[k] = fl[k];
        if (mfma_flops7) mfma_flops7[k] = mf[k];
        if (launches7) launches7[k] = n[k];
    }
    return IX_OK;
}

// Profiled launches of the pre-split fp16x3 contraction path (two operand-split kernels + the GEMM kernel per call, one
// event bracket around all three): summed time (ms), algorithmic FLOPs, calls.  Call before ix_gemm_prof_read.
extern "C" int ix_prof_x3(double* ms, double* flops, int64_t* calls) {
    double m = 0, f = 0;
    int64_t n = 0;
    for (size_t i = 0; i + 1 < g_ev_used && i / 2 < g_rec.size(); i += 2) {
        const ProfRec& r = g_rec[i / 2];
        if (r.bm != 3128) continue;
        hipEventSynchronize(g_ev[i + 1]);
        float t = 0.f;
        hipEventElapsedTime(&t, g_ev[i], g_ev[i + 1]);
        m += t; f += r.flops; n += 1;
    }
    if (ms) *ms = m;
    if (flops) *flops = f;
    if (calls) *calls = n;
    return IX_OK;
}

// Executed algorithmic FLOPs and launch count of the flash attention kernels since the last reset (always on).
extern "C" int ix_flash_stats(double* flops, int64_t* launches, int reset) {
    if (flops) *flops = g_flash_flops;
    if (launches) *launches = g_flash_launches;
    if (reset) { g_flash_flops = 0.0; g_flash_launches = 0; }
    return IX_OK;
}

// Sums the elapsed time of all recorded event pairs (blocks until they have completed); host pointers.
extern "C" int ix_gemm_prof_read(double* total_ms, int64_t* pairs) {
    double ms = 0.0;
    for (size_t i = 0; i + 1 < g_ev_used; i += 2) {
        if (hipEventSynchronize(g_ev[i + 1]) != hipSuccess) {
            ix_set_error("ix_gemm_prof_read: hipEventSynchronize failed");
            return IX_ERR_LAUNCH;
        }
        float t = 0.f;
        hipEventElapsedTime(&t, g_ev[i], g_ev[i + 1]);
        ms += t;
    }
    if (total_ms) *total_ms = ms;
    if (pairs) *pairs = (int64_t)(g_ev_used / 2);
    g_ev_used = 0;
    g_rec.clear();
    return IX_OK;
}

static inline void prof_mark(hipStream_t stream) {
    if (!g_prof_on) return;
    if (g_ev_used == g_ev.size()) {
        hipEvent_t e;
        hipEventCreate(&e);
        g_ev.push_back(e);
    }
    hipEventRecord(g_ev[g_ev_used++], stream);
}

// The same per-launch event bracket for the flash attention kernels (csrc/flash.hip): kind 2, `flops` = algorithmic
// (fp32-equivalent) FLOPs of the launch, products = [L, S] x hd products it evaluates (encoded in the record's K).
void ix_prof_begin(hipStream_t stream, int kind, double flops, double mfma_flops, int tag) {
    if (kind == 2) { g_flash_flops += flops; g_flash_launches += 1; }
    if (!g_prof_on) return;
    ProfRec r = {0, 0, tag, 0, 0, 0, 2000 + kind, 1};
    r.flops = flops;
    r.mfma_flops = mfma_flops;
    g_rec.push_back(r);
    prof_mark(stream);
}
void ix_prof_end(hipStream_t stream) { prof_mark(stream); }
// ... and for the activation x weight-planes contraction kernel (csrc/gemm_wp.hip): a contraction like the others (counted by
// ix_gemm_stats; tile code 1131; the fp16x3 form's arithmetic: three matrix instructions per fp32 multiply-add)
void ix_prof_begin_wp(hipStream_t stream, int M, int N, int K, int nbatch) {
    const double fl = 2.0 * (double)M * (double)N * (double)K * (double)nbatch;
    g_flops += fl;
    g_launches += 1;
    if (!g_prof_on) return;
    ProfRec r = {M, N, K, nbatch, 1, 1, 1131, 1};
    r.flops = fl;
    r.mfma_flops = 3.0 * fl;
    g_rec.push_back(r);
    prof_mark(stream);
}

// ... and for the bf16 GEMM of the 16-bit mode (csrc/gemm16.hip): tile code 1616, one matrix instruction per multiply-add, `bytes` =
// its algorithmic HBM bytes (bf16 operands, bf16 / fp32 result); counted by ix_gemm_stats like every contraction
void ix_prof_begin_b16(hipStream_t stream, int M, int N, int K, int nbatch, double bytes) {
    const double fl = 2.0 * (double)M * (double)N * (double)K * (double)nbatch;
    g_flops += fl;
    g_launches += 1;
    if (!g_prof_on) return;
    ProfRec r = {M, N, K, nbatch, 1, 1, 1616, 1};
    r.flops = fl;
    r.mfma_flops = fl;
    r.bytes = bytes;
    g_rec.push_back(r);
    prof_mark(stream);
}
// Profiled launches of the bf16 GEMM: summed event time (ms), algorithmic FLOPs (= executed: one term), algorithmic bytes, launches.
// Call before ix_gemm_prof_read (which clears the records).
extern "C" int ix_prof_b16(double* ms, double* flops, double* bytes, int64_t* launches) {
    double m = 0, f = 0, b = 0;
    int64_t n = 0;
    for (size_t i = 0; i + 1 < g_ev_used && i / 2 < g_rec.size(); i += 2) {
        const ProfRec& r = g_rec[i / 2];
        if (r.bm != 1616) continue;
        hipEventSynchronize(g_ev[i + 1]);
        float t = 0.f;
        hipEventElapsedTime(&t, g_ev[i], g_ev[i + 1]);
        m += t; f += r.flops; b += r.bytes; n += 1;
    }
    if (ms) *ms = m;
    if (flops) *flops = f;
    if (bytes) *bytes = b;
    if (launches) *launches = n;
    return IX_OK;
}

// Profiled launches of the weight-planes kernel alone (they are ALSO part of slot [2], the fp16x3 arithmetic, of
// ix_prof_contractions): summed event time (ms), algorithmic FLOPs, launches.  Call before ix_gemm_prof_read.
extern "C" int ix_prof_wp(double* ms, double* flops, int64_t* launches) {
    double m = 0, f = 0;
    int64_t n = 0;
    for (size_t i = 0; i + 1 < g_ev_used && i / 2 < g_rec.size(); i += 2) {
        const ProfRec& r = g_rec[i / 2];
        if (r.bm != 1131) continue;
        hipEventSynchronize(g_ev[i + 1]);
        float t = 0.f;
        hipEventElapsedTime(&t, g_ev[i], g_ev[i + 1]);
        m += t; f += r.flops; n += 1;
    }
    if (ms) *ms = m;
    if (flops) *flops = f;
    if (launches) *launches = n;
    return IX_OK;
}

// ---- tile / split-K plan ------------------------------------------------------------------------------------------
// Tile / split-K selection by a small cost model (cycles on the most loaded CU).  The MFMA pipes of a CU are the
// shared resource: a CU that receives n workgroups spends n * ksteps * step_cycles on MFMAs, while the fixed
// prologue/epilogue latencies of its (up to two) co-resident workgroups overlap.  Split-K adds a second launch
// and s passes over C.  This replaces "fill the chip" thresholds, which lose up to 2x to wave quantisation
// (e.g. 540 workgroups on 512 resident slots).
struct TilePlan {
    int bm, split, kps;
};
// cycles of one 128 x 128 x 32 step of the 12-wave kernel on one CU, as the cost model prices it: bf16x6 form 1900 (measured);
// fp16x3 form 950 (half the matrix instructions; swept 200 .. 2600 on the headline step and on the 2-episode step, round 3:
// 285.0 / 67.7 ms at 900 against 288.1 / 70.7 at 1900 and 290.4 / 75.2 at 200).  IX_X6_STEP_CYCLES overrides it (tuning runs).
static bool x3k_enabled();
static double x6_step_cycles() {
    static double forced = -1.0;
    if (forced < 0) {
        const char* e = getenv("IX_X6_STEP_CYCLES");
        forced = e ? atof(e) : 0.0;
    }
    if (forced >= 100.0) return forced;
    return x3k_enabled() ? 950.0 : 1900.0;
}
// what the cost model charges for the separate reduction launch of a split contraction, in cycles (IX_SPLITK_LAUNCH_CYCLES: tuning runs)
static double splitk_launch_cycles() {
    static double v = -1.0;
    if (v < 0) {
        const char* e = getenv("IX_SPLITK_LAUNCH_CYCLES");
        v = e ? atof(e) : 6000.0;
    }
    return v;
}
static TilePlan plan_tiles(int M, int N, int K, int nbatch, bool want_x6, int tile_hint, int split_k_hint, double step128 = 0.0) {
    if (step128 <= 0.0) step128 = x6_step_cycles();
    int bm = 64, split = 1;
    double best = 1e300;
    const int cand_tiles[2] = {128, 64};
    const int cand_split[14] = {1, 2, 3, 4, 5, 6, 8, 10, 12, 16, 24, 32, 48, 64};
    for (int ti = 0; ti < 2; ++ti) {
        const int t = cand_tiles[ti];
        if (tile_hint != 0 && (tile_hint == 1128 ? 128 : tile_hint) != t) continue;
        const int bk = t == 128 ? 32 : 64;
        // 64x64 tiles pull 2x the L2 bytes per flop; the bf16x6 128-tile step is 48 x 32 MFMA cycles + the split
        const double step_cycles = t == 128 ? (want_x6 ? step128 : 4096.0) : 2048.0 * 1.15;
        const int64_t tl = (int64_t)ix_div_up(M, t) * ix_div_up(N, t) * nbatch;
        for (int si = 0; si < 14; ++si) {
            int sp = cand_split[si];
            if (split_k_hint > 0) sp = split_k_hint;
            if (sp > 1 && K < 2 * bk * sp) break;
            int kps = ix_div_up(ix_div_up(K, sp), bk) * bk;
            if (kps < bk) kps = bk;
            const int real_split = K > 0 ? ix_div_up(K, kps) : 1;
            const int ksteps = ix_div_up(kps, bk);
            const int64_t blocks = tl * real_split;
            const int64_t per_cu = (blocks + 255) / 256;
            double cost = (double)per_cu * ksteps * step_cycles + (double)((per_cu + 1) / 2) * 7000.0;
            if (real_split > 1) {
                const double cbytes = 4.0 * (double)M * (double)N * (double)nbatch;
                cost += splitk_launch_cycles() + cbytes / 2048.0 + cbytes * real_split / 1024.0;  // reduction launch + partial planes
            }
            if (cost < best) {
                best = cost;
                bm = t;
                split = real_split;
            }
            if (split_k_hint > 0) break;
        }
    }
    const int bk = bm == 128 ? 32 : 64;
    int kps = ix_div_up(ix_div_up(K, split), bk) * bk;
    if (kps < bk) kps = bk;
    split = K > 0 ? ix_div_up(K, kps) : 1;
    return TilePlan{bm, split, kps};
}

// bytes of the partial planes a split-K contraction writes ([split][batch][M][N] + [split][batch_outer][M] row sums)
static size_t splitk_plane_bytes(int split, int nbatch, int batch_outer, int M, int N) {
    if (split <= 1) return 0;
    return ((size_t)split * (size_t)nbatch * (size_t)M * (size_t)N + (size_t)split * (size_t)batch_outer * (size_t)M) * sizeof(float);
}

// Deterministic split-K: the s-th split of a contraction writes its partial tile sums into plane s of a caller-provided
// workspace with plain stores; this kernel adds the planes IN ORDER (s = 0, 1, ...) into C.  Two runs of one binary give the
// same bits (the fp32 atomics this replaces rounded in arrival order).  blockIdx.y == nbatch: the partial row sums.
struct SplitEpi {   // fused BN affine (+ residual at C's own index) (+ ReLU) of the reduction; scale == null: none
    const float *scale, *shift, *res;
    int relu;
};
__global__ void splitk_reduce_kernel(const float* __restrict__ planes, float* __restrict__ C, int M, int N, int64_t ldc,
                                     int64_t sCo, int64_t sCi, int batch_inner, int nbatch, int split, int64_t sSplit, int vec,
                                     const float* __restrict__ rs_planes, float* __restrict__ rowsum, int64_t rowsum_stride,
                                     int64_t sSplitRowsum, int batch_outer, SplitEpi ep) {
    const int zb = blockIdx.y;
    const int64_t gs = (int64_t)gridDim.x * blockDim.x, t0 = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (zb == nbatch) {
        const int64_t total = (int64_t)batch_outer * M;
        for (int64_t i = t0; i < total; i += gs) {
            float v = rs_planes[i];
            for (int s = 1; s < split; ++s) v += rs_planes[s * sSplitRowsum + i];
            rowsum[(i / M) * rowsum_stride + (i % M)] = v;
        }
        return;
    }
    const float* src = planes + (int64_t)zb * M * N;
    float* c = C + (zb / batch_inner) * sCo + (zb % batch_inner) * sCi;
    const float* er = ep.res ? ep.res + (c - C) : nullptr;
    if (vec) {
        const int N4 = N >> 2;
        const int64_t total = (int64_t)M * N4;
        for (int64_t i = t0; i < total; i += gs) {
            float4 v = reinterpret_cast<const float4*>(src)[i];
            for (int s = 1; s < split; ++s) {
                const float4 u = reinterpret_cast<const float4*>(src + s * sSplit)[i];
                v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
            }
            const int64_t o = (i / N4) * ldc + (i % N4) * 4;
            if (ep.scale) {
                const int col = (int)(i % N4) * 4;
                const float4 sc = *reinterpret_cast<const float4*>(ep.scale + col), sh = *reinterpret_cast<const float4*>(ep.shift + col);
                v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
                if (er) {
                    const float4 rr = *reinterpret_cast<const float4*>(er + o);
                    v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
                }
                if (ep.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            }
            *reinterpret_cast<float4*>(c + o) = v;
        }
    } else {
        const int64_t total = (int64_t)M * N;
        for (int64_t i = t0; i < total; i += gs) {
            float v = src[i];
            for (int s = 1; s < split; ++s) v += src[s * sSplit + i];
            const int64_t o = (i / N) * ldc + (i % N);
            if (ep.scale) {
                v = v * ep.scale[i % N] + ep.shift[i % N];
                if (er) v += er[o];
                if (ep.relu) v = fmaxf(v, 0.f);
            }
            c[o] = v;
        }
    }
}

// Points a planned contraction at its split-K planes (workspace given) or falls back to atomics on a zero-filled C.
// Returns IX_OK / IX_ERR_WORKSPACE.  `real` keeps what splitk_finish needs.
struct SplitReal {
    float* C;
    int64_t ldc, sCo, sCi;
    int c_vec;
    float* rowsum;
    int64_t rowsum_stride;
    bool planes;
};
static int splitk_begin(GemmArgs& a, int nbatch, int batch_outer, bool rowsum_in_kernel, void* workspace, size_t workspace_bytes,
                        SplitReal& real, const char* who, hipStream_t stream) {
    real.planes = false;
    a.atomic = 0;
    a.sSplit = 0;
    a.sSplitRowsum = 0;
    if (a.split_k <= 1) return IX_OK;
    if (!workspace) {   // legacy: fp32 atomics onto a zero-filled C (order-dependent rounding)
        a.atomic = 1;
        dim3 zg(ix_grid_1d((int64_t)a.M * a.N, 256), nbatch);
        hipLaunchKernelGGL(zero_strided_kernel, zg, dim3(256), 0, stream, a.C, a.M, a.N, a.ldc, a.sCo, a.sCi, a.batch_inner);
        if (rowsum_in_kernel) {
            dim3 rg(ix_grid_1d((int64_t)a.M, 256), batch_outer);
            hipLaunchKernelGGL(zero_strided_kernel, rg, dim3(256), 0, stream, a.rowsum, 1, a.M, (int64_t)a.M, a.sRowsum, (int64_t)0, 1);
        }
        return IX_OK;
    }
    const size_t need = splitk_plane_bytes(a.split_k, nbatch, batch_outer, a.M, a.N);
    if (workspace_bytes < need || !aligned16(workspace)) {
        ix_set_error("%s: split-K workspace of %zu bytes (16-byte aligned) needed, %zu given", who, need, workspace_bytes);
        return IX_ERR_WORKSPACE;
    }
    real.planes = true;
    real.C = a.C; real.ldc = a.ldc; real.sCo = a.sCo; real.sCi = a.sCi; real.c_vec = a.c_vec;
    real.rowsum = a.rowsum; real.rowsum_stride = a.sRowsum;
    const int64_t mn = (int64_t)a.M * a.N;
    a.C = static_cast<float*>(workspace);
    a.ldc = a.N;
    a.sCo = (int64_t)a.batch_inner * mn;
    a.sCi = mn;
    a.sSplit = (int64_t)nbatch * mn;
    a.c_vec = (a.N % 4 == 0) ? 1 : 0;
    if (rowsum_in_kernel) {
        a.rowsum = a.C + (int64_t)a.split_k * a.sSplit;
        a.sRowsum = a.M;
        a.sSplitRowsum = (int64_t)batch_outer * a.M;
    }
    return IX_OK;
}
static void splitk_finish(const GemmArgs& a, int nbatch, int batch_outer, bool rowsum_in_kernel, const SplitReal& real,
                          hipStream_t stream, SplitEpi ep = SplitEpi{nullptr, nullptr, nullptr, 0}) {
    if (!real.planes) return;
    const int vec = (a.N % 4 == 0) && real.c_vec;
    dim3 grid(ix_grid_1d((int64_t)a.M * a.N / (vec ? 4 : 1), 256), nbatch + (rowsum_in_kernel ? 1 : 0));
    hipLaunchKernelGGL(splitk_reduce_kernel, grid, dim3(256), 0, stream, a.C, real.C, a.M, a.N, real.ldc, real.sCo, real.sCi,
                       a.batch_inner, nbatch, a.split_k, a.sSplit, vec, rowsum_in_kernel ? a.rowsum : nullptr, real.rowsum,
                       real.rowsum_stride, a.sSplitRowsum, batch_outer, ep);
}

// Fused epilogue request of the NEXT contraction issued by this thread (armed by ix_gemm_bn_act_f32 /
// ix_conv_gemm_bn_act_f32 around their call of the plain entry point).  A split-K launch applies it in its ordered
// reduction (any kernel family) and sets `applied`; otherwise the wrapper runs ix_channel_affine_f32 on the output
// afterwards.  (The affine in the 12-wave kernel's own store was built and measured in rounds 2 and 3: at 16 episodes the
// contraction kernels gain 6 ms -- the consumers' store phase is latency-bound and now also reads the residual -- where the
// separate, bandwidth-efficient affine launches cost 9: step 277.2 -> 278.3 ms, nothing won; on small, launch-bound problems
// the plan splits K anyway.)
struct EpiReq {
    const float *scale, *shift, *res;
    int relu, applied;
};
static thread_local EpiReq g_epi = {nullptr, nullptr, nullptr, 0, 0};
static int64_t g_epi_count[3] = {0, 0, 0};   // fused calls whose affine ran: in the split-K reduction | as a separate launch | in the kernel's store
extern "C" int ix_gemm_epilogue_stats(int64_t* in_reduction, int64_t* separate, int reset) {
    if (in_reduction) *in_reduction = g_epi_count[0];
    if (separate) *separate = g_epi_count[1];
    if (reset) g_epi_count[0] = g_epi_count[1] = 0;
    return IX_OK;
}
extern "C" int ix_gemm_epilogue_in_store(int64_t* count, int reset) {
    if (count) *count = g_epi_count[2];
    if (reset) g_epi_count[2] = 0;
    return IX_OK;
}
// The affine in the contraction kernel's own store (EPI instances of the fp16x3 kernel: forward contractions with both operands
// k-contiguous, plain and gathering).  Rounds 2 and 3 measured this behind a run-time flag in the shared kernel and found nothing
// won; as a separate instance it cannot touch the launches that do not use it.  0: never (IX_GEMM_EPI_IN_STORE=0, tests), 1: default.
static int g_epi_in_store = -1;
extern "C" int ix_gemm_set_epilogue_in_store(int on) {
    const int old = g_epi_in_store;
    g_epi_in_store = on ? 1 : 0;
    return old < 0 ? 1 : old;
}
static bool epi_in_store_enabled() {
    if (g_epi_in_store < 0) {
        const char* e = getenv("IX_GEMM_EPI_IN_STORE");
        g_epi_in_store = (e && e[0] == '0') ? 0 : 1;
    }
    return g_epi_in_store != 0;
}
// an unsplit launch of the fp16x3 kernel with a pending affine takes it in its store
static bool epi_in_store_possible(const GemmArgs& a, bool x3_forward_kind) {   // the whole predicate, nothing mutated
    if (!g_epi.scale || g_epi.applied || !x3_forward_kind || a.split_k != 1 || a.atomic || !a.c_vec || a.N % 4 || a.ldc % 4 || !epi_in_store_enabled()) return false;
    if (!aligned16(g_epi.scale) || !aligned16(g_epi.shift) || (g_epi.res && !aligned16(g_epi.res)) || a.bias) return false;
    return true;
}
static bool epi_in_store(GemmArgs& a, bool x3_forward_kind) {
    a.epi_scale = a.epi_shift = a.epi_res = nullptr;
    a.epi_relu = 0;
    if (!epi_in_store_possible(a, x3_forward_kind)) return false;
    a.epi_scale = g_epi.scale; a.epi_shift = g_epi.shift; a.epi_res = g_epi.res; a.epi_relu = g_epi.relu;
    g_epi.applied = 1;
    ++g_epi_count[2];
    return true;
}

// called once the plan is known and splitk_begin has run: a split-K launch takes the epilogue in its ordered reduction
static SplitEpi epi_place(const GemmArgs& a, const SplitReal& real) {
    SplitEpi ep = {nullptr, nullptr, nullptr, 0};
    if (!g_epi.scale) return ep;
    const bool al = aligned16(g_epi.scale) && aligned16(g_epi.shift) && (!g_epi.res || aligned16(g_epi.res)) && a.N % 4 == 0;
    if (al && !a.atomic && a.split_k > 1 && real.planes) {
        ep.scale = g_epi.scale; ep.shift = g_epi.shift; ep.res = g_epi.res; ep.relu = g_epi.relu;
        g_epi.applied = 1;
        ++g_epi_count[0];
    }
    return ep;
}

// Which contractions ix_gemm_f32_ws can run on the pre-split fp16x3 kernel (gemm_x3.hip): 16-byte-aligned operands, one
// batch level, no tile or split hint, enough work that two conversion launches pay (>= 0.25 GFLOP), K and N large enough
// for its 128 x {128, 64} tiles.  The caller opts in per call by passing a workspace (hipops: IX_GEMM_X3=1).
static int g_presplit = 0;   // ix_gemm_presplit_enable: route eligible ix_gemm_f32_ws calls to the pre-split fp16x3 kernel (opt-in)
extern "C" int ix_gemm_presplit_enable(int on) {
    g_presplit = on ? 1 : 0;
    return IX_OK;
}
static bool x3_eligible(int M, int N, int K, int batch_outer, int batch_inner, int64_t lda, int64_t ldb, int64_t sAo,
                        int64_t sBo, const float* A, const float* B, int tile_hint, int split_k_hint) {
    if (!g_presplit || g_x6 == 0 || tile_hint != 0 || split_k_hint != 0 || batch_inner != 1) return false;
    if (K < 64 || N < 48 || M < 64) return false;
    if ((lda % 4) || (ldb % 4) || (sAo % 4) || (sBo % 4) || !aligned16(A) || !aligned16(B)) return false;
    return 2.0 * (double)M * (double)N * (double)K * (double)batch_outer >= 0.25e9;
}

extern "C" int ix_workspace_bytes_gemm_f32(int M, int N, int K, int a_kcontig, int b_kcontig, int64_t lda, int64_t ldb,
                                           int batch_outer, int batch_inner, int64_t sAo, int64_t sBo, const float* A,
                                           const float* B, int tile_hint, int split_k_hint, size_t* out) {
    IX_CHECK_ARG(out != nullptr, "ix_workspace_bytes_gemm_f32: null out");
    (void)a_kcontig; (void)b_kcontig;
    if (x3_eligible(M, N, K, batch_outer, batch_inner, lda, ldb, sAo, sBo, A, B, tile_hint, split_k_hint)) {
        *out = IX_TICKET_BYTES + ix_x3_workspace_bytes(M, N, K, sAo ? batch_outer : 1, sBo ? batch_outer : 1);
        return IX_OK;
    }
    // split-K planes (+ partial row sums): an upper bound over the two kernel families the call may be planned for
    const int nbatch = batch_outer * batch_inner;
    size_t need = 0;
    if (M > 0 && N > 0 && nbatch > 0) {
        // (whichever kernel family and form of the 12-wave kernel the call is planned for when it is issued: the test hooks
        //  ix_gemm_set_mode / ix_gemm_set_x3 may switch between this query and the launch)
        const double steps[3] = {950.0, 1900.0, x6_step_cycles()};
        for (int x6 = 0; x6 < 2; ++x6)
            for (int si = 0; si < 3; ++si) {
                const TilePlan pl = plan_tiles(M, N, K, nbatch, x6 != 0, tile_hint, split_k_hint, steps[si]);
                const size_t b = splitk_plane_bytes(pl.split, nbatch, batch_outer, M, N);
                if (b > need) need = b;
            }
    }
    if (need) need += IX_TICKET_BYTES;
    // ix_gemm_rowsum_f32 on shapes the producers do not sum: a separate ordered column sum of A (K rows of M)
    if (!a_kcontig && batch_inner == 1 && M > 0 && K > 0) {
        size_t cs = 0;
        ix_workspace_bytes_colsum_f32(K, M, batch_outer, &cs);
        if (cs > need) need = cs;
    }
    *out = need;
    return IX_OK;
}

static int gemm_impl(const float* A, const float* B, float* C, const float* bias, int M, int N, int K,
                     int a_kcontig, int b_kcontig, int64_t lda, int64_t ldb, int64_t ldc, int batch_outer,
                     int batch_inner, int64_t sAo, int64_t sAi, int64_t sBo, int64_t sBi, int64_t sCo,
                     int64_t sCi, int64_t bias_stride_outer, float alpha, int tile_hint, int split_k_hint,
                     void* workspace, size_t workspace_bytes, float* rowsum, int64_t rowsum_stride, hipStream_t stream);

extern "C" int ix_gemm_f32(const float* A, const float* B, float* C, const float* bias, int M, int N, int K,
                           int a_kcontig, int b_kcontig, int64_t lda, int64_t ldb, int64_t ldc, int batch_outer,
                           int batch_inner, int64_t sAo, int64_t sAi, int64_t sBo, int64_t sBi, int64_t sCo,
                           int64_t sCi, int64_t bias_stride_outer, float alpha, int tile_hint, int split_k_hint,
                           hipStream_t stream) {
    return gemm_impl(A, B, C, bias, M, N, K, a_kcontig, b_kcontig, lda, ldb, ldc, batch_outer, batch_inner, sAo, sAi, sBo, sBi,
                     sCo, sCi, bias_stride_outer, alpha, tile_hint, split_k_hint, nullptr, 0, nullptr, 0, stream);
}

extern "C" int ix_gemm_f32_ws(const float* A, const float* B, float* C, const float* bias, int M, int N, int K,
                              int a_kcontig, int b_kcontig, int64_t lda, int64_t ldb, int64_t ldc, int batch_outer,
                              int batch_inner, int64_t sAo, int64_t sAi, int64_t sBo, int64_t sBi, int64_t sCo,
                              int64_t sCi, int64_t bias_stride_outer, float alpha, int tile_hint, int split_k_hint,
                              void* workspace, size_t workspace_bytes, hipStream_t stream) {
    return gemm_impl(A, B, C, bias, M, N, K, a_kcontig, b_kcontig, lda, ldb, ldc, batch_outer, batch_inner, sAo, sAi, sBo, sBi,
                     sCo, sCi, bias_stride_outer, alpha, tile_hint, split_k_hint, workspace, workspace_bytes, nullptr, 0, stream);
}

// The contraction plus, in the same launch, the row sums of A:  rowsum[bo * rowsum_stride + m] = sum_k A(m, k).  For A
// stored m-contiguous (a_kcontig = 0) on the bf16x6 kernel the A-producer waves accumulate them from the operand tiles
// they stream anyway -- the bias gradient colsum(dy) rides on the weight-gradient contraction dW = dy^T x for free
// (reference: torch.nn.functional.linear under autograd, every Linear of models/gpt.py and detr_models/transformer.py).
// Other layouts / kernels: a separate column-sum launch (needs A contiguous: lda = M, sAo = K * M, rowsum_stride = M).
extern "C" int ix_gemm_rowsum_f32(const float* A, const float* B, float* C, int M, int N, int K, int a_kcontig, int b_kcontig,
                                  int64_t lda, int64_t ldb, int64_t ldc, int batch_outer, int64_t sAo, int64_t sBo, int64_t sCo,
                                  float alpha, float* rowsum, int64_t rowsum_stride, void* workspace, size_t workspace_bytes,
                                  hipStream_t stream) {
    IX_CHECK_ARG(rowsum != nullptr, "ix_gemm_rowsum_f32: null rowsum");
    const int presplit = g_presplit;   // (the row sums ride on the bf16x6 / fp16x3 producers, never on the pre-split route)
    g_presplit = 0;
    const int rc = gemm_impl(A, B, C, nullptr, M, N, K, a_kcontig, b_kcontig, lda, ldb, ldc, batch_outer, 1, sAo, 0, sBo, 0, sCo, 0, 0,
                             alpha, 0, 0, workspace, workspace_bytes, rowsum, rowsum_stride, stream);
    g_presplit = presplit;
    return rc;
}

static int gemm_impl(const float* A, const float* B, float* C, const float* bias, int M, int N, int K,
                     int a_kcontig, int b_kcontig, int64_t lda, int64_t ldb, int64_t ldc, int batch_outer,
                     int batch_inner, int64_t sAo, int64_t sAi, int64_t sBo, int64_t sBi, int64_t sCo,
                     int64_t sCi, int64_t bias_stride_outer, float alpha, int tile_hint, int split_k_hint,
                     void* workspace, size_t workspace_bytes, float* rowsum, int64_t rowsum_stride, hipStream_t stream) {
    IX_CHECK_ARG(A && B && C, "ix_gemm_f32: null operand");
    IX_CHECK_ARG(M >= 0 && N >= 0 && K >= 0 && batch_outer >= 0 && batch_inner >= 1, "ix_gemm_f32: bad dims");
    const int nbatch = batch_outer * batch_inner;
    if (M == 0 || N == 0 || nbatch == 0) return IX_OK;
    IX_CHECK_ARG(nbatch <= 65535, "ix_gemm_f32: batch %d > 65535", nbatch);
    // workspace layout shared by every entry point: [IX_TICKET_BYTES of reduction tickets][scratch]
    void* const workspace_all = workspace;
    const size_t workspace_all_bytes = workspace_bytes;
    if (workspace) {
        IX_CHECK_ARG(workspace_bytes >= IX_TICKET_BYTES && aligned16(workspace), "ix_gemm_f32_ws: workspace below %d bytes or unaligned", IX_TICKET_BYTES);
        workspace = static_cast<char*>(workspace) + IX_TICKET_BYTES;
        workspace_bytes -= IX_TICKET_BYTES;
    }
    if (workspace && x3_eligible(M, N, K, batch_outer, batch_inner, lda, ldb, sAo, sBo, A, B, tile_hint, split_k_hint)) {
        const size_t need = ix_x3_workspace_bytes(M, N, K, sAo ? batch_outer : 1, sBo ? batch_outer : 1);
        if (workspace_bytes < need) {
            ix_set_error("ix_gemm_f32_ws: workspace of %zu bytes needed, %zu given", need, workspace_bytes);
            return IX_ERR_WORKSPACE;
        }
        X3Call c;
        c.A = A; c.B = B; c.C = C; c.bias = bias; c.M = M; c.N = N; c.K = K; c.a_kc = a_kcontig; c.b_kc = b_kcontig;
        c.nbatch = batch_outer; c.lda = lda; c.ldb = ldb; c.ldc = ldc; c.sA = sAo; c.sB = sBo; c.sC = sCo;
        c.sBias = bias_stride_outer; c.alpha = alpha;
        const double fl = 2.0 * (double)M * (double)N * (double)K * (double)nbatch;
        g_flops += fl;
        g_launches += 1;
        if (g_prof_on) {
            ProfRec r = {M, N, K, nbatch, a_kcontig, b_kcontig, 3128, 1};
            r.flops = fl;
            g_rec.push_back(r);
        }
        prof_mark(stream);
        ix_x3_gemm(c, workspace, stream);
        prof_mark(stream);
        IX_CHECK_LAUNCH("ix_gemm_f32_ws");
        return IX_OK;
    }
    GemmArgs a;
    a.cg = ConvGather();
    a.cg.mode_a = a.cg.mode_b = 0;
    a.rowsum = nullptr;
    a.sRowsum = 0;
    a.A = A; a.B = B; a.C = C; a.bias = bias;
    a.M = M; a.N = N; a.K = K;
    a.lda = lda; a.ldb = ldb; a.ldc = ldc;
    a.sAo = sAo; a.sAi = sAi; a.sBo = sBo; a.sBi = sBi; a.sCo = sCo; a.sCi = sCi;
    a.batch_inner = batch_inner;
    a.sBias = bias_stride_outer;
    a.alpha = alpha;
    const bool sa = (sAo % 4 == 0) && (sAi % 4 == 0) && (lda % 4 == 0) && aligned16(A);
    const bool sb = (sBo % 4 == 0) && (sBi % 4 == 0) && (ldb % 4 == 0) && aligned16(B);
    // a float4 may not straddle the valid extent of the contiguous dimension unless the tail is handled
    // element-wise; the loader falls back per vector, so only base alignment matters here.
    a.a_vec = sa ? 1 : 0;
    a.b_vec = sb ? 1 : 0;
    a.c_vec = ((sCo % 4 == 0) && (sCi % 4 == 0) && (ldc % 4 == 0) && aligned16(C)) ? 1 : 0;
    a.extA = a_kcontig ? (int64_t)(M - 1) * lda + K : (int64_t)(K - 1) * lda + M;
    a.extB = b_kcontig ? (int64_t)(N - 1) * ldb + K : (int64_t)(K - 1) * ldb + N;
    // bf16x6 kernel: 32-bit buffer offsets + 16-byte loads; its 3-stage ring and one-workgroup-per-CU residency only pay
    // off once there are enough K steps to stream (attention's K = 32 / 64 products stay on the fp32 kernel, where a
    // second resident workgroup hides the prologue)
    const bool x6_ok = sa && sb && K >= 32 && a.extA * 4 < (int64_t)1 << 31 && a.extB * 4 < (int64_t)1 << 31 &&
                       (int64_t)ix_div_up(M, 128) * ix_div_up(N, 32) * nbatch * 64 < ((int64_t)1 << 30);   // (item index fits an int)
    const bool want_x6 = x6_ok && (tile_hint == 1128 || (g_x6 && tile_hint != 128));

    const TilePlan plan = plan_tiles(M, N, K, nbatch, want_x6, tile_hint, split_k_hint);
    const int bm = plan.bm, split = plan.split;
    const bool use_x6 = bm == 128 && want_x6;
    const int bn = use_x6 ? (N > 64 ? 128 : (N > 32 ? 64 : 32)) : bm;   // narrow bf16x6 tiles for N = head dim
    a.tiles_m = ix_div_up(M, bm);
    a.tiles_n = ix_div_up(N, bn);
    a.split_k = split;
    a.k_per_split = plan.kps;
    // 256 x 128 tiles (gemm_f32_f16x3_w256_kernel) where they finish sooner: rounds of 256 workgroups x K steps x the
    // measured cost of a K step (w256_step_ratio() x the 128 x 128 kernel's for twice the products)
    bool use_w2 = false;
    if (use_x6 && bn == 128 && x3k_enabled() && g_x6 == 3 && !g_single_pass && w256_mode() > 0 &&
        (int64_t)a.tiles_m * a.tiles_n * nbatch * split < (1 << 30)) {
        const int tm2 = ix_div_up(M, 256);
        const int64_t it1 = (int64_t)a.tiles_m * a.tiles_n * nbatch * split, it2 = (int64_t)tm2 * a.tiles_n * nbatch * split;
        const double r1 = (double)(((it1 + 7) / 8 + 31) / 32), r2 = (double)(((it2 + 7) / 8 + 31) / 32);
        const double ks = (double)ix_div_up(plan.kps, 32);
        // fitted on tools/w256_bench.py (alternating best-of-3, profiles/r4z_w256_bench_fair.txt): a 256 x 128 K step costs 1.94 of
        // a 128 x 128 one with a k-contiguous A (0.97 per product), 2.15 with an m-contiguous A (its row sums and 4 x 4 register
        // transposes ride on ONE producer wave per SIMD); short-K, C-store-bound shapes (K < 128) gain nothing
        const double ratio = w256_step_ratio() > 0.0 ? w256_step_ratio() : (a_kcontig ? 1.94 : 2.15);
        use_w2 = w256_mode() == 2 || (K >= 128 && r2 * (ks * ratio + 1.0) < 0.97 * r1 * (ks + 1.5));
        if (use_w2) a.tiles_m = tm2;
    }
    bool rowsum_in_kernel = false;
    if (rowsum) {
        if (use_x6 && !a_kcontig && batch_inner == 1 && g_x6 == 3) {
            rowsum_in_kernel = true;
            a.rowsum = rowsum;
            a.sRowsum = rowsum_stride;
        } else {
            IX_CHECK_ARG(!a_kcontig && lda == M && batch_inner == 1 && (batch_outer == 1 || (sAo == (int64_t)K * M && rowsum_stride == M)),
                         "ix_gemm_rowsum_f32: the separate column-sum path needs a contiguous m-fastest A");
            const int rc = ix_colsum_f32(A, rowsum, K, M, batch_outer, workspace_all, workspace_all_bytes, stream);
            if (rc != IX_OK) return rc;
        }
    }
    // split-K partial sums: planes in the caller's workspace + an ordered reduction (deterministic); without a workspace
    // fp32 atomics onto a zero-filled C (a kernel, not hipMemsetAsync: inside the policy step's captured HIP graph a
    // memset node in front of the atomics replayed differently from eager)
    SplitReal real;
    {
        const int rc = splitk_begin(a, nbatch, batch_outer, rowsum_in_kernel, workspace, workspace_bytes, real, "ix_gemm_f32", stream);
        if (rc != IX_OK) return rc;
    }
    dim3 grid(a.tiles_m * a.tiles_n, nbatch, split);
    g_flops += 2.0 * (double)M * (double)N * (double)K * (double)nbatch;
    g_launches += 1;
    const bool use_x3 = use_x6 && bn == 128 && x3k_enabled() && g_x6 == 3;
    const SplitEpi sep = epi_place(a, real);
    // (the 256 x 128 tiles have no such instance: an unsplit launch with a pending affine goes back to 128 x 128 tiles -- the pass
    //  it saves is worth more than their 4-8 %)
    //  The move happens only when the affine WILL ride in the store -- the full predicate (alignment of scale / shift / residual,
    //  vector stores, N and ldc multiples of 4, persistent launch) evaluated on the 128-row plan first; otherwise the launch keeps its
    //  256 x 128 tiles and the affine runs as the library's separate pass.
    if (use_w2 && g_epi.scale && !g_epi.applied && split == 1) {
        GemmArgs t = a;
        t.tiles_m = ix_div_up(M, bm);
        if (epi_in_store_possible(t, use_x3 && a_kcontig && b_kcontig && batch_inner == 1 && persistent_ok_pre(t, nbatch, split))) {
            use_w2 = false;
            a.tiles_m = t.tiles_m;
            grid = dim3(a.tiles_m * a.tiles_n, nbatch, split);
        }
    }
    const bool epi_store = !use_w2 && epi_in_store(a, use_x3 && a_kcontig && b_kcontig && batch_inner == 1 && persistent_ok_pre(a, nbatch, split));
    if (g_prof_on) {
        ProfRec r = {M, N, K, nbatch, a_kcontig, b_kcontig, use_x3 ? 1129 : (use_x6 ? 1128 : bm), split};
        r.flops = 2.0 * (double)M * (double)N * (double)K * (double)nbatch;
        r.mfma_flops = r.flops * (use_x3 ? (g_single_pass ? 1.0 : 3.0) : (use_x6 ? 6.0 : 1.0));   // matrix instructions issued per fp32 multiply-add
        g_rec.push_back(r);
    }
    set_item_divs(a);
    prof_mark(stream);
    const bool persistent_ok = (int64_t)a.tiles_m * a.tiles_n * nbatch * split < (1 << 30);
    const int items = a.tiles_m * a.tiles_n * nbatch * split;
#ifdef X6_DIAG_MODES
    if (use_x6 && g_x6 == 5 && bn >= 64 && persistent_ok) {
        if (bn == 128) launch_x6q8_bn<128>(a, a_kcontig, b_kcontig, items, stream);
        else launch_x6q8_bn<64>(a, a_kcontig, b_kcontig, items, stream);
    } else if (use_x6 && g_x6 == 4 && persistent_ok)
        launch_x6q<true>(a, bn, a_kcontig, b_kcontig, items, stream);
    else if (use_x6 && g_x6 == 2 && persistent_ok)
        launch_x6p(a, bn, a_kcontig, b_kcontig, items, stream);
    else if (use_x6 && (g_x6 == 1 || !persistent_ok))
        launch_x6(a, bn, a_kcontig, b_kcontig, grid, stream);
    else
#endif
    if (use_w2)
        launch_w256(a, a_kcontig, b_kcontig, items, stream);
    else if (epi_store)
        launch_x3q_epi(a, items, stream);
    else if (use_x3)
        launch_x3q(a, a_kcontig, b_kcontig, items, stream);
    else if (use_x6)
        launch_x6q<false>(a, bn, a_kcontig, b_kcontig, items, stream);
    else if (bm == 128)
        launch_cfg<128, 128, 32>(a, a_kcontig, b_kcontig, grid, stream);
    else
        launch_cfg<64, 64, 64>(a, a_kcontig, b_kcontig, grid, stream);
    splitk_finish(a, nbatch, batch_outer, rowsum_in_kernel, real, stream, sep);
    prof_mark(stream);
    IX_CHECK_LAUNCH("ix_gemm_f32");
    return IX_OK;
}

// ------------------------------------------------------------------------------------------------------------
// Implicit-GEMM convolution (no patch matrix in HBM): forward, data gradient and weight gradient of a bias-free NHWC
// convolution with [out][kh][kw][in] weights, as three gather modes of the bf16x6 kernel's producer waves (ConvGather).
// The three are closed under differentiation (hipops.ConvFwd / ConvBwdData / ConvBwdWeight), so the MAML double
// backward through the adapted backbone stages runs on them as well.
//   kind 0  y[g][img][oy][ox][co]  = sum_{ky,kx,c} x[g][img][oy*s-p+ky*d][ox*s-p+kx*d][c] * w[g][co][ky][kx][c]
//   kind 1  dx[g][img][y][x][c]    = sum_{ky,kx,co} dy[g][img][(y+p-ky*d)/s][(x+p-kx*d)/s][co] * w[g][co][ky][kx][c]
//   kind 2  dw[g][co][ky][kx][c]   = sum_{img,oy,ox} dy[g][img][oy][ox][co] * x[g][img][oy*s-p+ky*d][ox*s-p+kx*d][c]
// `groups` = episodes with their own (fast) weights; shared weights: groups = 1 and imgs = all images.
// ------------------------------------------------------------------------------------------------------------
static int x6_pick_split(int M, int N, int K, int nbatch) {
    double best = 1e300;
    int bs = 1;
    const int cand[14] = {1, 2, 3, 4, 5, 6, 8, 10, 12, 16, 24, 32, 48, 64};
    const int64_t tl = (int64_t)ix_div_up(M, 128) * ix_div_up(N, 128) * nbatch;
    for (int si = 0; si < 14; ++si) {
        const int sp = cand[si];
        if (sp > 1 && K < 2 * 32 * sp) break;
        const int kps = ix_div_up(ix_div_up(K, sp), 32) * 32;
        const int real = ix_div_up(K, kps), ksteps = kps / 32;
        const int64_t per_cu = (tl * real + 255) / 256;
        double cost = (double)per_cu * ksteps * 1900.0 + (double)((per_cu + 1) / 2) * 7000.0;
        if (real > 1) {
            const double cbytes = 4.0 * (double)M * (double)N * (double)nbatch;
            cost += 6000.0 + cbytes / 2048.0 + cbytes * real / 1024.0;
        }
        if (cost < best) { best = cost; bs = real; }
    }
    return bs;
}

template <int BN>
static void launch_conv_bn(const GemmArgs& a, int kind, int items, hipStream_t stream, bool x3 = false, bool epi = false) {
    int g = (items + 7) / 8 * 8;
    if (g > 256) g = 256;
    const dim3 grid(g);
    if (epi && x3 && BN == 128) {   // forward convolution with the affine epilogue in its store
        if (g_single_pass) hipLaunchKernelGGL((gemm_f32_f16x3_p12_kernel<true, true, 1, 0, true, false, true>), grid, dim3(768), 0, stream, a, items);
        else hipLaunchKernelGGL((gemm_f32_f16x3_p12_kernel<true, true, 1, 0, false, false, true>), grid, dim3(768), 0, stream, a, items);
        return;
    }
    if (x3 && BN == 128 && a.cg.cmap) {   // a parity class of conv_bwd_data_s2 written in place (forward kind)
        if (g_single_pass) hipLaunchKernelGGL((gemm_f32_f16x3_p12_kernel<true, true, 1, 0, true, true>), grid, dim3(768), 0, stream, a, items);
        else hipLaunchKernelGGL((gemm_f32_f16x3_p12_kernel<true, true, 1, 0, false, true>), grid, dim3(768), 0, stream, a, items);
        return;
    }
    if (x3 && BN == 128 && g_single_pass) {   // ... its single-pass form (MODEL.COMPUTE_DTYPE: bf16 / fp16)
        if (kind == 0)
            hipLaunchKernelGGL((gemm_f32_f16x3_p12_kernel<true, true, 1, 0, true>), grid, dim3(768), 0, stream, a, items);
        else if (kind == 1)
            hipLaunchKernelGGL((gemm_f32_f16x3_p12_kernel<true, false, 1, 3, true>), grid, dim3(768), 0, stream, a, items);
        else
            hipLaunchKernelGGL((gemm_f32_f16x3_p12_kernel<false, false, 0, 2, true>), grid, dim3(768), 0, stream, a, items);
        return;
    }
    if (x3 && BN == 128) {   // fp16x3 form of the gathering producers (128-wide tiles only)
        if (kind == 0)
            hipLaunchKernelGGL((gemm_f32_f16x3_p12_kernel<true, true, 1, 0>), grid, dim3(768), 0, stream, a, items);
        else if (kind == 1)
            hipLaunchKernelGGL((gemm_f32_f16x3_p12_kernel<true, false, 1, 3>), grid, dim3(768), 0, stream, a, items);
        else
            hipLaunchKernelGGL((gemm_f32_f16x3_p12_kernel<false, false, 0, 2>), grid, dim3(768), 0, stream, a, items);
        return;
    }
    if (kind == 0)
        hipLaunchKernelGGL((gemm_f32_bf16x6_p12_kernel<BN, true, true, false, 4, 1, 0>), grid, dim3(768), 0, stream, a, items);
    else if (kind == 1)
        hipLaunchKernelGGL((gemm_f32_bf16x6_p12_kernel<BN, true, false, false, 4, 1, 3>), grid, dim3(768), 0, stream, a, items);
    else
        hipLaunchKernelGGL((gemm_f32_bf16x6_p12_kernel<BN, false, false, false, 4, 0, 2>), grid, dim3(768), 0, stream, a, items);
}

// 1 if ix_conv_gemm_f32 takes this convolution (else the caller keeps the patch-matrix path: ix_im2col_f32 + ix_gemm_f32)
extern "C" int ix_conv_gemm_supported(int groups, int imgs, int H, int W, int Cin, int OH, int OW, int Cout, int KH, int KW,
                                      int stride, int pad, int dil) {
    if (g_x6 == 0) return 0;
    if (groups < 1 || imgs < 1 || (Cin % 64) || (Cout % 64) || (stride != 1 && stride != 2 && stride != 4)) return 0;
    if (KH < 1 || KW < 1 || pad < 0 || dil < 1) return 0;
    const int64_t lim = ((int64_t)1 << 31) / 4 - 4160;   // (bytes < 2^31 - 16 640: SplitLoader::PixTap's out-of-range sentinel + 8 KB)
    if ((int64_t)imgs * H * W * Cin >= lim || (int64_t)imgs * OH * OW * Cout >= lim || (int64_t)Cout * KH * KW * Cin >= lim) return 0;
    if ((int64_t)imgs * H * W >= ((int64_t)1 << 30) || (int64_t)groups > 65535) return 0;
    return 1;
}

// (M, N, K) of the contraction behind one convolution kind
static void conv_gemm_dims(int kind, int imgs, int H, int W, int Cin, int OH, int OW, int Cout, int T, int* M, int* N, int* K) {
    if (kind == 0) { *M = imgs * OH * OW; *N = Cout; *K = T * Cin; }
    else if (kind == 1) { *M = imgs * H * W; *N = Cin; *K = T * Cout; }
    else { *M = Cout; *N = T * Cin; *K = imgs * OH * OW; }
}
static int conv_split(int M, int N, int K, int groups, int* kps_out) {
    int split = x6_pick_split(M, N, K, groups);
    const int kps = ix_div_up(ix_div_up(K, split), 32) * 32;
    *kps_out = kps;
    return ix_div_up(K, kps);
}

// ---- the stride-2 data gradient as stride-1 convolutions, one per parity class of the input pixels -------------------------
// dx[y][x] of a stride-2 convolution only receives the taps with (y + pad - ky) and (x + pad - kx) even: gathered over all
// input pixels (kind 1, qs = 1) three of four requested taps are the out-of-range zeros and the kernel still multiplies them
// (measured, tools/gemm_census.py: 69 TFLOP/s of useful work where the stride-1 convolutions of the same size run 245-275).
// Pixels (2i + py, 2j + px) of one parity class all see the SAME taps ky = ky0 + 2a, kx = kx0 + 2b, and over the class grid
// (i, j) that is a plain stride-1 convolution of dy with the sub-kernel, taps flipped:
//     dx[2i+py][2j+px][c] = sum_{a'',b'',co} dy[i - pad_y + a''][j - pad_x + b''][co] * wt[c][a''][b''][co]
//     wt[c][a''][b''][co] = w[co][ky0 + 2 (nky-1-a'')][kx0 + 2 (nkx-1-b'')][c],   pad_y = nky - 1 - (py + pad - ky0) / 2
// (3x3, pad 1: sub-kernels 1x1 / 1x2 / 2x1 / 2x2 with pads 0 -- 9 taps instead of 36; the 1x1 stride-2 downsample: one class
// with one tap, the other three are zero).  The regrouped weights and the class outputs live in the call's scratch; one
// interleaving pass writes dx (every element, so no memset).  Needs a workspace; without one the one-launch form runs.
static int g_conv_s2_in_place = 1;   // ix_conv_set_s2_split(mode | 4): classes through scratch + interleave even where the row map applies (A/B, tests)
static int g_conv_s2_split = -1;   // 0: never (A/B runs), 1: where it pays (default), 2: always (tests); IX_CONV_S2_SPLIT
extern "C" int ix_conv_set_s2_split(int mode) {
    g_conv_s2_in_place = (mode >= 0 && (mode & 4)) ? 0 : 1;
    if (mode >= 0) mode &= 3;
    g_conv_s2_split = mode < 0 ? 0 : (mode > 2 ? 2 : mode);
    return IX_OK;
}
static int conv_s2_split_mode() {
    if (g_conv_s2_split < 0) {
        const char* e = getenv("IX_CONV_S2_SPLIT");
        g_conv_s2_split = (e && e[0] >= '0' && e[0] <= '2') ? e[0] - '0' : 1;
    }
    return g_conv_s2_split;
}
struct S2Axis {
    int n, k0, pad, len;   // taps of the class along this axis, the first of them, the sub-convolution's pad, the class grid's extent
};
static S2Axis s2_axis(int extent, int parity, int K, int pad) {
    S2Axis x;
    x.k0 = (parity + pad) & 1;
    x.n = x.k0 < K ? (K - 1 - x.k0) / 2 + 1 : 0;
    x.pad = x.n - 1 - (parity + pad - x.k0) / 2;
    x.len = extent > parity ? (extent - parity + 1) / 2 : 0;
    return x;
}
struct S2Plan {
    S2Axis ay[4], ax[4];
    int64_t w_off[4], y_off[4];   // element offsets of the class's regrouped weights / output in their regions
    int64_t w_elems, y_elems;
    size_t planes;                // split-K planes: the largest any class launch needs
};
static size_t round256(size_t b) { return (b + 255) / 256 * 256; }
static S2Plan s2_plan(int groups, int imgs, int H, int W, int Cin, int Cout, int KH, int KW, int pad) {
    S2Plan p;
    p.w_elems = p.y_elems = 0;
    p.planes = 0;
    for (int c = 0; c < 4; ++c) {
        p.ay[c] = s2_axis(H, c >> 1, KH, pad);
        p.ax[c] = s2_axis(W, c & 1, KW, pad);
        p.w_off[c] = p.w_elems;
        p.y_off[c] = p.y_elems;
        const int64_t taps = (int64_t)p.ay[c].n * p.ax[c].n, px = (int64_t)p.ay[c].len * p.ax[c].len;
        if (taps == 0 || px == 0) continue;
        p.w_elems += (int64_t)groups * Cin * taps * Cout;
        p.y_elems += (int64_t)groups * imgs * px * Cin;
        int kps;
        const int M = (int)(imgs * px), K = (int)(taps * Cout);
        const size_t b = splitk_plane_bytes(conv_split(M, Cin, K, groups, &kps), groups, groups, M, Cin);
        if (b > p.planes) p.planes = b;
    }
    return p;
}
// ... where it pays: the class form is 2 (one class with taps) to 9 launches instead of one.  Measured break-even
// (tools/conv_s2_bench.py, tools/gemm_census.py): ~ 30 GFLOP executed by the one-launch form for a 3x3 (16 frames of 75 x 75 x
// 128: 75 us either way), far lower when a single class has taps (the 1x1 downsample: one quarter-size contraction + the zeros).
static bool conv_takes_s2_split(int kind, int stride, int dil, int groups, int imgs, int H, int W, int Cin, int Cout, int KH, int KW) {
    const int mode = conv_s2_split_mode();
    if (!(kind == 1 && stride == 2 && dil == 1 && H >= 2 && W >= 2 && mode != 0)) return false;
    if (mode == 2) return true;
    const double executed = 2.0 * groups * imgs * H * W * (double)Cin * Cout * KH * KW;
    return executed >= ((KH == 1 && KW == 1) ? 4e9 : 40e9);
}

extern "C" int ix_workspace_bytes_conv_gemm_f32(int kind, int groups, int imgs, int H, int W, int Cin, int OH, int OW, int Cout,
                                                int KH, int KW, int stride, int pad, int dil, size_t* out) {
    IX_CHECK_ARG(out != nullptr && kind >= 0 && kind <= 2, "ix_workspace_bytes_conv_gemm_f32: bad args");
    int M, N, K, kps;
    conv_gemm_dims(kind, imgs, H, W, Cin, OH, OW, Cout, KH * KW, &M, &N, &K);
    *out = splitk_plane_bytes(conv_split(M, N, K, groups, &kps), groups, groups, M, N);
    if (conv_takes_s2_split(kind, stride, dil, groups, imgs, H, W, Cin, Cout, KH, KW)) {   // whichever form runs when the call is issued (ix_conv_set_s2_split)
        const S2Plan p = s2_plan(groups, imgs, H, W, Cin, Cout, KH, KW, pad);
        const size_t b = round256(p.planes) + round256((size_t)p.w_elems * 4) + round256((size_t)p.y_elems * 4);
        if (b > *out) *out = b;
    }
    if (*out) *out += IX_TICKET_BYTES;
    return IX_OK;
}

// one convolution kind as ONE launch of the gathering kernel (+ its split-K reduction); `planes` = split-K scratch (past the tickets)
struct S2Map {   // a parity class written in place: the full image extent, the class's parities, elements of one group's dx
    int H, W, py, px;
    int64_t x_slice;
};
static bool conv_uses_x3(int kind, int N, int Cin) {   // the fp16x3 form of the gathering kernel (128-wide N tiles)
    const int bn = (kind == 2 ? (Cin % 128 == 0) : (N > 64)) ? 128 : 64;
    return bn == 128 && x3k_enabled() && g_x6 == 3;
}
static int conv_gemm_core(int kind, const float* src, const float* other, float* out, int groups, int imgs, int H, int W, int Cin,
                          int OH, int OW, int Cout, int KH, int KW, int stride, int pad_y, int pad_x, int dil, void* workspace,
                          size_t workspace_bytes, hipStream_t stream, const S2Map* map = nullptr) {
    const int T = KH * KW;
    const int qs = stride == 1 ? 0 : (stride == 2 ? 1 : 2);
    GemmArgs a;
    a.cg = ConvGather();
    ConvGather& g = a.cg;
    g.mode_a = g.mode_b = 0;
    g.cmap = 0; g.cH = g.cW = g.cpy = g.cpx = 0;
    a.rowsum = nullptr;
    a.sRowsum = 0;
    g.KW = KW;
    g.dKW = make_fastdiv(KW);
    g.bmod = 1; g.btap = 0;
    g.dBmod = make_fastdiv(1);
    a.bias = nullptr;
    a.sBias = 0;
    a.alpha = 1.f;
    a.batch_inner = 1;
    a.sAi = a.sBi = a.sCi = 0;
    a.a_vec = a.b_vec = a.c_vec = 1;
    const int64_t x_slice = (int64_t)imgs * H * W * Cin, y_slice = (int64_t)imgs * OH * OW * Cout, w_slice = (int64_t)Cout * T * Cin;
    int a_kc, b_kc;
    if (kind == 0) {          // y = conv(x, w):  A = x gathered over output pixels, B = w [co][(tap, c)]
        a.A = src; a.B = other; a.C = out;
        a.M = imgs * OH * OW; a.N = Cout; a.K = T * Cin;
        a.lda = Cin; a.ldb = (int64_t)T * Cin; a.ldc = Cout;
        a.sAo = x_slice; a.sBo = w_slice; a.sCo = y_slice;
        a.extA = x_slice; a.extB = w_slice;
        g.mode_a = 1;
        g.gH = OH; g.gW = OW; g.sH = H; g.sW = W; g.sC = Cin; g.a = stride; g.b = -pad_y; g.bx = -pad_x; g.d = dil; g.qs = 0;
        a_kc = 1; b_kc = 1;
    } else if (kind == 1) {   // dx = conv^T(dy, w):  A = dy gathered over input pixels, B rows (tap, co) remapped into w
        a.A = src; a.B = other; a.C = out;
        a.M = imgs * H * W; a.N = Cin; a.K = T * Cout;
        a.lda = Cout; a.ldb = (int64_t)T * Cin; a.ldc = Cin;
        a.sAo = y_slice; a.sBo = w_slice; a.sCo = x_slice;
        a.extA = y_slice; a.extB = w_slice;
        g.mode_a = 1; g.mode_b = 3;
        g.gH = H; g.gW = W; g.sH = OH; g.sW = OW; g.sC = Cout; g.a = 1; g.b = pad_y; g.bx = pad_x; g.d = -dil; g.qs = qs;
        g.bmod = Cout; g.btap = Cin; g.dBmod = make_fastdiv(Cout);
        a_kc = 1; b_kc = 0;
    } else {                  // dw = dy^T (x) x:  A = dy^T (co x pixels), B rows = pixels, columns (tap, c) gathered from x
        a.A = src; a.B = other; a.C = out;
        a.M = Cout; a.N = T * Cin; a.K = imgs * OH * OW;
        a.lda = Cout; a.ldb = Cin; a.ldc = (int64_t)T * Cin;
        a.sAo = y_slice; a.sBo = x_slice; a.sCo = w_slice;
        a.extA = y_slice; a.extB = x_slice;
        g.mode_b = 2;
        g.gH = OH; g.gW = OW; g.sH = H; g.sW = W; g.sC = Cin; g.a = stride; g.b = -pad_y; g.bx = -pad_x; g.d = dil; g.qs = 0;
        a_kc = 0; b_kc = 0;
    }
    (void)a_kc; (void)b_kc;
    g.dW = make_fastdiv(g.gW);
    g.dHW = make_fastdiv(g.gH * g.gW);
    g.dC = make_fastdiv(g.sC);
    // N tiles: 128 wide when the tile stays inside one tap (weight gradient: sC % BN == 0), else 64
    const int bn = (kind == 2 ? (Cin % 128 == 0) : (a.N > 64)) ? 128 : 64;
    a.tiles_m = ix_div_up(a.M, 128);
    a.tiles_n = ix_div_up(a.N, bn);
    int kps;
    const int split = conv_split(a.M, a.N, a.K, groups, &kps);
    a.split_k = split;
    a.k_per_split = kps;
    if (map) {   // (the caller checked: forward kind, fp16x3 form, no split-K)
        g.cmap = 1; g.cH = map->H; g.cW = map->W; g.cpy = map->py; g.cpx = map->px;
        a.sCo = map->x_slice;
    }
    SplitReal real;
    {
        const int rc = splitk_begin(a, groups, groups, false, workspace, workspace_bytes, real, "ix_conv_gemm_f32", stream);
        if (rc != IX_OK) return rc;
    }
    const int64_t items64 = (int64_t)a.tiles_m * a.tiles_n * groups * split;
    IX_CHECK_ARG(items64 < (1 << 30), "ix_conv_gemm_f32: too many tiles");
    const int items = (int)items64;
    const double fl = 2.0 * (double)a.M * (double)a.N * (double)a.K * (double)groups;
    g_flops += fl;
    g_launches += 1;
    const bool conv_x3 = bn == 128 && x3k_enabled() && g_x6 == 3;
    if (g_prof_on) {
        ProfRec r = {a.M, a.N, a.K, groups, a_kc, b_kc, conv_x3 ? 1129 : 1128, split};
        r.flops = fl;
        r.mfma_flops = fl * (conv_x3 ? (g_single_pass ? 1.0 : 3.0) : 6.0);
        g_rec.push_back(r);
    }
    set_item_divs(a);
    prof_mark(stream);
    const SplitEpi sep = epi_place(a, real);
    const bool epi_store = epi_in_store(a, kind == 0 && conv_x3 && !map);
    if (bn == 128) launch_conv_bn<128>(a, kind, items, stream, conv_x3, epi_store);
    else launch_conv_bn<64>(a, kind, items, stream);
    splitk_finish(a, groups, groups, false, real, stream, sep);
    prof_mark(stream);
    IX_CHECK_LAUNCH("ix_conv_gemm_f32");
    return IX_OK;
}

// wt[g][c][tap'][co] = w[g][co][ky][kx][c] for the taps of one parity class: [co][c] -> [c][co] through a 32 x 33 LDS tile
__global__ __launch_bounds__(256) void conv_s2_regroup_kernel(const float* __restrict__ w, float* __restrict__ wt, int Cin, int Cout,
                                                              int KW, int T, int nky, int nkx, int ky0, int kx0) {
    __shared__ float tile[32][33];
    const int taps = nky * nkx, gz = blockIdx.z / taps, tp = blockIdx.z - gz * taps;
    const int a = tp / nkx, b = tp - a * nkx;
    const int tap = (ky0 + 2 * (nky - 1 - a)) * KW + kx0 + 2 * (nkx - 1 - b);
    const int c0 = blockIdx.x * 32, co0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float* src = w + ((int64_t)gz * Cout * T + tap) * Cin;            // + co * T * Cin + c
    float* dst = wt + ((int64_t)gz * Cin * taps + tp) * Cout;               // + c * taps * Cout + co
#pragma unroll
    for (int r = ty; r < 32; r += 8) tile[r][tx] = src[(int64_t)(co0 + r) * T * Cin + c0 + tx];
    __syncthreads();
#pragma unroll
    for (int r = ty; r < 32; r += 8) dst[(int64_t)(c0 + r) * taps * Cout + co0 + tx] = tile[tx][r];
}

struct S2Interleave {
    const float* cls[4];   // class outputs [groups*imgs][len_y][len_x][Cin]; null: the class has no taps (zeros)
    int ly[4], lx[4];
    int in_place;          // bit c: class c wrote its pixels of dx itself (ConvGather::cmap) -- leave them alone
};
__global__ __launch_bounds__(256) void conv_s2_interleave_kernel(S2Interleave s, float* __restrict__ dx, int H, int W, int C4,
                                                                 int64_t total4) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total4; t += stride) {
        const int c4 = (int)(t % C4);
        const int64_t pix = t / C4;
        const int x = (int)(pix % W);
        const int64_t r = pix / W;
        const int y = (int)(r % H);
        const int64_t img = r / H;
        const int c = ((y & 1) << 1) | (x & 1);
        if ((s.in_place >> c) & 1) continue;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (s.cls[c])
            v = reinterpret_cast<const float4*>(s.cls[c])[((img * s.ly[c] + (y >> 1)) * s.lx[c] + (x >> 1)) * C4 + c4];
        reinterpret_cast<float4*>(dx)[t] = v;
    }
}

static int conv_bwd_data_s2(const float* dy, const float* w, float* dx, int groups, int imgs, int H, int W, int Cin, int OH, int OW,
                            int Cout, int KH, int KW, int pad, const S2Plan& p, char* scratch, hipStream_t stream) {
    char* planes = scratch;
    float* wt = reinterpret_cast<float*>(scratch + round256(p.planes));
    float* ys = reinterpret_cast<float*>(scratch + round256(p.planes) + round256((size_t)p.w_elems * 4));
    // In place where the launch can carry a row map (fp16x3 form of the forward-kind kernel, no split-K): the class writes its
    // pixels of dx itself and the interleaving pass only has the pixels of classes without taps left (none for a 3 x 3).
    bool direct = conv_uses_x3(0, Cin, Cout) && g_conv_s2_in_place != 0;
    for (int c = 0; c < 4 && direct; ++c) {
        const int64_t taps = (int64_t)p.ay[c].n * p.ax[c].n, px = (int64_t)p.ay[c].len * p.ax[c].len;
        int kps;
        if (taps && px && conv_split((int)(imgs * px), Cin, (int)(taps * Cout), groups, &kps) > 1) direct = false;
    }
    S2Interleave il;
    il.in_place = 0;
    int left = 0;   // classes the interleaving pass still has to write
    for (int c = 0; c < 4; ++c) {
        const S2Axis &ay = p.ay[c], &ax = p.ax[c];
        il.cls[c] = nullptr;
        il.ly[c] = ay.len; il.lx[c] = ax.len;
        if (ay.len * ax.len == 0) continue;
        if (ay.n * ax.n == 0) { ++left; continue; }
        float* wc = wt + p.w_off[c];
        float* yc = ys + p.y_off[c];
        il.cls[c] = yc;
        if (direct) il.in_place |= 1 << c; else ++left;
        hipLaunchKernelGGL(conv_s2_regroup_kernel, dim3(Cin / 32, Cout / 32, groups * ay.n * ax.n), dim3(256), 0, stream, w, wc, Cin,
                           Cout, KW, KH * KW, ay.n, ax.n, ay.k0, ax.k0);
        // the class as a forward convolution: source dy [imgs][OH][OW][Cout], "Cin" = Cout, "Cout" = Cin, output grid len_y x len_x
        const S2Map map = {H, W, c >> 1, c & 1, (int64_t)imgs * H * W * Cin};
        const int rc = conv_gemm_core(0, dy, wc, direct ? dx : yc, groups, imgs, OH, OW, Cout, ay.len, ax.len, Cin, ay.n, ax.n, 1, ay.pad,
                                      ax.pad, 1, p.planes ? planes : nullptr, p.planes, stream, direct ? &map : nullptr);
        if (rc != IX_OK) return rc;
    }
    const int64_t total4 = (int64_t)groups * imgs * H * W * Cin / 4;
    if (left)
        hipLaunchKernelGGL(conv_s2_interleave_kernel, dim3(ix_grid_1d(total4, 256)), dim3(256), 0, stream, il, dx, H, W, Cin / 4, total4);
    IX_CHECK_LAUNCH("ix_conv_gemm_f32 (stride-2 data gradient)");
    return IX_OK;
}

extern "C" int ix_conv_gemm_f32(int kind, const float* src, const float* other, float* out, int groups, int imgs, int H, int W,
                                int Cin, int OH, int OW, int Cout, int KH, int KW, int stride, int pad, int dil,
                                void* workspace, size_t workspace_bytes, hipStream_t stream) {
    IX_CHECK_ARG(src && other && out, "ix_conv_gemm_f32: null operand");
    IX_CHECK_ARG(kind >= 0 && kind <= 2, "ix_conv_gemm_f32: kind %d", kind);
    IX_CHECK_ARG(ix_conv_gemm_supported(groups, imgs, H, W, Cin, OH, OW, Cout, KH, KW, stride, pad, dil),
                 "ix_conv_gemm_f32: unsupported geometry (Cin %d, Cout %d must be multiples of 64; stride %d in {1, 2, 4}; a group's "
                 "tensors below 2 GiB) -- use ix_im2col_f32 + ix_gemm_f32",
                 Cin, Cout, stride);
    IX_CHECK_ARG(aligned16(src) && aligned16(other) && aligned16(out), "ix_conv_gemm_f32: operands must be 16-byte aligned");
    if (workspace) {   // [IX_TICKET_BYTES of reduction tickets][scratch], as every entry point
        IX_CHECK_ARG(workspace_bytes >= IX_TICKET_BYTES && aligned16(workspace), "ix_conv_gemm_f32: workspace below %d bytes or unaligned", IX_TICKET_BYTES);
        workspace = static_cast<char*>(workspace) + IX_TICKET_BYTES;
        workspace_bytes -= IX_TICKET_BYTES;
    }
    if (workspace && conv_takes_s2_split(kind, stride, dil, groups, imgs, H, W, Cin, Cout, KH, KW)) {
        const S2Plan p = s2_plan(groups, imgs, H, W, Cin, Cout, KH, KW, pad);
        const size_t need = round256(p.planes) + round256((size_t)p.w_elems * 4) + round256((size_t)p.y_elems * 4);
        if (workspace_bytes >= need && (reinterpret_cast<uintptr_t>(workspace) & 255) == 0)
            return conv_bwd_data_s2(src, other, out, groups, imgs, H, W, Cin, OH, OW, Cout, KH, KW, pad, p, static_cast<char*>(workspace),
                                    stream);
    }
    return conv_gemm_core(kind, src, other, out, groups, imgs, H, W, Cin, OH, OW, Cout, KH, KW, stride, pad, pad, dil, workspace,
                          workspace_bytes, stream);
}

// ---- contraction + frozen-BN affine (+ residual) (+ ReLU) as ONE launch where the kernel can take it -------------------------
// y = [relu]((A B) * scale[n] + shift[n] (+ residual[m][n])): reference models/detr_models/backbone.py:19-54 (FrozenBatchNorm2d
// behind every backbone convolution) with torchvision's Bottleneck tail `out += identity; out = relu(out)`.  A split-K
// launch applies it in its ordered reduction (one launch fewer where launches are what a step costs); an unsplit launch is
// followed by the affine pass as its own launch (ix_channel_affine_f32) -- the same result.  C: dense [batch][M][N].
extern "C" int ix_channel_affine_f32(const float* x, const float* scale, const float* shift, const float* residual, float* out,
                                     int64_t n, int C, int relu, hipStream_t stream);
static int epi_finish(int rc, float* C, int64_t total, int N, const float* scale, const float* shift, const float* residual,
                      int relu, hipStream_t stream) {
    const int applied = g_epi.applied;
    g_epi = EpiReq{nullptr, nullptr, nullptr, 0, 0};
    if (rc != IX_OK || applied) return rc;
    ++g_epi_count[1];
    return ix_channel_affine_f32(C, scale, shift, residual, C, total, N, relu, stream);
}

extern "C" int ix_gemm_bn_act_f32(const float* A, const float* B, float* C, int M, int N, int K, int a_kcontig, int b_kcontig,
                                  int64_t lda, int64_t ldb, int batch_outer, int64_t sAo, int64_t sBo, const float* scale,
                                  const float* shift, const float* residual, int relu, void* workspace, size_t workspace_bytes,
                                  hipStream_t stream) {
    IX_CHECK_ARG(scale && shift, "ix_gemm_bn_act_f32: null scale / shift");
    IX_CHECK_ARG(N % 4 == 0 && aligned16(C) && aligned16(scale) && aligned16(shift) && (!residual || aligned16(residual)),
                 "ix_gemm_bn_act_f32: N %% 4 == 0 and 16-byte aligned C / scale / shift / residual needed");
    g_epi = EpiReq{scale, shift, residual, relu, 0};
    const int rc = ix_gemm_f32_ws(A, B, C, nullptr, M, N, K, a_kcontig, b_kcontig, lda, ldb, N, batch_outer, 1, sAo, 0, sBo, 0,
                                  (int64_t)M * N, 0, 0, 1.f, 0, 0, workspace, workspace_bytes, stream);
    return epi_finish(rc, C, (int64_t)batch_outer * M * N, N, scale, shift, residual, relu, stream);
}

extern "C" int ix_conv_gemm_bn_act_f32(const float* x, const float* w, float* y, int groups, int imgs, int H, int W, int Cin,
                                       int OH, int OW, int Cout, int KH, int KW, int stride, int pad, int dil, const float* scale,
                                       const float* shift, const float* residual, int relu, void* workspace,
                                       size_t workspace_bytes, hipStream_t stream) {
    IX_CHECK_ARG(scale && shift, "ix_conv_gemm_bn_act_f32: null scale / shift");
    IX_CHECK_ARG(aligned16(scale) && aligned16(shift) && (!residual || aligned16(residual)),
                 "ix_conv_gemm_bn_act_f32: scale / shift / residual must be 16-byte aligned");
    g_epi = EpiReq{scale, shift, residual, relu, 0};
    const int rc = ix_conv_gemm_f32(0, x, w, y, groups, imgs, H, W, Cin, OH, OW, Cout, KH, KW, stride, pad, dil, workspace,
                                    workspace_bytes, stream);
    return epi_finish(rc, y, (int64_t)groups * imgs * OH * OW * Cout, Cout, scale, shift, residual, relu, stream);
}
