// fp32 GEMM on the fp16 matrix cores from PRE-SPLIT operands ("f16x3"): the second contraction kernel behind ix_gemm_f32
// (same reference call sites as gemm.hip: nn.Linear / conv-as-GEMM of models/detr_models/transformer.py:148-232,
// models/gpt.py:39-78, models/detr_models/backbone.py:88-90 and their autograd derivatives).
//
// Why.  The bf16x6 kernel (gemm.hip) converts every fp32 operand panel to bf16 planes on the fly, once per output tile it
// feeds (15-36 times per element), and issues six matrix instructions per 16 contracted elements; it is bound by the
// producers' conversion work and by the package power.  Here each operand is converted ONCE, by x3_split_kernel, into two
// fp16 planes  x 2^e = h + l  (22-23 significant bits) with one power-of-two scale per block of 32 rows taken over the
// WHOLE contracted extent, and the GEMM kernel streams 16-bit planes (no conversion arithmetic at all) and issues the three
// terms  l.h + h.l + h.h  on v_mfma_f32_32x32x16_f16 -- the arithmetic of the attention kernels' head-dim products
// (flash.hip), accuracy class of an fp32 dot product (tests/test_ops_gpu.py::test_f16x3_presplit_contraction_is_fp32_grade).
// The block scales of the two operands are undone by one multiply per 32x32 accumulator block in the epilogue.
//
// Canonical operand form: BOTH operands as row-major planes [batch][rows padded to 128][K padded to 32] with the
// contracted index contiguous ("TN"), whatever layout the fp32 operand had -- the split kernel transposes through LDS when
// the fp32 operand is stored with the other index contiguous.  One layout in the GEMM kernel, fragments = one
// ds_read_b128 per lane.
#include "common.h"
#include "gemm_x3.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void x3_split2h(float x0, float x1, unsigned& h, unsigned& l) {
    f32x2 v;
    v.x = x0; v.y = x1;
    const f16x2 hh = __builtin_convertvector(v, f16x2);
    const f32x2 back = __builtin_convertvector(hh, f32x2);
    f32x2 r;
    r.x = x0 - back.x; r.y = x1 - back.y;
    h = __builtin_bit_cast(unsigned, hh);
    l = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2));
}

// ------------------------------------------------------------------------------------------------------------
// split: fp32 operand view X(r, k), r < R, k < K  ->  planes [2][nb][Rp][Kp] fp16 + unscale [nb][Rp / 32]
//   KC: X(r, k) = X[r * ld + k];  !KC: X(r, k) = X[k * ld + r].   One workgroup per (32-row block, batch slice): pass 1
//   finds the block's largest magnitude over all K, pass 2 re-reads the slab (L2 for the usual extents) and converts.
// ------------------------------------------------------------------------------------------------------------
template <bool KC>
__global__ __launch_bounds__(256) void x3_split_kernel(const float* __restrict__ X, int64_t ld, int64_t sb, int R, int K, int Rp,
                                                       int Kp, unsigned short* __restrict__ planes, float* __restrict__ unscale,
                                                       int64_t plane_elems) {
    __shared__ float red[4];
    __shared__ float tile[32][33];
    const int tid = threadIdx.x, r0 = blockIdx.x * 32, b = blockIdx.y;
    const float* base = X + (int64_t)b * sb;
    // KC: thread = (row tid/8, four consecutive k at 4*(tid%8));  !KC: thread = (k line tid/8, four consecutive rows at 4*(tid%8))
    const int hi = tid >> 3, lo4 = (tid & 7) * 4;
    const int nkt = Kp / 32;
    auto load4 = [&](int kt) -> float4 {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (KC) {
            const int r = r0 + hi, k = kt * 32 + lo4;
            if (r < R && k < K) {
                const float* p = base + (int64_t)r * ld + k;
                if (k + 3 < K) v = *reinterpret_cast<const float4*>(p);
                else { v.x = p[0]; if (k + 1 < K) v.y = p[1]; if (k + 2 < K) v.z = p[2]; }
            }
        } else {
            const int k = kt * 32 + hi, r = r0 + lo4;
            if (k < K && r < R) {
                const float* p = base + (int64_t)k * ld + r;
                if (r + 3 < R) v = *reinterpret_cast<const float4*>(p);
                else { v.x = p[0]; if (r + 1 < R) v.y = p[1]; if (r + 2 < R) v.z = p[2]; }
            }
        }
        return v;
    };
    float mx = 0.f;
    for (int kt = 0; kt < nkt; ++kt) {
        const float4 v = load4(kt);
        mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    mx = ix_block_max_256(mx, red);
    const unsigned e = (__float_as_uint(mx) >> 23) & 0xffu;
    const bool tiny = e < 16u || e > 250u;                               // zero / denormal / inf block: unscaled
    const float sc = tiny ? 1.f : __uint_as_float((268u - e) << 23);     // block maximum into [2^14, 2^15)
    const float us = tiny ? 1.f : __uint_as_float((e - 14u) << 23);
    if (tid == 0) unscale[(int64_t)b * (Rp / 32) + blockIdx.x] = us;
    unsigned short* out = planes + ((int64_t)b * Rp + r0) * Kp;
    for (int kt = 0; kt < nkt; ++kt) {
        float4 v = load4(kt);
        if (!KC) {   // [k line][4 rows] -> [row][4 k] through LDS
            __syncthreads();
            tile[hi][lo4] = v.x; tile[hi][lo4 + 1] = v.y; tile[hi][lo4 + 2] = v.z; tile[hi][lo4 + 3] = v.w;
            __syncthreads();
            v.x = tile[lo4][hi]; v.y = tile[lo4 + 1][hi]; v.z = tile[lo4 + 2][hi]; v.w = tile[lo4 + 3][hi];
        }
        unsigned h0, l0, h1, l1;
        x3_split2h(v.x * sc, v.y * sc, h0, l0);
        x3_split2h(v.z * sc, v.w * sc, h1, l1);
        unsigned short* dst = out + (int64_t)hi * Kp + kt * 32 + lo4;
        *reinterpret_cast<uint2*>(dst) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(dst + plane_elems) = make_uint2(l0, l1);
    }
}

// ------------------------------------------------------------------------------------------------------------
// GEMM on the planes: C[b](M x N) = alpha * A[b] B[b]^T (+ bias), A planes [Mp][Kp], B planes [Np][Kp].
// 256 threads = 2 x 2 waves, tile 128 x BN x 32; one LDS image of the tile (2 operands x 2 planes, 80-byte rows), the
// next K tile travels through registers while this one is multiplied (two barriers per K tile; two to three workgroups
// per CU overlap them).  Split-K by fp32 atomics into a zeroed C.
// ------------------------------------------------------------------------------------------------------------
struct X3Args {
    const unsigned short *A, *B;     // plane 0 (h); plane 1 (l) at + a_plane / b_plane elements
    const float *usA, *usB;          // [nbA][Mp/32], [nbB][Np/32]
    float* C;
    const float* bias;
    int M, N, Mp, Np, Kp;
    int64_t a_plane, b_plane, a_batch, b_batch;   // elements
    int a_usb, b_usb;                             // unscale entries per batch slice (0: operand shared by all slices)
    int64_t ldc, sC, sBias;
    int tiles_m, tiles_n, split_k, kt_per_split;
    float alpha;
};

constexpr int X3_ROWB = 80, X3_BM = 128;

__device__ __forceinline__ int x3_xcd_swizzle(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

template <int BN>
__global__ __launch_bounds__(256, 2) void gemm_f16x3_kernel(X3Args p) {
    constexpr int BM = X3_BM, WM = 64, WN = BN / 2, TM = 2, TN = WN / 32;
    constexpr int PA = BM * X3_ROWB, PB = BN * X3_ROWB;            // bytes per plane image
    constexpr int OFF_B = 2 * PA, BYTES = 2 * PA + 2 * PB;
    constexpr int NCA = BM * 4 / 256, NCB = BN * 4 / 256;          // 16-byte chunks per thread and plane (rows x 4 chunks of 8 k)
    constexpr int CP = WN + 4;                                     // epilogue strip pitch (floats)
    static_assert(4 * 32 * CP * 4 <= BYTES, "epilogue strips must fit the operand image");
    __shared__ __attribute__((aligned(16))) unsigned char lds[BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 31, a = lane >> 5;
    const int nwg = p.tiles_m * p.tiles_n;
    const int tile = x3_xcd_swizzle(blockIdx.x, nwg);
    constexpr int GROUP_M = 8;
    const int group_size = GROUP_M * p.tiles_n;
    const int first_m = (tile / group_size) * GROUP_M;
    const int gm = min(p.tiles_m - first_m, GROUP_M);
    const int m0 = (first_m + (tile % group_size) % gm) * BM, n0 = ((tile % group_size) / gm) * BN;
    const int zb = blockIdx.y, ks = blockIdx.z;
    const int kt0 = ks * p.kt_per_split, kt1 = min(p.Kp / 32, kt0 + p.kt_per_split);
    const unsigned short* Ab = p.A + (int64_t)zb * p.a_batch + (int64_t)m0 * p.Kp;
    const unsigned short* Bb = p.B + (int64_t)zb * p.b_batch + (int64_t)n0 * p.Kp;
    const int wm = (wave >> 1) * WM, wn = (wave & 1) * WN;

    // staging registers: plain named variables (a value loaded before a barrier and stored to LDS after it must not live
    // in an array, see flash.hip)
    uint4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;   // A: plane h chunks 0,1, plane l chunks 0,1; B likewise (BN 64: one chunk)
#define X3_SRC(BASE, PLANE, C, KT) ((BASE) + (PLANE) + (int64_t)((C) >> 2) * p.Kp + (KT) * 32 + ((C) & 3) * 8)
#define X3_LOAD(KT)                                                                                   \
    ra0 = *reinterpret_cast<const uint4*>(X3_SRC(Ab, 0, tid, KT));                                    \
    ra1 = *reinterpret_cast<const uint4*>(X3_SRC(Ab, 0, tid + 256, KT));                              \
    ra2 = *reinterpret_cast<const uint4*>(X3_SRC(Ab, p.a_plane, tid, KT));                            \
    ra3 = *reinterpret_cast<const uint4*>(X3_SRC(Ab, p.a_plane, tid + 256, KT));                      \
    rb0 = *reinterpret_cast<const uint4*>(X3_SRC(Bb, 0, tid, KT));                                    \
    rb2 = *reinterpret_cast<const uint4*>(X3_SRC(Bb, p.b_plane, tid, KT));                            \
    if (NCB == 2) {                                                                                   \
        rb1 = *reinterpret_cast<const uint4*>(X3_SRC(Bb, 0, tid + 256, KT));                          \
        rb3 = *reinterpret_cast<const uint4*>(X3_SRC(Bb, p.b_plane, tid + 256, KT));                  \
    }
#define X3_DST(OFF, C) (lds + (OFF) + ((C) >> 2) * X3_ROWB + ((C) & 3) * 16)
#define X3_STORE()                                                                                    \
    *reinterpret_cast<uint4*>(X3_DST(0, tid)) = ra0;                                                  \
    *reinterpret_cast<uint4*>(X3_DST(0, tid + 256)) = ra1;                                            \
    *reinterpret_cast<uint4*>(X3_DST(PA, tid)) = ra2;                                                 \
    *reinterpret_cast<uint4*>(X3_DST(PA, tid + 256)) = ra3;                                           \
    *reinterpret_cast<uint4*>(X3_DST(OFF_B, tid)) = rb0;                                              \
    *reinterpret_cast<uint4*>(X3_DST(OFF_B + PB, tid)) = rb2;                                         \
    if (NCB == 2) {                                                                                   \
        *reinterpret_cast<uint4*>(X3_DST(OFF_B, tid + 256)) = rb1;                                    \
        *reinterpret_cast<uint4*>(X3_DST(OFF_B + PB, tid + 256)) = rb3;                               \
    }
    static_assert(NCA == 2 && (NCB == 2 || NCB == 1), "tile shape");
    rb1 = make_uint4(0, 0, 0, 0); rb3 = rb1;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (kt0 < kt1) {
        X3_LOAD(kt0)
        X3_STORE()
    }
    __syncthreads();
    for (int kt = kt0; kt < kt1; ++kt) {
        X3_LOAD(min(kt + 1, kt1 - 1))   // (the last iteration re-requests its own tile: unconditional code)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            u32x4 fa[TM][2], fb[TN][2];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl)
                    fa[i][pl] = *reinterpret_cast<const u32x4*>(lds + pl * PA + (wm + i * 32 + lr) * X3_ROWB + (s * 16 + 8 * a) * 2);
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl)
                    fb[j][pl] = *reinterpret_cast<const u32x4*>(lds + OFF_B + pl * PB + (wn + j * 32 + lr) * X3_ROWB + (s * 16 + 8 * a) * 2);
            // C[m, n]: MFMA A operand = rows of A (m), B operand = rows of B (n); smallest terms first
#define X3_TERM(PA_, PB_)                                                                                              \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j) acc[i][j] =           \
        __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[i][PA_]), __builtin_bit_cast(f16x8, fb[j][PB_]), acc[i][j], 0, 0, 0);
            X3_TERM(1, 0) X3_TERM(0, 1) X3_TERM(0, 0)
#undef X3_TERM
        }
        __syncthreads();   // every wave is done with this K tile
        X3_STORE()
        __syncthreads();   // the next one is visible
    }
#undef X3_LOAD
#undef X3_STORE
#undef X3_SRC
#undef X3_DST

    // ---- epilogue: undo the block scales, alpha, bias; row-major 16-byte stores through a private LDS strip per wave ----
    const float* usA = p.usA + (int64_t)zb * p.a_usb + (m0 + wm) / 32;
    const float* usB = p.usB + (int64_t)zb * p.b_usb + (n0 + wn) / 32;
    float* C = p.C + (int64_t)zb * p.sC;
    const float* bias = p.bias ? p.bias + (int64_t)zb * p.sBias : nullptr;
    const bool add_bias = bias != nullptr && ks == 0;
    float* strip = reinterpret_cast<float*>(lds) + wave * 32 * CP;
    const bool vec = p.split_k == 1 && (p.ldc % 4 == 0) && ((reinterpret_cast<uintptr_t>(C) & 15) == 0);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const float sa = usA[i] * p.alpha;
        const int r0 = m0 + wm + i * 32;
        if (vec && r0 + 32 <= p.M && n0 + wn + WN <= p.N) {   // interior strip (wave-uniform)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const float sc = sa * usB[j];
                const int col = n0 + wn + j * 32 + lr;
                const float bv = add_bias ? bias[col] : 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) strip[((r & 3) + 8 * (r >> 2) + 4 * a) * CP + j * 32 + lr] = acc[i][j][r] * sc + bv;
            }
            __builtin_amdgcn_wave_barrier();
            constexpr int CPR = WN / 4, NQ = 32 * CPR / 64;
            float4 v[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int c = lane + 64 * q;
                v[q] = *reinterpret_cast<const float4*>(&strip[(c / CPR) * CP + (c % CPR) * 4]);
            }
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int c = lane + 64 * q;
                *reinterpret_cast<float4*>(C + (int64_t)(r0 + c / CPR) * p.ldc + n0 + wn + (c % CPR) * 4) = v[q];
            }
            __builtin_amdgcn_wave_barrier();
        } else {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const float sc = sa * usB[j];
                const int col = n0 + wn + j * 32 + lr;
                if (col >= p.N) continue;
                const float bv = add_bias ? bias[col] : 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = r0 + (r & 3) + 8 * (r >> 2) + 4 * a;
                    if (row < p.M) {
                        float* dst = C + (int64_t)row * p.ldc + col;
                        const float val = acc[i][j][r] * sc + bv;
                        if (p.split_k > 1) unsafeAtomicAdd(dst, val);
                        else *dst = val;
                    }
                }
            }
        }
    }
}

__global__ void x3_zero_kernel(float* C, int M, int N, int64_t ldc, int64_t sC) {
    float* c = C + (int64_t)blockIdx.y * sC;
    const int64_t total = (int64_t)M * N;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
        c[(i / N) * ldc + (i % N)] = 0.f;
}

static inline int64_t x3_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

// bytes of workspace for one call: planes of A and B (shared operands once) + their unscale factors
size_t ix_x3_workspace_bytes(int M, int N, int K, int nbA, int nbB) {
    const int64_t Mp = x3_up(M, 128), Np = x3_up(N, 128), Kp = x3_up(K, 32);
    const int64_t pa = x3_up((int64_t)nbA * Mp * Kp * 2 * 2, 256), pb = x3_up((int64_t)nbB * Np * Kp * 2 * 2, 256);
    const int64_t ua = x3_up((int64_t)nbA * (Mp / 32) * 4, 256), ub = x3_up((int64_t)nbB * (Np / 32) * 4, 256);
    return (size_t)(pa + pb + ua + ub);
}

int ix_x3_gemm(const X3Call& c, void* workspace, hipStream_t stream) {
    const int64_t Mp = x3_up(c.M, 128), Np = x3_up(c.N, 128), Kp = x3_up(c.K, 32);
    const int nbA = c.sA ? c.nbatch : 1, nbB = c.sB ? c.nbatch : 1;
    unsigned char* w = (unsigned char*)workspace;
    unsigned short* pA = (unsigned short*)w;                       w += x3_up((int64_t)nbA * Mp * Kp * 4, 256);
    unsigned short* pB = (unsigned short*)w;                       w += x3_up((int64_t)nbB * Np * Kp * 4, 256);
    float* uA = (float*)w;                                         w += x3_up((int64_t)nbA * (Mp / 32) * 4, 256);
    float* uB = (float*)w;
    const int64_t planeA = (int64_t)nbA * Mp * Kp, planeB = (int64_t)nbB * Np * Kp;
    // A(m, k): a_kcontig -> A[m * lda + k];  B(k, n): b_kcontig -> B[n * ldb + k] (rows of the canonical form = n)
    if (c.a_kc) hipLaunchKernelGGL(x3_split_kernel<true>, dim3(Mp / 32, nbA), dim3(256), 0, stream, c.A, c.lda, c.sA, c.M, c.K, (int)Mp, (int)Kp, pA, uA, planeA);
    else hipLaunchKernelGGL(x3_split_kernel<false>, dim3(Mp / 32, nbA), dim3(256), 0, stream, c.A, c.lda, c.sA, c.M, c.K, (int)Mp, (int)Kp, pA, uA, planeA);
    if (c.b_kc) hipLaunchKernelGGL(x3_split_kernel<true>, dim3(Np / 32, nbB), dim3(256), 0, stream, c.B, c.ldb, c.sB, c.N, c.K, (int)Np, (int)Kp, pB, uB, planeB);
    else hipLaunchKernelGGL(x3_split_kernel<false>, dim3(Np / 32, nbB), dim3(256), 0, stream, c.B, c.ldb, c.sB, c.N, c.K, (int)Np, (int)Kp, pB, uB, planeB);
    X3Args a;
    a.A = pA; a.B = pB; a.usA = uA; a.usB = uB; a.C = c.C; a.bias = c.bias;
    a.M = c.M; a.N = c.N; a.Mp = (int)Mp; a.Np = (int)Np; a.Kp = (int)Kp;
    a.a_plane = planeA; a.b_plane = planeB;
    a.a_batch = c.sA ? Mp * Kp : 0; a.b_batch = c.sB ? Np * Kp : 0;
    a.a_usb = c.sA ? (int)(Mp / 32) : 0; a.b_usb = c.sB ? (int)(Np / 32) : 0;
    a.ldc = c.ldc; a.sC = c.sC; a.sBias = c.sBias; a.alpha = c.alpha;
    const int bn = c.N > 64 ? 128 : 64;
    a.tiles_m = (int)(Mp / 128); a.tiles_n = (c.N + bn - 1) / bn;
    // split-K: fill the chip for skinny outputs (weight gradients: small M x N, K = tokens)
    const int nkt = (int)(Kp / 32);
    const int64_t wgs = (int64_t)a.tiles_m * a.tiles_n * c.nbatch;
    int split = 1;
    if (wgs < 256 && nkt >= 32) {
        split = (int)((512 + wgs - 1) / wgs);
        if (split > nkt / 8) split = nkt / 8;
        if (split < 1) split = 1;
    }
    a.kt_per_split = (nkt + split - 1) / split;
    split = (nkt + a.kt_per_split - 1) / a.kt_per_split;
    a.split_k = split;
    if (split > 1) {
        int g = (int)(((int64_t)c.M * c.N + 255) / 256);
        if (g > 2048) g = 2048;
        hipLaunchKernelGGL(x3_zero_kernel, dim3(g, c.nbatch), dim3(256), 0, stream, c.C, c.M, c.N, c.ldc, c.sC);
    }
    dim3 grid(a.tiles_m * a.tiles_n, c.nbatch, split);
    if (bn == 128) hipLaunchKernelGGL(gemm_f16x3_kernel<128>, grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(gemm_f16x3_kernel<64>, grid, dim3(256), 0, stream, a);
    return 0;
}
