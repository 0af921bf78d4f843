// HBM-bound elementwise / channel / row kernels of the 16-bit activation mode (MODEL.COMPUTE_DTYPE: bf16, b16.py): the bf16 twins of
// csrc/elementwise.hip and of the LayerNorm kernels of csrc/rowwise.hip.  Tensors are bf16 in HBM (2 bytes per element instead of 4:
// these passes are bandwidth-bound, so that is their speed-up), every value is widened to fp32 in registers, the arithmetic is the
// fp32 kernels' (same formulas, same dropout hash of (seed, element index): a bf16 result is the fp32 kernel's result of the same
// inputs, rounded once), statistics / parameter gradients / per-channel vectors stay fp32.
// Reference sites: residual adds, ReLU / GELU / dropout of models/detr_models/transformer.py:148-232 and models/gpt.py:39-78, the
// FrozenBatchNorm2d affine of models/detr_models/backbone.py:44-54, nn.LayerNorm of both transformers, and their first derivatives.
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned short bf16_t;

#define E16_BLOCK 256

__device__ __forceinline__ float e16_lo(unsigned v) { return __uint_as_float(v << 16); }
__device__ __forceinline__ float e16_hi(unsigned v) { return __uint_as_float(v & 0xffff0000u); }
__device__ __forceinline__ float e16_f(bf16_t v) { return __uint_as_float((unsigned)v << 16); }
__device__ __forceinline__ unsigned e16_pack(float a, float b) {   // round to nearest even
    unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
    ua = (ua + 0x7fffu + ((ua >> 16) & 1u)) >> 16;
    ub = (ub + 0x7fffu + ((ub >> 16) & 1u)) & 0xffff0000u;
    return ua | ub;
}
__device__ __forceinline__ bf16_t e16_b(float a) { return (bf16_t)(e16_pack(a, 0.f) & 0xffffu); }
struct F8 { float v[8]; };
__device__ __forceinline__ F8 e16_ld8(const bf16_t* p) {
    const u32x4 r = *reinterpret_cast<const u32x4*>(p);
    F8 o;
    o.v[0] = e16_lo(r.x); o.v[1] = e16_hi(r.x); o.v[2] = e16_lo(r.y); o.v[3] = e16_hi(r.y);
    o.v[4] = e16_lo(r.z); o.v[5] = e16_hi(r.z); o.v[6] = e16_lo(r.w); o.v[7] = e16_hi(r.w);
    return o;
}
__device__ __forceinline__ void e16_st8(bf16_t* p, const F8& o) {
    *reinterpret_cast<u32x4*>(p) = u32x4{e16_pack(o.v[0], o.v[1]), e16_pack(o.v[2], o.v[3]), e16_pack(o.v[4], o.v[5]), e16_pack(o.v[6], o.v[7])};
}
__device__ __forceinline__ uint32_t e16_mix32(uint64_t z) {   // the dropout hash of csrc/elementwise.hip
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return (uint32_t)((z ^ (z >> 31)) >> 32);
}
__device__ __forceinline__ bool e16_keep(uint64_t seed, int64_t k, uint32_t thresh) {
    return e16_mix32(seed ^ ((uint64_t)k * 0xD6E8FEB86659FD93ull)) >= thresh;
}
__device__ __forceinline__ float e16_gelu_cdf(float x) { return 0.5f * (1.f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float e16_gelu_pdf(float x) { return 0.39894228040143267794f * __expf(-0.5f * x * x); }

// ---- ix_map_b16: out[k] = f_op(a[k], b[k], c[k]) ---------------------------------------------------------------------------------
enum {
    E16_ADD = 0,          // a + b
    E16_AXPBY = 1,        // p0 a + p1 b
    E16_SCALE = 2,        // p0 a
    E16_RELU = 3,         // max(a, 0)
    E16_RELU_BWD = 4,     // a [b > 0] p0            (a = dy, b = y; p0 = 1 or 1 / keep)
    E16_RELU_BWD_SUM = 5, // (a + b) [c > 0]
    E16_GELU = 6,         // a Phi(a)
    E16_GELU_BWD = 7,     // a (Phi(b) + b phi(b))   (a = dy, b = x)
    E16_DROPOUT = 8,      // keep(k) a / (1 - p0)
    E16_RELU_DROPOUT = 9, // keep(k) max(a, 0) / (1 - p0)
    E16_ADD_DROPOUT = 10, // a + keep(k) b / (1 - p0)
    E16_OPS = 11
};
struct MapArgs {
    const bf16_t *a, *b, *c;
    bf16_t* out;
    int64_t n;
    float p0, p1;
    uint64_t seed;
    uint32_t thresh;
    const uint64_t* salt;
};
template <int OP>
__device__ __forceinline__ float e16_apply(float a, float b, float c, const MapArgs& p, float inv_keep, uint64_t seed, int64_t k) {
    switch (OP) {
        case E16_ADD: return a + b;
        case E16_AXPBY: return p.p0 * a + p.p1 * b;
        case E16_SCALE: return p.p0 * a;
        case E16_RELU: return a > 0.f ? a : 0.f;
        case E16_RELU_BWD: return b > 0.f ? a * p.p0 : 0.f;
        case E16_RELU_BWD_SUM: return c > 0.f ? a + b : 0.f;
        case E16_GELU: return a * e16_gelu_cdf(a);
        case E16_GELU_BWD: return a * (e16_gelu_cdf(b) + b * e16_gelu_pdf(b));
        case E16_DROPOUT: return e16_keep(seed, k, p.thresh) ? a * inv_keep : 0.f;
        case E16_RELU_DROPOUT: return (e16_keep(seed, k, p.thresh) && a > 0.f) ? a * inv_keep : 0.f;
        case E16_ADD_DROPOUT: return a + (e16_keep(seed, k, p.thresh) ? b * inv_keep : 0.f);
    }
    return 0.f;
}
template <int OP>
__global__ __launch_bounds__(E16_BLOCK) void map_b16_kernel(MapArgs p) {
    constexpr bool HAS_B = OP == E16_ADD || OP == E16_AXPBY || OP == E16_RELU_BWD || OP == E16_RELU_BWD_SUM || OP == E16_GELU_BWD || OP == E16_ADD_DROPOUT;
    constexpr bool HAS_C = OP == E16_RELU_BWD_SUM;
    uint64_t seed = p.seed;
    if (OP >= E16_DROPOUT && p.salt) seed ^= *p.salt;
    const float inv_keep = OP >= E16_DROPOUT ? 1.f / (1.f - p.p0) : 1.f;
    const int64_t n8 = p.n >> 3, stride = (int64_t)gridDim.x * E16_BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * E16_BLOCK + threadIdx.x; i < n8; i += stride) {
        const F8 a = e16_ld8(p.a + 8 * i);
        F8 b = a, c = a, o;
        if (HAS_B) b = e16_ld8(p.b + 8 * i);
        if (HAS_C) c = e16_ld8(p.c + 8 * i);
#pragma unroll
        for (int e = 0; e < 8; ++e) o.v[e] = e16_apply<OP>(a.v[e], b.v[e], c.v[e], p, inv_keep, seed, 8 * i + e);
        e16_st8(p.out + 8 * i, o);
    }
    if (blockIdx.x == 0)
        for (int64_t k = (n8 << 3) + threadIdx.x; k < p.n; k += E16_BLOCK)
            p.out[k] = e16_b(e16_apply<OP>(e16_f(p.a[k]), HAS_B ? e16_f(p.b[k]) : 0.f, HAS_C ? e16_f(p.c[k]) : 0.f, p, inv_keep, seed, k));
}

extern "C" int ix_map_b16(int op, const void* a, const void* b, const void* c, void* out, int64_t n, float p0, float p1, uint64_t seed,
                          hipStream_t stream) {
    if (n <= 0) return IX_OK;
    IX_CHECK_ARG(op >= 0 && op < E16_OPS, "ix_map_b16: unknown op %d", op);
    IX_CHECK_ARG(a && out && ix_al16(a) && ix_al16(out) && (!b || ix_al16(b)) && (!c || ix_al16(c)), "ix_map_b16: null or unaligned pointer");
    const bool need_b = op == E16_ADD || op == E16_AXPBY || op == E16_RELU_BWD || op == E16_RELU_BWD_SUM || op == E16_GELU_BWD || op == E16_ADD_DROPOUT;
    IX_CHECK_ARG((!need_b || b) && (op != E16_RELU_BWD_SUM || c), "ix_map_b16: op %d needs more operands", op);
    IX_CHECK_ARG(op < E16_DROPOUT || (p0 >= 0.f && p0 < 1.f), "ix_map_b16: p=%f outside [0,1)", p0);
    MapArgs m = {(const bf16_t*)a, (const bf16_t*)b, (const bf16_t*)c, (bf16_t*)out, n, p0, p1, seed, (uint32_t)((double)p0 * 4294967296.0), ix_g_salt};
    const dim3 grid(ix_grid_1d(n / 8 + 1, E16_BLOCK)), block(E16_BLOCK);
#define E16_CASE(OP) case OP: hipLaunchKernelGGL(map_b16_kernel<OP>, grid, block, 0, stream, m); break;
    switch (op) {
        E16_CASE(E16_ADD) E16_CASE(E16_AXPBY) E16_CASE(E16_SCALE) E16_CASE(E16_RELU) E16_CASE(E16_RELU_BWD) E16_CASE(E16_RELU_BWD_SUM)
        E16_CASE(E16_GELU) E16_CASE(E16_GELU_BWD) E16_CASE(E16_DROPOUT) E16_CASE(E16_RELU_DROPOUT) E16_CASE(E16_ADD_DROPOUT)
    }
#undef E16_CASE
    IX_CHECK_LAUNCH("ix_map_b16");
    return IX_OK;
}

// ---- out = ((a0 + a1) + a2) + ... over 2..8 bf16 tensors (hipops.SumN: the gradient of a tensor with several consumers) --------------
struct SumN16 { const bf16_t* src[8]; int n; };
__global__ __launch_bounds__(E16_BLOCK) void sum_n_b16_kernel(SumN16 a, bf16_t* __restrict__ o, int64_t count) {
    const int64_t n8 = count >> 3, stride = (int64_t)gridDim.x * E16_BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * E16_BLOCK + threadIdx.x; i < n8; i += stride) {
        F8 s = e16_ld8(a.src[0] + 8 * i);
#pragma unroll
        for (int t = 1; t < 8; ++t)
            if (t < a.n) {
                const F8 x = e16_ld8(a.src[t] + 8 * i);
#pragma unroll
                for (int e = 0; e < 8; ++e) s.v[e] += x.v[e];
            }
        e16_st8(o + 8 * i, s);
    }
    if (blockIdx.x == 0)
        for (int64_t k = (n8 << 3) + threadIdx.x; k < count; k += E16_BLOCK) {
            float s = e16_f(a.src[0][k]);
            for (int t = 1; t < a.n; ++t) s += e16_f(a.src[t][k]);
            o[k] = e16_b(s);
        }
}
extern "C" int ix_sum_n_b16(const void* const* srcs, int n, void* out, int64_t count, hipStream_t stream) {
    if (count <= 0) return IX_OK;
    IX_CHECK_ARG(srcs && out && n >= 2 && n <= 8 && ix_al16(out), "ix_sum_n_b16: 2..8 source tensors");
    SumN16 a;
    for (int t = 0; t < 8; ++t) {
        a.src[t] = (const bf16_t*)(t < n ? srcs[t] : srcs[0]);
        IX_CHECK_ARG(a.src[t] != nullptr && ix_al16(a.src[t]), "ix_sum_n_b16: null or unaligned source");
    }
    a.n = n;
    hipLaunchKernelGGL(sum_n_b16_kernel, dim3(ix_grid_1d(count / 8 + 1, E16_BLOCK)), dim3(E16_BLOCK), 0, stream, a, (bf16_t*)out, count);
    IX_CHECK_LAUNCH("ix_sum_n_b16");
    return IX_OK;
}

// ---- per-channel ops on [rows, C] (C % 8 == 0): fp32 vectors per channel, bf16 activations -----------------------------------------
enum {
    C16_AFFINE = 0,           // [relu](x scale[c] + shift[c] (+ y))     (y = residual or null)
    C16_RELU_BWD_SCALE = 1,   // x [y > 0] scale[c]                       (x = dy, y = the ReLU output)
    C16_SCALE = 2,            // x scale[c]
    C16_ADD_VEC = 3,          // x + scale[g][c]                          (scale = a row vector per group of rows_per_group rows)
    C16_OPS = 4
};
template <int OP>
__global__ __launch_bounds__(E16_BLOCK) void channel_b16_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ y,
                                                                const float* __restrict__ scale, const float* __restrict__ shift,
                                                                bf16_t* __restrict__ o, int64_t n8, int C8, int relu) {
    const int64_t stride = (int64_t)gridDim.x * E16_BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * E16_BLOCK + threadIdx.x; i < n8; i += stride) {
        const int64_t c = 8 * (i % C8);
        const F8 a = e16_ld8(x + 8 * i);
        F8 r;
        const f32x4 s0 = *reinterpret_cast<const f32x4*>(scale + c), s1 = *reinterpret_cast<const f32x4*>(scale + c + 4);
        const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
        if (OP == C16_AFFINE) {
            const f32x4 h0 = *reinterpret_cast<const f32x4*>(shift + c), h1 = *reinterpret_cast<const f32x4*>(shift + c + 4);
            const float sh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
            F8 res = a;
            if (y) res = e16_ld8(y + 8 * i);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float v = a.v[e] * sc[e] + sh[e];
                if (y) v += res.v[e];
                r.v[e] = (relu && v < 0.f) ? 0.f : v;
            }
        } else if (OP == C16_RELU_BWD_SCALE) {
            const F8 yy = e16_ld8(y + 8 * i);
#pragma unroll
            for (int e = 0; e < 8; ++e) r.v[e] = yy.v[e] > 0.f ? a.v[e] * sc[e] : 0.f;
        } else if (OP == C16_SCALE) {
#pragma unroll
            for (int e = 0; e < 8; ++e) r.v[e] = a.v[e] * sc[e];
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) r.v[e] = a.v[e] + sc[e];
        }
        e16_st8(o + 8 * i, r);
    }
}
// `groups` (C16_ADD_VEC only): the rows are `groups` equal contiguous slabs and slab g adds scale[g * C ..] (one learned query table per
// episode's fast weights): one launch per slab.
extern "C" int ix_channel_b16(int op, const void* x, const void* y, const float* scale, const float* shift, void* out, int64_t rows,
                              int C, int relu, int groups, hipStream_t stream) {
    if (rows <= 0 || C <= 0) return IX_OK;
    IX_CHECK_ARG(op >= 0 && op < C16_OPS, "ix_channel_b16: unknown op %d", op);
    IX_CHECK_ARG(x && out && scale && ix_al16(x) && ix_al16(out) && ix_al16(scale) && (!y || ix_al16(y)) && (!shift || ix_al16(shift)),
                 "ix_channel_b16: null or unaligned pointer");
    IX_CHECK_ARG(C % 8 == 0, "ix_channel_b16: C=%d must be a multiple of 8", C);
    IX_CHECK_ARG(op != C16_AFFINE || shift, "ix_channel_b16: the affine needs a shift vector");
    IX_CHECK_ARG(op != C16_RELU_BWD_SCALE || y, "ix_channel_b16: the ReLU derivative needs the ReLU output");
    if (groups < 1) groups = 1;
    IX_CHECK_ARG(op == C16_ADD_VEC || groups == 1, "ix_channel_b16: groups only with the row-vector add");
    IX_CHECK_ARG(rows % groups == 0, "ix_channel_b16: rows must divide into the groups");
    const int64_t rpg = rows / groups, n8 = rpg * (C / 8);
    const dim3 grid(ix_grid_1d(n8, E16_BLOCK)), block(E16_BLOCK);
    for (int g = 0; g < groups; ++g) {
        const bf16_t* xg = (const bf16_t*)x + (int64_t)g * rpg * C;
        const bf16_t* yg = y ? (const bf16_t*)y + (int64_t)g * rpg * C : nullptr;
        bf16_t* og = (bf16_t*)out + (int64_t)g * rpg * C;
        const float* sg = scale + (int64_t)g * C;
#define C16_CASE(OP) case OP: hipLaunchKernelGGL(channel_b16_kernel<OP>, grid, block, 0, stream, xg, yg, sg, shift, og, n8, C / 8, relu); break;
        switch (op) { C16_CASE(C16_AFFINE) C16_CASE(C16_RELU_BWD_SCALE) C16_CASE(C16_SCALE) C16_CASE(C16_ADD_VEC) }
#undef C16_CASE
    }
    IX_CHECK_LAUNCH("ix_channel_b16");
    return IX_OK;
}

// ---- LayerNorm over the last dim D (D % 4 == 0, D <= 1024), one wave per row; x, y, dy, dx bf16; gamma, beta, statistics, parameter
// gradients fp32.  Lane l owns columns 256 j + 4 l .. + 3 (8-byte accesses).
#define LN16_ROWS 4   // rows (waves) per workgroup
template <int NV>   // NV = ceil(D / 256)
__global__ __launch_bounds__(256) void ln_fwd_b16_kernel(const bf16_t* __restrict__ x, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, bf16_t* __restrict__ y,
                                                         float* __restrict__ mean, float* __restrict__ rstd, int64_t rows, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * LN16_ROWS + (threadIdx.x >> 6);
    if (row >= rows) return;
    const bf16_t* xr = x + row * D;
    float v[NV][4];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = 256 * j + 4 * lane;
        if (c < D) {
            const u32x2 r = *reinterpret_cast<const u32x2*>(xr + c);
            v[j][0] = e16_lo(r.x); v[j][1] = e16_hi(r.x); v[j][2] = e16_lo(r.y); v[j][3] = e16_hi(r.y);
        } else v[j][0] = v[j][1] = v[j][2] = v[j][3] = 0.f;
        s += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
    }
    const float mu = ix_wave_sum(s) / D;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j)
        if (256 * j + 4 * lane < D)
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = v[j][e] - mu; q += d * d; }
    const float r = rsqrtf(ix_wave_sum(q) / D + eps);
    if (lane == 0) { mean[row] = mu; rstd[row] = r; }
    bf16_t* yr = y + row * D;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = 256 * j + 4 * lane;
        if (c < D) {
            const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c), b = *reinterpret_cast<const f32x4*>(beta + c);
            *reinterpret_cast<u32x2*>(yr + c) = u32x2{e16_pack((v[j][0] - mu) * r * g.x + b.x, (v[j][1] - mu) * r * g.y + b.y),
                                                     e16_pack((v[j][2] - mu) * r * g.z + b.z, (v[j][3] - mu) * r * g.w + b.w)};
        }
    }
}
extern "C" int ix_layernorm_fwd_b16(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, int64_t rows,
                                    int D, float eps, hipStream_t stream) {
    if (rows <= 0) return IX_OK;
    IX_CHECK_ARG(x && gamma && beta && y && mean && rstd && ix_al16(x) && ix_al16(y) && ix_al16(gamma) && ix_al16(beta), "ix_layernorm_fwd_b16: bad args");
    IX_CHECK_ARG(D > 0 && D <= 1024 && D % 4 == 0, "ix_layernorm_fwd_b16: D=%d unsupported (multiples of 4 up to 1024)", D);
    const dim3 grid((unsigned)((rows + LN16_ROWS - 1) / LN16_ROWS)), block(256);
#define LN(N) hipLaunchKernelGGL(ln_fwd_b16_kernel<N>, grid, block, 0, stream, (const bf16_t*)x, gamma, beta, (bf16_t*)y, mean, rstd, rows, D, eps)
    if (D <= 256) LN(1);
    else if (D <= 512) LN(2);
    else LN(4);
#undef LN
    IX_CHECK_LAUNCH("ix_layernorm_fwd_b16");
    return IX_OK;
}

// Backward: g = dy gamma; dx = r (g - mean(g) - xhat mean(g xhat)); dgamma = sum_rows dy xhat; dbeta = sum_rows dy.
// A workgroup walks LN16_GROUP rows per wave and leaves ONE partial row of (dgamma, dbeta) in `part` ([blocks][2][D], fp32);
// ln_bwd_b16_reduce_kernel adds the partials in block order (deterministic; no tickets, no atomics).
#define LN16_GROUP 16
template <int NV>
__global__ __launch_bounds__(256) void ln_bwd_b16_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                         const float* __restrict__ gamma, const float* __restrict__ mean,
                                                         const float* __restrict__ rstd, bf16_t* __restrict__ dx, float* __restrict__ part,
                                                         int64_t rows, int D) {
    __shared__ float sg[LN16_ROWS][256 * NV];
    __shared__ float sb[LN16_ROWS][256 * NV];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float ag[NV][4], ab[NV][4], gm[NV][4];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = 256 * j + 4 * lane;
        f32x4 g = {0.f, 0.f, 0.f, 0.f};
        if (c < D) g = *reinterpret_cast<const f32x4*>(gamma + c);
        gm[j][0] = g.x; gm[j][1] = g.y; gm[j][2] = g.z; gm[j][3] = g.w;
#pragma unroll
        for (int e = 0; e < 4; ++e) ag[j][e] = ab[j][e] = 0.f;
    }
    for (int t = 0; t < LN16_GROUP; ++t) {
        const int64_t row = ((int64_t)blockIdx.x * LN16_GROUP + t) * LN16_ROWS + w;
        if (row >= rows) break;
        const float mu = mean[row], r = rstd[row];
        float xh[NV][4], gg[NV][4], dv[NV][4];
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int c = 256 * j + 4 * lane;
            if (c < D) {
                const u32x2 d = *reinterpret_cast<const u32x2*>(dy + row * D + c), xx = *reinterpret_cast<const u32x2*>(x + row * D + c);
                dv[j][0] = e16_lo(d.x); dv[j][1] = e16_hi(d.x); dv[j][2] = e16_lo(d.y); dv[j][3] = e16_hi(d.y);
                xh[j][0] = (e16_lo(xx.x) - mu) * r; xh[j][1] = (e16_hi(xx.x) - mu) * r;
                xh[j][2] = (e16_lo(xx.y) - mu) * r; xh[j][3] = (e16_hi(xx.y) - mu) * r;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) dv[j][e] = xh[j][e] = 0.f;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                gg[j][e] = dv[j][e] * gm[j][e];
                a += gg[j][e];
                b += gg[j][e] * xh[j][e];
                ag[j][e] += dv[j][e] * xh[j][e];
                ab[j][e] += dv[j][e];
            }
        }
        a = ix_wave_sum(a) / D;
        b = ix_wave_sum(b) / D;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int c = 256 * j + 4 * lane;
            if (c < D)
                *reinterpret_cast<u32x2*>(dx + row * D + c) = u32x2{e16_pack(r * (gg[j][0] - a - xh[j][0] * b), r * (gg[j][1] - a - xh[j][1] * b)),
                                                                   e16_pack(r * (gg[j][2] - a - xh[j][2] * b), r * (gg[j][3] - a - xh[j][3] * b))};
        }
    }
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) { sg[w][256 * j + 4 * lane + e] = ag[j][e]; sb[w][256 * j + 4 * lane + e] = ab[j][e]; }
    __syncthreads();
    float* P = part + (int64_t)blockIdx.x * 2 * D;
    for (int c = threadIdx.x; c < D; c += 256) {
        P[c] = (sg[0][c] + sg[1][c]) + (sg[2][c] + sg[3][c]);
        P[D + c] = (sb[0][c] + sb[1][c]) + (sb[2][c] + sb[3][c]);
    }
}
__global__ __launch_bounds__(256) void ln_bwd_b16_reduce_kernel(const float* __restrict__ part, float* __restrict__ dgamma,
                                                                float* __restrict__ dbeta, int nblk, int D) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= 2 * D) return;
    float s = 0.f;
    int b = 0;
    for (; b + 8 <= nblk; b += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = part[(int64_t)(b + u) * 2 * D + c];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; b < nblk; ++b) s += part[(int64_t)b * 2 * D + c];
    if (c < D) dgamma[c] = s;
    else dbeta[c - D] = s;
}
extern "C" int ix_workspace_bytes_layernorm_bwd_b16(int64_t rows, int D, size_t* out) {
    IX_CHECK_ARG(out && rows >= 0 && D > 0, "ix_workspace_bytes_layernorm_bwd_b16: bad args");
    const int64_t nblk = (rows + LN16_ROWS * LN16_GROUP - 1) / (LN16_ROWS * LN16_GROUP);
    *out = IX_TICKET_BYTES + (size_t)nblk * 2 * D * sizeof(float);
    return IX_OK;
}
extern "C" int ix_layernorm_bwd_b16(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd, void* dx,
                                    float* dgamma, float* dbeta, int64_t rows, int D, void* workspace, size_t workspace_bytes,
                                    hipStream_t stream) {
    if (rows <= 0) return IX_OK;
    IX_CHECK_ARG(dy && x && gamma && mean && rstd && dx && dgamma && dbeta && ix_al16(dy) && ix_al16(x) && ix_al16(dx) && ix_al16(gamma),
                 "ix_layernorm_bwd_b16: bad args");
    IX_CHECK_ARG(D > 0 && D <= 1024 && D % 4 == 0, "ix_layernorm_bwd_b16: D=%d unsupported (multiples of 4 up to 1024)", D);
    const int64_t nblk = (rows + LN16_ROWS * LN16_GROUP - 1) / (LN16_ROWS * LN16_GROUP);
    const size_t need = IX_TICKET_BYTES + (size_t)nblk * 2 * D * sizeof(float);
    if (!workspace || workspace_bytes < need) {
        ix_set_error("ix_layernorm_bwd_b16: workspace of %zu bytes needed (ix_workspace_bytes_layernorm_bwd_b16)", need);
        return IX_ERR_WORKSPACE;
    }
    float* part = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(workspace) + IX_TICKET_BYTES);
    const dim3 grid((unsigned)nblk), block(256);
#define LN(N) hipLaunchKernelGGL(ln_bwd_b16_kernel<N>, grid, block, 0, stream, (const bf16_t*)dy, (const bf16_t*)x, gamma, mean, rstd, (bf16_t*)dx, part, rows, D)
    if (D <= 256) LN(1);
    else if (D <= 512) LN(2);
    else LN(4);
#undef LN
    hipLaunchKernelGGL(ln_bwd_b16_reduce_kernel, dim3((2 * D + 255) / 256), dim3(256), 0, stream, part, dgamma, dbeta, (int)nblk, D);
    IX_CHECK_LAUNCH("ix_layernorm_bwd_b16");
    return IX_OK;
}

// ---- column sums of a bf16 [G][rows][C] tensor -> fp32 [G][C] (bias gradients): each workgroup sums CS16_ROWS rows of a 512-column
// slab into one fp32 partial row, colsum_b16_reduce_kernel adds the partial rows in order (deterministic; C % 8 == 0) -----------------
#define CS16_ROWS 128
__global__ __launch_bounds__(256) void colsum_b16_kernel(const bf16_t* __restrict__ x, float* __restrict__ part, int64_t rows, int C, int nblk) {
    // thread t: column group (t % 64) of this slab (8 columns), row phase t / 64 (4 phases)
    __shared__ float red[4][512];
    const int cg = threadIdx.x & 63, ph = threadIdx.x >> 6;
    const int c0 = blockIdx.y * 512 + cg * 8;
    const int g = blockIdx.z;
    const bf16_t* xg = x + (int64_t)g * rows * C;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (c0 < C) {
        const int64_t r0 = (int64_t)blockIdx.x * CS16_ROWS, r1 = min(rows, r0 + CS16_ROWS);
        for (int64_t r = r0 + ph; r < r1; r += 4) {
            const F8 v = e16_ld8(xg + r * C + c0);
#pragma unroll
            for (int e = 0; e < 8; ++e) s[e] += v.v[e];
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[ph][cg * 8 + e] = s[e];
    __syncthreads();
    for (int c = threadIdx.x; c < 512; c += 256) {
        const int col = blockIdx.y * 512 + c;
        if (col < C) part[((int64_t)g * nblk + blockIdx.x) * C + col] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
    }
}
// 64 columns per workgroup, four row phases: thread (c, ph) adds partial rows ph, ph + 4, ... in order, the four phase sums are added
// in phase order (a fixed order: the same bits every time)
__global__ __launch_bounds__(256) void colsum_b16_reduce_kernel(const float* __restrict__ part, float* __restrict__ out, int nblk, int C, int G) {
    __shared__ float red[4][64];
    const int cl = threadIdx.x & 63, ph = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 64 + cl;     // flat (group, column)
    float s = 0.f;
    if (i < (int64_t)G * C) {
        const int g = (int)(i / C), c = (int)(i - (int64_t)g * C);
        const float* p = part + (int64_t)g * nblk * C + c;
        for (int b = ph; b < nblk; b += 4) s += p[(int64_t)b * C];
    }
    red[ph][cl] = s;
    __syncthreads();
    if (ph == 0 && i < (int64_t)G * C) out[i] = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
}
extern "C" int ix_workspace_bytes_colsum_b16(int64_t rows, int C, int groups, size_t* out) {
    IX_CHECK_ARG(out && rows >= 0 && C > 0 && groups > 0, "ix_workspace_bytes_colsum_b16: bad args");
    const int64_t nblk = (rows + CS16_ROWS - 1) / CS16_ROWS;
    *out = IX_TICKET_BYTES + (size_t)groups * nblk * C * sizeof(float);
    return IX_OK;
}
extern "C" int ix_colsum_b16(const void* x, float* out, int64_t rows, int C, int groups, void* workspace, size_t workspace_bytes,
                             hipStream_t stream) {
    if (rows <= 0 || C <= 0 || groups <= 0) return IX_OK;
    IX_CHECK_ARG(x && out && ix_al16(x) && C % 8 == 0 && groups <= 65535, "ix_colsum_b16: null / unaligned pointer or C %% 8 != 0");
    const int64_t nblk = (rows + CS16_ROWS - 1) / CS16_ROWS;
    const size_t need = IX_TICKET_BYTES + (size_t)groups * nblk * C * sizeof(float);
    if (!workspace || workspace_bytes < need) {
        ix_set_error("ix_colsum_b16: workspace of %zu bytes needed (ix_workspace_bytes_colsum_b16)", need);
        return IX_ERR_WORKSPACE;
    }
    float* part = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(workspace) + IX_TICKET_BYTES);
    hipLaunchKernelGGL(colsum_b16_kernel, dim3((unsigned)nblk, (C + 511) / 512, groups), dim3(256), 0, stream, (const bf16_t*)x, part, rows, C, (int)nblk);
    hipLaunchKernelGGL(colsum_b16_reduce_kernel, dim3((unsigned)(((int64_t)groups * C + 63) / 64)), dim3(256), 0, stream, part, out, (int)nblk, C, groups);
    IX_CHECK_LAUNCH("ix_colsum_b16");
    return IX_OK;
}
