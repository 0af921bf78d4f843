// Wavefront-reduction kernels: row softmax and LayerNorm with their first- and second-order backward forms.
// One 64-lane wave owns one row; lane l touches elements l, l+64, ... (coalesced), the row lives in registers
// between the reduction and the write so every operand is read from HBM exactly once.
//
// reference sites: softmax inside nn.MultiheadAttention (models/detr_models/transformer.py:153,216,219) and the
// explicit softmax of models/gpt.py:48-52; nn.LayerNorm in transformer.py:139-140,199-201 and gpt.py:64-65,98.
// The bwd_bwd kernels are the analytic double-backward that autograd's SoftmaxBackwardDataBackward0 /
// NativeLayerNormBackwardBackward0 nodes compute for the MAML meta-gradient (models/interactron.py:99-123).
#include "common.h"

#define ROWS_PER_BLOCK 4  // 256 threads = 4 waves = 4 rows

// ------------------------------------------------------------------------------------------------------------
// softmax
// ------------------------------------------------------------------------------------------------------------
template <int NREG>
__global__ __launch_bounds__(256) void softmax_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                          int64_t rows, int len, int64_t ld,
                                                          const uint8_t* __restrict__ mask, int rows_per_mask,
                                                          int64_t mask_ld) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * ld;
    float* yr = y + row * ld;
    const uint8_t* mr = mask ? mask + (row / rows_per_mask) * mask_ld : nullptr;
    float v[NREG];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < NREG; ++i) {
        const int c = lane + 64 * i;
        float t = -INFINITY;
        if (c < len) {
            t = xr[c];
            if (mr && mr[c]) t = -INFINITY;
        }
        v[i] = t;
        mx = fmaxf(mx, t);
    }
    mx = ix_wave_max(mx);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NREG; ++i) {
        v[i] = __expf(v[i] - mx);  // exp(-inf) = 0 for padded / masked slots
        s += v[i];
    }
    s = ix_wave_sum(s);
    const float inv = 1.f / s;
#pragma unroll
    for (int i = 0; i < NREG; ++i) {
        const int c = lane + 64 * i;
        if (c < len) yr[c] = v[i] * inv;
    }
}

// streaming fallback for rows longer than the register-resident variants (three passes over L2-resident data)
__global__ __launch_bounds__(256) void softmax_fwd_stream_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                 int64_t rows, int len, int64_t ld,
                                                                 const uint8_t* __restrict__ mask, int rows_per_mask,
                                                                 int64_t mask_ld) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * ld;
    float* yr = y + row * ld;
    const uint8_t* mr = mask ? mask + (row / rows_per_mask) * mask_ld : nullptr;
    float mx = -INFINITY;
    for (int c = lane; c < len; c += 64) {
        const float t = (mr && mr[c]) ? -INFINITY : xr[c];
        mx = fmaxf(mx, t);
    }
    mx = ix_wave_max(mx);
    float s = 0.f;
    for (int c = lane; c < len; c += 64) {
        const float t = (mr && mr[c]) ? -INFINITY : xr[c];
        s += __expf(t - mx);
    }
    s = ix_wave_sum(s);
    const float inv = 1.f / s;
    for (int c = lane; c < len; c += 64) {
        const float t = (mr && mr[c]) ? -INFINITY : xr[c];
        yr[c] = __expf(t - mx) * inv;
    }
}

// y = softmax(x + (-inf where mask)) along the last dim. x, y: [rows, len] with row pitch ld.
// mask (optional): uint8 [n_mask_rows, mask_ld], row r uses mask row r / rows_per_mask (key-padding mask per frame).
extern "C" int ix_softmax_fwd_f32(const float* x, float* y, int64_t rows, int len, int64_t ld, const uint8_t* mask,
                                  int rows_per_mask, int64_t mask_ld, hipStream_t stream) {
    if (rows <= 0 || len <= 0) return IX_OK;
    IX_CHECK_ARG(x && y && ld >= len, "ix_softmax_fwd_f32: bad args");
    IX_CHECK_ARG(!mask || rows_per_mask > 0, "ix_softmax_fwd_f32: rows_per_mask must be > 0 with a mask");
    dim3 grid((unsigned)((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK)), block(256);
#define SM(N) hipLaunchKernelGGL(softmax_fwd_kernel<N>, grid, block, 0, stream, x, y, rows, len, ld, mask, rows_per_mask, mask_ld)
    if (len <= 64) SM(1);
    else if (len <= 256) SM(4);
    else if (len <= 512) SM(8);
    else if (len <= 1024) SM(16);
    else if (len <= 2304) SM(36);
    else hipLaunchKernelGGL(softmax_fwd_stream_kernel, grid, block, 0, stream, x, y, rows, len, ld, mask, rows_per_mask, mask_ld);
#undef SM
    IX_CHECK_LAUNCH("ix_softmax_fwd_f32");
    return IX_OK;
}

// dx = y * (dy - sum(y*dy))
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy,
                                                          float* __restrict__ dx, int64_t rows, int len, int64_t ld) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* yr = y + row * ld;
    const float* gr = dy + row * ld;
    float* o = dx + row * ld;
    float s = 0.f;
    for (int c = lane; c < len; c += 64) s += yr[c] * gr[c];
    s = ix_wave_sum(s);
    for (int c = lane; c < len; c += 64) o[c] = yr[c] * (gr[c] - s);
}

extern "C" int ix_softmax_bwd_f32(const float* y, const float* dy, float* dx, int64_t rows, int len, int64_t ld,
                                  hipStream_t stream) {
    if (rows <= 0 || len <= 0) return IX_OK;
    IX_CHECK_ARG(y && dy && dx && ld >= len, "ix_softmax_bwd_f32: bad args");
    dim3 grid((unsigned)((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK));
    hipLaunchKernelGGL(softmax_bwd_kernel, grid, dim3(256), 0, stream, y, dy, dx, rows, len, ld);
    IX_CHECK_LAUNCH("ix_softmax_bwd_f32");
    return IX_OK;
}

// Node: dx = y*(dy - s), s = sum(y*dy).  Given G = dL/d(dx):
//   grad_dy = y*(G - t),            t = sum(G*y)
//   grad_y  = G*(dy - s) - dy*t
__global__ __launch_bounds__(256) void softmax_bwd_bwd_kernel(const float* __restrict__ G, const float* __restrict__ y,
                                                              const float* __restrict__ dy, float* __restrict__ grad_y,
                                                              float* __restrict__ grad_dy, int64_t rows, int len,
                                                              int64_t ld) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int64_t off = row * ld;
    float s = 0.f, t = 0.f;
    for (int c = lane; c < len; c += 64) {
        const float yy = y[off + c];
        s += yy * dy[off + c];
        t += yy * G[off + c];
    }
    s = ix_wave_sum(s);
    t = ix_wave_sum(t);
    for (int c = lane; c < len; c += 64) {
        const float yy = y[off + c], g = G[off + c], d = dy[off + c];
        grad_dy[off + c] = yy * (g - t);
        grad_y[off + c] = g * (d - s) - d * t;
    }
}

extern "C" int ix_softmax_bwd_bwd_f32(const float* G, const float* y, const float* dy, float* grad_y, float* grad_dy,
                                      int64_t rows, int len, int64_t ld, hipStream_t stream) {
    if (rows <= 0 || len <= 0) return IX_OK;
    IX_CHECK_ARG(G && y && dy && grad_y && grad_dy && ld >= len, "ix_softmax_bwd_bwd_f32: bad args");
    dim3 grid((unsigned)((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK));
    hipLaunchKernelGGL(softmax_bwd_bwd_kernel, grid, dim3(256), 0, stream, G, y, dy, grad_y, grad_dy, rows, len, ld);
    IX_CHECK_LAUNCH("ix_softmax_bwd_bwd_f32");
    return IX_OK;
}

// ------------------------------------------------------------------------------------------------------------
// Attention probabilities: softmax + dropout as ONE node and its first / second backward.  The [L, S] score tensors are
// the largest tensors of the step (2 GB per fusion layer at 16 episodes); every separate elementwise node on them is
// a full HBM round trip, so the dropout mask (a pure function of (seed, element index), same hash as dropout_kernel)
// is applied inside the softmax kernels instead of in passes of its own.
//   forward      y = softmax(x [+ key mask]),  d = m * y / keep
//   backward     gs = y * (gy - sum(y * gy)),  gy = m * gd / keep
//   double bwd   given G = dL/d(gs) and HD = dL/d(d) (both [L, S]):
//                  s = sum(y gy), t = sum(y G)
//                  HgD = m / keep * y (G - t)                              = dL/d(gd)
//                  HY  = G (gy - s) - gy t + m / keep * HD                 = dL/d(y), all paths
//                  HS  = y * (HY - sum(y HY))                              = dL/d(x) through the softmax
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t attn_mix32(uint64_t z) {   // == mix32 of elementwise.hip
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return (uint32_t)((z ^ (z >> 31)) >> 32);
}
__device__ __forceinline__ float attn_keep(uint64_t seed, int64_t k, uint32_t thresh, float scale) {
    if (thresh == 0) return 1.f;
    return attn_mix32(seed ^ ((uint64_t)k * 0xD6E8FEB86659FD93ull)) >= thresh ? scale : 0.f;
}

__global__ void attn_prob_fwd_stream_kernel(const float*, float*, float*, int64_t, int, int64_t, const uint8_t*, int, int64_t, uint32_t, float, uint64_t, const uint64_t*);
__global__ void attn_prob_bwd_stream_kernel(const float*, const float*, float*, int64_t, int, int64_t, uint32_t, float, uint64_t, const uint64_t*);
__global__ void attn_prob_bwd_bwd_stream_kernel(const float*, const float*, const float*, const float*, const float*, float*, float*, int64_t, int, int64_t, uint32_t, float, uint64_t, const uint64_t*);
template <int NREG>
__global__ __launch_bounds__(256) void attn_prob_fwd_kernel(const float* x, float* y, float* __restrict__ d, int64_t rows,
                                                            int len, int64_t ld, const uint8_t* __restrict__ mask,
                                                            int rows_per_mask, int64_t mask_ld, uint32_t thresh, float scale,
                                                            uint64_t seed, const uint64_t* __restrict__ salt) {
    if (salt) seed ^= *salt;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * ld;   // (y may alias x: a row is read completely before it is written)
    float* yr = y + row * ld;
    const uint8_t* mr = mask ? mask + (row / rows_per_mask) * mask_ld : nullptr;
    float v[NREG];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < NREG; ++i) {
        const int c = lane + 64 * i;
        float t = -INFINITY;
        if (c < len) {
            t = xr[c];
            if (mr && mr[c]) t = -INFINITY;
        }
        v[i] = t;
        mx = fmaxf(mx, t);
    }
    mx = ix_wave_max(mx);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NREG; ++i) {
        v[i] = __expf(v[i] - mx);
        s += v[i];
    }
    s = ix_wave_sum(s);
    const float inv = 1.f / s;
#pragma unroll
    for (int i = 0; i < NREG; ++i) {
        const int c = lane + 64 * i;
        if (c < ld) {   // pad columns [len, ld) are written as zeros: the tensors need no separate fill
            const float val = c < len ? v[i] * inv : 0.f;
            yr[c] = val;
            if (d) d[row * ld + c] = val * attn_keep(seed, row * ld + c, thresh, scale);
        }
    }
}

extern "C" int ix_attn_prob_fwd_f32(const float* x, float* y, float* d, int64_t rows, int len, int64_t ld, const uint8_t* mask,
                                    int rows_per_mask, int64_t mask_ld, float p, uint64_t seed, hipStream_t stream) {
    if (rows <= 0 || len <= 0) return IX_OK;
    IX_CHECK_ARG(x && y && ld >= len, "ix_attn_prob_fwd_f32: bad args");
    IX_CHECK_ARG(!mask || rows_per_mask > 0, "ix_attn_prob_fwd_f32: rows_per_mask must be > 0 with a mask");
    IX_CHECK_ARG(p >= 0.f && p < 1.f, "ix_attn_prob_fwd_f32: p=%f outside [0,1)", p);
    const uint32_t thresh = (uint32_t)((double)p * 4294967296.0);
    const float scale = 1.f / (1.f - p);
    dim3 grid((unsigned)((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK)), block(256);
#define SM(N) hipLaunchKernelGGL(attn_prob_fwd_kernel<N>, grid, block, 0, stream, x, y, d, rows, len, ld, mask, rows_per_mask, mask_ld, thresh, scale, seed, ix_g_salt)
    if (ld <= 64) SM(1);
    else if (ld <= 256) SM(4);
    else if (ld <= 512) SM(8);
    else if (ld <= 1024) SM(16);
    else if (ld <= 2304) SM(36);
    else hipLaunchKernelGGL(attn_prob_fwd_stream_kernel, grid, block, 0, stream, x, y, d, rows, len, ld, mask, rows_per_mask, mask_ld, thresh, scale, seed, ix_g_salt);
#undef SM
    IX_CHECK_LAUNCH("ix_attn_prob_fwd_f32");
    return IX_OK;
}

// Rows up to 64 * NREG entries stay in registers between the reduction and the write: every operand is read once.
template <int NREG>
__global__ __launch_bounds__(256) void attn_prob_bwd_kernel(const float* __restrict__ y, const float* __restrict__ gd,
                                                            float* __restrict__ gs, int64_t rows, int len, int64_t ld,
                                                            uint32_t thresh, float scale, uint64_t seed, const uint64_t* __restrict__ salt) {
    if (salt) seed ^= *salt;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int64_t off = row * ld;
    float yv[NREG], gy[NREG];
    // all loads first (branch-free, clamped column), then the mask hash and the arithmetic: the row's 2 * NREG loads
    // are in flight together
#pragma unroll
    for (int i = 0; i < NREG; ++i) {
        const int c = min(lane + 64 * i, len - 1);
        yv[i] = y[off + c];
        gy[i] = gd[off + c];
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NREG; ++i) {
        const int c = lane + 64 * i;
        const bool in = c < len;
        yv[i] = in ? yv[i] : 0.f;
        gy[i] = in ? gy[i] * attn_keep(seed, off + c, thresh, scale) : 0.f;
        s += yv[i] * gy[i];
    }
    s = ix_wave_sum(s);
#pragma unroll
    for (int i = 0; i < NREG; ++i) {
        const int c = lane + 64 * i;
        if (c < ld) gs[off + c] = c < len ? yv[i] * (gy[i] - s) : 0.f;
    }
}

extern "C" int ix_attn_prob_bwd_f32(const float* y, const float* gd, float* gs, int64_t rows, int len, int64_t ld, float p,
                                    uint64_t seed, hipStream_t stream) {
    if (rows <= 0 || len <= 0) return IX_OK;
    IX_CHECK_ARG(y && gd && gs && ld >= len, "ix_attn_prob_bwd_f32: bad args");
    IX_CHECK_ARG(p >= 0.f && p < 1.f, "ix_attn_prob_bwd_f32: p=%f outside [0,1)", p);
    const uint32_t thresh = (uint32_t)((double)p * 4294967296.0);
    const float scale = 1.f / (1.f - p);
    dim3 grid((unsigned)((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK)), block(256);
#define SM(N) hipLaunchKernelGGL(attn_prob_bwd_kernel<N>, grid, block, 0, stream, y, gd, gs, rows, len, ld, thresh, scale, seed, ix_g_salt)
    if (ld <= 64) SM(1);
    else if (ld <= 256) SM(4);
    else if (ld <= 512) SM(8);
    else if (ld <= 1024) SM(16);
    else if (ld <= 2304) SM(36);
    else hipLaunchKernelGGL(attn_prob_bwd_stream_kernel, grid, block, 0, stream, y, gd, gs, rows, len, ld, thresh, scale, seed, ix_g_salt);
#undef SM
    IX_CHECK_LAUNCH("ix_attn_prob_bwd_f32");
    return IX_OK;
}

// G = G1 + G2 (either may be null), HD may be null (no cotangent reached d); writes HgD and HS (see the block comment)
template <int NREG>
__global__ __launch_bounds__(256) void attn_prob_bwd_bwd_kernel(const float* __restrict__ G1, const float* __restrict__ G2,
                                                                const float* __restrict__ y, const float* __restrict__ gd,
                                                                const float* __restrict__ HD, float* __restrict__ HgD,
                                                                float* __restrict__ HS, int64_t rows, int len, int64_t ld,
                                                                uint32_t thresh, float scale, uint64_t seed, const uint64_t* __restrict__ salt) {
    if (salt) seed ^= *salt;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int64_t off = row * ld;
    // one pass over the operands (kept in registers): with hy = g (gy - s) - gy t + hd m,
    //   u = sum(y hy) = sum(y g gy) - 2 s t + sum(y hd m)
    float yv[NREG], gv[NREG], gy[NREG], hd[NREG];
    uint64_t kept = 0;
    float s = 0.f, t = 0.f, a = 0.f;
    // loads first (branch-free, clamped column; null operands are wave-uniform), then mask hash + arithmetic
#pragma unroll
    for (int i = 0; i < NREG; ++i) {
        const int c = min(lane + 64 * i, len - 1);
        yv[i] = y[off + c];
        gy[i] = gd[off + c];
        gv[i] = G1 ? G1[off + c] : 0.f;
        hd[i] = HD ? HD[off + c] : 0.f;
    }
    if (G2) {
#pragma unroll
        for (int i = 0; i < NREG; ++i) gv[i] += G2[off + min(lane + 64 * i, len - 1)];
    }
#pragma unroll
    for (int i = 0; i < NREG; ++i) {
        const int c = lane + 64 * i;
        const bool in = c < len;
        const float m = in ? attn_keep(seed, off + c, thresh, scale) : 0.f;
        if (m != 0.f) kept |= 1ull << i;
        yv[i] = in ? yv[i] : 0.f;
        gv[i] = in ? gv[i] : 0.f;
        gy[i] *= m;
        hd[i] *= m;
        s += yv[i] * gy[i];
        t += yv[i] * gv[i];
        a += yv[i] * gv[i] * gy[i] + yv[i] * hd[i];
    }
    s = ix_wave_sum(s);
    t = ix_wave_sum(t);
    const float u = ix_wave_sum(a) - 2.f * s * t;
#pragma unroll
    for (int i = 0; i < NREG; ++i) {
        const int c = lane + 64 * i;
        if (c < ld) {
            const float m = ((kept >> i) & 1) ? scale : 0.f;
            const float hy = gv[i] * (gy[i] - s) - gy[i] * t + hd[i];
            HgD[off + c] = c < len ? (yv[i] * (gv[i] - t)) * m : 0.f;
            HS[off + c] = c < len ? yv[i] * (hy - u) : 0.f;
        }
    }
}

extern "C" int ix_attn_prob_bwd_bwd_f32(const float* G1, const float* G2, const float* y, const float* gd, const float* HD,
                                        float* HgD, float* HS, int64_t rows, int len, int64_t ld, float p, uint64_t seed,
                                        hipStream_t stream) {
    if (rows <= 0 || len <= 0) return IX_OK;
    IX_CHECK_ARG(y && gd && HgD && HS && ld >= len, "ix_attn_prob_bwd_bwd_f32: bad args");
    IX_CHECK_ARG(p >= 0.f && p < 1.f, "ix_attn_prob_bwd_bwd_f32: p=%f outside [0,1)", p);
    const uint32_t thresh = (uint32_t)((double)p * 4294967296.0);
    const float scale = 1.f / (1.f - p);
    dim3 grid((unsigned)((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK)), block(256);
#define SM(N) hipLaunchKernelGGL(attn_prob_bwd_bwd_kernel<N>, grid, block, 0, stream, G1, G2, y, gd, HD, HgD, HS, rows, len, ld, thresh, scale, seed, ix_g_salt)
    if (ld <= 64) SM(1);
    else if (ld <= 256) SM(4);
    else if (ld <= 512) SM(8);
    else if (ld <= 1024) SM(16);
    else if (ld <= 2304) SM(36);
    else hipLaunchKernelGGL(attn_prob_bwd_bwd_stream_kernel, grid, block, 0, stream, G1, G2, y, gd, HD, HgD, HS, rows, len, ld, thresh, scale, seed, ix_g_salt);
#undef SM
    IX_CHECK_LAUNCH("ix_attn_prob_bwd_bwd_f32");
    return IX_OK;
}

// Streaming forms of the three attention-probability kernels for rows longer than the register-resident variants
// (800x800 frames: fusion T = 12 755, detector S = 2 500): operands are read twice (three times in the forward), the
// second pass out of L2.
__global__ __launch_bounds__(256) void attn_prob_fwd_stream_kernel(const float* x, float* y, float* __restrict__ d,
                                                                   int64_t rows, int len, int64_t ld,
                                                                   const uint8_t* __restrict__ mask, int rows_per_mask,
                                                                   int64_t mask_ld, uint32_t thresh, float scale, uint64_t seed, const uint64_t* __restrict__ salt) {
    if (salt) seed ^= *salt;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int64_t off = row * ld;
    const uint8_t* mr = mask ? mask + (row / rows_per_mask) * mask_ld : nullptr;
    float mx = -INFINITY;
    for (int c = lane; c < len; c += 64) mx = fmaxf(mx, (mr && mr[c]) ? -INFINITY : x[off + c]);
    mx = ix_wave_max(mx);
    float s = 0.f;
    for (int c = lane; c < len; c += 64) s += __expf(((mr && mr[c]) ? -INFINITY : x[off + c]) - mx);
    s = ix_wave_sum(s);
    const float inv = 1.f / s;
    for (int c = lane; c < ld; c += 64) {   // (y may alias x: element c is read and written by the same lane)
        const float val = c < len ? __expf(((mr && mr[c]) ? -INFINITY : x[off + c]) - mx) * inv : 0.f;
        y[off + c] = val;
        if (d) d[off + c] = val * attn_keep(seed, off + c, thresh, scale);
    }
}

__global__ __launch_bounds__(256) void attn_prob_bwd_stream_kernel(const float* __restrict__ y, const float* __restrict__ gd,
                                                                   float* __restrict__ gs, int64_t rows, int len, int64_t ld,
                                                                   uint32_t thresh, float scale, uint64_t seed, const uint64_t* __restrict__ salt) {
    if (salt) seed ^= *salt;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int64_t off = row * ld;
    float s = 0.f;
    for (int c = lane; c < len; c += 64) s += y[off + c] * (gd[off + c] * attn_keep(seed, off + c, thresh, scale));
    s = ix_wave_sum(s);
    for (int c = lane; c < ld; c += 64)
        gs[off + c] = c < len ? y[off + c] * (gd[off + c] * attn_keep(seed, off + c, thresh, scale) - s) : 0.f;
}

__global__ __launch_bounds__(256) void attn_prob_bwd_bwd_stream_kernel(const float* __restrict__ G1, const float* __restrict__ G2,
                                                                       const float* __restrict__ y, const float* __restrict__ gd,
                                                                       const float* __restrict__ HD, float* __restrict__ HgD,
                                                                       float* __restrict__ HS, int64_t rows, int len, int64_t ld,
                                                                       uint32_t thresh, float scale, uint64_t seed, const uint64_t* __restrict__ salt) {
    if (salt) seed ^= *salt;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int64_t off = row * ld;
    float s = 0.f, t = 0.f, a = 0.f;
    for (int c = lane; c < len; c += 64) {
        const float yy = y[off + c], m = attn_keep(seed, off + c, thresh, scale);
        const float g = (G1 ? G1[off + c] : 0.f) + (G2 ? G2[off + c] : 0.f);
        const float gy = gd[off + c] * m;
        s += yy * gy;
        t += yy * g;
        a += yy * g * gy + (HD ? yy * (HD[off + c] * m) : 0.f);
    }
    s = ix_wave_sum(s);
    t = ix_wave_sum(t);
    const float u = ix_wave_sum(a) - 2.f * s * t;
    for (int c = lane; c < ld; c += 64) {
        float hgd = 0.f, hs = 0.f;
        if (c < len) {
            const float yy = y[off + c], m = attn_keep(seed, off + c, thresh, scale);
            const float g = (G1 ? G1[off + c] : 0.f) + (G2 ? G2[off + c] : 0.f);
            const float gy = gd[off + c] * m;
            const float hy = g * (gy - s) - gy * t + (HD ? HD[off + c] * m : 0.f);
            hgd = (yy * (g - t)) * m;
            hs = yy * (hy - u);
        }
        HgD[off + c] = hgd;
        HS[off + c] = hs;
    }
}

// ------------------------------------------------------------------------------------------------------------
// LayerNorm over the last dim D (D <= 64*NREG), eps inside the sqrt, biased variance (torch semantics)
// ------------------------------------------------------------------------------------------------------------
template <int NREG>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float* __restrict__ y,
                                                     float* __restrict__ mean, float* __restrict__ rstd, int64_t rows,
                                                     int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    if (row >= rows) return;
    {   // blockIdx.y = group (episode): activations advance by rows*D, the affine by D
        const int64_t go = (int64_t)blockIdx.y * rows;
        x += go * D; y += go * D; mean += go; rstd += go;
        gamma += (int64_t)blockIdx.y * D; beta += (int64_t)blockIdx.y * D;
    }
    const float* xr = x + row * D;
    float v[NREG];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NREG; ++i) {
        const int c = lane + 64 * i;
        v[i] = c < D ? xr[c] : 0.f;
        s += v[i];
    }
    const float mu = ix_wave_sum(s) / D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NREG; ++i) {
        const int c = lane + 64 * i;
        const float d = c < D ? v[i] - mu : 0.f;
        q += d * d;
    }
    const float r = rsqrtf(ix_wave_sum(q) / D + eps);
    if (lane == 0) {
        mean[row] = mu;
        rstd[row] = r;
    }
    float* yr = y + row * D;
#pragma unroll
    for (int i = 0; i < NREG; ++i) {
        const int c = lane + 64 * i;
        if (c < D) yr[c] = (v[i] - mu) * r * gamma[c] + beta[c];
    }
}

extern "C" int ix_layernorm_fwd_f32(const float* x, const float* gamma, const float* beta, float* y, float* mean,
                                    float* rstd, int64_t rows, int D, float eps, int groups, hipStream_t stream) {
    if (rows <= 0 || groups <= 0) return IX_OK;
    IX_CHECK_ARG(x && gamma && beta && y && mean && rstd && groups <= 65535, "ix_layernorm_fwd_f32: bad args");
    IX_CHECK_ARG(D > 0 && D <= 1024, "ix_layernorm_fwd_f32: D=%d unsupported (1..1024)", D);
    dim3 grid((unsigned)((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK), groups), block(256);
#define LN(N) hipLaunchKernelGGL(ln_fwd_kernel<N>, grid, block, 0, stream, x, gamma, beta, y, mean, rstd, rows, D, eps)
    if (D <= 256) LN(4);
    else if (D <= 512) LN(8);
    else LN(16);
#undef LN
    IX_CHECK_LAUNCH("ix_layernorm_fwd_f32");
    return IX_OK;
}

// Backward.  g = dy*gamma; dx = r*(g - mean(g) - xhat*mean(g*xhat)); dgamma += dy*xhat; dbeta += dy
// Column reductions: per-block LDS partials over its 4 rows x many row-groups, then one atomic per column.
// VEC: a lane owns four consecutive columns per 256-column slab (16-byte loads and stores; D % 4 == 0, 16-byte aligned rows) --
// register i <-> column 256 (i / 4) + 4 lane + (i % 4); else register i <-> column lane + 64 i
template <int NREG, bool VEC>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, float* __restrict__ dx,
                                                     float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                     int64_t rows, int D, int row_groups, float* __restrict__ part,
                                                     unsigned int* tickets) {
    __shared__ float sg[ROWS_PER_BLOCK][64 * NREG];
    __shared__ float sb[ROWS_PER_BLOCK][64 * NREG];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    {   // blockIdx.y = group
        const int64_t go = (int64_t)blockIdx.y * rows;
        dy += go * D; x += go * D; dx += go * D; mean += go; rstd += go;
        gamma += (int64_t)blockIdx.y * D; dgamma += (int64_t)blockIdx.y * D; dbeta += (int64_t)blockIdx.y * D;
    }
    auto col = [&](int i) { return VEC ? 256 * (i >> 2) + 4 * lane + (i & 3) : lane + 64 * i; };
    float ag[NREG], ab[NREG], gm[NREG];
#pragma unroll
    for (int i = 0; i < NREG; ++i) {
        ag[i] = ab[i] = 0.f;
        gm[i] = col(i) < D ? gamma[col(i)] : 0.f;   // (the same columns in every row of this wave)
    }
    for (int g = 0; g < row_groups; ++g) {
        const int64_t row = ((int64_t)blockIdx.x * row_groups + g) * ROWS_PER_BLOCK + w;
        if (row >= rows) break;
        const float mu = mean[row], r = rstd[row];
        const float* xr = x + row * D;
        const float* gr = dy + row * D;
        float xh[NREG], gg[NREG], dv[NREG], xv[NREG];
        if (VEC) {
#pragma unroll
            for (int v = 0; v < NREG / 4; ++v) {
                const int c0 = 256 * v + 4 * lane;
                float4 d4 = make_float4(0.f, 0.f, 0.f, 0.f), x4 = d4;
                if (c0 < D) { d4 = *reinterpret_cast<const float4*>(gr + c0); x4 = *reinterpret_cast<const float4*>(xr + c0); }
                dv[4 * v] = d4.x; dv[4 * v + 1] = d4.y; dv[4 * v + 2] = d4.z; dv[4 * v + 3] = d4.w;
                xv[4 * v] = x4.x; xv[4 * v + 1] = x4.y; xv[4 * v + 2] = x4.z; xv[4 * v + 3] = x4.w;
            }
        } else {
#pragma unroll
            for (int i = 0; i < NREG; ++i) {
                const int c = lane + 64 * i;
                dv[i] = c < D ? gr[c] : 0.f;
                xv[i] = c < D ? xr[c] : 0.f;
            }
        }
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int i = 0; i < NREG; ++i) {
            if (col(i) < D) {
                xh[i] = (xv[i] - mu) * r;
                gg[i] = dv[i] * gm[i];
                ag[i] += dv[i] * xh[i];
                ab[i] += dv[i];
            } else {
                xh[i] = gg[i] = 0.f;
            }
            a += gg[i];
            b += gg[i] * xh[i];
        }
        a = ix_wave_sum(a) / D;
        b = ix_wave_sum(b) / D;
        float* o = dx + row * D;
        if (VEC) {
#pragma unroll
            for (int v = 0; v < NREG / 4; ++v) {
                const int c0 = 256 * v + 4 * lane;
                if (c0 < D)
                    *reinterpret_cast<float4*>(o + c0) = make_float4(r * (gg[4 * v] - a - xh[4 * v] * b), r * (gg[4 * v + 1] - a - xh[4 * v + 1] * b),
                                                                    r * (gg[4 * v + 2] - a - xh[4 * v + 2] * b), r * (gg[4 * v + 3] - a - xh[4 * v + 3] * b));
            }
        } else {
#pragma unroll
            for (int i = 0; i < NREG; ++i) {
                const int c = lane + 64 * i;
                if (c < D) o[c] = r * (gg[i] - a - xh[i] * b);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NREG; ++i) {
        sg[w][col(i)] = ag[i];
        sb[w][col(i)] = ab[i];
    }
    __syncthreads();
    // column sums over the rows of a group: one partial per workgroup, added in workgroup order by the last one to arrive
    if (gridDim.x == 1) {
        for (int c = threadIdx.x; c < D; c += 256) {
            dgamma[c] = sg[0][c] + sg[1][c] + sg[2][c] + sg[3][c];
            dbeta[c] = sb[0][c] + sb[1][c] + sb[2][c] + sb[3][c];
        }
        return;
    }
    // part: [group][part1: gridDim.x x 2D | part2: cohorts x 2D]
    const int ncoh = (gridDim.x + IX_COHORT - 1) / IX_COHORT;
    float* p1 = part + (int64_t)blockIdx.y * (gridDim.x + ncoh) * 2 * D;
    float* pg = p1 + (int64_t)blockIdx.x * 2 * D;
    for (int c = threadIdx.x; c < D; c += 256) {
        ix_store_agent(pg + c, sg[0][c] + sg[1][c] + sg[2][c] + sg[3][c]);
        ix_store_agent(pg + D + c, sb[0][c] + sb[1][c] + sb[2][c] + sb[3][c]);
    }
    ix_ordered_colsum(p1, p1 + (int64_t)gridDim.x * 2 * D, tickets + (int64_t)blockIdx.y * (ncoh + 1), blockIdx.x, gridDim.x, 2 * D,
                      [=](int c, float t) { if (c < D) dgamma[c] = t; else dbeta[c - D] = t; });
}

static int ln_row_groups(int64_t rows) { return rows > 4096 ? 8 : (rows > 512 ? 2 : 1); }
static unsigned ln_grid_x(int64_t rows) {
    const int rg = ln_row_groups(rows);
    return (unsigned)((rows + ROWS_PER_BLOCK * rg - 1) / (ROWS_PER_BLOCK * rg));
}
// scratch of the LayerNorm backward / double backward: IX_TICKET_BYTES of tickets (zero on entry, left zero) + one partial
// (dgamma, dbeta) pair per workgroup.  rows = rows PER GROUP.
extern "C" int ix_workspace_bytes_layernorm_bwd(int64_t rows, int D, int groups, size_t* out) {
    IX_CHECK_ARG(out != nullptr, "ix_workspace_bytes_layernorm_bwd: null out");
    const unsigned gx = rows > 0 ? ln_grid_x(rows) : 1;
    *out = gx > 1 ? IX_TICKET_BYTES + sizeof(float) * 2 * (size_t)D * (size_t)(gx + ix_cohorts(gx)) * (size_t)(groups > 0 ? groups : 1) : 0;
    return IX_OK;
}

static int ln_scratch(const char* who, int64_t rows, int D, int groups, int vecs, void* workspace, size_t workspace_bytes,
                      float** part, unsigned int** tickets) {
    *part = nullptr;
    *tickets = nullptr;
    if (rows <= 0 || ln_grid_x(rows) <= 1) return IX_OK;
    const unsigned gx = ln_grid_x(rows);
    const size_t need = IX_TICKET_BYTES + sizeof(float) * vecs * (size_t)D * (size_t)(gx + ix_cohorts(gx)) * (size_t)groups;
    if (!workspace || workspace_bytes < need || !ix_al16(workspace) || (int64_t)groups * (ix_cohorts(gx) + 1) > IX_MAX_TICKETS) {
        ix_set_error("%s: workspace of %zu bytes (16-byte aligned) needed, %zu given", who, need, workspace ? workspace_bytes : (size_t)0);
        return IX_ERR_WORKSPACE;
    }
    *tickets = static_cast<unsigned int*>(workspace);
    *part = reinterpret_cast<float*>(static_cast<char*>(workspace) + IX_TICKET_BYTES);
    return IX_OK;
}

__global__ void ln_zero_kernel(float* __restrict__ p, int64_t n) {
    const int64_t gs = (int64_t)gridDim.x * blockDim.x;
    for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < n; k += gs) p[k] = 0.f;
}

extern "C" int ix_layernorm_bwd_f32(const float* dy, const float* x, const float* gamma, const float* mean,
                                    const float* rstd, float* dx, float* dgamma, float* dbeta, int64_t rows, int D,
                                    int groups, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    IX_CHECK_ARG(dgamma && dbeta && groups >= 1 && groups <= 65535, "ix_layernorm_bwd_f32: bad dgamma/dbeta/groups");
    IX_CHECK_ARG(D > 0 && D <= 1024, "ix_layernorm_bwd_f32: D=%d unsupported (1..1024)", D);
    if (rows <= 0) {
        const int64_t n = (int64_t)D * groups;
        hipLaunchKernelGGL(ln_zero_kernel, dim3(ix_grid_1d(n, 256)), dim3(256), 0, stream, dgamma, n);
        hipLaunchKernelGGL(ln_zero_kernel, dim3(ix_grid_1d(n, 256)), dim3(256), 0, stream, dbeta, n);
        return IX_OK;
    }
    IX_CHECK_ARG(dy && x && gamma && mean && rstd && dx, "ix_layernorm_bwd_f32: null pointer");
    float* part;
    unsigned int* tickets;
    const int rc = ln_scratch("ix_layernorm_bwd_f32", rows, D, groups, 2, workspace, workspace_bytes, &part, &tickets);
    if (rc != IX_OK) return rc;
    const int row_groups = ln_row_groups(rows);
    dim3 grid(ln_grid_x(rows), groups), block(256);
#define LNB(N, V) hipLaunchKernelGGL((ln_bwd_kernel<N, V>), grid, block, 0, stream, dy, x, gamma, mean, rstd, dx, dgamma, dbeta, rows, D, row_groups, part, tickets)
    const bool vec = D % 4 == 0 && ix_al16(dy) && ix_al16(x) && ix_al16(dx);
    if (vec) {
        if (D <= 256) LNB(4, true);
        else if (D <= 512) LNB(8, true);
        else LNB(16, true);
    } else {
        if (D <= 256) LNB(4, false);
        else if (D <= 512) LNB(8, false);
        else LNB(16, false);
    }
#undef LNB
    IX_CHECK_LAUNCH("ix_layernorm_bwd_f32");
    return IX_OK;
}

// Double backward of the node (dy, x, gamma) -> (dx, dgamma, dbeta).  Upstream grads: Gx [rows,D] (for dx),
// Gg [D] (for dgamma), Gb [D] (for dbeta); any may be null (= zero).  With m(.) the feature mean, per row:
//   g = dy*gamma, a = m(g), b = m(g*xh), dx = r*(g - a - xh*b), P(v) = r*(v - m(v) - xh*m(v*xh))
//   grad_dy    = P(Gx)*gamma + Gg*xh + Gb
//   grad_gamma = sum_rows P(Gx)*dy
//   Xh         = -r*b*Gx - r*g*m(Gx*xh) + Gg*dy
//   grad_x     = P(Xh) - r*xh*m(Gx*dx)
template <int NREG, bool VEC>   // (VEC as in ln_bwd_kernel: four consecutive columns per lane and 256-column slab)
__global__ __launch_bounds__(256) void ln_bwd_bwd_kernel(const float* __restrict__ Gx, const float* __restrict__ Gg,
                                                         const float* __restrict__ Gb, const float* __restrict__ dy,
                                                         const float* __restrict__ x, const float* __restrict__ gamma,
                                                         const float* __restrict__ mean,
                                                         const float* __restrict__ rstd, float* __restrict__ grad_dy,
                                                         float* __restrict__ grad_x, float* __restrict__ grad_gamma,
                                                         int64_t rows, int D, int row_groups, float* __restrict__ part,
                                                         unsigned int* tickets) {
    __shared__ float sg[ROWS_PER_BLOCK][64 * NREG];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    {   // blockIdx.y = group
        const int64_t go = (int64_t)blockIdx.y * rows;
        if (Gx) Gx += go * D;
        if (Gg) Gg += (int64_t)blockIdx.y * D;
        if (Gb) Gb += (int64_t)blockIdx.y * D;
        dy += go * D; x += go * D; grad_dy += go * D; grad_x += go * D; mean += go; rstd += go;
        gamma += (int64_t)blockIdx.y * D; grad_gamma += (int64_t)blockIdx.y * D;
    }
    auto col = [&](int i) { return VEC ? 256 * (i >> 2) + 4 * lane + (i & 3) : lane + 64 * i; };
    float ag[NREG], gm[NREG], ggv[NREG], gbv[NREG];
#pragma unroll
    for (int i = 0; i < NREG; ++i) {
        const int c = col(i);
        ag[i] = 0.f;
        gm[i] = c < D ? gamma[c] : 0.f;   // (the same columns in every row of this wave)
        ggv[i] = (Gg && c < D) ? Gg[c] : 0.f;
        gbv[i] = (Gb && c < D) ? Gb[c] : 0.f;
    }
    for (int grp = 0; grp < row_groups; ++grp) {
        const int64_t row = ((int64_t)blockIdx.x * row_groups + grp) * ROWS_PER_BLOCK + w;
        if (row >= rows) break;
        const float mu = mean[row], r = rstd[row];
        const int64_t off = row * D;
        float xh[NREG], g[NREG], gx[NREG], d[NREG];
        float a = 0.f, b = 0.f, mgx = 0.f, mgxxh = 0.f;
        if (VEC) {
#pragma unroll
            for (int v = 0; v < NREG / 4; ++v) {
                const int c0 = 256 * v + 4 * lane;
                float4 d4 = make_float4(0.f, 0.f, 0.f, 0.f), x4 = d4, q4 = d4;
                if (c0 < D) {
                    d4 = *reinterpret_cast<const float4*>(dy + off + c0);
                    x4 = *reinterpret_cast<const float4*>(x + off + c0);
                    if (Gx) q4 = *reinterpret_cast<const float4*>(Gx + off + c0);
                }
                d[4 * v] = d4.x; d[4 * v + 1] = d4.y; d[4 * v + 2] = d4.z; d[4 * v + 3] = d4.w;
                xh[4 * v] = x4.x; xh[4 * v + 1] = x4.y; xh[4 * v + 2] = x4.z; xh[4 * v + 3] = x4.w;
                gx[4 * v] = q4.x; gx[4 * v + 1] = q4.y; gx[4 * v + 2] = q4.z; gx[4 * v + 3] = q4.w;
            }
        } else {
#pragma unroll
            for (int i = 0; i < NREG; ++i) {
                const int c = lane + 64 * i;
                d[i] = c < D ? dy[off + c] : 0.f;
                xh[i] = c < D ? x[off + c] : 0.f;
                gx[i] = (Gx && c < D) ? Gx[off + c] : 0.f;
            }
        }
#pragma unroll
        for (int i = 0; i < NREG; ++i) {
            if (col(i) < D) {
                xh[i] = (xh[i] - mu) * r;
                g[i] = d[i] * gm[i];
            } else {
                d[i] = xh[i] = g[i] = gx[i] = 0.f;
            }
            a += g[i];
            b += g[i] * xh[i];
            mgx += gx[i];
            mgxxh += gx[i] * xh[i];
        }
        a = ix_wave_sum(a) / D;
        b = ix_wave_sum(b) / D;
        mgx = ix_wave_sum(mgx) / D;
        mgxxh = ix_wave_sum(mgxxh) / D;
        // second round of row statistics: m(Gx*dx), m(Xh), m(Xh*xh)
        float Xh[NREG];
        float mgxdx = 0.f, mX = 0.f, mXxh = 0.f;
#pragma unroll
        for (int i = 0; i < NREG; ++i) {
            if (col(i) < D) {
                const float dxv = r * (g[i] - a - xh[i] * b);
                const float gg = ggv[i];
                Xh[i] = -r * b * gx[i] - r * g[i] * mgxxh + gg * d[i];
                mgxdx += gx[i] * dxv;
                mX += Xh[i];
                mXxh += Xh[i] * xh[i];
            } else {
                Xh[i] = 0.f;
            }
        }
        mgxdx = ix_wave_sum(mgxdx) / D;
        mX = ix_wave_sum(mX) / D;
        mXxh = ix_wave_sum(mXxh) / D;
        float o1[NREG], o2[NREG];
#pragma unroll
        for (int i = 0; i < NREG; ++i) {
            const float pgx = r * (gx[i] - mgx - xh[i] * mgxxh);
            o1[i] = pgx * gm[i] + ggv[i] * xh[i] + gbv[i];
            o2[i] = r * (Xh[i] - mX - xh[i] * mXxh) - r * xh[i] * mgxdx;
            if (col(i) < D) ag[i] += pgx * d[i];
        }
        if (VEC) {
#pragma unroll
            for (int v = 0; v < NREG / 4; ++v) {
                const int c0 = 256 * v + 4 * lane;
                if (c0 < D) {
                    *reinterpret_cast<float4*>(grad_dy + off + c0) = make_float4(o1[4 * v], o1[4 * v + 1], o1[4 * v + 2], o1[4 * v + 3]);
                    *reinterpret_cast<float4*>(grad_x + off + c0) = make_float4(o2[4 * v], o2[4 * v + 1], o2[4 * v + 2], o2[4 * v + 3]);
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < NREG; ++i) {
                const int c = lane + 64 * i;
                if (c < D) { grad_dy[off + c] = o1[i]; grad_x[off + c] = o2[i]; }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NREG; ++i) sg[w][col(i)] = ag[i];
    __syncthreads();
    if (gridDim.x == 1) {
        for (int c = threadIdx.x; c < D; c += 256) grad_gamma[c] = sg[0][c] + sg[1][c] + sg[2][c] + sg[3][c];
        return;
    }
    const int ncoh = (gridDim.x + IX_COHORT - 1) / IX_COHORT;
    float* p1 = part + (int64_t)blockIdx.y * (gridDim.x + ncoh) * D;
    float* pg = p1 + (int64_t)blockIdx.x * D;
    for (int c = threadIdx.x; c < D; c += 256) ix_store_agent(pg + c, sg[0][c] + sg[1][c] + sg[2][c] + sg[3][c]);
    ix_ordered_colsum(p1, p1 + (int64_t)gridDim.x * D, tickets + (int64_t)blockIdx.y * (ncoh + 1), blockIdx.x, gridDim.x, D,
                      [=](int c, float t) { grad_gamma[c] = t; });
}

extern "C" int ix_layernorm_bwd_bwd_f32(const float* Gx, const float* Gg, const float* Gb, const float* dy,
                                        const float* x, const float* gamma, const float* mean, const float* rstd,
                                        float* grad_dy, float* grad_x, float* grad_gamma, int64_t rows, int D,
                                        int groups, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    IX_CHECK_ARG(grad_gamma && groups >= 1 && groups <= 65535, "ix_layernorm_bwd_bwd_f32: bad grad_gamma/groups");
    IX_CHECK_ARG(D > 0 && D <= 1024, "ix_layernorm_bwd_bwd_f32: D=%d unsupported (1..1024)", D);
    if (rows <= 0) {
        const int64_t n = (int64_t)D * groups;
        hipLaunchKernelGGL(ln_zero_kernel, dim3(ix_grid_1d(n, 256)), dim3(256), 0, stream, grad_gamma, n);
        return IX_OK;
    }
    IX_CHECK_ARG(dy && x && gamma && mean && rstd && grad_dy && grad_x, "ix_layernorm_bwd_bwd_f32: null pointer");
    float* part;
    unsigned int* tickets;
    const int rc = ln_scratch("ix_layernorm_bwd_bwd_f32", rows, D, groups, 1, workspace, workspace_bytes, &part, &tickets);
    if (rc != IX_OK) return rc;
    const int row_groups = ln_row_groups(rows);
    dim3 grid(ln_grid_x(rows), groups), block(256);
#define LNBB(N, V) hipLaunchKernelGGL((ln_bwd_bwd_kernel<N, V>), grid, block, 0, stream, Gx, Gg, Gb, dy, x, gamma, mean, rstd, grad_dy, grad_x, grad_gamma, rows, D, row_groups, part, tickets)
    const bool vec = D % 4 == 0 && ix_al16(dy) && ix_al16(x) && ix_al16(grad_dy) && ix_al16(grad_x) && (!Gx || ix_al16(Gx));
    if (vec) {
        if (D <= 256) LNBB(4, true);
        else if (D <= 512) LNBB(8, true);
        else LNBB(16, true);
    } else {
        if (D <= 256) LNBB(4, false);
        else if (D <= 512) LNBB(8, false);
        else LNBB(16, false);
    }
#undef LNBB
    IX_CHECK_LAUNCH("ix_layernorm_bwd_bwd_f32");
    return IX_OK;
}
