// HBM-bound elementwise / broadcast / column-reduction kernels of the Interactron hot path.
// All are grid-stride, 16-byte vectorised when the pointers allow, and hold no state.
//
// reference sites: residual adds and ReLU/GELU/sigmoid/dropout in models/detr_models/transformer.py:148-232,
// models/gpt.py:39-78, models/detr_models/detr.py:299-311; FrozenBatchNorm2d affine in
// models/detr_models/backbone.py:44-54.  The *_bwd / *_bwd_bwd forms are what the MAML meta-gradient
// (models/interactron.py:99-123, create_graph=True) differentiates through a second time.
#include "common.h"

#define EW_BLOCK 256

static inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// ---- generic unary / binary / ternary map with float4 bulk + scalar tail ------------------------------------
template <typename F>
__global__ void map1_kernel(const float* __restrict__ a, float* __restrict__ o, int64_t n, bool vec, F f) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (vec) {
        const int64_t n4 = n >> 2;
        for (int64_t k = i; k < n4; k += stride) {
            float4 x = reinterpret_cast<const float4*>(a)[k];
            float4 y = make_float4(f(x.x), f(x.y), f(x.z), f(x.w));
            reinterpret_cast<float4*>(o)[k] = y;
        }
        for (int64_t k = (n4 << 2) + i; k < n; k += stride) o[k] = f(a[k]);
    } else {
        for (int64_t k = i; k < n; k += stride) o[k] = f(a[k]);
    }
}

template <typename F>
__global__ void map2_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ o, int64_t n,
                            bool vec, F f) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (vec) {
        const int64_t n4 = n >> 2;
        for (int64_t k = i; k < n4; k += stride) {
            float4 x = reinterpret_cast<const float4*>(a)[k];
            float4 y = reinterpret_cast<const float4*>(b)[k];
            reinterpret_cast<float4*>(o)[k] = make_float4(f(x.x, y.x), f(x.y, y.y), f(x.z, y.z), f(x.w, y.w));
        }
        for (int64_t k = (n4 << 2) + i; k < n; k += stride) o[k] = f(a[k], b[k]);
    } else {
        for (int64_t k = i; k < n; k += stride) o[k] = f(a[k], b[k]);
    }
}

// two outputs from three inputs (the *_bwd_bwd kernels)
template <typename F>
__global__ void map3x2_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c,
                              float* __restrict__ o0, float* __restrict__ o1, int64_t n, F f) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < n; k += stride) {
        float r0, r1;
        f(a[k], b[k], c[k], r0, r1);
        o0[k] = r0;
        o1[k] = r1;
    }
}

#define LAUNCH1(name, a, o, n, stream, ...)                                                                  \
    do {                                                                                                     \
        if ((n) <= 0) return IX_OK;                                                                          \
        IX_CHECK_ARG((a) && (o), name ": null pointer");                                                     \
        const bool vec__ = al16(a) && al16(o);                                                               \
        hipLaunchKernelGGL(map1_kernel, dim3(ix_grid_1d(((n) + 3) / 4, EW_BLOCK)), dim3(EW_BLOCK), 0, stream, \
                           a, o, (int64_t)(n), vec__, __VA_ARGS__);                                          \
        IX_CHECK_LAUNCH(name);                                                                               \
        return IX_OK;                                                                                        \
    } while (0)

#define LAUNCH2(name, a, b, o, n, stream, ...)                                                               \
    do {                                                                                                     \
        if ((n) <= 0) return IX_OK;                                                                          \
        IX_CHECK_ARG((a) && (b) && (o), name ": null pointer");                                              \
        const bool vec__ = al16(a) && al16(b) && al16(o);                                                    \
        hipLaunchKernelGGL(map2_kernel, dim3(ix_grid_1d(((n) + 3) / 4, EW_BLOCK)), dim3(EW_BLOCK), 0, stream, \
                           a, b, o, (int64_t)(n), vec__, __VA_ARGS__);                                       \
        IX_CHECK_LAUNCH(name);                                                                               \
        return IX_OK;                                                                                        \
    } while (0)

// out = ((a0 + a1) + a2) + ... (2 <= n <= 8 tensors of one shape, summed left to right): the gradient of a tensor with several
// consumers in ONE pass (hipops.Fanout / SumN) instead of autograd's n - 1 two-operand adds (reference: every residual
// connection and every shared activation of models/detr_models/transformer.py:148-232, models/gpt.py:60-78, backbone
// bottlenecks; under autograd each is a chain of aten::add).  16-byte loads when every pointer allows.
struct SumNArgs {
    const float* src[8];
    int n;
};
__global__ void sum_n_kernel(SumNArgs a, float* __restrict__ o, int64_t count, bool vec) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (vec) {
        const int64_t n4 = count >> 2;
        for (int64_t k = i; k < n4; k += stride) {
            float4 s = reinterpret_cast<const float4*>(a.src[0])[k];
#pragma unroll
            for (int t = 1; t < 8; ++t) {
                if (t < a.n) {
                    const float4 x = reinterpret_cast<const float4*>(a.src[t])[k];
                    s.x += x.x; s.y += x.y; s.z += x.z; s.w += x.w;
                }
            }
            reinterpret_cast<float4*>(o)[k] = s;
        }
        for (int64_t k = (n4 << 2) + i; k < count; k += stride) {
            float s = a.src[0][k];
            for (int t = 1; t < a.n; ++t) s += a.src[t][k];
            o[k] = s;
        }
    } else {
        for (int64_t k = i; k < count; k += stride) {
            float s = a.src[0][k];
            for (int t = 1; t < a.n; ++t) s += a.src[t][k];
            o[k] = s;
        }
    }
}
extern "C" int ix_sum_n_f32(const float* const* srcs, int n, float* out, int64_t count, hipStream_t stream) {
    if (count <= 0) return IX_OK;
    IX_CHECK_ARG(srcs && out && n >= 2 && n <= 8, "ix_sum_n_f32: 2..8 source tensors");
    SumNArgs a;
    bool vec = al16(out);
    for (int t = 0; t < 8; ++t) {
        a.src[t] = t < n ? srcs[t] : srcs[0];
        IX_CHECK_ARG(a.src[t] != nullptr, "ix_sum_n_f32: null source");
        vec = vec && al16(a.src[t]);
    }
    a.n = n;
    hipLaunchKernelGGL(sum_n_kernel, dim3(ix_grid_1d((count + 3) / 4, EW_BLOCK)), dim3(EW_BLOCK), 0, stream, a, out, count, vec);
    IX_CHECK_LAUNCH("ix_sum_n_f32");
    return IX_OK;
}

// out = alpha*a + beta*b
extern "C" int ix_axpby_f32(const float* a, const float* b, float* out, int64_t n, float alpha, float beta,
                            hipStream_t stream) {
    LAUNCH2("ix_axpby_f32", a, b, out, n, stream, [=] __device__(float x, float y) { return alpha * x + beta * y; });
}

extern "C" int ix_mul_f32(const float* a, const float* b, float* out, int64_t n, hipStream_t stream) {
    LAUNCH2("ix_mul_f32", a, b, out, n, stream, [] __device__(float x, float y) { return x * y; });
}

extern "C" int ix_scale_f32(const float* x, float* out, int64_t n, float alpha, hipStream_t stream) {
    LAUNCH1("ix_scale_f32", x, out, n, stream, [=] __device__(float v) { return alpha * v; });
}

// out = x * s[0], s a device scalar (keeps scalar second-order terms on the device)
extern "C" int ix_scale_dev_f32(const float* x, const float* s, float* out, int64_t n, hipStream_t stream) {
    IX_CHECK_ARG(s, "ix_scale_dev_f32: null scalar");
    LAUNCH1("ix_scale_dev_f32", x, out, n, stream, [=] __device__(float v) { return v * s[0]; });
}

extern "C" int ix_relu_f32(const float* x, float* out, int64_t n, hipStream_t stream) {
    LAUNCH1("ix_relu_f32", x, out, n, stream, [] __device__(float v) { return v > 0.f ? v : 0.f; });
}

// dx = dy * [y > 0]   (y = relu output; linear in dy, so it is its own second-order form)
extern "C" int ix_relu_bwd_f32(const float* dy, const float* y, float* dx, int64_t n, hipStream_t stream) {
    LAUNCH2("ix_relu_bwd_f32", dy, y, dx, n, stream, [] __device__(float g, float v) { return v > 0.f ? g : 0.f; });
}

__device__ __forceinline__ float gelu_cdf(float x) { return 0.5f * (1.f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_pdf(float x) { return 0.39894228040143267794f * __expf(-0.5f * x * x); }

// exact (erf) GELU, as nn.GELU() in reference models/gpt.py:68
extern "C" int ix_gelu_f32(const float* x, float* out, int64_t n, hipStream_t stream) {
    LAUNCH1("ix_gelu_f32", x, out, n, stream, [] __device__(float v) { return v * gelu_cdf(v); });
}

extern "C" int ix_gelu_bwd_f32(const float* dy, const float* x, float* dx, int64_t n, hipStream_t stream) {
    LAUNCH2("ix_gelu_bwd_f32", dy, x, dx, n, stream,
            [] __device__(float g, float v) { return g * (gelu_cdf(v) + v * gelu_pdf(v)); });
}

// given G = dL/d(dx) of the gelu_bwd node: grad_dy = G * gelu'(x), grad_x = G * dy * gelu''(x),
// gelu''(x) = pdf(x) * (2 - x^2)
extern "C" int ix_gelu_bwd_bwd_f32(const float* G, const float* dy, const float* x, float* grad_dy, float* grad_x,
                                   int64_t n, hipStream_t stream) {
    if (n <= 0) return IX_OK;
    IX_CHECK_ARG(G && dy && x && grad_dy && grad_x, "ix_gelu_bwd_bwd_f32: null pointer");
    hipLaunchKernelGGL(map3x2_kernel, dim3(ix_grid_1d(n, EW_BLOCK)), dim3(EW_BLOCK), 0, stream, G, dy, x, grad_dy,
                       grad_x, n, [] __device__(float g, float d, float v, float& r0, float& r1) {
                           const float pdf = gelu_pdf(v);
                           r0 = g * (gelu_cdf(v) + v * pdf);
                           r1 = g * d * pdf * (2.f - v * v);
                       });
    IX_CHECK_LAUNCH("ix_gelu_bwd_bwd_f32");
    return IX_OK;
}

extern "C" int ix_sigmoid_f32(const float* x, float* out, int64_t n, hipStream_t stream) {
    LAUNCH1("ix_sigmoid_f32", x, out, n, stream, [] __device__(float v) { return 1.f / (1.f + __expf(-v)); });
}

extern "C" int ix_sigmoid_bwd_f32(const float* dy, const float* y, float* dx, int64_t n, hipStream_t stream) {
    LAUNCH2("ix_sigmoid_bwd_f32", dy, y, dx, n, stream, [] __device__(float g, float v) { return g * v * (1.f - v); });
}

// G = dL/d(dx): grad_dy = G*y*(1-y), grad_y = G*dy*(1-2y)
extern "C" int ix_sigmoid_bwd_bwd_f32(const float* G, const float* dy, const float* y, float* grad_dy, float* grad_y,
                                      int64_t n, hipStream_t stream) {
    if (n <= 0) return IX_OK;
    IX_CHECK_ARG(G && dy && y && grad_dy && grad_y, "ix_sigmoid_bwd_bwd_f32: null pointer");
    hipLaunchKernelGGL(map3x2_kernel, dim3(ix_grid_1d(n, EW_BLOCK)), dim3(EW_BLOCK), 0, stream, G, dy, y, grad_dy,
                       grad_y, n, [] __device__(float g, float d, float v, float& r0, float& r1) {
                           r0 = g * v * (1.f - v);
                           r1 = g * d * (1.f - 2.f * v);
                       });
    IX_CHECK_LAUNCH("ix_sigmoid_bwd_bwd_f32");
    return IX_OK;
}

// ---- dropout: counter-hash mask regenerated from (seed, element index), so backward needs no stored mask ----
__device__ __forceinline__ uint32_t mix32(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return (uint32_t)((z ^ (z >> 31)) >> 32);
}

__global__ void dropout_kernel(const float* __restrict__ x, float* __restrict__ o, int64_t n, uint32_t thresh,
                               float scale, uint64_t seed, const uint64_t* __restrict__ salt) {
    if (salt) seed ^= *salt;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < n; k += stride) {
        const uint32_t r = mix32(seed ^ ((uint64_t)k * 0xD6E8FEB86659FD93ull));
        o[k] = r >= thresh ? x[k] * scale : 0.f;
    }
}

// out = x * keep / (1-p); the same (seed) applied to a gradient is the exact adjoint (the op is linear).
extern "C" int ix_dropout_f32(const float* x, float* out, int64_t n, float p, uint64_t seed, hipStream_t stream) {
    if (n <= 0) return IX_OK;
    IX_CHECK_ARG(x && out, "ix_dropout_f32: null pointer");
    IX_CHECK_ARG(p >= 0.f && p < 1.f, "ix_dropout_f32: p=%f outside [0,1)", p);
    const uint32_t thresh = (uint32_t)((double)p * 4294967296.0);
    hipLaunchKernelGGL(dropout_kernel, dim3(ix_grid_1d(n, EW_BLOCK)), dim3(EW_BLOCK), 0, stream, x, out, n, thresh,
                       1.f / (1.f - p), seed, ix_g_salt);
    IX_CHECK_LAUNCH("ix_dropout_f32");
    return IX_OK;
}

// relu + dropout as one pass (the transformer FFNs: dropout(relu(linear1(x))), reference transformer.py:158,229), and
// its backward dx = dy * [y > 0] / keep: y = m relu(x) / keep is positive exactly where both the mask and the relu pass,
// so the backward needs neither the mask hash nor x.
__global__ void relu_dropout_kernel(const float* __restrict__ x, float* __restrict__ o, int64_t n, uint32_t thresh,
                                    float scale, uint64_t seed, const uint64_t* __restrict__ salt) {
    if (salt) seed ^= *salt;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < n; k += stride) {
        const uint32_t r = mix32(seed ^ ((uint64_t)k * 0xD6E8FEB86659FD93ull));
        const float v = x[k];
        o[k] = (r >= thresh && v > 0.f) ? v * scale : 0.f;
    }
}

extern "C" int ix_relu_dropout_f32(const float* x, float* out, int64_t n, float p, uint64_t seed, hipStream_t stream) {
    if (n <= 0) return IX_OK;
    IX_CHECK_ARG(x && out, "ix_relu_dropout_f32: null pointer");
    IX_CHECK_ARG(p >= 0.f && p < 1.f, "ix_relu_dropout_f32: p=%f outside [0,1)", p);
    hipLaunchKernelGGL(relu_dropout_kernel, dim3(ix_grid_1d(n, EW_BLOCK)), dim3(EW_BLOCK), 0, stream, x, out, n,
                       (uint32_t)((double)p * 4294967296.0), 1.f / (1.f - p), seed, ix_g_salt);
    IX_CHECK_LAUNCH("ix_relu_dropout_f32");
    return IX_OK;
}

// x + dropout(a): the residual adds of the transformer blocks (transformer.py:157-160,222-231, gpt.py:75-77) in one pass
__global__ void add_dropout_kernel(const float* __restrict__ x, const float* __restrict__ a, float* __restrict__ o, int64_t n,
                                   uint32_t thresh, float scale, uint64_t seed, const uint64_t* __restrict__ salt) {
    if (salt) seed ^= *salt;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < n; k += stride) {
        const uint32_t r = mix32(seed ^ ((uint64_t)k * 0xD6E8FEB86659FD93ull));
        o[k] = x[k] + (r >= thresh ? a[k] * scale : 0.f);
    }
}

extern "C" int ix_add_dropout_f32(const float* x, const float* a, float* out, int64_t n, float p, uint64_t seed,
                                  hipStream_t stream) {
    if (n <= 0) return IX_OK;
    IX_CHECK_ARG(x && a && out, "ix_add_dropout_f32: null pointer");
    IX_CHECK_ARG(p >= 0.f && p < 1.f, "ix_add_dropout_f32: p=%f outside [0,1)", p);
    hipLaunchKernelGGL(add_dropout_kernel, dim3(ix_grid_1d(n, EW_BLOCK)), dim3(EW_BLOCK), 0, stream, x, a, out, n,
                       (uint32_t)((double)p * 4294967296.0), 1.f / (1.f - p), seed, ix_g_salt);
    IX_CHECK_LAUNCH("ix_add_dropout_f32");
    return IX_OK;
}

extern "C" int ix_relu_bwd_scaled_f32(const float* dy, const float* y, float* dx, int64_t n, float scale, hipStream_t stream) {
    LAUNCH2("ix_relu_bwd_scaled_f32", dy, y, dx, n, stream, [scale] __device__(float g, float v) { return v > 0.f ? g * scale : 0.f; });
}

// (a + b) * [y > 0]: the ReLU derivative of a bottleneck tail whose output feeds two consumers (the next block's first
// convolution and its identity branch) -- their two gradients are summed HERE instead of by a pass of their own (hipops.ReluBwdSum)
__global__ void relu_bwd_sum_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ y,
                                    float* __restrict__ o, int64_t n, bool vec) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x, i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    int64_t done = 0;
    if (vec) {
        const int64_t n4 = n >> 2;
        for (int64_t k = i; k < n4; k += stride) {
            const float4 p = reinterpret_cast<const float4*>(a)[k], q = reinterpret_cast<const float4*>(b)[k];
            const float4 v = reinterpret_cast<const float4*>(y)[k];
            reinterpret_cast<float4*>(o)[k] = make_float4(v.x > 0.f ? p.x + q.x : 0.f, v.y > 0.f ? p.y + q.y : 0.f,
                                                          v.z > 0.f ? p.z + q.z : 0.f, v.w > 0.f ? p.w + q.w : 0.f);
        }
        done = n4 << 2;
    }
    for (int64_t k = done + i; k < n; k += stride) o[k] = y[k] > 0.f ? a[k] + b[k] : 0.f;
}
extern "C" int ix_relu_bwd_sum_f32(const float* a, const float* b, const float* y, float* out, int64_t n, hipStream_t stream) {
    if (n <= 0) return IX_OK;
    IX_CHECK_ARG(a && b && y && out, "ix_relu_bwd_sum_f32: null pointer");
    const bool vec = al16(a) && al16(b) && al16(y) && al16(out);
    hipLaunchKernelGGL(relu_bwd_sum_kernel, dim3(ix_grid_1d((n + 3) / 4, EW_BLOCK)), dim3(EW_BLOCK), 0, stream, a, b, y, out, n, vec);
    IX_CHECK_LAUNCH("ix_relu_bwd_sum_f32");
    return IX_OK;
}

// ---- row-vector broadcast and column reductions ----------------------------------------------------------
__global__ void add_rowvec_kernel(const float* __restrict__ a, const float* __restrict__ v, float* __restrict__ o,
                                  int64_t n, int C) {
    // blockIdx.y = group: a/o advance by n, v by C
    a += (int64_t)blockIdx.y * n;
    o += (int64_t)blockIdx.y * n;
    v += (int64_t)blockIdx.y * C;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < n; k += stride) o[k] = a[k] + v[k % C];
}

// out[r, c] = a[r, c] + v[c]   (a may be null: pure broadcast of v over R rows)
extern "C" int ix_add_rowvec_f32(const float* a, const float* v, float* out, int64_t rows, int C, int groups,
                                 hipStream_t stream) {
    const int64_t n = rows * C;
    if (n <= 0 || groups <= 0) return IX_OK;
    IX_CHECK_ARG(a && v && out && groups <= 65535, "ix_add_rowvec_f32: bad args");
    hipLaunchKernelGGL(add_rowvec_kernel, dim3(ix_grid_1d(n, EW_BLOCK), groups), dim3(EW_BLOCK), 0, stream, a, v, out, n, C);
    IX_CHECK_LAUNCH("ix_add_rowvec_f32");
    return IX_OK;
}

__global__ void bcast_rows_kernel(const float* __restrict__ v, float* __restrict__ o, int64_t n, int C) {
    o += (int64_t)blockIdx.y * n;
    v += (int64_t)blockIdx.y * C;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < n; k += stride) o[k] = v[k % C];
}

// out[r, c] = v[c]  -- the adjoint of ix_colsum_f32
extern "C" int ix_bcast_rows_f32(const float* v, float* out, int64_t rows, int C, int groups, hipStream_t stream) {
    const int64_t n = rows * C;
    if (n <= 0 || groups <= 0) return IX_OK;
    IX_CHECK_ARG(v && out && groups <= 65535, "ix_bcast_rows_f32: bad args");
    hipLaunchKernelGGL(bcast_rows_kernel, dim3(ix_grid_1d(n, EW_BLOCK), groups), dim3(EW_BLOCK), 0, stream, v, out, n, C);
    IX_CHECK_LAUNCH("ix_bcast_rows_f32");
    return IX_OK;
}

// Each block owns a band of rows and a 256-wide band of columns; lanes walk down rows (coalesced across columns).  With
// several row bands the band sums go to `part[group][band][c]` and the last band to finish adds them in band order
// (ix_last_block); part == nullptr: one atomic per column per block onto a zero-filled out (legacy, order-dependent).
// scratch of one (column band, group): [part1: gridDim.y x 256 | part2: cohorts x 256] floats
__device__ __forceinline__ float* colsum_slot(float* part) {
    const int ncoh = (gridDim.y + IX_COHORT - 1) / IX_COHORT;
    return part + ((int64_t)blockIdx.z * gridDim.x + blockIdx.x) * (gridDim.y + ncoh) * 256;
}
__device__ __forceinline__ void colsum_finish(float* __restrict__ out, float* slot, unsigned int* tickets, int C) {
    const int ncoh = (gridDim.y + IX_COHORT - 1) / IX_COHORT;
    const int c0 = blockIdx.x * 256;
    float* og = out + (int64_t)blockIdx.z * C + c0;
    const int ncols = min(256, C - c0);
    ix_ordered_colsum(slot, slot + (int64_t)gridDim.y * 256, tickets + ((int64_t)blockIdx.z * gridDim.x + blockIdx.x) * (ncoh + 1),
                      blockIdx.y, gridDim.y, 256, [=](int c, float t) { if (c < ncols) og[c] = t; });
}

__global__ void colsum_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t rows, int C,
                              int rows_per_block, float* __restrict__ part, unsigned int* tickets) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const float* xg = x + (int64_t)blockIdx.z * rows * C;   // blockIdx.z = group
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    float s = 0.f;
    if (c < C)
        for (int64_t r = r0; r < r1; ++r) s += xg[r * C + c];
    if (gridDim.y == 1) {
        if (c < C) out[(int64_t)blockIdx.z * C + c] = s;
        return;
    }
    if (!part) {
        if (c < C) unsafeAtomicAdd(&out[(int64_t)blockIdx.z * C + c], s);
        return;
    }
    float* slot = colsum_slot(part);
    ix_store_agent(slot + (int64_t)blockIdx.y * 256 + threadIdx.x, s);
    colsum_finish(out, slot, tickets, C);
}

// 16-byte form (C % 4 == 0, aligned rows): a thread owns 4 adjacent columns, the block's 4 waves take rows r, r+1, r+2,
// r+3 of the band with two independent accumulator sets (8 row loads in flight per thread), LDS-reduced per block.
__global__ __launch_bounds__(256) void colsum_vec_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t rows,
                                                         int C, int rows_per_block, float* __restrict__ part,
                                                         unsigned int* tickets) {
    __shared__ float4 red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = (blockIdx.x * 64 + lane) * 4;
    x += (int64_t)blockIdx.z * rows * C;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
    if (c < C) {
        int64_t r = r0 + wave;
        for (; r + 4 < r1; r += 8) {
            const float4 a = *reinterpret_cast<const float4*>(x + r * C + c);
            const float4 b = *reinterpret_cast<const float4*>(x + (r + 4) * C + c);
            s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
            s1.x += b.x; s1.y += b.y; s1.z += b.z; s1.w += b.w;
        }
        if (r < r1) {
            const float4 a = *reinterpret_cast<const float4*>(x + r * C + c);
            s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
        }
    }
    red[wave][lane] = make_float4(s0.x + s1.x, s0.y + s1.y, s0.z + s1.z, s0.w + s1.w);
    __syncthreads();
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    if (wave == 0 && c < C) {
        const float4 a = red[0][lane], b = red[1][lane], d = red[2][lane], e = red[3][lane];
        t = make_float4((a.x + b.x) + (d.x + e.x), (a.y + b.y) + (d.y + e.y), (a.z + b.z) + (d.z + e.z), (a.w + b.w) + (d.w + e.w));
    }
    float* og = out + (int64_t)blockIdx.z * C;
    if (gridDim.y == 1) {
        if (wave == 0 && c < C) *reinterpret_cast<float4*>(og + c) = t;
        return;
    }
    if (!part) {
        if (wave == 0 && c < C) {
            unsafeAtomicAdd(&og[c + 0], t.x);
            unsafeAtomicAdd(&og[c + 1], t.y);
            unsafeAtomicAdd(&og[c + 2], t.z);
            unsafeAtomicAdd(&og[c + 3], t.w);
        }
        return;
    }
    float* slot = colsum_slot(part);
    if (wave == 0) {
        float* ps = slot + (int64_t)blockIdx.y * 256 + lane * 4;
        ix_store_agent(ps + 0, t.x); ix_store_agent(ps + 1, t.y); ix_store_agent(ps + 2, t.z); ix_store_agent(ps + 3, t.w);
    }
    colsum_finish(out, slot, tickets, C);
}

__global__ void zero_f32_kernel(float* __restrict__ p, int64_t n) {
    const int64_t gs = (int64_t)gridDim.x * blockDim.x;
    for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < n; k += gs) p[k] = 0.f;
}

static size_t colsum_part_bytes(int nby, int C, int groups) {
    return sizeof(float) * 256 * (size_t)(nby + ix_cohorts(nby)) * (size_t)ix_div_up(C, 256) * (size_t)groups;
}
static int colsum_bands(int64_t rows, int* rpb_out) {
    // few rows: one band (single pass, nothing to combine); else bands of >= 64 rows, at most 2048 of them
    int rpb = rows <= 256 ? 256 : 64;
    while ((rows + rpb - 1) / rpb > 2048) rpb *= 2;
    *rpb_out = rpb;
    return (int)((rows + rpb - 1) / rpb);
}

extern "C" int ix_workspace_bytes_colsum_f32(int64_t rows, int C, int groups, size_t* out) {
    IX_CHECK_ARG(out != nullptr, "ix_workspace_bytes_colsum_f32: null out");
    int rpb;
    const int nby = rows > 0 ? colsum_bands(rows, &rpb) : 1;
    *out = nby > 1 ? IX_TICKET_BYTES + colsum_part_bytes(nby, C, groups > 0 ? groups : 1) : 0;
    return IX_OK;
}

// out[g, c] = sum_r x[g, r, c].  workspace (ix_workspace_bytes_colsum_f32; first IX_TICKET_BYTES zero on entry, left zero):
// ordered, run-to-run identical sums in ONE launch; NULL: atomics onto a zero-filled out.
extern "C" int ix_colsum_f32(const float* x, float* out, int64_t rows, int C, int groups, void* workspace, size_t workspace_bytes,
                             hipStream_t stream) {
    IX_CHECK_ARG(out && C >= 0 && groups >= 0 && groups <= 65535, "ix_colsum_f32: bad args");
    if (C == 0 || groups == 0) return IX_OK;
    if (rows <= 0) {
        hipLaunchKernelGGL(zero_f32_kernel, dim3(ix_grid_1d((int64_t)C * groups, 256)), dim3(256), 0, stream, out, (int64_t)C * groups);
        return IX_OK;
    }
    IX_CHECK_ARG(x, "ix_colsum_f32: null input");
    int rpb;
    const int nby = colsum_bands(rows, &rpb);
    dim3 grid(ix_div_up(C, 256), (unsigned)nby, groups);
    float* part = nullptr;
    unsigned int* tickets = nullptr;
    if (nby > 1 && workspace) {
        const size_t need = IX_TICKET_BYTES + colsum_part_bytes(nby, C, groups);
        if (workspace_bytes < need || !ix_al16(workspace) || (int64_t)grid.x * groups * (ix_cohorts(nby) + 1) > IX_MAX_TICKETS) {
            ix_set_error("ix_colsum_f32: workspace of %zu bytes (16-byte aligned) needed, %zu given", need, workspace_bytes);
            return IX_ERR_WORKSPACE;
        }
        tickets = static_cast<unsigned int*>(workspace);
        part = reinterpret_cast<float*>(static_cast<char*>(workspace) + IX_TICKET_BYTES);
    } else if (nby > 1) {
        hipLaunchKernelGGL(zero_f32_kernel, dim3(ix_grid_1d((int64_t)C * groups, 256)), dim3(256), 0, stream, out, (int64_t)C * groups);
    }
    if ((C & 3) == 0 && al16(x) && al16(out))
        hipLaunchKernelGGL(colsum_vec_kernel, grid, dim3(256), 0, stream, x, out, rows, C, rpb, part, tickets);
    else
        hipLaunchKernelGGL(colsum_kernel, grid, dim3(256), 0, stream, x, out, rows, C, rpb, part, tickets);
    IX_CHECK_LAUNCH("ix_colsum_f32");
    return IX_OK;
}

// block partials -> part[block]; the last block adds them in order (ix_last_block).  tickets[0] is this launch's counter.
__global__ void dot_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                           int64_t n, float* __restrict__ part, unsigned int* tickets) {
    __shared__ float red[4];
    float s = 0.f;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < n; k += stride) s += a[k] * b[k];
    s = ix_block_sum_256(s, red);
    if (gridDim.x == 1) {
        if (threadIdx.x == 0) *out = s;
        return;
    }
    if (threadIdx.x == 0) ix_store_agent(part + blockIdx.x, s);
    if (!ix_last_block(tickets, gridDim.x)) return;
    float t = 0.f;
    for (unsigned int i = threadIdx.x; i < gridDim.x; i += 256) t += ix_load_agent(part + i);   // (gridDim.x <= 256)
    t = ix_block_sum_256(t, red);
    if (threadIdx.x == 0) *out = t;
}

// out[0] = sum_i a[i]*b[i].  workspace: IX_TICKET_BYTES (zero on entry, left zero) + 1 KiB of partials; required once the
// vectors are long enough for several workgroups (n > 4096).
extern "C" int ix_dot_f32(const float* a, const float* b, float* out, int64_t n, void* workspace, size_t workspace_bytes,
                          hipStream_t stream) {
    IX_CHECK_ARG(out, "ix_dot_f32: null output");
    IX_CHECK_ARG(n <= 0 || (a && b), "ix_dot_f32: null input");
    int g = n > 4096 ? ix_grid_1d(n, 1024) : 1;
    if (g > 256) g = 256;
    if (g > 1) IX_CHECK_ARG(workspace && workspace_bytes >= IX_TICKET_BYTES + 1024 && ix_al16(workspace), "ix_dot_f32: workspace of %d bytes needed", IX_TICKET_BYTES + 1024);
    hipLaunchKernelGGL(dot_kernel, dim3(g), dim3(256), 0, stream, a, b, out, n > 0 ? n : 0,
                       g > 1 ? reinterpret_cast<float*>(static_cast<char*>(workspace) + IX_TICKET_BYTES) : nullptr,
                       static_cast<unsigned int*>(workspace));
    IX_CHECK_LAUNCH("ix_dot_f32");
    return IX_OK;
}

// ---- sum of row L2 norms: the learned loss of a chunk of episodes in ONE launch ---------------------------------------
// total = sum_e ||x_e||, x [E, n] (reference models/interactron.py:96: `learned_loss = torch.norm(fusion_out["loss"])`, once per
// task; the episode-batched step needs the E norms of a chunk and their sum: E dot products + E square roots + a stack + a
// sum before).  One workgroup: wave w takes rows w, w + 4, ...; the row sums are added in row order (deterministic).  The
// backward y = g x_e / ||x_e|| and ITS backward (the MAML meta-gradient differentiates the inner gradient) are the two
// kernels below; a zero row has gradient 0, as torch.norm's backward masks it.
__device__ __forceinline__ float rn_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__global__ __launch_bounds__(256) void rownorm_sum_kernel(const float* __restrict__ x, float* __restrict__ norms,
                                                          float* __restrict__ total, int E, int n) {
    __shared__ float sn[1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int e = wave; e < E; e += 4) {
        float s = 0.f;
        for (int j = lane; j < n; j += 64) { const float v = x[(int64_t)e * n + j]; s += v * v; }
        s = sqrtf(rn_wave_sum(s));
        if (lane == 0) { norms[e] = s; sn[e] = s; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int e = 0; e < E; ++e) t += sn[e];
        *total = t;
    }
}
// gx[e][j] = g * x[e][j] / norms[e]   (g: one device scalar)
__global__ void rownorm_sum_bwd_kernel(const float* __restrict__ x, const float* __restrict__ norms, const float* __restrict__ g,
                                       float* __restrict__ gx, int E, int n) {
    // grid-stride: ix_grid_1d caps the grid at 4 096 workgroups (E * n beyond 2^20 elements must still be written)
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < (int64_t)E * n; i += (int64_t)gridDim.x * blockDim.x) {
        const float nm = norms[i / n];
        gx[i] = nm > 0.f ? *g * x[i] / nm : 0.f;
    }
}
// cotangent H of y = g x / ||x_e||:  Gg = sum_e <H_e, x_e> / n_e;  Gx[e] = g (H_e / n_e - x_e <H_e, x_e> / n_e^3)
__global__ __launch_bounds__(256) void rownorm_sum_bwd_bwd_kernel(const float* __restrict__ x, const float* __restrict__ norms,
                                                                  const float* __restrict__ g, const float* __restrict__ H,
                                                                  float* __restrict__ Gx, float* __restrict__ Gg, int E, int n) {
    __shared__ float sd[1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float gv = *g;
    for (int e = wave; e < E; e += 4) {
        const float nm = norms[e];
        float d = 0.f;
        for (int j = lane; j < n; j += 64) d += H[(int64_t)e * n + j] * x[(int64_t)e * n + j];
        d = rn_wave_sum(d);
        const float inv = nm > 0.f ? 1.f / nm : 0.f;
        for (int j = lane; j < n; j += 64) {
            const int64_t k = (int64_t)e * n + j;
            Gx[k] = gv * (H[k] * inv - x[k] * d * inv * inv * inv);
        }
        if (lane == 0) sd[e] = d * inv;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int e = 0; e < E; ++e) t += sd[e];
        *Gg = t;
    }
}
extern "C" int ix_rownorm_sum_f32(const float* x, float* norms, float* total, int E, int n, hipStream_t stream) {
    IX_CHECK_ARG(x && norms && total && E > 0 && E <= 1024 && n > 0, "ix_rownorm_sum_f32: bad args (1..1024 rows)");
    hipLaunchKernelGGL(rownorm_sum_kernel, dim3(1), dim3(256), 0, stream, x, norms, total, E, n);
    IX_CHECK_LAUNCH("ix_rownorm_sum_f32");
    return IX_OK;
}
extern "C" int ix_rownorm_sum_bwd_f32(const float* x, const float* norms, const float* g, float* gx, int E, int n, hipStream_t stream) {
    IX_CHECK_ARG(x && norms && g && gx && E > 0 && n > 0, "ix_rownorm_sum_bwd_f32: bad args");
    hipLaunchKernelGGL(rownorm_sum_bwd_kernel, dim3(ix_grid_1d((int64_t)E * n, 256)), dim3(256), 0, stream, x, norms, g, gx, E, n);
    IX_CHECK_LAUNCH("ix_rownorm_sum_bwd_f32");
    return IX_OK;
}
extern "C" int ix_rownorm_sum_bwd_bwd_f32(const float* x, const float* norms, const float* g, const float* H, float* Gx, float* Gg,
                                          int E, int n, hipStream_t stream) {
    IX_CHECK_ARG(x && norms && g && H && Gx && Gg && E > 0 && E <= 1024 && n > 0, "ix_rownorm_sum_bwd_bwd_f32: bad args (1..1024 rows)");
    hipLaunchKernelGGL(rownorm_sum_bwd_bwd_kernel, dim3(1), dim3(256), 0, stream, x, norms, g, H, Gx, Gg, E, n);
    IX_CHECK_LAUNCH("ix_rownorm_sum_bwd_bwd_f32");
    return IX_OK;
}

// ---- FrozenBatchNorm2d folded into a per-channel affine, NHWC (channel = fastest dim) ------------------------
__global__ void bn_fold_kernel(const float* w, const float* b, const float* rm, const float* rv, float* scale,
                               float* shift, int C, float eps) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float s = w[c] * rsqrtf(rv[c] + eps);
    scale[c] = s;
    shift[c] = b[c] - rm[c] * s;
}

// scale = w * rsqrt(var + eps), shift = b - mean * scale   (reference backbone.py:44-54)
extern "C" int ix_bn_fold_f32(const float* w, const float* b, const float* rm, const float* rv, float* scale,
                              float* shift, int C, float eps, hipStream_t stream) {
    if (C <= 0) return IX_OK;
    IX_CHECK_ARG(w && b && rm && rv && scale && shift, "ix_bn_fold_f32: null pointer");
    hipLaunchKernelGGL(bn_fold_kernel, dim3(ix_div_up(C, 256)), dim3(256), 0, stream, w, b, rm, rv, scale, shift, C,
                       eps);
    IX_CHECK_LAUNCH("ix_bn_fold_f32");
    return IX_OK;
}

template <bool RELU, bool RES, bool SHIFT>
__global__ void channel_affine_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                      const float* __restrict__ shift, const float* __restrict__ res,
                                      float* __restrict__ o, int64_t n4, int C4) {
    // C % 4 == 0: each thread owns one float4 = 4 consecutive channels
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < n4; k += stride) {
        const int c4 = (int)(k % C4);
        float4 v = reinterpret_cast<const float4*>(x)[k];
        const float4 s = reinterpret_cast<const float4*>(scale)[c4];
        v.x *= s.x; v.y *= s.y; v.z *= s.z; v.w *= s.w;
        if (SHIFT) {
            const float4 t = reinterpret_cast<const float4*>(shift)[c4];
            v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
        }
        if (RES) {
            const float4 r = reinterpret_cast<const float4*>(res)[k];
            v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
        }
        if (RELU) {
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
        reinterpret_cast<float4*>(o)[k] = v;
    }
}

// out = [y > 0] * g * scale[c]: backward of relu(x*scale + shift) w.r.t. x in one pass (NHWC, C % 4 == 0)
__global__ void relu_bwd_channel_scale_kernel(const float* __restrict__ g, const float* __restrict__ y,
                                              const float* __restrict__ scale, float* __restrict__ o, int64_t n4, int C4) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < n4; k += stride) {
        const float4 v = reinterpret_cast<const float4*>(g)[k], yy = reinterpret_cast<const float4*>(y)[k];
        const float4 s = reinterpret_cast<const float4*>(scale)[(int)(k % C4)];
        reinterpret_cast<float4*>(o)[k] = make_float4(yy.x > 0.f ? v.x * s.x : 0.f, yy.y > 0.f ? v.y * s.y : 0.f,
                                                      yy.z > 0.f ? v.z * s.z : 0.f, yy.w > 0.f ? v.w * s.w : 0.f);
    }
}

extern "C" int ix_relu_bwd_channel_scale_f32(const float* g, const float* y, const float* scale, float* out, int64_t n, int C,
                                             hipStream_t stream) {
    if (n <= 0) return IX_OK;
    IX_CHECK_ARG(g && y && scale && out, "ix_relu_bwd_channel_scale_f32: null pointer");
    IX_CHECK_ARG(C % 4 == 0 && n % C == 0, "ix_relu_bwd_channel_scale_f32: need C %% 4 == 0 and n %% C == 0 (C=%d)", C);
    IX_CHECK_ARG(al16(g) && al16(y) && al16(out) && al16(scale), "ix_relu_bwd_channel_scale_f32: pointers must be 16-byte aligned");
    const int64_t n4 = n / 4;
    hipLaunchKernelGGL(relu_bwd_channel_scale_kernel, dim3(ix_grid_1d(n4, EW_BLOCK)), dim3(EW_BLOCK), 0, stream, g, y, scale,
                       out, n4, C / 4);
    IX_CHECK_LAUNCH("ix_relu_bwd_channel_scale_f32");
    return IX_OK;
}

// out[o][n][r] = x[o][n][r] * scale[n]: the frozen-BN scale on the WEIGHT side (w [(E,) N, R] with R = K or KH*KW*Cin, R % 4 == 0).
// hipops.RowScale: the backward of a convolution + frozen BN multiplies the weights (and the weight gradient) by the scale
// instead of the activation-sized gradient -- a tenth of the bytes even with per-episode weights.
__global__ __launch_bounds__(64) void row_scale_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                       float* __restrict__ o, int64_t rows, int N, int R4) {
    // one wave per weight row (o, n): no per-element index arithmetic, the row's scale read once
    for (int64_t row = blockIdx.x; row < rows; row += gridDim.x) {
        const float s = scale[(int)(row % N)];
        const float4* src = reinterpret_cast<const float4*>(x) + row * R4;
        float4* dst = reinterpret_cast<float4*>(o) + row * R4;
        for (int k = threadIdx.x; k < R4; k += 64) {
            float4 v = src[k];
            v.x *= s; v.y *= s; v.z *= s; v.w *= s;
            dst[k] = v;
        }
    }
}
extern "C" int ix_row_scale_f32(const float* x, const float* scale, float* out, int64_t outer, int N, int64_t R, hipStream_t stream) {
    if (outer <= 0 || N <= 0 || R <= 0) return IX_OK;
    IX_CHECK_ARG(x && scale && out && R % 4 == 0 && R / 4 < (1 << 30), "ix_row_scale_f32: null pointer or row length %lld not a multiple of 4", (long long)R);
    IX_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)out & 15) == 0, "ix_row_scale_f32: pointers must be 16-byte aligned");
    const int64_t rows = outer * N;
    hipLaunchKernelGGL(row_scale_kernel, dim3((unsigned)(rows < (1 << 20) ? rows : (1 << 20))), dim3(64), 0, stream, x, scale, out, rows, N,
                       (int)(R / 4));
    IX_CHECK_LAUNCH("ix_row_scale_f32");
    return IX_OK;
}

// out = [relu]( x*scale[c] (+ shift[c]) (+ residual) ), x NHWC with C % 4 == 0.  shift / residual may be null.
extern "C" int ix_channel_affine_f32(const float* x, const float* scale, const float* shift, const float* residual,
                                     float* out, int64_t n, int C, int relu, hipStream_t stream) {
    if (n <= 0) return IX_OK;
    IX_CHECK_ARG(x && scale && out, "ix_channel_affine_f32: null pointer");
    IX_CHECK_ARG(C % 4 == 0 && n % C == 0, "ix_channel_affine_f32: need C %% 4 == 0 and n %% C == 0 (C=%d)", C);
    IX_CHECK_ARG(al16(x) && al16(out) && al16(scale) && (!shift || al16(shift)) && (!residual || al16(residual)),
                 "ix_channel_affine_f32: pointers must be 16-byte aligned");
    const int64_t n4 = n / 4;
    dim3 g(ix_grid_1d(n4, EW_BLOCK)), b(EW_BLOCK);
#define CA(R, S, H) hipLaunchKernelGGL((channel_affine_kernel<R, S, H>), g, b, 0, stream, x, scale, shift, residual, out, n4, C / 4)
    const bool has_shift = shift != nullptr;
    if (relu) {
        if (residual) { if (has_shift) CA(true, true, true); else CA(true, true, false); }
        else { if (has_shift) CA(true, false, true); else CA(true, false, false); }
    } else {
        if (residual) { if (has_shift) CA(false, true, true); else CA(false, true, false); }
        else { if (has_shift) CA(false, false, true); else CA(false, false, false); }
    }
#undef CA
    IX_CHECK_LAUNCH("ix_channel_affine_f32");
    return IX_OK;
}
