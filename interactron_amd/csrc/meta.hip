// MAML fast-weight update and the outer optimiser step as fused multi-tensor / flat-buffer kernels.
//   fast = theta - clamp(lr * g, -clip, +clip)            reference utils/meta_utils.py:135-142
//   clip_grad_norm_ + Adam                                reference engine/interactron_trainer.py:107-111
// The 199 adapted tensors are updated by a handful of launches (pointer tables travel in the kernel-argument
// segment), so the adapt loop never round-trips to the host.
#include "common.h"

#define MT_MAX 48  // tensors per launch: 48 * (3 ptr + size + block offset) stays well inside the 4 KB kernarg limit
#define MT_CHUNK 4096  // elements per block

struct MultiArgs {
    const float* a[MT_MAX];
    const float* b[MT_MAX];
    float* o[MT_MAX];
    int64_t n[MT_MAX];
    int block_start[MT_MAX + 1];
    int count;
};

template <typename F>
__global__ __launch_bounds__(256) void multi_map2_kernel(MultiArgs m, F f) {
    // locate the tensor this block belongs to (count <= 48: linear scan in SGPRs)
    int t = 0;
    while (t + 1 < m.count && (int)blockIdx.x >= m.block_start[t + 1]) ++t;
    const int64_t base = (int64_t)(blockIdx.x - m.block_start[t]) * MT_CHUNK;
    const int64_t end = base + MT_CHUNK < m.n[t] ? base + MT_CHUNK : m.n[t];
    const float* a = m.a[t];
    const float* b = m.b[t];
    float* o = m.o[t];
    // (16-byte accesses where the three tensors allow it -- they do for every parameter of the detector: at 16 episodes this
    //  kernel moves 7 GB per step, and with 4-byte accesses it did so at 2.6 TB/s)
    if (((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(o)) & 15) == 0) {
        const int64_t end4 = base + ((end - base) & ~(int64_t)3);
        for (int64_t k = base + 4 * threadIdx.x; k < end4; k += 1024) {
            const float4 x = *reinterpret_cast<const float4*>(a + k), y = *reinterpret_cast<const float4*>(b + k);
            *reinterpret_cast<float4*>(o + k) = make_float4(f(x.x, y.x), f(x.y, y.y), f(x.z, y.z), f(x.w, y.w));
        }
        for (int64_t k = end4 + threadIdx.x; k < end; k += 256) o[k] = f(a[k], b[k]);
        return;
    }
    for (int64_t k = base + threadIdx.x; k < end; k += 256) o[k] = f(a[k], b[k]);
}

template <typename F>
static int launch_multi(const char* name, const float* const* a, const float* const* b, float* const* o,
                        const int64_t* sizes, int ntensors, hipStream_t stream, F f) {
    IX_CHECK_ARG(ntensors >= 0 && (ntensors == 0 || (a && b && o && sizes)), "%s: bad args", name);
    int i = 0;
    while (i < ntensors) {
        MultiArgs m;
        int cnt = 0, blocks = 0;
        while (i < ntensors && cnt < MT_MAX) {
            if (sizes[i] > 0) {
                IX_CHECK_ARG(a[i] && b[i] && o[i], "%s: null tensor %d", name, i);
                m.a[cnt] = a[i]; m.b[cnt] = b[i]; m.o[cnt] = o[i]; m.n[cnt] = sizes[i];
                m.block_start[cnt] = blocks;
                blocks += (int)((sizes[i] + MT_CHUNK - 1) / MT_CHUNK);
                ++cnt;
            }
            ++i;
        }
        if (cnt == 0) break;
        m.block_start[cnt] = blocks;
        m.count = cnt;
        hipLaunchKernelGGL(multi_map2_kernel, dim3(blocks), dim3(256), 0, stream, m, f);
        IX_CHECK_LAUNCH(name);
    }
    return IX_OK;
}

// Episode expansion of the fast weights and its adjoint, all tensors of a parameter list in one launch set
// (reference models/interactron.py:86-90: theta_task = clone(theta) per task; here the E copies of a chunk):
//   expand:  out_i[e * n_i + k] = src_i[k]            (e < E)        m.n = E * n_i output elements
//   reduce:  out_i[k] = sum_e src_i[e * n_i + k]                     m.n = n_i output elements (fixed order: deterministic)
__global__ __launch_bounds__(256) void multi_expand_kernel(MultiArgs m, int E) {
    int t = 0;
    while (t + 1 < m.count && (int)blockIdx.x >= m.block_start[t + 1]) ++t;
    const int64_t base = (int64_t)(blockIdx.x - m.block_start[t]) * MT_CHUNK;
    const int64_t end = base + MT_CHUNK < m.n[t] ? base + MT_CHUNK : m.n[t];
    const int64_t n = m.n[t] / E;
    const float* a = m.a[t];
    float* o = m.o[t];
    if (((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(o)) & 15) == 0 && (n & 3) == 0) {
        // 16-byte copies; the source index wraps by subtraction (one 64-bit remainder per thread instead of one per element)
        int64_t k = base + 4 * threadIdx.x, sidx = k % n;
        for (; k < end; k += 1024) {
            *reinterpret_cast<float4*>(o + k) = *reinterpret_cast<const float4*>(a + sidx);
            sidx += 1024;
            while (sidx >= n) sidx -= n;
        }
        return;
    }
    for (int64_t k = base + threadIdx.x; k < end; k += 256) o[k] = a[k % n];
}

__global__ __launch_bounds__(256) void multi_reduce_kernel(MultiArgs m, int E) {
    int t = 0;
    while (t + 1 < m.count && (int)blockIdx.x >= m.block_start[t + 1]) ++t;
    const int64_t base = (int64_t)(blockIdx.x - m.block_start[t]) * MT_CHUNK;
    const int64_t n = m.n[t];
    const int64_t end = base + MT_CHUNK < n ? base + MT_CHUNK : n;
    const float* a = m.a[t];
    float* o = m.o[t];
    if (((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(o)) & 15) == 0 && (n & 3) == 0) {   // (same sums, same order)
        for (int64_t k = base + 4 * threadIdx.x; k < end; k += 1024) {
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int e = 0; e < E; ++e) {
                const float4 x = *reinterpret_cast<const float4*>(a + (int64_t)e * n + k);
                s.x += x.x; s.y += x.y; s.z += x.z; s.w += x.w;
            }
            *reinterpret_cast<float4*>(o + k) = s;
        }
        return;
    }
    for (int64_t k = base + threadIdx.x; k < end; k += 256) {
        float s = 0.f;
        for (int e = 0; e < E; ++e) s += a[(int64_t)e * n + k];
        o[k] = s;
    }
}

// sizes[i] = elements of ONE copy (n_i); expand: src [n_i] -> out [E, n_i]; reduce: src [E, n_i] -> out [n_i]
static int launch_multi_e(const char* name, bool expand, const float* const* src, float* const* out, const int64_t* sizes,
                          int ntensors, int E, hipStream_t stream) {
    IX_CHECK_ARG(ntensors >= 0 && E >= 1 && (ntensors == 0 || (src && out && sizes)), "%s: bad args", name);
    int i = 0;
    while (i < ntensors) {
        MultiArgs m;
        int cnt = 0, blocks = 0;
        while (i < ntensors && cnt < MT_MAX) {
            if (sizes[i] > 0) {
                IX_CHECK_ARG(src[i] && out[i], "%s: null tensor %d", name, i);
                const int64_t n_out = expand ? sizes[i] * E : sizes[i];
                m.a[cnt] = src[i]; m.b[cnt] = nullptr; m.o[cnt] = out[i]; m.n[cnt] = n_out;
                m.block_start[cnt] = blocks;
                blocks += (int)((n_out + MT_CHUNK - 1) / MT_CHUNK);
                ++cnt;
            }
            ++i;
        }
        if (cnt == 0) break;
        m.block_start[cnt] = blocks;
        m.count = cnt;
        if (expand)
            hipLaunchKernelGGL(multi_expand_kernel, dim3(blocks), dim3(256), 0, stream, m, E);
        else
            hipLaunchKernelGGL(multi_reduce_kernel, dim3(blocks), dim3(256), 0, stream, m, E);
        IX_CHECK_LAUNCH(name);
    }
    return IX_OK;
}

extern "C" int ix_expand_multi_f32(const float* const* src, float* const* out, const int64_t* sizes, int ntensors, int E,
                                   hipStream_t stream) {
    return launch_multi_e("ix_expand_multi_f32", true, src, out, sizes, ntensors, E, stream);
}

extern "C" int ix_reduce_multi_f32(const float* const* src, float* const* out, const int64_t* sizes, int ntensors, int E,
                                   hipStream_t stream) {
    return launch_multi_e("ix_reduce_multi_f32", false, src, out, sizes, ntensors, E, stream);
}

// out_i = p_i - clamp(lr*g_i, -clip, clip) for every tensor i (host arrays of device pointers)
extern "C" int ix_sgd_clip_multi_f32(const float* const* p, const float* const* g, float* const* out,
                                     const int64_t* sizes, int ntensors, float lr, float clip, hipStream_t stream) {
    return launch_multi("ix_sgd_clip_multi_f32", p, g, out, sizes, ntensors, stream,
                        [=] __device__(float pv, float gv) { return pv - fminf(fmaxf(lr * gv, -clip), clip); });
}

// Adjoint w.r.t. g of the update above: out_i = -lr * G_i * [ |lr*g_i| <= clip ]   (torch.clip passes the gradient
// on the closed interval)
extern "C" int ix_sgd_clip_bwd_multi_f32(const float* const* G, const float* const* g, float* const* out,
                                         const int64_t* sizes, int ntensors, float lr, float clip,
                                         hipStream_t stream) {
    return launch_multi("ix_sgd_clip_bwd_multi_f32", G, g, out, sizes, ntensors, stream, [=] __device__(float Gv, float gv) {
        const float s = lr * gv;
        return (s >= -clip && s <= clip) ? -lr * Gv : 0.f;
    });
}

// ---- flat-buffer outer step ------------------------------------------------------------------------------
__global__ void sumsq_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ out, float* __restrict__ part,
                             unsigned int* tickets) {
    __shared__ float red[4];
    float s = 0.f;
    const int64_t gs = (int64_t)gridDim.x * blockDim.x;
    const int64_t n4 = n >> 2;
    for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < n4; k += gs) {
        const float4 v = reinterpret_cast<const float4*>(x)[k];
        s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    for (int64_t k = (n4 << 2) + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < n; k += gs) s += x[k] * x[k];
    s = ix_block_sum_256(s, red);
    if (gridDim.x == 1) {
        if (threadIdx.x == 0) *out += s;
        return;
    }
    if (threadIdx.x == 0) ix_store_agent(part + blockIdx.x, s);
    if (!ix_last_block(tickets, gridDim.x)) return;   // the last workgroup adds the partials in a fixed order
    float t = 0.f;
    for (unsigned int i = threadIdx.x; i < gridDim.x; i += 256) t += ix_load_agent(part + i);
    t = ix_block_sum_256(t, red);
    if (threadIdx.x == 0) *out += t;
}

// out[0] += sum x^2   (caller zeroes out; lets several buffers accumulate into one total norm).  workspace:
// IX_TICKET_BYTES of tickets (zero on entry, left zero) + 4 KiB of partials -- ordered, run-to-run identical sum.
extern "C" int ix_sumsq_accum_f32(const float* x, int64_t n, float* out, void* workspace, size_t workspace_bytes,
                                  hipStream_t stream) {
    if (n <= 0) return IX_OK;
    IX_CHECK_ARG(x && out && ((uintptr_t)x & 15) == 0, "ix_sumsq_accum_f32: bad args (x must be 16-byte aligned)");
    int g = ix_grid_1d((n + 3) / 4, 256);
    if (g > 1024) g = 1024;
    if (g > 1)
        IX_CHECK_ARG(workspace && workspace_bytes >= IX_TICKET_BYTES + 4096 && ix_al16(workspace),
                     "ix_sumsq_accum_f32: workspace of %d bytes needed", IX_TICKET_BYTES + 4096);
    hipLaunchKernelGGL(sumsq_kernel, dim3(g), dim3(256), 0, stream, x, n, out,
                       g > 1 ? reinterpret_cast<float*>(static_cast<char*>(workspace) + IX_TICKET_BYTES) : nullptr,
                       static_cast<unsigned int*>(workspace));
    IX_CHECK_LAUNCH("ix_sumsq_accum_f32");
    return IX_OK;
}

// Adam with the clip_grad_norm_ coefficient folded in: c = min(1, max_norm / (sqrt(sumsq[0]) + 1e-6)), g' = c*g
// (torch.nn.utils.clip_grad_norm_ + torch.optim.Adam, no weight decay, no amsgrad).  Grad buffer is left untouched
// except when zero_grad != 0 (then it is cleared for the next accumulation).
__global__ void adam_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            int64_t n, float lr, float b1, float b2, float eps, float bc1, float bc2,
                            const float* __restrict__ sumsq, float max_norm, int zero_grad) {
    float c = 1.f;
    if (sumsq) {
        const float cc = max_norm / (sqrtf(sumsq[0]) + 1e-6f);
        c = cc < 1.f ? cc : 1.f;
    }
    const int64_t gs = (int64_t)gridDim.x * blockDim.x;
    for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < n; k += gs) {
        const float gg = g[k] * c;
        const float mm = b1 * m[k] + (1.f - b1) * gg;
        const float vv = b2 * v[k] + (1.f - b2) * gg * gg;
        m[k] = mm;
        v[k] = vv;
        const float denom = sqrtf(vv) / sqrtf(bc2) + eps;
        p[k] -= (lr / bc1) * (mm / denom);
        if (zero_grad) g[k] = 0.f;
    }
}

extern "C" int ix_adam_step_f32(float* p, float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                                float eps, int step, const float* sumsq, float max_norm, int zero_grad,
                                hipStream_t stream) {
    if (n <= 0) return IX_OK;
    IX_CHECK_ARG(p && g && m && v && step >= 1, "ix_adam_step_f32: bad args");
    const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
    hipLaunchKernelGGL(adam_kernel, dim3(ix_grid_1d(n, 256)), dim3(256), 0, stream, p, g, m, v, n, lr, beta1, beta2,
                       eps, bc1, bc2, sumsq, max_norm, zero_grad);
    IX_CHECK_LAUNCH("ix_adam_step_f32");
    return IX_OK;
}
