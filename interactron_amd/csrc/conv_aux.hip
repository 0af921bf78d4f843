// Gather/scatter halves of the ResNet-50-DC5 convolutions (reference models/detr_models/backbone.py:88-90 ->
// torchvision resnet50, v1.5 bottlenecks, layer4 dilated).  Activations are NHWC so the GEMM contraction
// dimension (kh, kw, cin) is contiguous for both the patch matrix and the permuted weights:
//     conv(x, W) = im2col(x) [n*OH*OW, KH*KW*C]  x  W_krsc^T [KH*KW*C, Cout]      (ix_gemm_f32)
// im2col and col2im are exact adjoints, so the data-gradient, the weight-gradient and every second-order term of
// the MAML meta-gradient are again {im2col, col2im, GEMM} compositions.  1x1 stride-1 convolutions skip this file
// entirely (they are plain GEMMs over the NHWC tensor).
#include "common.h"

struct ConvGeom {
    int n, H, W, C;
    int KH, KW, stride, pad, dil;
    int OH, OW;
    int K;   // KH*KW*C
    int Kp;  // row pitch of the patch matrix (>= K, padded columns are zero-filled)
};

// One thread per 4 consecutive patch-matrix columns (never straddles a (kh,kw) block because C % 4 == 0).
__global__ void im2col_vec_kernel(const float* __restrict__ x, float* __restrict__ cols, ConvGeom g, int64_t total4) {
    const int K4 = g.Kp >> 2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total4; t += stride) {
        const int64_t row = t / K4;
        const int col = (int)(t % K4) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (col < g.K) {
            const int ci = col % g.C, kk = col / g.C;
            const int kw = kk % g.KW, kh = kk / g.KW;
            const int ow = (int)(row % g.OW);
            const int64_t r2 = row / g.OW;
            const int oh = (int)(r2 % g.OH), b = (int)(r2 / g.OH);
            const int ih = oh * g.stride - g.pad + kh * g.dil, iw = ow * g.stride - g.pad + kw * g.dil;
            if (ih >= 0 && ih < g.H && iw >= 0 && iw < g.W)
                v = *reinterpret_cast<const float4*>(x + (((int64_t)b * g.H + ih) * g.W + iw) * g.C + ci);
        }
        reinterpret_cast<float4*>(cols)[t] = v;
    }
}

// Scalar variant with arbitrary input strides (used for the NCHW 3-channel stem input).
__global__ void im2col_scalar_kernel(const float* __restrict__ x, float* __restrict__ cols, ConvGeom g, int64_t sxn,
                                     int64_t sxh, int64_t sxw, int64_t sxc, int64_t total) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += stride) {
        const int64_t row = t / g.Kp;
        const int col = (int)(t % g.Kp);
        float v = 0.f;
        if (col < g.K) {
            const int ci = col % g.C, kk = col / g.C;
            const int kw = kk % g.KW, kh = kk / g.KW;
            const int ow = (int)(row % g.OW);
            const int64_t r2 = row / g.OW;
            const int oh = (int)(r2 % g.OH), b = (int)(r2 / g.OH);
            const int ih = oh * g.stride - g.pad + kh * g.dil, iw = ow * g.stride - g.pad + kw * g.dil;
            if (ih >= 0 && ih < g.H && iw >= 0 && iw < g.W) v = x[b * sxn + ih * sxh + iw * sxw + ci * sxc];
        }
        cols[t] = v;
    }
}

// The same gather, four consecutive patch-matrix columns per thread (Kp % 4 == 0, fewer than 2^31 float4 chunks): the row is
// decomposed once per chunk in 32-bit arithmetic and (ci, kw, kh) advance by carries -- the one-element-per-thread form spends
// ~150 instructions (five divisions, two of them 64-bit) per 4 bytes written (the 3-channel NCHW stem: 1.0 ms per 300^2 step).
__global__ void im2col_scalar4_kernel(const float* __restrict__ x, float* __restrict__ cols, ConvGeom g, int64_t sxn,
                                      int64_t sxh, int64_t sxw, int64_t sxc, int total4) {
    const int K4 = g.Kp >> 2;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total4; t += gridDim.x * blockDim.x) {
        const int row = t / K4, col0 = (t - row * K4) * 4;
        const int ow = row % g.OW, r2 = row / g.OW;
        const int oh = r2 % g.OH, b = r2 / g.OH;
        int kk = col0 / g.C, ci = col0 - kk * g.C;
        int kh = kk / g.KW, kw = kk - kh * g.KW;
        const int ih0 = oh * g.stride - g.pad, iw0 = ow * g.stride - g.pad;
        const float* xb = x + b * sxn;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ih = ih0 + kh * g.dil, iw = iw0 + kw * g.dil;
            v[j] = (col0 + j < g.K && ih >= 0 && ih < g.H && iw >= 0 && iw < g.W) ? xb[ih * sxh + iw * sxw + ci * sxc] : 0.f;
            if (++ci == g.C) { ci = 0; if (++kw == g.KW) { kw = 0; ++kh; } }
        }
        reinterpret_cast<float4*>(cols)[t] = make_float4(v[0], v[1], v[2], v[3]);
    }
}

static int fill_geom(ConvGeom& g, int n, int H, int W, int C, int KH, int KW, int stride, int pad, int dil, int Kp) {
    g.n = n; g.H = H; g.W = W; g.C = C; g.KH = KH; g.KW = KW; g.stride = stride; g.pad = pad; g.dil = dil;
    g.OH = (H + 2 * pad - dil * (KH - 1) - 1) / stride + 1;
    g.OW = (W + 2 * pad - dil * (KW - 1) - 1) / stride + 1;
    g.K = KH * KW * C;
    g.Kp = Kp;
    return (g.OH > 0 && g.OW > 0 && Kp >= g.K) ? 0 : -1;
}

// cols[(b,oh,ow), (kh,kw,ci)] = x[b, oh*s-p+kh*d, ow*s-p+kw*d, ci] (0 outside), x addressed with element strides
// (sxn, sxh, sxw, sxc); cols row pitch Kp >= KH*KW*C, pad columns zeroed.
extern "C" int ix_im2col_f32(const float* x, float* cols, int n, int H, int W, int C, int64_t sxn, int64_t sxh,
                             int64_t sxw, int64_t sxc, int KH, int KW, int stride, int pad, int dil, int Kp,
                             hipStream_t stream) {
    ConvGeom g;
    IX_CHECK_ARG(x && cols && n >= 0 && stride > 0 && dil > 0, "ix_im2col_f32: bad args");
    IX_CHECK_ARG(fill_geom(g, n, H, W, C, KH, KW, stride, pad, dil, Kp) == 0, "ix_im2col_f32: bad geometry");
    const int64_t rows = (int64_t)n * g.OH * g.OW;
    if (rows == 0) return IX_OK;
    const bool nhwc = sxc == 1 && sxw == C && sxh == (int64_t)W * C && sxn == (int64_t)H * W * C;
    const bool vec = nhwc && C % 4 == 0 && Kp % 4 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)cols & 15) == 0;
    if (vec) {
        const int64_t total4 = rows * (Kp / 4);
        hipLaunchKernelGGL(im2col_vec_kernel, dim3(ix_grid_1d(total4, 256)), dim3(256), 0, stream, x, cols, g, total4);
    } else if (Kp % 4 == 0 && ((uintptr_t)cols & 15) == 0 && rows * (Kp / 4) < ((int64_t)1 << 31) - 65536 * 256) {
        const int total4 = (int)(rows * (Kp / 4));
        hipLaunchKernelGGL(im2col_scalar4_kernel, dim3(ix_grid_1d(total4, 256)), dim3(256), 0, stream, x, cols, g, sxn, sxh, sxw, sxc,
                           total4);
    } else {
        const int64_t total = rows * Kp;
        hipLaunchKernelGGL(im2col_scalar_kernel, dim3(ix_grid_1d(total, 256)), dim3(256), 0, stream, x, cols, g, sxn,
                           sxh, sxw, sxc, total);
    }
    IX_CHECK_LAUNCH("ix_im2col_f32");
    return IX_OK;
}

// Adjoint of im2col in gather form (no atomics): every input pixel sums the <= KH*KW patch entries that read it.
template <int V>
__global__ void col2im_kernel(const float* __restrict__ cols, float* __restrict__ dx, ConvGeom g, int64_t total) {
    const int CV = g.C / V;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += stride) {
        const int ci = (int)(t % CV) * V;
        int64_t r = t / CV;
        const int iw = (int)(r % g.W);
        r /= g.W;
        const int ih = (int)(r % g.H), b = (int)(r / g.H);
        float acc[V];
#pragma unroll
        for (int j = 0; j < V; ++j) acc[j] = 0.f;
        for (int kh = 0; kh < g.KH; ++kh) {
            const int th = ih + g.pad - kh * g.dil;
            if (th < 0 || th % g.stride) continue;
            const int oh = th / g.stride;
            if (oh >= g.OH) continue;
            for (int kw = 0; kw < g.KW; ++kw) {
                const int tw = iw + g.pad - kw * g.dil;
                if (tw < 0 || tw % g.stride) continue;
                const int ow = tw / g.stride;
                if (ow >= g.OW) continue;
                const float* p = cols + (((int64_t)b * g.OH + oh) * g.OW + ow) * g.Kp + (kh * g.KW + kw) * g.C + ci;
                if (V == 4) {
                    const float4 v = *reinterpret_cast<const float4*>(p);
                    acc[0] += v.x; acc[1] += v.y; acc[2] += v.z; acc[3] += v.w;
                } else {
                    acc[0] += p[0];
                }
            }
        }
        float* o = dx + (((int64_t)b * g.H + ih) * g.W + iw) * g.C + ci;
        if (V == 4) *reinterpret_cast<float4*>(o) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        else o[0] = acc[0];
    }
}

// dx [n,H,W,C] (NHWC contiguous) = im2col^T(cols)
extern "C" int ix_col2im_f32(const float* cols, float* dx, int n, int H, int W, int C, int KH, int KW, int stride,
                             int pad, int dil, int Kp, hipStream_t stream) {
    ConvGeom g;
    IX_CHECK_ARG(cols && dx && n >= 0 && stride > 0 && dil > 0, "ix_col2im_f32: bad args");
    IX_CHECK_ARG(fill_geom(g, n, H, W, C, KH, KW, stride, pad, dil, Kp) == 0, "ix_col2im_f32: bad geometry");
    const int64_t px = (int64_t)n * H * W;
    if (px == 0) return IX_OK;
    const bool vec = C % 4 == 0 && Kp % 4 == 0 && ((uintptr_t)dx & 15) == 0 && ((uintptr_t)cols & 15) == 0;
    if (vec) {
        const int64_t total = px * (C / 4);
        hipLaunchKernelGGL(col2im_kernel<4>, dim3(ix_grid_1d(total, 256)), dim3(256), 0, stream, cols, dx, g, total);
    } else {
        const int64_t total = px * C;
        hipLaunchKernelGGL(col2im_kernel<1>, dim3(ix_grid_1d(total, 256)), dim3(256), 0, stream, cols, dx, g, total);
    }
    IX_CHECK_LAUNCH("ix_col2im_f32");
    return IX_OK;
}

// 3x3/s2/p1-style max pooling on NHWC (the ResNet stem; frozen region, forward only)
__global__ void maxpool_nhwc_kernel(const float* __restrict__ x, float* __restrict__ y, int n, int H, int W, int C,
                                    int k, int stride, int pad, int OH, int OW, int64_t total4) {
    const int C4 = C >> 2;
    const int64_t gs = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total4; t += gs) {
        const int c = (int)(t % C4) * 4;
        int64_t r = t / C4;
        const int ow = (int)(r % OW);
        r /= OW;
        const int oh = (int)(r % OH), b = (int)(r / OH);
        float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        for (int kh = 0; kh < k; ++kh) {
            const int ih = oh * stride - pad + kh;
            if (ih < 0 || ih >= H) continue;
            for (int kw = 0; kw < k; ++kw) {
                const int iw = ow * stride - pad + kw;
                if (iw < 0 || iw >= W) continue;
                const float4 v = *reinterpret_cast<const float4*>(x + (((int64_t)b * H + ih) * W + iw) * C + c);
                m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
            }
        }
        reinterpret_cast<float4*>(y)[t] = m;
    }
}

extern "C" int ix_maxpool_nhwc_f32(const float* x, float* y, int n, int H, int W, int C, int k, int stride, int pad,
                                   hipStream_t stream) {
    IX_CHECK_ARG(x && y && C % 4 == 0 && k > 0 && stride > 0, "ix_maxpool_nhwc_f32: bad args (C %% 4 must be 0)");
    const int OH = (H + 2 * pad - k) / stride + 1, OW = (W + 2 * pad - k) / stride + 1;
    const int64_t total4 = (int64_t)n * OH * OW * (C / 4);
    if (total4 <= 0) return IX_OK;
    hipLaunchKernelGGL(maxpool_nhwc_kernel, dim3(ix_grid_1d(total4, 256)), dim3(256), 0, stream, x, y, n, H, W, C, k,
                       stride, pad, OH, OW, total4);
    IX_CHECK_LAUNCH("ix_maxpool_nhwc_f32");
    return IX_OK;
}

// out[b, h, w, c] (NHWC) <-> in[b, c, h, w] (NCHW): layout change at the drop-in boundary
// (DETR returns image_features / embedded_memory_features as NCHW, reference detr.py:73-74).
__global__ void nhwc_to_nchw_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t HW, int C,
                                    int64_t total) {
    const int64_t gs = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += gs) {
        // t indexes the NCHW output: ((b*C + c)*HW + p)
        const int64_t p = t % HW;
        const int64_t r = t / HW;
        const int c = (int)(r % C);
        const int64_t b = r / C;
        y[t] = x[(b * HW + p) * C + c];
    }
}

extern "C" int ix_nhwc_to_nchw_f32(const float* x, float* y, int n, int64_t HW, int C, hipStream_t stream) {
    const int64_t total = (int64_t)n * HW * C;
    if (total <= 0) return IX_OK;
    IX_CHECK_ARG(x && y, "ix_nhwc_to_nchw_f32: null pointer");
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(ix_grid_1d(total, 256)), dim3(256), 0, stream, x, y, HW, C, total);
    IX_CHECK_LAUNCH("ix_nhwc_to_nchw_f32");
    return IX_OK;
}
