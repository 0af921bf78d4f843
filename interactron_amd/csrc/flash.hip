// Flash-style attention on the bf16 matrix cores with fp32-grade arithmetic: forward, backward and double backward of
//     O = dropout(softmax(scale Q K^T + key bias)) V        per (batch, head)
// without ever writing an [L, S] tensor to HBM.  Replaces the attention core of reference models/gpt.py:39-57
// (att = softmax(q k^T / sqrt(hd)); att = drop(att); y = att v) and of nn.MultiheadAttention as called from
// models/detr_models/transformer.py:148-161,211-232, plus the autograd derivatives the MAML meta-gradient takes of it
// (models/interactron.py:99-123: grad(create_graph=True) then backward).
//
// Arithmetic: fp32-grade on the 16-bit matrix cores, two exact operand splits.
//   * "tr" operands and every [L, S]-shaped intermediate (probabilities, score cotangents -- split in registers,
//     straight out of the MFMA accumulators): three bf16 values x = h + m + l (24 significant bits), a product is the
//     six terms  l.h + h.l + m.m + m.h + h.m + h.h  on v_mfma_f32_32x32x16_bf16 -- the scheme of the bf16x6 contraction
//     kernel in gemm.hip (bf16 has fp32's exponent range, so nothing needs scaling).
//   * "row" operands (the products that contract over the head dim: q k^T, dO v^T and their second-order cousins, both
//     operands straight from HBM): two fp16 values x 2^e = h + l (22 significant bits) with ONE power-of-two scale per
//     block of 32 rows, a product is the three terms  l.h + h.l + h.h  on v_mfma_f32_32x32x16_f16 (the dropped l.l is
//     2^-22 relative; fp16 subnormals are honoured by the MFMA, tools/micro/f16_denorm.hip), and the two block scales
//     are undone by one wave-uniform multiply of the accumulator.  Measured against float64 this is the accuracy class
//     of an fp32 dot product of the same length (tests/test_ops_gpu.py) at half the matrix instructions.
//   * tr_form 1 (the default since round 3): the second-stage products on the fp16 form too.  The tr planes then hold the
//     SAME scaled fp16 pairs as the row planes (transposed), and an [L, S] intermediate x is brought into fp16 range in
//     registers: x . (block unscale of the tr tile) . F with F a power of two that the owning lane keeps per ACCUMULATOR --
//     lowered (and the accumulator rescaled, exactly) whenever a tile's largest magnitude would leave [.., 2^15), never
//     raised -- and undone when the accumulator is stored.  An element then carries max(2^-22 |x|, 2^-39 max|x| of its output
//     row so far): the accuracy class of the first-stage products, at three matrix instructions instead of six and two
//     planes of LDS traffic instead of three (measured: -13 ... -22 % per kernel, profiles/README.md round 3).
// Operands that come from HBM are split ONCE by ix_attn_split_f32 into the planes the kernels consume.
//
// Layouts.  An [L, S] tile is always computed TRANSPOSED relative to its owner: a workgroup that owns query rows
// (forward, dQ-type outputs) computes T^T[key, query] = Kside[key, :] . Qside[query, :], so a lane of the 32x32 MFMA
// accumulator holds ONE query (column lane & 31) and 16 keys (rows (r & 3) + 8 (r >> 2) + 4 (lane >> 5)): softmax row
// statistics are in-lane reductions plus one exchange with lane ^ 32, and the accumulator registers ARE the B operand
// of the next product  Out^T[d, query] = X^T[d, key] . T^T[key, query]  once the keys of X^T are stored in the order the
// accumulator delivers them (within every 16 keys, bits 2 and 3 of the index swapped -- done once by the split kernel,
// "tr" layout).  Workgroups that own key rows (dK / dV-type outputs) use the mirror image (lane = key, registers =
// queries).  No [L, S] value ever goes through LDS.
//
//   row layout  [2 planes][batch*head][Rp][hd]   fp16   fragment of 8 consecutive d of one row  (contraction over d)
//               + unscale factors [batch*head][Rp / 32] f32 (2^-e of each 32-row block)
//   tr  layout  [3 planes][batch*head][hd][Rp]   bf16   rows permuted within 16-groups          (contraction over rows)
//        form 1 [2 planes][batch*head][hd][Rp]   fp16   the row planes' values, transposed and permuted likewise
//   Rp = R rounded up to 128 (zero rows); keys are walked in tiles of 32 up to ceil(S / 32) * 32.
#include "flash_common.h"

// ------------------------------------------------------------------------------------------------------------
// split kernel: fp32 [n][R][ld] (head h at columns off + h*hd) -> fp16 row planes (+ block unscale factors) and
// bf16 tr planes.  One workgroup = one block of 32 rows of one (batch, head).
// ------------------------------------------------------------------------------------------------------------
// Up to three operands per launch (blockIdx.z; the q, k, v -- or hq, hk, hv -- of one attention call: one launch instead of
// three in a step whose small-batch form is bound by its launch count).
struct SplitOp {
    const float* x;
    unsigned short *rowp, *trp;
    float* unscale;
    int R, Rp, off;
    int64_t ld, plane_elems;
    // optional ride-along (ix_attn_split_dot_f32): t[bh][row] = sum_d x[row, h, d] * y[row, h, d] (delta = dO . O of the backward
    // pass: the split of dO reads dO anyway), rows R..Rp written as 0
    const float* dot_y;
    float* dot_out;
    int64_t dot_ld;
    int dot_off;
};
struct SplitOps {
    SplitOp op[3];
};

// IN16: the operand is bf16 in HBM (the 16-bit activation mode: every value fits the h plane exactly, the l plane is zero)
template <int HD, bool IN16 = false>
__global__ __launch_bounds__(256) void attn_split_kernel(SplitOps ops, int H, int tr_form) {
    const SplitOp& o = ops.op[blockIdx.z];
    if ((int)blockIdx.x * 32 >= o.Rp) return;
    const float* __restrict__ X = o.x;
    unsigned short* __restrict__ rowp = o.rowp;
    unsigned short* __restrict__ trp = o.trp;
    float* __restrict__ unscale = o.unscale;
    const int R = o.R, Rp = o.Rp, off = o.off;
    const int64_t ld = o.ld, plane_elems = o.plane_elems;
    constexpr int EPT = 32 * HD / 256;          // elements per thread: 8 (hd 64) / 4 (hd 32)
    constexpr int TPR = HD / EPT;               // threads per row: 8
    __shared__ __attribute__((aligned(16))) unsigned short lt[3][HD][32 + 8];   // [plane][d][permuted row], 80-byte rows
    __shared__ float red[4];
    const int tid = threadIdx.x, r0 = blockIdx.x * 32, bh = blockIdx.y;
    const int b = bh / H, h = bh % H;
    const int row = tid / TPR, c0 = (tid % TPR) * EPT;
    float v[EPT];
    const bool in = r0 + row < R;
    if (IN16) {
        const unsigned short* src = reinterpret_cast<const unsigned short*>(X) + ((int64_t)b * R + r0 + row) * ld + off + h * HD + c0;
#pragma unroll
        for (int i = 0; i < EPT; i += 4) {
            uint2 t = make_uint2(0u, 0u);
            if (in) t = *reinterpret_cast<const uint2*>(src + i);
            v[i] = __uint_as_float(t.x << 16); v[i + 1] = __uint_as_float(t.x & 0xffff0000u);
            v[i + 2] = __uint_as_float(t.y << 16); v[i + 3] = __uint_as_float(t.y & 0xffff0000u);
        }
    } else {
        const float* src = X + ((int64_t)b * R + r0 + row) * ld + off + h * HD + c0;
#pragma unroll
        for (int i = 0; i < EPT; i += 4) {
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            if (in) t = *reinterpret_cast<const float4*>(src + i);
            v[i] = t.x; v[i + 1] = t.y; v[i + 2] = t.z; v[i + 3] = t.w;
        }
    }
    if (o.dot_y) {   // (block-uniform) the TPR threads of a row are consecutive lanes: partial dot products, then a butterfly
        float d = 0.f;
        if (in) {
            const float* ys = o.dot_y + ((int64_t)b * R + r0 + row) * o.dot_ld + o.dot_off + h * HD + c0;
#pragma unroll
            for (int i = 0; i < EPT; i += 4) {
                const float4 t = *reinterpret_cast<const float4*>(ys + i);
                d += v[i] * t.x + v[i + 1] * t.y + v[i + 2] * t.z + v[i + 3] * t.w;
            }
        }
#pragma unroll
        for (int m = 1; m < TPR; m <<= 1) d += __shfl_xor(d, m, 64);
        if ((tid % TPR) == 0) o.dot_out[(int64_t)bh * Rp + r0 + row] = d;
    }
    unsigned ph[EPT / 2], pm[EPT / 2], pl[EPT / 2];
    if (rowp || (trp && tr_form == 1)) {
        // block scale: the largest magnitude of the 32 x HD block lands in [2^14, 2^15) (fp16 overflows at 65504); the
        // scale itself stops at 2^60 (blocks whose maximum is below 2^-46 keep fewer bits): the kernels' running factors stop
        // at 2^60 too, so unscale . factor cannot underflow to zero
        float mx = 0.f;
#pragma unroll
        for (int i = 0; i < EPT; ++i) mx = fmaxf(mx, fabsf(v[i]));
        mx = ix_block_max_256(mx, red);
        unsigned e = (__float_as_uint(mx) >> 23) & 0xffu;                       // biased exponent of the maximum
        const bool odd = e > 250u;                                              // (inf / nan block: unscaled)
        // a zero (or nearly zero) block gets the SMALLEST unscale factor, not 1: the kernels size their running factors by
        // bound . unscale, and a block that contributes nothing must not drag them down for the blocks that do
        e = max(e, 127u + 14u - 60u);
        const float sc = odd ? 1.f : __uint_as_float((268u - e) << 23);         // 2^(14 - E)
        const float us = odd ? 1.f : __uint_as_float((e - 14u) << 23);          // 2^(E - 14)
        if (tid == 0) unscale[(int64_t)bh * (Rp / 32) + blockIdx.x] = us;
        unsigned short* dst = rowp + ((int64_t)bh * Rp + r0 + row) * HD + c0;
#pragma unroll
        for (int i = 0; i < EPT / 2; ++i) {
            fl_split2h(v[2 * i] * sc, v[2 * i + 1] * sc, ph[i], pm[i]);
            if (rowp) {
                reinterpret_cast<unsigned*>(dst)[i] = ph[i];
                reinterpret_cast<unsigned*>(dst + plane_elems)[i] = pm[i];
            }
        }
    }
    if (!trp) return;
    if (tr_form == 0) {
#pragma unroll
        for (int i = 0; i < EPT / 2; ++i) fl_split3(v[2 * i], v[2 * i + 1], ph[i], pm[i], pl[i]);
    } else {
#pragma unroll
        for (int i = 0; i < EPT / 2; ++i) pl[i] = 0u;
    }
    const int ntp = tr_form == 0 ? 3 : 2;
    const int prow = (row & 16) | fl_perm16(row & 15);
#pragma unroll
    for (int i = 0; i < EPT / 2; ++i) {
        lt[0][c0 + 2 * i][prow] = (unsigned short)(ph[i] & 0xffff); lt[0][c0 + 2 * i + 1][prow] = (unsigned short)(ph[i] >> 16);
        lt[1][c0 + 2 * i][prow] = (unsigned short)(pm[i] & 0xffff); lt[1][c0 + 2 * i + 1][prow] = (unsigned short)(pm[i] >> 16);
        lt[2][c0 + 2 * i][prow] = (unsigned short)(pl[i] & 0xffff); lt[2][c0 + 2 * i + 1][prow] = (unsigned short)(pl[i] >> 16);
    }
    __syncthreads();
    // tr layout: per plane HD rows of 32 16-bit values (64 bytes = four 16-byte chunks)
    for (int c = tid; c < ntp * HD * 4; c += 256) {
        const int pl_ = c / (HD * 4), d = (c / 4) % HD, ch = c % 4;
        const uint4 val = *reinterpret_cast<const uint4*>(&lt[pl_][d][ch * 8]);
        *reinterpret_cast<uint4*>(trp + pl_ * plane_elems + ((int64_t)bh * HD + d) * Rp + r0 + ch * 8) = val;
    }
}

static int fl_split_launch(const char* who, int count, const float* const* x, void* const* row_planes, float* const* unscale,
                           void* const* tr_planes, int tr_form, int n, const int* R, const int* Rp, const int64_t* ld,
                           const int* off, int H, int hd, hipStream_t stream, const float* dot_y = nullptr, int64_t dot_ld = 0,
                           int dot_off = 0, float* dot_out = nullptr, bool in16 = false) {
    IX_CHECK_ARG(count >= 1 && count <= 3, "%s: %d operands (1..3)", who, count);
    IX_CHECK_ARG(tr_form == 0 || tr_form == 1, "%s: tr_form %d (0 = three bf16 planes, 1 = two fp16 planes)", who, tr_form);
    IX_CHECK_ARG(hd == 32 || hd == 64, "%s: head dim %d (32 or 64)", who, hd);
    SplitOps ops;
    memset(&ops, 0, sizeof(ops));
    int maxRp = 0;
    for (int i = 0; i < count; ++i) {
        IX_CHECK_ARG(x[i] && (row_planes[i] || tr_planes[i]), "%s: null pointer (operand %d)", who, i);
        IX_CHECK_ARG(unscale[i] || !(row_planes[i] || (tr_planes[i] && tr_form == 1)),
                     "%s: fp16 planes (row, or tr of form 1) come with their block unscale factors", who);
        IX_CHECK_ARG(R[i] > 0 && Rp[i] % 128 == 0 && Rp[i] >= R[i], "%s: Rp=%d must be R=%d rounded up to 128", who, Rp[i], R[i]);
        IX_CHECK_ARG(ld[i] % 4 == 0 && off[i] % 4 == 0 && ((uintptr_t)x[i] & 15) == 0, "%s: rows must be 16-byte aligned", who);
        SplitOp& o = ops.op[i];
        o.x = x[i]; o.rowp = (unsigned short*)row_planes[i]; o.trp = (unsigned short*)tr_planes[i]; o.unscale = unscale[i];
        o.R = R[i]; o.Rp = Rp[i]; o.off = off[i]; o.ld = ld[i];
        o.plane_elems = (int64_t)n * H * Rp[i] * hd;
        maxRp = Rp[i] > maxRp ? Rp[i] : maxRp;
    }
    if (dot_y) {   // rides on operand 0
        IX_CHECK_ARG(dot_out && dot_ld % 4 == 0 && dot_off % 4 == 0 && ((uintptr_t)dot_y & 15) == 0, "%s: the dot operand's rows must be 16-byte aligned", who);
        ops.op[0].dot_y = dot_y; ops.op[0].dot_out = dot_out; ops.op[0].dot_ld = dot_ld; ops.op[0].dot_off = dot_off;
    }
    dim3 grid(maxRp / 32, n * H, count);
    if (in16) {
        if (hd == 64) hipLaunchKernelGGL((attn_split_kernel<64, true>), grid, dim3(256), 0, stream, ops, H, tr_form);
        else hipLaunchKernelGGL((attn_split_kernel<32, true>), grid, dim3(256), 0, stream, ops, H, tr_form);
    } else if (hd == 64)
        hipLaunchKernelGGL(attn_split_kernel<64>, grid, dim3(256), 0, stream, ops, H, tr_form);
    else
        hipLaunchKernelGGL(attn_split_kernel<32>, grid, dim3(256), 0, stream, ops, H, tr_form);
    IX_CHECK_LAUNCH(who);
    return IX_OK;
}

extern "C" int ix_attn_split_f32(const float* x, void* row_planes, float* row_unscale, void* tr_planes, int tr_form, int n, int R,
                                 int Rp, int64_t ld, int off, int H, int hd, hipStream_t stream) {
    if (n <= 0 || R <= 0) return IX_OK;
    return fl_split_launch("ix_attn_split_f32", 1, &x, &row_planes, &row_unscale, &tr_planes, tr_form, n, &R, &Rp, &ld, &off, H, hd,
                           stream);
}

// split of x with t[bh][row] = sum_d x[row, h, d] y[row, h, d] riding along (rows R..Rp of t: 0) -- the backward pass's dO planes and
// delta = dO . O in one read of dO (ix_attn_split_f32 + ix_attn_rowdot_f32 otherwise)
extern "C" int ix_attn_split_dot_f32(const float* x, void* row_planes, float* row_unscale, void* tr_planes, int tr_form, int n, int R,
                                     int Rp, int64_t ld, int off, int H, int hd, const float* y, int64_t ldy, int offy, float* t,
                                     hipStream_t stream) {
    if (n <= 0 || R <= 0) return IX_OK;
    IX_CHECK_ARG(y && t, "ix_attn_split_dot_f32: null dot operand / output");
    return fl_split_launch("ix_attn_split_dot_f32", 1, &x, &row_planes, &row_unscale, &tr_planes, tr_form, n, &R, &Rp, &ld, &off, H, hd,
                           stream, y, ldy, offy, t);
}

// the same for up to three operands of one attention call in ONE launch (arrays of `count` entries)
extern "C" int ix_attn_split_multi_f32(int count, const float* const* x, void* const* row_planes, float* const* row_unscale,
                                       void* const* tr_planes, int tr_form, int n, const int* R, const int* Rp, const int64_t* ld,
                                       const int* off, int H, int hd, hipStream_t stream) {
    if (n <= 0 || count <= 0) return IX_OK;
    IX_CHECK_ARG(x && row_planes && row_unscale && tr_planes && R && Rp && ld && off, "ix_attn_split_multi_f32: null array");
    return fl_split_launch("ix_attn_split_multi_f32", count, x, row_planes, row_unscale, tr_planes, tr_form, n, R, Rp, ld, off, H, hd,
                           stream);
}

// ... and for bf16 operands (the 16-bit activation mode): x points to 2-byte elements, ld / off in elements (multiples of 4); the dot
// operand y of the single-operand form stays fp32 (the attention output as the forward kernel wrote it)
extern "C" int ix_attn_split_multi_b16(int count, const void* const* x, void* const* row_planes, float* const* row_unscale,
                                       void* const* tr_planes, int tr_form, int n, const int* R, const int* Rp, const int64_t* ld,
                                       const int* off, int H, int hd, hipStream_t stream) {
    if (n <= 0 || count <= 0) return IX_OK;
    IX_CHECK_ARG(x && row_planes && row_unscale && tr_planes && R && Rp && ld && off, "ix_attn_split_multi_b16: null array");
    return fl_split_launch("ix_attn_split_multi_b16", count, reinterpret_cast<const float* const*>(x), row_planes, row_unscale, tr_planes,
                           tr_form, n, R, Rp, ld, off, H, hd, stream, nullptr, 0, 0, nullptr, true);
}
extern "C" int ix_attn_split_dot_b16(const void* x, void* row_planes, float* row_unscale, void* tr_planes, int tr_form, int n, int R,
                                     int Rp, int64_t ld, int off, int H, int hd, const float* y, int64_t ldy, int offy, float* t,
                                     hipStream_t stream) {
    if (n <= 0 || R <= 0) return IX_OK;
    IX_CHECK_ARG(y && t, "ix_attn_split_dot_b16: null dot operand / output");
    const float* xf = reinterpret_cast<const float*>(x);
    return fl_split_launch("ix_attn_split_dot_b16", 1, &xf, &row_planes, &row_unscale, &tr_planes, tr_form, n, &R, &Rp, &ld, &off, H, hd,
                           stream, y, ldy, offy, t, true);
}

// additive key bias [n][Sp]: 0 for a valid key, -inf for a padded (mask != 0) key and for the tail S..Sp
__global__ void attn_bias_kernel(const uint8_t* __restrict__ mask, float* __restrict__ bias, int S, int Sp, int64_t mask_ld) {
    const int b = blockIdx.y;
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < Sp; c += gridDim.x * blockDim.x)
        bias[(int64_t)b * Sp + c] = (c < S && !(mask && mask[(int64_t)b * mask_ld + c])) ? 0.f : -INFINITY;
}

extern "C" int ix_attn_bias_f32(const uint8_t* mask, float* bias, int n, int S, int Sp, int64_t mask_ld, hipStream_t stream) {
    if (n <= 0 || Sp <= 0) return IX_OK;
    IX_CHECK_ARG(bias && Sp >= S, "ix_attn_bias_f32: bad args");
    hipLaunchKernelGGL(attn_bias_kernel, dim3(ix_div_up(Sp, 256), n), dim3(256), 0, stream, mask, bias, S, Sp, mask_ld);
    IX_CHECK_LAUNCH("ix_attn_bias_f32");
    return IX_OK;
}


// ------------------------------------------------------------------------------------------------------------
// common machinery of the six kernels
// ------------------------------------------------------------------------------------------------------------
// A tile of 32 rows as staged in LDS: "row" segments [2][32][HD + 8] fp16 (fragments along d) and "tr" segments
// [TP][HD][32 + 8] (TP = 3 bf16 planes, or 2 fp16 planes in tr form 1; fragments along the 32 rows); the 16-byte row padding keeps ds_read_b128 fragment reads
// conflict-free (row pitch / 16 odd).  One unit = one plane of one segment = 4 * HD 16-byte chunks.
template <int HD, int TP>
struct FlSeg {
    static constexpr int RROW = (HD + 8) * 2, RPLANE = 32 * RROW, RBYTES = 2 * RPLANE;
    static constexpr int TROW = (32 + 8) * 2, TPLANE = HD * TROW, TBYTES = TP * TPLANE;
    // chunk c of a unit: global element offset from the tile's first row / LDS byte offset inside the plane
    static __device__ __forceinline__ int64_t row_src(int c) { return (int64_t)c * 8; }
    static __device__ __forceinline__ int row_dst(int c) { return (c / (HD / 8)) * RROW + (c % (HD / 8)) * 16; }
    static __device__ __forceinline__ int64_t tr_src(int c, int Rp) { return (int64_t)(c >> 2) * Rp + (c & 3) * 8; }
    static __device__ __forceinline__ int tr_dst(int c) { return (c >> 2) * TROW + (c & 3) * 16; }
};

// Staging registers: twelve plain named variables per list (FL_DECL_REGS), visited by macro expansion with literal
// indices.  (Neither an indexed array nor a struct of registers survives here: the compiler keeps either in scratch
// memory as soon as a value is loaded before a barrier and stored to LDS after it, and a load whose destination is in
// scratch is waited for on the spot -- every staging load then costs a full memory round trip.)
#define FL_DECL_REGS(P) uint4 P##0, P##1, P##2, P##3, P##4, P##5, P##6, P##7, P##8, P##9, P##10, P##11;
// global -> staging registers (requested early) -> LDS (after the barrier that frees the destination) for the units
// FIRST .. FIRST + COUNT - 1 (COUNT <= 12) of a tile whose first NROW segments are row segments (2 units each) followed
// by tr segments (TP units each).  One chunk per thread and unit at HD 64, two units per pass at HD 32 (an odd unit out is
// copied twice).  SRC / DST are expressions in seg_ (segment), pl_ (plane), c_ (chunk).
#define FL_NREGS(COUNT) (HD == 64 ? (COUNT) : ((COUNT) + 1) / 2)
#define FL_UNIT_DECODE(I, FIRST, COUNT, NROW)                                                                          \
    const int u_ = (FIRST) + (HD == 64 ? (I) : min(2 * (I) + (tid >> 7), (COUNT) - 1));                                \
    const int c_ = HD == 64 ? tid : (tid & 127);                                                                       \
    const bool isrow_ = u_ < 2 * (NROW);                                                                               \
    const int seg_ = isrow_ ? u_ / 2 : (NROW) + (u_ - 2 * (NROW)) / TP, pl_ = isrow_ ? u_ % 2 : (u_ - 2 * (NROW)) % TP; \
    (void)isrow_;
#define FL_LOAD1(P, I, FIRST, COUNT, NROW, SRC)                                                                        \
    if ((I) < FL_NREGS(COUNT)) {                                                                                       \
        FL_UNIT_DECODE(I, FIRST, COUNT, NROW)                                                                          \
        P##I = *reinterpret_cast<const uint4*>(SRC);                                                                   \
    }
#define FL_STORE1(P, I, FIRST, COUNT, NROW, DST)                                                                       \
    if ((I) < FL_NREGS(COUNT)) {                                                                                       \
        FL_UNIT_DECODE(I, FIRST, COUNT, NROW)                                                                          \
        *reinterpret_cast<uint4*>(DST) = P##I;                                                                         \
    }
#define FL_STAGE_LOAD(P, FIRST, COUNT, NROW, SRC)                                                                      \
    FL_LOAD1(P, 0, FIRST, COUNT, NROW, SRC) FL_LOAD1(P, 1, FIRST, COUNT, NROW, SRC) FL_LOAD1(P, 2, FIRST, COUNT, NROW, SRC)   \
    FL_LOAD1(P, 3, FIRST, COUNT, NROW, SRC) FL_LOAD1(P, 4, FIRST, COUNT, NROW, SRC) FL_LOAD1(P, 5, FIRST, COUNT, NROW, SRC)   \
    FL_LOAD1(P, 6, FIRST, COUNT, NROW, SRC) FL_LOAD1(P, 7, FIRST, COUNT, NROW, SRC) FL_LOAD1(P, 8, FIRST, COUNT, NROW, SRC)   \
    FL_LOAD1(P, 9, FIRST, COUNT, NROW, SRC) FL_LOAD1(P, 10, FIRST, COUNT, NROW, SRC) FL_LOAD1(P, 11, FIRST, COUNT, NROW, SRC)
#define FL_STAGE_STORE(P, FIRST, COUNT, NROW, DST)                                                                     \
    FL_STORE1(P, 0, FIRST, COUNT, NROW, DST) FL_STORE1(P, 1, FIRST, COUNT, NROW, DST) FL_STORE1(P, 2, FIRST, COUNT, NROW, DST) \
    FL_STORE1(P, 3, FIRST, COUNT, NROW, DST) FL_STORE1(P, 4, FIRST, COUNT, NROW, DST) FL_STORE1(P, 5, FIRST, COUNT, NROW, DST) \
    FL_STORE1(P, 6, FIRST, COUNT, NROW, DST) FL_STORE1(P, 7, FIRST, COUNT, NROW, DST) FL_STORE1(P, 8, FIRST, COUNT, NROW, DST) \
    FL_STORE1(P, 9, FIRST, COUNT, NROW, DST) FL_STORE1(P, 10, FIRST, COUNT, NROW, DST) FL_STORE1(P, 11, FIRST, COUNT, NROW, DST)

// three-term product of one k-slice on fp16 planes (h, l): acc += A . B, smallest terms first
#define FL_MMA3(ACC, A, B) \
    ACC = fl_mfma_h(A[1], B[0], ACC); ACC = fl_mfma_h(A[0], B[1], ACC); ACC = fl_mfma_h(A[0], B[0], ACC);
// six-term product of one k-slice on bf16 planes (h, m, l)
#define FL_MMA6(ACC, A, B)                                                             \
    ACC = fl_mfma(A[2], B[0], ACC); ACC = fl_mfma(A[0], B[2], ACC); ACC = fl_mfma(A[1], B[1], ACC); \
    ACC = fl_mfma(A[1], B[0], ACC); ACC = fl_mfma(A[0], B[1], ACC); ACC = fl_mfma(A[0], B[0], ACC);
// A fragment (2 fp16 planes) of k-slice KS, row ROWIDX of the row segment at OFF
#define FL_ROWFRAG(DST, BASE, OFF, ROWIDX, KS)                                                                         \
    _Pragma("unroll") for (int pl_ = 0; pl_ < 2; ++pl_) DST[pl_] = *reinterpret_cast<const u32x4*>(                    \
        (BASE) + (OFF) + pl_ * G::RPLANE + (ROWIDX) * G::RROW + ((KS) * 16 + 8 * a) * 2);
// B fragments (2 fp16 planes x NKS slices) of one row of a row-layout tensor in HBM
#define FL_BFRAGS(DST, PTR, ELEM_OFF, PLANE)                                                                           \
    _Pragma("unroll") for (int ks_ = 0; ks_ < NKS; ++ks_) _Pragma("unroll") for (int pl_ = 0; pl_ < 2; ++pl_)           \
        DST[ks_][pl_] = *reinterpret_cast<const u32x4*>((PTR) + (ELEM_OFF) + pl_ * (PLANE) + ks_ * 16);
// 16 accumulator-layout values -> B fragments (TP planes: 3 bf16, or 2 fp16 of values already in fp16 range) of the two
// 16-row slices
#define FL_SPLIT16(PLANES, X)                                                                                          \
    _Pragma("unroll") for (int s2_ = 0; s2_ < 2; ++s2_) _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) {              \
        unsigned hh_, mm_, ll_ = 0u;                                                                                   \
        if (TP == 3) fl_split3(X[8 * s2_ + 2 * j_], X[8 * s2_ + 2 * j_ + 1], hh_, mm_, ll_);                           \
        else fl_split2h(X[8 * s2_ + 2 * j_], X[8 * s2_ + 2 * j_ + 1], hh_, mm_);                                       \
        PLANES[s2_][0][j_] = hh_; PLANES[s2_][1][j_] = mm_; PLANES[s2_][2][j_] = ll_;                                  \
    }
// ACC[db] += X^T[d, row] . PLANES[row, col]   for the tr segment at OFF
#define FL_STAGE2(ACC, BASE, OFF, ROWIDX, PLANES)                                                                      \
    _Pragma("unroll") for (int s2_ = 0; s2_ < 2; ++s2_) _Pragma("unroll") for (int db_ = 0; db_ < NDB; ++db_) {         \
        u32x4 tf_[3];                                                                                                  \
        _Pragma("unroll") for (int pl_ = 0; pl_ < TP; ++pl_) tf_[pl_] = *reinterpret_cast<const u32x4*>(               \
            (BASE) + (OFF) + pl_ * G::TPLANE + (db_ * 32 + (ROWIDX)) * G::TROW + (s2_ * 16 + 8 * a) * 2);              \
        if (TP == 3) { FL_MMA6(ACC[db_], tf_, PLANES[s2_]) } else { FL_MMA3(ACC[db_], tf_, PLANES[s2_]) }              \
    }
// ---- tr form 1: an [L, S] intermediate into fp16 range ----
// X (16 accumulator-layout values of one output column per lane) *= US . F, where US is the wave-uniform unscale factor of
// the tr tile it is about to be multiplied with and F the running power-of-two factor of the accumulator ACC it goes into
// (one per lane; the two half-waves of a column agree).  F only ever decreases: when the tile's largest magnitude MX (per
// lane, or any bound of it) times US . F would reach 2^15, F drops to put it into [2^14, 2^15) and ACC is rescaled (exact).
// Rare after the first tiles of a row, so the whole update sits behind one wave-wide branch.
#define FL_FIT_BOUND(MXUS, F, ACC)                                                                                      \
    if (__builtin_amdgcn_ballot_w64((MXUS) * F >= 32768.f)) {                                                          \
        const unsigned e_ = (__float_as_uint(MXUS) >> 23) & 0xffu;                                                     \
        const float fn_ = (MXUS) * F >= 32768.f ? __uint_as_float(min(268u - e_, 187u) << 23) : F;                     \
        const int d_ = (int)(__float_as_uint(fn_) >> 23) - (int)(__float_as_uint(F) >> 23) + 127;                      \
        const float rs_ = d_ > 0 ? __uint_as_float((unsigned)d_ << 23) : 0.f;                                          \
        _Pragma("unroll") for (int db_ = 0; db_ < NDB; ++db_) _Pragma("unroll") for (int r_ = 0; r_ < 16; ++r_)         \
            ACC[db_][r_] *= rs_;                                                                                       \
        F = fn_;                                                                                                       \
    }
#define FL_FIT16(X, US, F, ACC)                                                                                        \
    if (TP == 2) {                                                                                                     \
        float mx_ = 0.f;                                                                                               \
        _Pragma("unroll") for (int r_ = 0; r_ < 16; ++r_) mx_ = fmaxf(mx_, fabsf(X[r_]));                              \
        mx_ = fmaxf(mx_, __shfl_xor(mx_, 32, 64)) * (US);                                                              \
        FL_FIT_BOUND(mx_, F, ACC)                                                                                      \
        const float c_ = (US) * F;                                                                                     \
        _Pragma("unroll") for (int r_ = 0; r_ < 16; ++r_) X[r_] *= c_;                                                 \
    }
// the same for values known to lie in [0, BOUND] (probabilities): no maximum is taken
#define FL_FIT16_BOUNDED(X, BOUND, US, F, ACC)                                                                         \
    if (TP == 2) {                                                                                                     \
        const float mxb_ = (BOUND) * (US);                                                                             \
        FL_FIT_BOUND(mxb_, F, ACC)                                                                                     \
        const float c_ = (US) * F;                                                                                     \
        _Pragma("unroll") for (int r_ = 0; r_ < 16; ++r_) X[r_] *= c_;                                                 \
    }
// LLVM's instruction-group scheduling strategies as a hint at the top of a tile loop (scheduling only): measured per kernel
// at L = S = 12 755, hd 64 (tools/flash_ab.sh): strategy 0 -7.8 % on the dq / ddO pass and -2.8 % on the forward, +50 % on the
// statistics pass; strategy 2 (MFMA / exp interleave) -10.8 % on the statistics pass, -4.4 % on dq / ddO; the key-owning
// passes gain from neither (strategy 1 aborts the compiler).  MODE < 0: none.
#define FL_IGLP(MODE)                                                  \
    if ((MODE) == 0) __builtin_amdgcn_iglp_opt(0);                     \
    else if ((MODE) == 2) __builtin_amdgcn_iglp_opt(2);
// accumulator-layout output (lane = row of the output tensor, registers = d) -> fp32 rows
#define FL_STORE_ROWS(ACC, DSTPTR, MUL)                                                                                \
    _Pragma("unroll") for (int db_ = 0; db_ < NDB; ++db_) _Pragma("unroll") for (int g_ = 0; g_ < 4; ++g_) {            \
        f32x4 v_;                                                                                                      \
        v_.x = ACC[db_][4 * g_] * (MUL); v_.y = ACC[db_][4 * g_ + 1] * (MUL);                                          \
        v_.z = ACC[db_][4 * g_ + 2] * (MUL); v_.w = ACC[db_][4 * g_ + 3] * (MUL);                                      \
        *reinterpret_cast<f32x4*>((DSTPTR) + db_ * 32 + 8 * g_) = v_;                                                  \
    }


// dropout keep flags of the 16 accumulator registers of a tile whose lane holds ONE row id and whose registers walk the
// OTHER index in accumulator order (PAIRS: registers 2i, 2i+1 are the two keys of one hash).  bool, not float: the
// sixteen flags live as lane masks in scalar register pairs instead of sixteen vector registers.
// FL_M(r, x) = M_r x (the mask with its 1/keep);  FL_K(r, x) = x where kept, 0 where dropped (1/keep applied elsewhere)
#define FL_M(R, X) (DROP ? (kp[R] ? (X) * p.inv_keep : 0.f) : (X))
#define FL_K(R, X) (DROP ? (kp[R] ? (X) : 0.f) : (X))
#define FL_MASK_KEYS_IN_REGS(MK, RID, T0)                                                                              \
    _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) {                                                                 \
        const int key_ = (T0) + ((2 * i_) & 3) + 8 * ((2 * i_) >> 2) + 4 * a;                                          \
        const unsigned hsh_ = fl_hash(p.seed_lo, p.seed_hi, (RID), (unsigned)key_ >> 1);                               \
        MK[2 * i_] = (hsh_ & 0xffffu) >= p.thr16;                                                                      \
        MK[2 * i_ + 1] = (hsh_ >> 16) >= p.thr16;                                                                      \
    }
// (key-owning: neighbouring lanes hold keys 2 j, 2 j + 1 -- the same hash for every query row, different halves of it.  Each lane
//  computes the hashes of half of its sixteen rows (registers with bit 1 == key & 1) and reads the other half from its partner
//  (DPP quad_perm); the lane's half is masked in place and compared against thr16 or thr16 << 16.  Same mask, half the hashes:
//  csrc/flash16.hip M16_MASK_QUERIES, measured there -7.5 % on the first-order pass.)
#define FL_MASK_QUERIES_IN_REGS(MK, KEY, T0)                                                                           \
    {                                                                                                                  \
        const unsigned odd_ = (unsigned)(KEY) & 1u, half_ = odd_ ? 0xffff0000u : 0xffffu;                              \
        const unsigned thr_ = odd_ ? p.thr16 << 16 : p.thr16;                                                          \
        unsigned hq_[4][2];                                                                                            \
        _Pragma("unroll") for (int k_ = 0; k_ < 4; ++k_) _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_) {             \
            const int q_ = (T0) + 2 * (int)odd_ + j_ + 8 * k_ + 4 * a;                                                 \
            hq_[k_][j_] = fl_hash(p.seed_lo, p.seed_hi, (unsigned)(bh * p.L + q_), (unsigned)(KEY) >> 1);              \
        }                                                                                                              \
        _Pragma("unroll") for (int r_ = 0; r_ < 16; ++r_) {                                                            \
            const unsigned own_ = hq_[r_ >> 2][r_ & 1];   /* registers 4k + {0, 1}: the even lane's; 4k + {2, 3}: the odd lane's */ \
            const unsigned h_ = (r_ & 2) ? (unsigned)__builtin_amdgcn_mov_dpp((int)own_, 0xF5, 0xf, 0xf, true)   /* quad_perm [1,1,3,3] */ \
                                         : (unsigned)__builtin_amdgcn_mov_dpp((int)own_, 0xA0, 0xf, 0xf, true);  /* quad_perm [0,0,2,2] */ \
            MK[r_] = (h_ & half_) >= thr_;                                                                             \
        }                                                                                                              \
    }

// ------------------------------------------------------------------------------------------------------------
// forward: query-owning workgroups (4 waves x 32 queries), key tiles of 32, double-buffered LDS, one barrier per tile
// ------------------------------------------------------------------------------------------------------------
// LSE_ONLY: only the row normalisers are produced (q k^T + online softmax statistics, no P v): what the derivative kernels
// need when the forward OUTPUT came from the fp8 kernel, whose normalisers belong to fp8 scores, not to the fp16 scores
// the derivative kernels recompute.
template <int HD, bool DROP, bool LSE_ONLY = false, int TP = 3>
__global__ __launch_bounds__(256, 2) void flash_fwd_kernel(FlashArgs p) {
    if (DROP && p.salt) { p.seed_lo ^= p.salt[0]; p.seed_hi ^= p.salt[1]; }
    constexpr int NKS = HD / 16, NDB = HD / 32;
    typedef FlSeg<HD, TP> G;
    constexpr int OFF_K = 0, OFF_VT = G::RBYTES, BYTES = LSE_ONLY ? G::RBYTES : G::RBYTES + G::TBYTES, NU = LSE_ONLY ? 2 : 2 + TP, NROW = 1;
    __shared__ __attribute__((aligned(16))) unsigned char ldsb[2][BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane & 31, a = lane >> 5;
    const int bh = blockIdx.y, b = bh / p.H, h = bh % p.H;
    const int q0 = blockIdx.x * 128 + wave * 32;
    const int ntiles = (p.S + 31) / 32;
    const int64_t kro = (int64_t)bh * p.Sp * HD, kto = (int64_t)bh * HD * p.Sp;
    const float* bias = p.bias + (int64_t)b * p.Sp;
    const float* kus = p.k_us + (int64_t)bh * (p.Sp / 32);
    const float* vus = TP == 2 ? p.v_us + (int64_t)bh * (p.Sp / 32) : kus;
    float fo = FL_F0;   // (tr form 1) running factor of o

    u32x4 qf[NKS][2];
    FL_BFRAGS(qf, p.q_row, ((int64_t)bh * p.Lp + q0 + lq) * HD + 8 * a, p.q_plane)
    const float c2 = p.scale_log2e * p.q_us[(int64_t)bh * (p.Lp / 32) + q0 / 32];
    f32x16 o[NDB];
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
    float m = -1e30f, l = 0.f;
    const unsigned rid = (unsigned)(bh * p.L + q0 + lq);

    FL_DECL_REGS(sv)
#define FLF_SRC(T0)                                                                                                    \
    (seg_ == 0 ? p.k_row + kro + pl_ * p.k_plane + (int64_t)(T0) * HD + G::row_src(c_)                                \
               : p.v_tr + kto + pl_ * p.k_plane + (T0) + G::tr_src(c_, p.Sp))
#define FLF_DST(BUF) (seg_ == 0 ? (BUF) + OFF_K + pl_ * G::RPLANE + G::row_dst(c_) : (BUF) + OFF_VT + pl_ * G::TPLANE + G::tr_dst(c_))
    FL_STAGE_LOAD(sv, 0, NU, NROW, FLF_SRC(0))
    FL_STAGE_STORE(sv, 0, NU, NROW, FLF_DST(ldsb[0]))
    __syncthreads();

    for (int t = 0; t < ntiles; ++t) {
        FL_IGLP(0)
        const unsigned char* lds = ldsb[t & 1];
        const int t0 = t * 32;
        // key bias of this lane's 16 keys: register r <-> key t0 + (r & 3) + 8 (r >> 2) + 4 a.  Requested BEFORE the
        // prefetch: vmcnt retires in order, so a wait for these (L1 hits) must not sit behind the next tile's HBM loads
        f32x4 kb[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) kb[g] = *reinterpret_cast<const f32x4*>(bias + t0 + 8 * g + 4 * a);
        const float cs = c2 * kus[t], usv = vus[t];
        // next tile's operands: requested now, written to the other LDS buffer after this tile's products (the last
        // iteration re-requests its own tile: unconditional code keeps the staging registers out of scratch memory)
        FL_STAGE_LOAD(sv, 0, NU, NROW, FLF_SRC(min(t0 + 32, ntiles * 32 - 32)))
        // ---- S^T[key, query] = K . Q^T ----
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            u32x4 kf[2];
            FL_ROWFRAG(kf, lds, OFF_K, lq, ks)
            FL_MMA3(s, kf, qf[ks])
        }
        // ---- online softmax (base 2), one query per lane ----
        float x[16], tmax = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            x[r] = s[r] * cs + kb[r >> 2][r & 3];
            tmax = fmaxf(tmax, x[r]);
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float mn = fmaxf(m, tmax);
        const float alpha = fl_exp2(m - mn);
        m = mn;
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            x[r] = fl_exp2(x[r] - mn);
            ps += x[r];
        }
        l = l * alpha + ps;
        if (!LSE_ONLY && __builtin_amdgcn_ballot_w64(alpha != 1.f)) {   // (a row maximum moved somewhere in this wave: rare after the first tiles)
#pragma unroll
            for (int db = 0; db < NDB; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
        }
        if (!LSE_ONLY) {
            if (DROP) {
                bool kp[16];
                FL_MASK_KEYS_IN_REGS(kp, rid, t0)
#pragma unroll
                for (int r = 0; r < 16; ++r) x[r] = kp[r] ? x[r] : 0.f;   // (1 / keep is applied once, to O)
            }
            // ---- O^T[d, query] += V^T[d, key] . P^T[key, query] ----
            u32x4 pp[2][3];
            FL_FIT16_BOUNDED(x, 1.f, usv, fo, o)   // (exp2(. - running max) <= 1)
            FL_SPLIT16(pp, x)
            FL_STAGE2(o, lds, OFF_VT, lq, pp)
        }
        FL_STAGE_STORE(sv, 0, NU, NROW, FLF_DST(ldsb[(t + 1) & 1]))
        __syncthreads();
    }
#undef FLF_SRC
#undef FLF_DST
    l += __shfl_xor(l, 32, 64);
    const int q = q0 + lq;
    if (q < p.L) {
        if (!LSE_ONLY) {
            const float inv = TP == 2 ? p.inv_keep / (l * fo) : p.inv_keep / l;
            float* dst = p.o1 + ((int64_t)b * p.L + q) * p.ld1 + p.off1 + h * HD + 4 * a;
            FL_STORE_ROWS(o, dst, inv)
        }
        if (a == 0) p.lse[(int64_t)bh * p.Lp + q] = (m + log2f(l)) * FL_LN2;
    } else if (a == 0) {
        p.lse[(int64_t)bh * p.Lp + q] = INFINITY;   // padded query rows: the derivative kernels then see P = 0 there
    }
}

// ============================================================================================================
// Backward.  With  P = softmax rows,  M = keep mask / keep,  Pd = P o M,  O = Pd V:
//     gd = dO V^T      gy = M o gd      t_i = sum_j P_ij gy_ij = dO_i . O_i      gs = P o (gy - t)
//     gQ = scale gs K      gK = scale gs^T Q      gV = Pd^T dO
// P is recomputed from (q, k, lse); two kernels: query-owning workgroups produce gQ, key-owning workgroups gK and gV.
// ============================================================================================================

// t[bh][q] = sum_d dO[q, h, d] O[q, h, d]   (one thread per (bh, q); hd floats each from two rows)
__global__ void attn_rowdot_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ t, int H, int L,
                                   int Lp, int hd, int64_t lda, int offa, int64_t ldb, int offb) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x, bh = blockIdx.y;
    if (q >= Lp) return;
    float s = 0.f;
    if (q < L) {
        const int bb = bh / H, h = bh % H;
        const float* pa = a + ((int64_t)bb * L + q) * lda + offa + h * hd;
        const float* pb = b + ((int64_t)bb * L + q) * ldb + offb + h * hd;
        for (int d = 0; d < hd; d += 4) {
            const float4 x = *reinterpret_cast<const float4*>(pa + d), y = *reinterpret_cast<const float4*>(pb + d);
            s += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
        }
    }
    t[(int64_t)bh * Lp + q] = s;
}

extern "C" int ix_attn_rowdot_f32(const float* a, const float* b, float* t, int n, int H, int L, int Lp, int hd, int64_t lda,
                                  int offa, int64_t ldb, int offb, hipStream_t stream) {
    if (n <= 0 || L <= 0) return IX_OK;
    IX_CHECK_ARG(a && b && t && hd % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0 && offa % 4 == 0 && offb % 4 == 0,
                 "ix_attn_rowdot_f32: bad args");
    hipLaunchKernelGGL(attn_rowdot_kernel, dim3(ix_div_up(Lp, 256), n * H), dim3(256), 0, stream, a, b, t, H, L, Lp, hd, lda,
                       offa, ldb, offb);
    IX_CHECK_LAUNCH("ix_attn_rowdot_f32");
    return IX_OK;
}

// ---- query-owning workgroup: gQ (o1) ---------------------------------------------------------------------------------
template <int HD, bool DROP, int TP = 3>
__global__ __launch_bounds__(256, 2) void flash_bwd_q_kernel(FlashArgs p) {
    if (DROP && p.salt) { p.seed_lo ^= p.salt[0]; p.seed_hi ^= p.salt[1]; }
    constexpr int NKS = HD / 16, NDB = HD / 32;
    typedef FlSeg<HD, TP> G;
    constexpr int OFF_K = 0, OFF_V = G::RBYTES, OFF_KT = 2 * G::RBYTES, BYTES = 2 * G::RBYTES + G::TBYTES, NU = 4 + TP, NROW = 2;
    __shared__ __attribute__((aligned(16))) unsigned char ldsb[2][BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane & 31, a = lane >> 5;
    const int bh = blockIdx.y, b = bh / p.H, h = bh % p.H;
    const int q0 = blockIdx.x * 128 + wave * 32;
    const int ntiles = (p.S + 31) / 32;
    const int64_t kro = (int64_t)bh * p.Sp * HD, kto = (int64_t)bh * HD * p.Sp;
    const float* bias = p.bias + (int64_t)b * p.Sp;
    const float* kus = p.k_us + (int64_t)bh * (p.Sp / 32);
    const float* vus = p.v_us + (int64_t)bh * (p.Sp / 32);

    u32x4 qf[NKS][2], df[NKS][2];
    FL_BFRAGS(qf, p.q_row, ((int64_t)bh * p.Lp + q0 + lq) * HD + 8 * a, p.q_plane)
    FL_BFRAGS(df, p.do_row, ((int64_t)bh * p.Lp + q0 + lq) * HD + 8 * a, p.q_plane)
    const int64_t qb = (int64_t)bh * (p.Lp / 32) + q0 / 32;
    const float c2 = p.scale_log2e * p.q_us[qb], usd = p.do_us[qb];
    const float lse2 = p.lse[(int64_t)bh * p.Lp + q0 + lq] * FL_LOG2E;
    const float dl = p.delta[(int64_t)bh * p.Lp + q0 + lq];
    const unsigned rid = (unsigned)(bh * p.L + q0 + lq);
    f32x16 gq[NDB];
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) gq[db][r] = 0.f;
    float fq = FL_F0;

    FL_DECL_REGS(sv)
#define FLQ_SRC(T0)                                                                                                    \
    (seg_ == 0 ? p.k_row + kro + pl_ * p.k_plane + (int64_t)(T0) * HD + G::row_src(c_)                                \
   : seg_ == 1 ? p.v_row + kro + pl_ * p.k_plane + (int64_t)(T0) * HD + G::row_src(c_)                                \
               : p.k_tr + kto + pl_ * p.k_plane + (T0) + G::tr_src(c_, p.Sp))
#define FLQ_DST(BUF)                                                                                                   \
    (seg_ < 2 ? (BUF) + seg_ * G::RBYTES + pl_ * G::RPLANE + G::row_dst(c_) : (BUF) + OFF_KT + pl_ * G::TPLANE + G::tr_dst(c_))
    FL_STAGE_LOAD(sv, 0, NU, NROW, FLQ_SRC(0))
    FL_STAGE_STORE(sv, 0, NU, NROW, FLQ_DST(ldsb[0]))
    __syncthreads();

    for (int t = 0; t < ntiles; ++t) {
        const unsigned char* lds = ldsb[t & 1];
        const int t0 = t * 32;
        f32x4 kb[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) kb[g] = *reinterpret_cast<const f32x4*>(bias + t0 + 8 * g + 4 * a);
        const float usk = kus[t];
        const float cs = c2 * usk, cg = usd * vus[t];
        FL_STAGE_LOAD(sv, 0, NU, NROW, FLQ_SRC(min(t0 + 32, ntiles * 32 - 32)))
        // ---- S^T = K Q^T and gd^T = V dO^T (two independent accumulator chains) ----
        f32x16 s, gd;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; gd[r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            u32x4 kf[2], vf[2];
            FL_ROWFRAG(kf, lds, OFF_K, lq, ks)
            FL_ROWFRAG(vf, lds, OFF_V, lq, ks)
            FL_MMA3(s, kf, qf[ks])
            FL_MMA3(gd, vf, df[ks])
        }
        // ---- gs = P o (M o gd - t) ----
        float x[16];
        bool kp[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) kp[r] = true;
        if (DROP) { FL_MASK_KEYS_IN_REGS(kp, rid, t0) }
#pragma unroll
        for (int r = 0; r < 16; ++r)
            x[r] = fl_exp2(s[r] * cs + kb[r >> 2][r & 3] - lse2) * (FL_M(r, gd[r] * cg) - dl);
        u32x4 pp[2][3];
        FL_FIT16(x, usk, fq, gq)
        FL_SPLIT16(pp, x)
        // ---- gQ^T[d, query] += K^T[d, key] gs^T[key, query] ----
        FL_STAGE2(gq, lds, OFF_KT, lq, pp)
        FL_STAGE_STORE(sv, 0, NU, NROW, FLQ_DST(ldsb[(t + 1) & 1]))
        __syncthreads();
    }
#undef FLQ_SRC
#undef FLQ_DST
    const int q = q0 + lq;
    if (q < p.L) {
        float* dst = p.o1 + ((int64_t)b * p.L + q) * p.ld1 + p.off1 + h * HD + 4 * a;
        const float mq = TP == 2 ? p.scale / fq : p.scale;
        FL_STORE_ROWS(gq, dst, mq)
    }
}

// ---- key-owning workgroup: gK (o2), gV (o3); tiles oriented [query, key]: lane = key, registers = queries ------------
// One LDS buffer, two phases per tile (as the second-order passes below): the ROW region (q, dO rows) feeds the two
// [query, key] tiles while the next tile's rows are in flight, the TR region (q, dO transposed + lse, delta) feeds the two
// output products while the next tile's tr operands are in flight.  49 KB of LDS and <= 256 registers: two workgroups per CU.
template <int HD, bool DROP, int TP = 3>
__global__ __launch_bounds__(256, 2) void flash_bwd_kv_kernel(FlashArgs p) {
    if (DROP && p.salt) { p.seed_lo ^= p.salt[0]; p.seed_hi ^= p.salt[1]; }
    constexpr int NKS = HD / 16, NDB = HD / 32;
    typedef FlSeg<HD, TP> G;
    constexpr int OFF_Q = 0, OFF_D = G::RBYTES, OFF_QT = 2 * G::RBYTES, OFF_DT = 2 * G::RBYTES + G::TBYTES;
    constexpr int OFF_ST = 2 * G::RBYTES + 2 * G::TBYTES, BYTES = OFF_ST + 256, NROW = 2, NUR = 4, NUT = 2 * TP;   // + lse[32], delta[32]
    __shared__ __attribute__((aligned(16))) unsigned char lds[BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lk = lane & 31, a = lane >> 5;
    const int bh = blockIdx.y, b = bh / p.H, h = bh % p.H;
    const int k0 = blockIdx.x * 128 + wave * 32;
    const int ntiles = (p.L + 31) / 32;
    const int64_t qro = (int64_t)bh * p.Lp * HD, qto = (int64_t)bh * HD * p.Lp, sto = (int64_t)bh * p.Lp;
    const float* qus = p.q_us + (int64_t)bh * (p.Lp / 32);
    const float* dus = p.do_us + (int64_t)bh * (p.Lp / 32);

    u32x4 kf[NKS][2], vf[NKS][2];
    FL_BFRAGS(kf, p.k_row, ((int64_t)bh * p.Sp + k0 + lk) * HD + 8 * a, p.k_plane)
    FL_BFRAGS(vf, p.v_row, ((int64_t)bh * p.Sp + k0 + lk) * HD + 8 * a, p.k_plane)
    const int64_t kbk = (int64_t)bh * (p.Sp / 32) + k0 / 32;
    const float c2 = p.scale_log2e * p.k_us[kbk], usv = p.v_us[kbk];
    const float kbias = p.bias[(int64_t)b * p.Sp + k0 + lk];
    const int key = k0 + lk;
    f32x16 gk[NDB], gv[NDB];
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) { gk[db][r] = 0.f; gv[db][r] = 0.f; }
    float fk = FL_F0, fv = FL_F0;

    FL_DECL_REGS(sv)    // row units (phase 1)
    FL_DECL_REGS(svt)   // tr units (phase 2)
    float sst = 0.f;    // staged statistic: threads 0..31 carry lse, 32..63 delta of the next tile
#define FLK_SRC(T0)                                                                                                    \
    (seg_ == 0 ? p.q_row + qro + pl_ * p.q_plane + (int64_t)(T0) * HD + G::row_src(c_)                                \
   : seg_ == 1 ? p.do_row + qro + pl_ * p.q_plane + (int64_t)(T0) * HD + G::row_src(c_)                               \
   : seg_ == 2 ? p.q_tr + qto + pl_ * p.q_plane + (T0) + G::tr_src(c_, p.Lp)                                           \
               : p.do_tr + qto + pl_ * p.q_plane + (T0) + G::tr_src(c_, p.Lp))
#define FLK_DST                                                                                                        \
    (seg_ < 2 ? lds + seg_ * G::RBYTES + pl_ * G::RPLANE + G::row_dst(c_)                                              \
              : lds + 2 * G::RBYTES + (seg_ - 2) * G::TBYTES + pl_ * G::TPLANE + G::tr_dst(c_))
#define FLK_STAT_LOAD(T0) if (tid < 64) sst = tid < 32 ? p.lse[sto + (T0) + tid] : p.delta[sto + (T0) + tid - 32];
    FL_STAGE_LOAD(sv, 0, NUR, NROW, FLK_SRC(0))
    FL_STAGE_STORE(sv, 0, NUR, NROW, FLK_DST)
    FL_STAGE_LOAD(svt, NUR, NUT, NROW, FLK_SRC(0))
    FLK_STAT_LOAD(0)
    FL_STAGE_STORE(svt, NUR, NUT, NROW, FLK_DST)
    if (tid < 64) reinterpret_cast<float*>(lds + OFF_ST)[tid] = sst;
    __syncthreads();

    for (int t = 0; t < ntiles; ++t) {
        const int t0 = t * 32, tn = min(t0 + 32, ntiles * 32 - 32);
        const float usq = qus[t], usd = dus[t];
        const float cs = c2 * usq, cg = usv * usd;
        FL_STAGE_LOAD(sv, 0, NUR, NROW, FLK_SRC(tn))
        // ---- S[query, key] = Q K^T and gd = dO V^T ----
        f32x16 s, gd;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; gd[r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            u32x4 qa[2], da[2];
            FL_ROWFRAG(qa, lds, OFF_Q, lk, ks)
            FL_ROWFRAG(da, lds, OFF_D, lk, ks)
            FL_MMA3(s, qa, kf[ks])
            FL_MMA3(gd, da, vf[ks])
        }
        __syncthreads();   // (A) the ROW region is free, the TR region (and the statistics) are complete
        FL_STAGE_STORE(sv, 0, NUR, NROW, FLK_DST)
        FL_STAGE_LOAD(svt, NUR, NUT, NROW, FLK_SRC(tn))
        FLK_STAT_LOAD(tn)
        // statistics of this lane's 16 queries: register r <-> query t0 + (r & 3) + 8 (r >> 2) + 4 a
#define FLK_ST(WHICH, R) (reinterpret_cast<const float*>(lds + OFF_ST + (WHICH) * 128)[((R) & 3) + 8 * ((R) >> 2) + 4 * a])
        float pd[16], gs[16];
        bool kp[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) kp[r] = true;
        if (DROP) { FL_MASK_QUERIES_IN_REGS(kp, key, t0) }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float pr = fl_exp2(s[r] * cs + kbias - FLK_ST(0, r) * FL_LOG2E);
            pd[r] = FL_K(r, pr);                                          // (x 1/keep at the end, on gV)
            gs[r] = pr * (FL_M(r, gd[r] * cg) - FLK_ST(1, r));
        }
#undef FLK_ST
        // ---- gV^T[d, key] += dO^T[d, query] Pd[query, key];  gK^T[d, key] += Q^T[d, query] gs[query, key] ----
        u32x4 pp[2][3];
        FL_FIT16_BOUNDED(pd, 1.f, usd, fv, gv)
        FL_SPLIT16(pp, pd)
        FL_STAGE2(gv, lds, OFF_DT, lk, pp)
        FL_FIT16(gs, usq, fk, gk)
        FL_SPLIT16(pp, gs)
        FL_STAGE2(gk, lds, OFF_QT, lk, pp)
        __syncthreads();   // (B) every wave is done with the TR region; the next tile's rows are visible
        FL_STAGE_STORE(svt, NUR, NUT, NROW, FLK_DST)
        if (tid < 64) reinterpret_cast<float*>(lds + OFF_ST)[tid] = sst;
    }
#undef FLK_SRC
#undef FLK_DST
#undef FLK_STAT_LOAD
    if (key < p.S) {
        float* dk = p.o2 + ((int64_t)b * p.S + key) * p.ld2 + p.off2 + h * HD + 4 * a;
        float* dv = p.o3 + ((int64_t)b * p.S + key) * p.ld3 + p.off3 + h * HD + 4 * a;
        const float mk = TP == 2 ? p.scale / fk : p.scale, mv = TP == 2 ? p.inv_keep / fv : p.inv_keep;
        FL_STORE_ROWS(gk, dk, mk)
        FL_STORE_ROWS(gv, dv, mv)
    }
}

// ============================================================================================================
// Double backward: the backward above as a function (q, k, v, dO) -> (gQ, gK, gV), differentiated once more.  With
// cotangents (hq, hk, hv) of (gQ, gK, gV), scale c, and P, M, gy, t, gs as above:
//     G  = c (hq k^T + q hk^T)            = dL/d gs                HD = dO hv^T             = dL/d Pd
//     u_i = sum_j P_ij G_ij               HgD = M o P o (G - u)    = dL/d gd
//     HY = G (gy - t) - gy u + M o HD     w_i = sum_j P_ij HY_ij   HS = P o (HY - w)        = dL/d S
//     dq = c (gs hk + HS k)     dk = c (gs^T hq + HS^T q)     dv = HgD^T dO     ddO = Pd hv + HgD v
// Three passes, all recomputing the [L, S] tiles from the operand planes: (1) query-owning, row statistics
//     u_i,  w_i = sum_j P G gy - 2 t u + sum_j Pd HD        (HY is linear in u, sum_j P gy = t)
// (2) query-owning, dq and ddO;  (3) key-owning, dk and dv.
// ============================================================================================================

// passes 1 and 2 (query-owning).
// STATS (pass 1): only u, w are produced; four row segments, double-buffered LDS, one barrier per tile.
// pass 2 (dq -> o1, ddO -> o4): row + tr segments fill the LDS once, so a tile is two phases around two barriers -- phase 1
// forms the [key, query] tiles out of the ROW region while the next tile's rows are in flight, phase 2 runs the four output
// products out of the TR region while the next tile's tr operands are in flight; each region is refilled right after the
// barrier that ends its phase, from one shared set of staging registers.
template <int HD, bool DROP, bool STATS, int TP = 3>
__global__ __launch_bounds__(256, (STATS || HD == 32) ? 2 : 1) void flash_bb_q_kernel(FlashArgs p) {
    if (DROP && p.salt) { p.seed_lo ^= p.salt[0]; p.seed_hi ^= p.salt[1]; }
    constexpr int NKS = HD / 16, NDB = HD / 32;
    typedef FlSeg<HD, TP> G;
    // row segments k, hk, v, hv; then (pass 2) tr segments hk, k, hv, v
    constexpr int OFF_K = 0, OFF_HK = G::RBYTES, OFF_V = 2 * G::RBYTES, OFF_HV = 3 * G::RBYTES;
    constexpr int OFF_HKT = 4 * G::RBYTES, OFF_KT = OFF_HKT + G::TBYTES, OFF_HVT = OFF_KT + G::TBYTES, OFF_VT = OFF_HVT + G::TBYTES;
    // pass 2 also keeps the dO rows of each wave's 32 queries resident behind the tile (their B fragments are read from
    // LDS per k-slice: 32 registers fewer, which is what keeps the staging loads out of scratch memory)
    constexpr int OFF_RES = 4 * G::RBYTES + 4 * G::TBYTES;
    constexpr int BYTES = STATS ? 4 * G::RBYTES : OFF_RES + 4 * G::RBYTES;
    constexpr int NBUF = STATS ? 2 : 1, NROW = 4, NUR = 8, NUT = 4 * TP;
    __shared__ __attribute__((aligned(16))) unsigned char ldsq[NBUF][BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane & 31, a = lane >> 5;
    const int bh = blockIdx.y, b = bh / p.H, h = bh % p.H;
    const int q0 = blockIdx.x * 128 + wave * 32;
    const int ntiles = (p.S + 31) / 32;
    const int64_t kro = (int64_t)bh * p.Sp * HD, kto = (int64_t)bh * HD * p.Sp, kbo = (int64_t)bh * (p.Sp / 32);
    const float* bias = p.bias + (int64_t)b * p.Sp;

    u32x4 qf[NKS][2], hqf[NKS][2], df[STATS ? NKS : 1][2];
    FL_BFRAGS(qf, p.q_row, ((int64_t)bh * p.Lp + q0 + lq) * HD + 8 * a, p.q_plane)
    FL_BFRAGS(hqf, p.hq_row, ((int64_t)bh * p.Lp + q0 + lq) * HD + 8 * a, p.q_plane)
    const int res = OFF_RES + wave * G::RBYTES;
    if (STATS) {
        FL_BFRAGS(df, p.do_row, ((int64_t)bh * p.Lp + q0 + lq) * HD + 8 * a, p.q_plane)
    } else {
        for (int c = lane; c < 2 * 4 * HD; c += 64) {   // 2 planes x (32 rows x HD fp16 = 4 HD chunks)
            const int pl = c / (4 * HD), ch = c % (4 * HD);
            const unsigned short* src = p.do_row + pl * p.q_plane + ((int64_t)bh * p.Lp + q0) * HD + G::row_src(ch);
            *reinterpret_cast<uint4*>(ldsq[0] + res + pl * G::RPLANE + G::row_dst(ch)) = *reinterpret_cast<const uint4*>(src);
        }
    }
    const int64_t qb = (int64_t)bh * (p.Lp / 32) + q0 / 32;
    const float usq = p.q_us[qb], usd = p.do_us[qb], ushq = p.hq_us[qb];
    const int64_t so = (int64_t)bh * p.Lp + q0 + lq;
    const float lse2 = p.lse[so] * FL_LOG2E, dl = p.delta[so];
    float uu = 0.f, ww = 0.f, aa = 0.f, bq = 0.f;
    if (!STATS) { uu = p.u[so]; ww = p.w[so]; }
    const unsigned rid = (unsigned)(bh * p.L + q0 + lq);
    f32x16 dq[NDB], ddo[NDB];
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dq[db][r] = 0.f; ddo[db][r] = 0.f; }
    float fdq = FL_F0, fddo = FL_F0;

    FL_DECL_REGS(sv)    // row units (phase 1)
    FL_DECL_REGS(svt)   // tr units (phase 2)
#define FLB_SRC(T0)                                                                                                    \
    (seg_ == 0 ? p.k_row + kro + pl_ * p.k_plane + (int64_t)(T0) * HD + G::row_src(c_)                                \
   : seg_ == 1 ? p.hk_row + kro + pl_ * p.k_plane + (int64_t)(T0) * HD + G::row_src(c_)                               \
   : seg_ == 2 ? p.v_row + kro + pl_ * p.k_plane + (int64_t)(T0) * HD + G::row_src(c_)                                \
   : seg_ == 3 ? p.hv_row + kro + pl_ * p.k_plane + (int64_t)(T0) * HD + G::row_src(c_)                               \
   : seg_ == 4 ? p.hk_tr + kto + pl_ * p.k_plane + (T0) + G::tr_src(c_, p.Sp)                                          \
   : seg_ == 5 ? p.k_tr + kto + pl_ * p.k_plane + (T0) + G::tr_src(c_, p.Sp)                                           \
   : seg_ == 6 ? p.hv_tr + kto + pl_ * p.k_plane + (T0) + G::tr_src(c_, p.Sp)                                          \
               : p.v_tr + kto + pl_ * p.k_plane + (T0) + G::tr_src(c_, p.Sp))
#define FLB_DST(BUF)                                                                                                   \
    (seg_ < 4 ? (BUF) + seg_ * G::RBYTES + pl_ * G::RPLANE + G::row_dst(c_)                                            \
              : (BUF) + 4 * G::RBYTES + (seg_ - 4) * G::TBYTES + pl_ * G::TPLANE + G::tr_dst(c_))
    FL_STAGE_LOAD(sv, 0, NUR, NROW, FLB_SRC(0))
    FL_STAGE_STORE(sv, 0, NUR, NROW, FLB_DST(ldsq[0]))
    if (!STATS) {
        FL_STAGE_LOAD(svt, NUR, NUT, NROW, FLB_SRC(0))
        FL_STAGE_STORE(svt, NUR, NUT, NROW, FLB_DST(ldsq[0]))
    }
    __syncthreads();

    for (int t = 0; t < ntiles; ++t) {
        FL_IGLP(STATS ? 2 : 0)
        unsigned char* lds = ldsq[STATS ? (t & 1) : 0];
        const int t0 = t * 32, tn = min(t0 + 32, ntiles * 32 - 32);
        f32x4 kb[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) kb[g] = *reinterpret_cast<const f32x4*>(bias + t0 + 8 * g + 4 * a);
        const float usk = p.k_us[kbo + t], ushk = p.hk_us[kbo + t], usv = p.v_us[kbo + t], ushv = p.hv_us[kbo + t];
        const float cs = p.scale_log2e * usq * usk, cg = usd * usv, c1 = p.scale * ushq * usk, c3 = p.scale * usq * ushk, ch = usd * ushv;
        FL_STAGE_LOAD(sv, 0, NUR, NROW, FLB_SRC(tn))
        // ---- the [key, query] tiles: S, gd, G (two chains: hq.k and q.hk carry different block scales), HD ----
        f32x16 s, gd, g1, g2, hd_;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; gd[r] = 0.f; g1[r] = 0.f; g2[r] = 0.f; hd_[r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            u32x4 kf[2], hkf[2], vf[2], hvf[2], dfl[2];
            FL_ROWFRAG(kf, lds, OFF_K, lq, ks)
            FL_ROWFRAG(hkf, lds, OFF_HK, lq, ks)
            FL_ROWFRAG(vf, lds, OFF_V, lq, ks)
            FL_ROWFRAG(hvf, lds, OFF_HV, lq, ks)
            if (STATS) { dfl[0] = df[STATS ? ks : 0][0]; dfl[1] = df[STATS ? ks : 0][1]; } else { FL_ROWFRAG(dfl, lds, res, lq, ks) }
            FL_MMA3(s, kf, qf[ks])
            FL_MMA3(gd, vf, dfl)
            FL_MMA3(g1, kf, hqf[ks])
            FL_MMA3(hd_, hvf, dfl)
            FL_MMA3(g2, hkf, qf[ks])
        }
        float pr[16];
        bool kp[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) kp[r] = true;
        if (DROP) { FL_MASK_KEYS_IN_REGS(kp, rid, t0) }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            pr[r] = fl_exp2(s[r] * cs + kb[r >> 2][r & 3] - lse2);
            g1[r] = g1[r] * c1 + g2[r] * c3;   // G
            gd[r] = FL_M(r, gd[r] * cg);       // gy = M o gd
            hd_[r] = FL_M(r, hd_[r] * ch);     // M o HD
        }
        if (STATS) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pg = pr[r] * g1[r];
                uu += pg;
                aa += pg * gd[r];
                bq += pr[r] * hd_[r];
            }
            FL_STAGE_STORE(sv, 0, NUR, NROW, FLB_DST(ldsq[(t + 1) & 1]))
            __syncthreads();
        } else {
            __syncthreads();   // (A) every wave has formed its tiles: the ROW region is free, the TR region is complete
            FL_STAGE_STORE(sv, 0, NUR, NROW, FLB_DST(lds))
            FL_STAGE_LOAD(svt, NUR, NUT, NROW, FLB_SRC(tn))
            float x[16];
            u32x4 pp[2][3];
            // gs = P (gy - t)                                   dq += hk^T gs
#pragma unroll
            for (int r = 0; r < 16; ++r) x[r] = pr[r] * (gd[r] - dl);
            FL_FIT16(x, ushk, fdq, dq)
            FL_SPLIT16(pp, x)
            FL_STAGE2(dq, lds, OFF_HKT, lq, pp)
            // HS = P (G (gy - t) - gy u + M HD - w)             dq += k^T HS
#pragma unroll
            for (int r = 0; r < 16; ++r) x[r] = pr[r] * (g1[r] * (gd[r] - dl) - gd[r] * uu + hd_[r] - ww);
            FL_FIT16(x, usk, fdq, dq)
            FL_SPLIT16(pp, x)
            FL_STAGE2(dq, lds, OFF_KT, lq, pp)
            // Pd = M P                                          ddO += hv^T Pd
#pragma unroll
            for (int r = 0; r < 16; ++r) x[r] = FL_M(r, pr[r]);
            FL_FIT16_BOUNDED(x, p.inv_keep, ushv, fddo, ddo)
            FL_SPLIT16(pp, x)
            FL_STAGE2(ddo, lds, OFF_HVT, lq, pp)
            // HgD = M P (G - u)                                 ddO += v^T HgD
#pragma unroll
            for (int r = 0; r < 16; ++r) x[r] = FL_M(r, pr[r] * (g1[r] - uu));
            FL_FIT16(x, usv, fddo, ddo)
            FL_SPLIT16(pp, x)
            FL_STAGE2(ddo, lds, OFF_VT, lq, pp)
            __syncthreads();   // (B) every wave is done with the TR region; the next tile's rows are visible
            FL_STAGE_STORE(svt, NUR, NUT, NROW, FLB_DST(lds))
        }
    }
#undef FLB_SRC
#undef FLB_DST
    const int q = q0 + lq;
    if (STATS) {
        uu += __shfl_xor(uu, 32, 64);
        aa += __shfl_xor(aa, 32, 64);
        bq += __shfl_xor(bq, 32, 64);
        if (a == 0) {   // (padded queries: P = 0 -> zeros; the whole [BH][Lp] workspace is written)
            p.u[so] = uu;
            p.w[so] = aa - 2.f * dl * uu + bq;
        }
        return;
    }
    if (q < p.L) {
        float* d1 = p.o1 + ((int64_t)b * p.L + q) * p.ld1 + p.off1 + h * HD + 4 * a;
        float* d4 = p.o4 + ((int64_t)b * p.L + q) * p.ld4 + p.off4 + h * HD + 4 * a;
        const float m1 = TP == 2 ? p.scale / fdq : p.scale, m4 = TP == 2 ? 1.f / fddo : 1.f;
        FL_STORE_ROWS(dq, d1, m1)
        FL_STORE_ROWS(ddo, d4, m4)
    }
}

// pass 3 (key-owning): dk (o2), dv (o3).  Tiles [query, key]: lane = key, registers = queries.  Two-phase tile as in pass 2.
template <int HD, bool DROP, int TP = 3>
__global__ __launch_bounds__(256, HD == 32 ? 2 : 1) void flash_bb_kv_kernel(FlashArgs p) {
    if (DROP && p.salt) { p.seed_lo ^= p.salt[0]; p.seed_hi ^= p.salt[1]; }
    constexpr int NKS = HD / 16, NDB = HD / 32;
    typedef FlSeg<HD, TP> G;
    // row segments q, hq, dO; tr segments hq, q, dO; statistics lse, delta, u, w [32] each (part of the TR phase)
    constexpr int OFF_Q = 0, OFF_HQ = G::RBYTES, OFF_D = 2 * G::RBYTES;
    constexpr int OFF_HQT = 3 * G::RBYTES, OFF_QT = OFF_HQT + G::TBYTES, OFF_DT = OFF_QT + G::TBYTES, OFF_ST = OFF_DT + G::TBYTES;
    // + resident: the hk and hv rows of each wave's 32 keys (their B fragments are read from here per k-slice instead of
    // living in 64 registers: with them in registers the staging loads spill, and a spilled in-flight load is waited for)
    constexpr int OFF_RES = OFF_ST + 512, BYTES = OFF_RES + 4 * 2 * G::RBYTES, NROW = 3, NUR = 6, NUT = 3 * TP;
    __shared__ __attribute__((aligned(16))) unsigned char lds[BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lk = lane & 31, a = lane >> 5;
    const int bh = blockIdx.y, b = bh / p.H, h = bh % p.H;
    const int k0 = blockIdx.x * 128 + wave * 32;
    const int ntiles = (p.L + 31) / 32;
    const int64_t qro = (int64_t)bh * p.Lp * HD, qto = (int64_t)bh * HD * p.Lp, sto = (int64_t)bh * p.Lp, qbo = (int64_t)bh * (p.Lp / 32);

    u32x4 kf[NKS][2], vf[NKS][2];
    FL_BFRAGS(kf, p.k_row, ((int64_t)bh * p.Sp + k0 + lk) * HD + 8 * a, p.k_plane)
    FL_BFRAGS(vf, p.v_row, ((int64_t)bh * p.Sp + k0 + lk) * HD + 8 * a, p.k_plane)
    const int res = OFF_RES + wave * 2 * G::RBYTES;   // this wave's resident rows: hk then hv, row-segment layout
    for (int c = lane; c < 2 * 2 * 4 * HD; c += 64) {   // 2 tensors x 2 planes x (32 rows x HD fp16 = 4 HD chunks)
        const int tsr = c / (2 * 4 * HD), pl = (c / (4 * HD)) & 1, ch = c % (4 * HD);
        const unsigned short* src = (tsr ? p.hv_row : p.hk_row) + pl * p.k_plane + ((int64_t)bh * p.Sp + k0) * HD + G::row_src(ch);
        *reinterpret_cast<uint4*>(lds + res + tsr * G::RBYTES + pl * G::RPLANE + G::row_dst(ch)) = *reinterpret_cast<const uint4*>(src);
    }
    const int64_t kbk = (int64_t)bh * (p.Sp / 32) + k0 / 32;
    const float usk = p.k_us[kbk], ushk = p.hk_us[kbk], usv = p.v_us[kbk], ushv = p.hv_us[kbk];
    const float kbias = p.bias[(int64_t)b * p.Sp + k0 + lk];
    const int key = k0 + lk;
    f32x16 dk[NDB], dv[NDB];
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[db][r] = 0.f; dv[db][r] = 0.f; }
    float fdk = FL_F0, fdv = FL_F0;

    FL_DECL_REGS(sv)    // row units (phase 1)
    FL_DECL_REGS(svt)   // tr units (phase 2)
    float sst = 0.f;   // staged statistics: threads 0..127 carry lse | delta | u | w of the next tile
#define FLC_SRC(T0)                                                                                                    \
    (seg_ == 0 ? p.q_row + qro + pl_ * p.q_plane + (int64_t)(T0) * HD + G::row_src(c_)                                \
   : seg_ == 1 ? p.hq_row + qro + pl_ * p.q_plane + (int64_t)(T0) * HD + G::row_src(c_)                               \
   : seg_ == 2 ? p.do_row + qro + pl_ * p.q_plane + (int64_t)(T0) * HD + G::row_src(c_)                               \
   : seg_ == 3 ? p.hq_tr + qto + pl_ * p.q_plane + (T0) + G::tr_src(c_, p.Lp)                                          \
   : seg_ == 4 ? p.q_tr + qto + pl_ * p.q_plane + (T0) + G::tr_src(c_, p.Lp)                                           \
               : p.do_tr + qto + pl_ * p.q_plane + (T0) + G::tr_src(c_, p.Lp))
#define FLC_DST                                                                                                        \
    (seg_ < 3 ? lds + seg_ * G::RBYTES + pl_ * G::RPLANE + G::row_dst(c_)                                              \
              : lds + 3 * G::RBYTES + (seg_ - 3) * G::TBYTES + pl_ * G::TPLANE + G::tr_dst(c_))
#define FLC_STAT_LOAD(T0)                                                                                              \
    if (tid < 128) {                                                                                                   \
        const float* sp_ = tid < 32 ? p.lse : tid < 64 ? p.delta : tid < 96 ? p.u : p.w;                               \
        sst = sp_[sto + (T0) + (tid & 31)];                                                                            \
    }
    FL_STAGE_LOAD(sv, 0, NUR, NROW, FLC_SRC(0))
    FL_STAGE_STORE(sv, 0, NUR, NROW, FLC_DST)
    FL_STAGE_LOAD(svt, NUR, NUT, NROW, FLC_SRC(0))
    FLC_STAT_LOAD(0)
    FL_STAGE_STORE(svt, NUR, NUT, NROW, FLC_DST)
    if (tid < 128) reinterpret_cast<float*>(lds + OFF_ST)[tid] = sst;
    __syncthreads();

    for (int t = 0; t < ntiles; ++t) {
        const int t0 = t * 32, tn = min(t0 + 32, ntiles * 32 - 32);
        const float usq = p.q_us[qbo + t], ushq = p.hq_us[qbo + t], usd = p.do_us[qbo + t];
        const float cs = p.scale_log2e * usq * usk, cg = usd * usv, c1 = p.scale * ushq * usk, c3 = p.scale * usq * ushk, ch = usd * ushv;
        FL_STAGE_LOAD(sv, 0, NUR, NROW, FLC_SRC(tn))
        f32x16 s, gd, g1, g2, hd_;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; gd[r] = 0.f; g1[r] = 0.f; g2[r] = 0.f; hd_[r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            u32x4 qa[2], hqa[2], da[2], hkf[2], hvf[2];
            FL_ROWFRAG(qa, lds, OFF_Q, lk, ks)
            FL_ROWFRAG(hqa, lds, OFF_HQ, lk, ks)
            FL_ROWFRAG(da, lds, OFF_D, lk, ks)
            FL_ROWFRAG(hkf, lds, res, lk, ks)
            FL_ROWFRAG(hvf, lds, res + G::RBYTES, lk, ks)
            FL_MMA3(s, qa, kf[ks])
            FL_MMA3(gd, da, vf[ks])
            FL_MMA3(g1, hqa, kf[ks])
            FL_MMA3(hd_, da, hvf)
            FL_MMA3(g2, qa, hkf)
        }
        __syncthreads();   // (A) the ROW region is free, the TR region (and the statistics) are complete
        FL_STAGE_STORE(sv, 0, NUR, NROW, FLC_DST)
        FL_STAGE_LOAD(svt, NUR, NUT, NROW, FLC_SRC(tn))
        FLC_STAT_LOAD(tn)
        // statistics of this lane's 16 queries: register r <-> query t0 + (r & 3) + 8 (r >> 2) + 4 a; read per use
#define FLC_ST(WHICH, R) (reinterpret_cast<const float*>(lds + OFF_ST + (WHICH) * 128)[((R) & 3) + 8 * ((R) >> 2) + 4 * a])
        float pr[16], x[16];
        bool kp[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) kp[r] = true;
        if (DROP) { FL_MASK_QUERIES_IN_REGS(kp, key, t0) }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            pr[r] = fl_exp2(s[r] * cs + kbias - FLC_ST(0, r) * FL_LOG2E);
            g1[r] = g1[r] * c1 + g2[r] * c3;   // G
            gd[r] = FL_M(r, gd[r] * cg);       // gy = M o gd
            hd_[r] = FL_M(r, hd_[r] * ch);     // M o HD
        }
        u32x4 pp[2][3];
        // gs                                                     dk += hq^T gs
#pragma unroll
        for (int r = 0; r < 16; ++r) x[r] = pr[r] * (gd[r] - FLC_ST(1, r));
        FL_FIT16(x, ushq, fdk, dk)
        FL_SPLIT16(pp, x)
        FL_STAGE2(dk, lds, OFF_HQT, lk, pp)
        // HS                                                     dk += q^T HS
#pragma unroll
        for (int r = 0; r < 16; ++r)
            x[r] = pr[r] * (g1[r] * (gd[r] - FLC_ST(1, r)) - gd[r] * FLC_ST(2, r) + hd_[r] - FLC_ST(3, r));
        FL_FIT16(x, usq, fdk, dk)
        FL_SPLIT16(pp, x)
        FL_STAGE2(dk, lds, OFF_QT, lk, pp)
        // HgD                                                    dv += dO^T HgD
#pragma unroll
        for (int r = 0; r < 16; ++r) x[r] = FL_M(r, pr[r] * (g1[r] - FLC_ST(2, r)));
        FL_FIT16(x, usd, fdv, dv)
        FL_SPLIT16(pp, x)
        FL_STAGE2(dv, lds, OFF_DT, lk, pp)
#undef FLC_ST
        __syncthreads();   // (B) every wave is done with the TR region; the next tile's rows are visible
        FL_STAGE_STORE(svt, NUR, NUT, NROW, FLC_DST)
        if (tid < 128) reinterpret_cast<float*>(lds + OFF_ST)[tid] = sst;
    }
#undef FLC_SRC
#undef FLC_DST
#undef FLC_STAT_LOAD
    if (key < p.S) {
        float* d2 = p.o2 + ((int64_t)b * p.S + key) * p.ld2 + p.off2 + h * HD + 4 * a;
        float* d3 = p.o3 + ((int64_t)b * p.S + key) * p.ld3 + p.off3 + h * HD + 4 * a;
        const float m2 = TP == 2 ? p.scale / fdk : p.scale, m3 = TP == 2 ? 1.f / fdv : 1.f;
        FL_STORE_ROWS(dk, d2, m2)
        FL_STORE_ROWS(dv, d3, m3)
    }
}

// ------------------------------------------------------------------------------------------------------------
// C-ABI
// ------------------------------------------------------------------------------------------------------------
// One operand's planes as written by ix_attn_split_f32 (any member may be null where an entry point does not use it)
struct ix_attn_planes {
    const void* row;      // fp16 row planes   [2][n*H][Rp][hd]
    const float* unscale; // block unscales    [n*H][Rp / 32]
    const void* tr;       // tr planes         [3][n*H][hd][Rp] bf16 (tr_form 0) / [2][n*H][hd][Rp] fp16 (tr_form 1)
    int tr_form;          // what ix_attn_split_f32 was asked to write; all operands of one call agree
};
// second-stage matrix instructions per algorithmic product: 6 (three bf16 planes) or 3 (two fp16 planes)
#define FL_S2(FORM) ((FORM) == 1 ? 3 : 6)

// head dim 64, fp16 form: the 16x16x32 passes of flash16.hip (eight waves per workgroup, row planes only).  Process-wide
// switch for A/B runs and for the tests that pin both kernel families (ix_flash_set_m16; IX_FLASH_M16=0 in the environment).
void fl16_launch_fwd(const FlashArgs& a, dim3 grid, hipStream_t stream);
void fl16_launch_bwd_q(const FlashArgs& a, dim3 grid, hipStream_t stream);
void fl16_launch_bwd_kv(const FlashArgs& a, dim3 grid, hipStream_t stream);
void fl16_launch_bb_stats(const FlashArgs& a, dim3 grid, hipStream_t stream);
void fl16_launch_bb_q(const FlashArgs& a, dim3 grid, hipStream_t stream);
void fl16_launch_bb_kv(const FlashArgs& a, dim3 grid, hipStream_t stream);
// ... and their single-term twins (flash16.hip built with -DM16_ONE): the h planes only, for operands that are 16-bit values (the
// 16-bit activation mode).  ix_flash_set_single_term(1) routes the head-dim-64 fp16-form passes there; returns the previous setting.
void fl16_launch_fwd_one(const FlashArgs& a, dim3 grid, hipStream_t stream);
void fl16_launch_bwd_q_one(const FlashArgs& a, dim3 grid, hipStream_t stream);
void fl16_launch_bwd_kv_one(const FlashArgs& a, dim3 grid, hipStream_t stream);
void fl16_launch_bb_stats_one(const FlashArgs& a, dim3 grid, hipStream_t stream);
void fl16_launch_bb_q_one(const FlashArgs& a, dim3 grid, hipStream_t stream);
void fl16_launch_bb_kv_one(const FlashArgs& a, dim3 grid, hipStream_t stream);
static int g_fl_one = 0;
extern "C" int ix_flash_set_single_term(int on) {
    const int old = g_fl_one;
    if (on == 0 || on == 1) g_fl_one = on;
    return old;
}
static int fl_m16_default() {
    const char* e = getenv("IX_FLASH_M16");
    return !(e && e[0] == '0');
}
static int g_fl_m16 = fl_m16_default();
extern "C" int ix_flash_set_m16(int on) {
    const int old = g_fl_m16;
    if (on == 0 || on == 1) g_fl_m16 = on;
    return old;
}
static inline bool fl_m16(int hd, int form) { return g_fl_m16 && hd == 64 && form == 1; }

// bias == NULL ("no key is masked") is taken by the head-dim-64 fp16-form passes of flash16.hip only: they then neither load nor
// add a bias and blank the keys >= S of the last tile themselves
static int fl_common(FlashArgs& a, const char* who, const float* bias, bool m16, int n, int H, int L, int Lp, int S, int Sp, int hd,
                     float scale, float p_drop, uint64_t seed) {
    IX_CHECK_ARG(bias != nullptr || m16, "%s: null key bias (only the head-dim-64 fp16-form kernels run without one)", who);
    IX_CHECK_ARG(hd == 32 || hd == 64, "%s: head dim %d (32 or 64)", who, hd);
    IX_CHECK_ARG(S > 0 && Lp % 128 == 0 && Sp % 128 == 0 && Lp >= L && Sp >= S, "%s: bad padded sizes", who);
    IX_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f, "%s: p=%f outside [0,1)", who, p_drop);
    IX_CHECK_ARG((int64_t)n * H * L < ((int64_t)1 << 32) && n * H <= 65535, "%s: too many rows", who);
    memset(&a, 0, sizeof(a));
    a.bias = bias;
    a.H = H; a.L = L; a.Lp = Lp; a.S = S; a.Sp = Sp;
    a.q_plane = (int64_t)n * H * Lp * hd; a.k_plane = (int64_t)n * H * Sp * hd;
    a.scale = scale; a.scale_log2e = scale * FL_LOG2E;
    a.thr16 = (unsigned)((double)p_drop * 65536.0 + 0.5);
    a.inv_keep = a.thr16 ? 65536.f / (float)(65536u - a.thr16) : 1.f;
    a.seed_lo = (unsigned)seed; a.seed_hi = (unsigned)(seed >> 32);
    a.salt = reinterpret_cast<const unsigned*>(ix_g_salt);
    return IX_OK;
}
#define FL_OUT_OK(LD, OFF) ((LD) % 4 == 0 && (OFF) % 4 == 0)
// algorithmic FLOPs of ONE [L, S] x hd product over all (batch, head) pairs -- the unit the launch statistics count in
#define FL_PRODUCT_FLOPS (2.0 * (double)n * (double)H * (double)L * (double)S * (double)hd)
#define FL_DISPATCH2(KERNEL, GRID, HD_, TP_)                                                           \
    if (a.thr16) hipLaunchKernelGGL((KERNEL<HD_, true, TP_>), GRID, dim3(256), 0, stream, a);          \
    else hipLaunchKernelGGL((KERNEL<HD_, false, TP_>), GRID, dim3(256), 0, stream, a);
#define FL_DISPATCH(KERNEL, GRID, FORM)                                                                \
    if (hd == 64) {                                                                                    \
        if ((FORM) == 1) { FL_DISPATCH2(KERNEL, GRID, 64, 2) } else { FL_DISPATCH2(KERNEL, GRID, 64, 3) } \
    } else {                                                                                           \
        if ((FORM) == 1) { FL_DISPATCH2(KERNEL, GRID, 32, 2) } else { FL_DISPATCH2(KERNEL, GRID, 32, 3) } \
    }

#define FL_DISPATCH_FWD(HD_, TP_)                                                                      \
    if (a.thr16) hipLaunchKernelGGL((flash_fwd_kernel<HD_, true, false, TP_>), grid, dim3(256), 0, stream, a); \
    else hipLaunchKernelGGL((flash_fwd_kernel<HD_, false, false, TP_>), grid, dim3(256), 0, stream, a);
extern "C" int ix_flash_fwd_f32(const ix_attn_planes* q, const ix_attn_planes* k, const ix_attn_planes* v, const float* bias,
                                float* out, float* lse, int n, int H, int L, int Lp, int S, int Sp, int hd, int64_t ld_out,
                                int off_out, float scale, float p_drop, uint64_t seed, hipStream_t stream) {
    if (n <= 0 || L <= 0) return IX_OK;
    IX_CHECK_ARG(q && k && q->row && q->unscale && k->row && k->unscale && lse, "ix_flash_fwd_f32: null pointer");
    const bool m16 = out && v && fl_m16(hd, v->tr_form);
    IX_CHECK_ARG(!out || (v && (m16 ? v->row != nullptr : v->tr != nullptr)), "ix_flash_fwd_f32: v planes missing (%s)", m16 ? "row" : "tr");
    IX_CHECK_ARG(FL_OUT_OK(ld_out, off_out) && ((uintptr_t)out & 15) == 0, "ix_flash_fwd_f32: output rows must be 16-byte aligned");
    FlashArgs a;
    const int rc = fl_common(a, "ix_flash_fwd_f32", bias, m16, n, H, L, Lp, S, Sp, hd, scale, p_drop, seed);
    if (rc) return rc;
    a.q_row = (const unsigned short*)q->row; a.q_us = q->unscale;
    a.k_row = (const unsigned short*)k->row; a.k_us = k->unscale;
    a.v_tr = out ? (const unsigned short*)v->tr : nullptr;
    a.v_row = out ? (const unsigned short*)v->row : nullptr;
    const int form = out ? v->tr_form : 0;
    IX_CHECK_ARG(form == 0 || (form == 1 && v->unscale), "ix_flash_fwd_f32: v tr planes of form %d lack their unscale factors", form);
    a.v_us = out ? v->unscale : nullptr;
    a.o1 = out; a.ld1 = ld_out; a.off1 = off_out; a.lse = lse;
    dim3 grid((L + 127) / 128, n * H);
    if (!out) {   // row normalisers only
        ix_prof_begin(stream, 2, 1.0 * FL_PRODUCT_FLOPS, 3 * FL_PRODUCT_FLOPS, 1);
        if (hd == 64) hipLaunchKernelGGL((flash_fwd_kernel<64, false, true>), grid, dim3(256), 0, stream, a);
        else hipLaunchKernelGGL((flash_fwd_kernel<32, false, true>), grid, dim3(256), 0, stream, a);
        ix_prof_end(stream);
        IX_CHECK_LAUNCH("ix_flash_fwd_f32");
        return IX_OK;
    }
    ix_prof_begin(stream, 2, 2.0 * FL_PRODUCT_FLOPS, (1 * 3 + 1 * FL_S2(form)) * FL_PRODUCT_FLOPS, 1);
    if (m16) {
        if (g_fl_one) fl16_launch_fwd_one(a, grid, stream); else fl16_launch_fwd(a, grid, stream);
    } else if (hd == 64) {
        if (form == 1) { FL_DISPATCH_FWD(64, 2) } else { FL_DISPATCH_FWD(64, 3) }
    } else {
        if (form == 1) { FL_DISPATCH_FWD(32, 2) } else { FL_DISPATCH_FWD(32, 3) }
    }
    ix_prof_end(stream);
    IX_CHECK_LAUNCH("ix_flash_fwd_f32");
    return IX_OK;
}

extern "C" int ix_flash_bwd_f32(const ix_attn_planes* q, const ix_attn_planes* k, const ix_attn_planes* v,
                                const ix_attn_planes* d_out, const float* bias, const float* lse, const float* delta, float* gq,
                                float* gk, float* gv, int n, int H, int L, int Lp, int S, int Sp, int hd, int64_t ld_q,
                                int off_q, int64_t ld_k, int off_k, int64_t ld_v, int off_v, float scale, float p_drop,
                                uint64_t seed, hipStream_t stream) {
    if (n <= 0 || L <= 0 || S <= 0) return IX_OK;
    IX_CHECK_ARG(q && k && v && d_out && q->row && q->unscale && k->row && k->unscale && v->row && v->unscale &&
                 d_out->row && d_out->unscale && lse && delta, "ix_flash_bwd_f32: null operand");
    const bool m16 = fl_m16(hd, q->tr_form);
    IX_CHECK_ARG(m16 || (q->tr && k->tr && d_out->tr), "ix_flash_bwd_f32: tr planes missing");
    IX_CHECK_ARG(gq || (gk && gv), "ix_flash_bwd_f32: no output requested");
    IX_CHECK_ARG(FL_OUT_OK(ld_q, off_q) && FL_OUT_OK(ld_k, off_k) && FL_OUT_OK(ld_v, off_v), "ix_flash_bwd_f32: output rows must be 16-byte aligned");
    FlashArgs a;
    const int rc = fl_common(a, "ix_flash_bwd_f32", bias, m16, n, H, L, Lp, S, Sp, hd, scale, p_drop, seed);
    if (rc) return rc;
    a.q_row = (const unsigned short*)q->row; a.q_us = q->unscale; a.q_tr = (const unsigned short*)q->tr;
    a.do_row = (const unsigned short*)d_out->row; a.do_us = d_out->unscale; a.do_tr = (const unsigned short*)d_out->tr;
    a.k_row = (const unsigned short*)k->row; a.k_us = k->unscale; a.k_tr = (const unsigned short*)k->tr;
    a.v_row = (const unsigned short*)v->row; a.v_us = v->unscale;
    a.lse = const_cast<float*>(lse); a.delta = delta;
    const int form = q->tr_form;
    IX_CHECK_ARG((form == 0 || form == 1) && k->tr_form == form && d_out->tr_form == form,
                 "ix_flash_bwd_f32: operands were split with different tr forms");
    a.o1 = gq; a.ld1 = ld_q; a.off1 = off_q; a.o2 = gk; a.ld2 = ld_k; a.off2 = off_k; a.o3 = gv; a.ld3 = ld_v; a.off3 = off_v;
    // algorithmic FLOPs = the products the REFERENCE's graph evaluates (first derivative of softmax(q k^T) v: dP = dO v^T,
    // dV = P^T dO, dQ = dS k, dK = dS^T q -- four; two per kernel); the recomputed S (and gd in the key-owning pass) only count as
    // executed matrix instructions (third argument)
    if (gq) {   // S, gd, gQ
        dim3 grid((L + 127) / 128, n * H);
        ix_prof_begin(stream, 2, 2.0 * FL_PRODUCT_FLOPS, (2 * 3 + 1 * FL_S2(form)) * FL_PRODUCT_FLOPS, 2);
        if (m16) { if (g_fl_one) fl16_launch_bwd_q_one(a, grid, stream); else fl16_launch_bwd_q(a, grid, stream); }
        else { FL_DISPATCH(flash_bwd_q_kernel, grid, form) }
        ix_prof_end(stream);
    }
    if (gk && gv) {   // S, gd, gK, gV
        dim3 grid((S + 127) / 128, n * H);
        ix_prof_begin(stream, 2, 2.0 * FL_PRODUCT_FLOPS, (2 * 3 + 2 * FL_S2(form)) * FL_PRODUCT_FLOPS, 3);
        if (m16) { if (g_fl_one) fl16_launch_bwd_kv_one(a, grid, stream); else fl16_launch_bwd_kv(a, grid, stream); }
        else { FL_DISPATCH(flash_bwd_kv_kernel, grid, form) }
        ix_prof_end(stream);
    }
    IX_CHECK_LAUNCH("ix_flash_bwd_f32");
    return IX_OK;
}

extern "C" int ix_flash_bwd_bwd_f32(const ix_attn_planes* q, const ix_attn_planes* k, const ix_attn_planes* v,
                                    const ix_attn_planes* d_out, const ix_attn_planes* hq, const ix_attn_planes* hk,
                                    const ix_attn_planes* hv, const float* bias, const float* lse, const float* delta, float* dq,
                                    float* dk, float* dv, float* ddo, int n, int H, int L, int Lp, int S, int Sp, int hd,
                                    int64_t ld_q, int off_q, int64_t ld_k, int off_k, int64_t ld_v, int off_v, int64_t ld_do,
                                    int off_do, float scale, float p_drop, uint64_t seed, void* workspace, size_t workspace_bytes,
                                    hipStream_t stream) {
    if (n <= 0 || L <= 0 || S <= 0) return IX_OK;
    const ix_attn_planes* ops[7] = {q, k, v, d_out, hq, hk, hv};
    for (int i = 0; i < 7; ++i)
        IX_CHECK_ARG(ops[i] && ops[i]->row && ops[i]->unscale, "ix_flash_bwd_bwd_f32: operand %d lacks planes", i);
    const int form = q->tr_form;
    const bool m16 = fl_m16(hd, form);
    for (int i = 0; i < 7; ++i) IX_CHECK_ARG(m16 || ops[i]->tr, "ix_flash_bwd_bwd_f32: operand %d lacks tr planes", i);
    for (int i = 0; i < 7; ++i)
        IX_CHECK_ARG((form == 0 || form == 1) && ops[i]->tr_form == form, "ix_flash_bwd_bwd_f32: operand %d was split with another tr form", i);
    IX_CHECK_ARG(lse && delta && dq && dk && dv && ddo, "ix_flash_bwd_bwd_f32: null pointer");
    IX_CHECK_ARG(FL_OUT_OK(ld_q, off_q) && FL_OUT_OK(ld_k, off_k) && FL_OUT_OK(ld_v, off_v) && FL_OUT_OK(ld_do, off_do),
                 "ix_flash_bwd_bwd_f32: output rows must be 16-byte aligned");
    const size_t need = (size_t)2 * n * H * Lp * sizeof(float);
    if (!workspace || workspace_bytes < need) {
        ix_set_error("ix_flash_bwd_bwd_f32: workspace of %zu bytes needed, %zu given", need, workspace_bytes);
        return IX_ERR_WORKSPACE;
    }
    FlashArgs a;
    const int rc = fl_common(a, "ix_flash_bwd_bwd_f32", bias, m16, n, H, L, Lp, S, Sp, hd, scale, p_drop, seed);
    if (rc) return rc;
#define FL_SET(NAME, SRC) a.NAME##_row = (const unsigned short*)(SRC)->row; a.NAME##_us = (SRC)->unscale; a.NAME##_tr = (const unsigned short*)(SRC)->tr;
    FL_SET(q, q) FL_SET(k, k) FL_SET(v, v) FL_SET(do, d_out) FL_SET(hq, hq) FL_SET(hk, hk) FL_SET(hv, hv)
#undef FL_SET
    a.lse = const_cast<float*>(lse); a.delta = delta;
    a.u = (float*)workspace; a.w = a.u + (size_t)n * H * Lp;
    a.o1 = dq; a.ld1 = ld_q; a.off1 = off_q; a.o2 = dk; a.ld2 = ld_k; a.off2 = off_k;
    a.o3 = dv; a.ld3 = ld_v; a.off3 = off_v; a.o4 = ddo; a.ld4 = ld_do; a.off4 = off_do;
    const dim3 gq((L + 127) / 128, n * H), gk((S + 127) / 128, n * H), blk(256);
    // tiles formed per pass (G counts twice: hq k^T + q hk^T): statistics 5; dq + ddO 5 + 4; dk + dv 5 + 3 -- the executed
    // matrix work (third argument).  Algorithmic FLOPs = the ten products of the materialised double-backward graph (hipops.
    // AttentionCoreBwd.backward = what autograd evaluates for the reference): G 2 + HD 1 in the statistics pass, the 4 and 3
    // output products of the other two; recomputed S / gd / G / HD tiles are not algorithmic work
#define FL_BB_LAUNCH2(HD_, DR_, TP_)                                                                   \
    ix_prof_begin(stream, 2, 3.0 * FL_PRODUCT_FLOPS, (5 * 3) * FL_PRODUCT_FLOPS, 4);                      \
    hipLaunchKernelGGL((flash_bb_q_kernel<HD_, DR_, true, 3>), gq, blk, 0, stream, a);                 \
    ix_prof_end(stream);                                                                               \
    ix_prof_begin(stream, 2, 4.0 * FL_PRODUCT_FLOPS, (5 * 3 + 4 * FL_S2(form)) * FL_PRODUCT_FLOPS, 5);    \
    hipLaunchKernelGGL((flash_bb_q_kernel<HD_, DR_, false, TP_>), gq, blk, 0, stream, a);              \
    ix_prof_end(stream);                                                                               \
    ix_prof_begin(stream, 2, 3.0 * FL_PRODUCT_FLOPS, (5 * 3 + 3 * FL_S2(form)) * FL_PRODUCT_FLOPS, 6);    \
    hipLaunchKernelGGL((flash_bb_kv_kernel<HD_, DR_, TP_>), gk, blk, 0, stream, a);                    \
    ix_prof_end(stream);
#define FL_BB_LAUNCH(HD_, DR_) if (form == 1) { FL_BB_LAUNCH2(HD_, DR_, 2) } else { FL_BB_LAUNCH2(HD_, DR_, 3) }
    if (m16) {
        ix_prof_begin(stream, 2, 3.0 * FL_PRODUCT_FLOPS, (5 * 3) * FL_PRODUCT_FLOPS, 4);
        if (g_fl_one) fl16_launch_bb_stats_one(a, gq, stream); else fl16_launch_bb_stats(a, gq, stream);
        ix_prof_end(stream);
        ix_prof_begin(stream, 2, 4.0 * FL_PRODUCT_FLOPS, (5 * 3 + 4 * 3) * FL_PRODUCT_FLOPS, 5);
        if (g_fl_one) fl16_launch_bb_q_one(a, gq, stream); else fl16_launch_bb_q(a, gq, stream);
        ix_prof_end(stream);
        ix_prof_begin(stream, 2, 3.0 * FL_PRODUCT_FLOPS, (5 * 3 + 3 * 3) * FL_PRODUCT_FLOPS, 6);
        if (g_fl_one) fl16_launch_bb_kv_one(a, gk, stream); else fl16_launch_bb_kv(a, gk, stream);
        ix_prof_end(stream);
    } else if (hd == 64) {
        if (a.thr16) { FL_BB_LAUNCH(64, true) } else { FL_BB_LAUNCH(64, false) }
    } else {
        if (a.thr16) { FL_BB_LAUNCH(32, true) } else { FL_BB_LAUNCH(32, false) }
    }
#undef FL_BB_LAUNCH
#undef FL_BB_LAUNCH2
    IX_CHECK_LAUNCH("ix_flash_bwd_bwd_f32");
    return IX_OK;
}

extern "C" int ix_workspace_bytes_flash_bwd_bwd(int n, int H, int L, size_t* out) {
    IX_CHECK_ARG(out && n >= 0 && H >= 0 && L >= 0, "ix_workspace_bytes_flash_bwd_bwd: bad args");
    *out = (size_t)2 * n * H * ((L + 127) / 128 * 128) * sizeof(float);
    return IX_OK;
}

// The flash kernels' dropout mask as a tensor (tests only): m[bh][q][key] = 1/keep where kept, 0 where dropped.
__global__ void flash_dropmask_kernel(float* __restrict__ m, int L, int S, unsigned thr16, float inv_keep, unsigned seed_lo,
                                      unsigned seed_hi, const unsigned* __restrict__ salt) {
    if (salt) { seed_lo ^= salt[0]; seed_hi ^= salt[1]; }
    const int bh = blockIdx.z, q = blockIdx.y;
    for (int key = blockIdx.x * blockDim.x + threadIdx.x; key < S; key += gridDim.x * blockDim.x) {
        const unsigned hsh = fl_hash(seed_lo, seed_hi, (unsigned)(bh * L + q), (unsigned)key >> 1);
        m[((int64_t)bh * L + q) * S + key] = ((key & 1) ? (hsh >> 16) : (hsh & 0xffffu)) >= thr16 ? inv_keep : 0.f;
    }
}

extern "C" int ix_flash_dropmask_f32(float* m, int BH, int L, int S, float p_drop, uint64_t seed, hipStream_t stream) {
    IX_CHECK_ARG(m && BH > 0 && L > 0 && S > 0 && L <= 65535 && BH <= 65535, "ix_flash_dropmask_f32: bad args");
    const unsigned thr16 = (unsigned)((double)p_drop * 65536.0 + 0.5);
    const float inv_keep = thr16 ? 65536.f / (float)(65536u - thr16) : 1.f;
    hipLaunchKernelGGL(flash_dropmask_kernel, dim3(ix_div_up(S, 256), L, BH), dim3(256), 0, stream, m, L, S, thr16, inv_keep,
                       (unsigned)seed, (unsigned)(seed >> 32), reinterpret_cast<const unsigned*>(ix_g_salt));
    IX_CHECK_LAUNCH("ix_flash_dropmask_f32");
    return IX_OK;
}

// ============================================================================================================
// fp8 forward (opt-in; BASELINE.json configs[4]: 1600 long edge, 200 queries, "fp8 MFMA attention"): the two products of
// the forward pass -- q k^T and P v (reference models/gpt.py:48-53) -- on v_mfma_f32_32x32x16_fp8_fp8 (OCP e4m3 operands,
// fp32 accumulate), ONE matrix instruction per 16 contracted elements instead of three / six.  Operands are quantised
// once by ix_attn_split_fp8_f32 (one plane, per-32-row block scale so that the block maximum lands in [128, 256)), the
// probabilities in registers (x 256).  Softmax statistics, the output and lse stay fp32; the derivative kernels above are
// unchanged (they recompute P from the fp16 planes), so this is an approximation of the FORWARD values only, with a
// stated tolerance (tests/test_ops_gpu.py::test_flash_forward_fp8).
// ============================================================================================================
__device__ __forceinline__ unsigned fl_pack_fp8x4(float a, float b, float c, float d) {
    int w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    return (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
}

template <int HD>
__global__ __launch_bounds__(256) void attn_split_fp8_kernel(const float* __restrict__ X, unsigned char* __restrict__ rowp,
                                                             float* __restrict__ unscale, unsigned char* __restrict__ trp, int R,
                                                             int Rp, int64_t ld, int off, int H) {
    constexpr int EPT = 32 * HD / 256, TPR = HD / EPT;
    __shared__ __attribute__((aligned(16))) unsigned char lt[HD][32 + 8];   // [d][permuted row]
    __shared__ float red[4];
    const int tid = threadIdx.x, r0 = blockIdx.x * 32, bh = blockIdx.y;
    const int b = bh / H, h = bh % H;
    const int row = tid / TPR, c0 = (tid % TPR) * EPT;
    float v[EPT];
    const bool in = r0 + row < R;
    const float* src = X + ((int64_t)b * R + r0 + row) * ld + off + h * HD + c0;
#pragma unroll
    for (int i = 0; i < EPT; i += 4) {
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        if (in) t = *reinterpret_cast<const float4*>(src + i);
        v[i] = t.x; v[i + 1] = t.y; v[i + 2] = t.z; v[i + 3] = t.w;
    }
    float mx = 0.f;
#pragma unroll
    for (int i = 0; i < EPT; ++i) mx = fmaxf(mx, fabsf(v[i]));
    mx = ix_block_max_256(mx, red);
    const unsigned e = (__float_as_uint(mx) >> 23) & 0xffu;
    const bool tiny = e < 16u || e > 250u;
    const float sc = tiny ? 1.f : __uint_as_float((261u - e) << 23);   // 2^(7 - E): block maximum in [128, 256) (e4m3 max 448)
    const float us = tiny ? 1.f : __uint_as_float((e - 7u) << 23);
    if (tid == 0) unscale[(int64_t)bh * (Rp / 32) + blockIdx.x] = us;
    unsigned w[EPT / 4];
#pragma unroll
    for (int i = 0; i < EPT / 4; ++i) w[i] = fl_pack_fp8x4(v[4 * i] * sc, v[4 * i + 1] * sc, v[4 * i + 2] * sc, v[4 * i + 3] * sc);
    if (rowp) {
        unsigned* dst = reinterpret_cast<unsigned*>(rowp + ((int64_t)bh * Rp + r0 + row) * HD + c0);
#pragma unroll
        for (int i = 0; i < EPT / 4; ++i) dst[i] = w[i];
    }
    if (!trp) return;
    const int prow = (row & 16) | fl_perm16(row & 15);
#pragma unroll
    for (int i = 0; i < EPT; ++i) lt[c0 + i][prow] = (unsigned char)(w[i / 4] >> (8 * (i & 3)));
    __syncthreads();
    for (int c = tid; c < HD * 4; c += 256) {   // HD rows of 32 bytes = four 8-byte chunks
        const int d = c / 4, ch = c % 4;
        *reinterpret_cast<uint2*>(trp + ((int64_t)bh * HD + d) * Rp + r0 + ch * 8) = *reinterpret_cast<const uint2*>(&lt[d][ch * 8]);
    }
}

extern "C" int ix_attn_split_fp8_f32(const float* x, void* row_plane, void* tr_plane, float* unscale, int n, int R, int Rp,
                                     int64_t ld, int off, int H, int hd, hipStream_t stream) {
    if (n <= 0 || R <= 0) return IX_OK;
    IX_CHECK_ARG(x && unscale && (row_plane || tr_plane), "ix_attn_split_fp8_f32: null pointer");
    IX_CHECK_ARG(hd == 32 || hd == 64, "ix_attn_split_fp8_f32: head dim %d (32 or 64)", hd);
    IX_CHECK_ARG(Rp % 128 == 0 && Rp >= R, "ix_attn_split_fp8_f32: Rp=%d must be R=%d rounded up to 128", Rp, R);
    IX_CHECK_ARG(ld % 4 == 0 && off % 4 == 0 && ((uintptr_t)x & 15) == 0, "ix_attn_split_fp8_f32: rows must be 16-byte aligned");
    dim3 grid(Rp / 32, n * H);
    if (hd == 64)
        hipLaunchKernelGGL(attn_split_fp8_kernel<64>, grid, dim3(256), 0, stream, x, (unsigned char*)row_plane, unscale,
                           (unsigned char*)tr_plane, R, Rp, ld, off, H);
    else
        hipLaunchKernelGGL(attn_split_fp8_kernel<32>, grid, dim3(256), 0, stream, x, (unsigned char*)row_plane, unscale,
                           (unsigned char*)tr_plane, R, Rp, ld, off, H);
    IX_CHECK_LAUNCH("ix_attn_split_fp8_f32");
    return IX_OK;
}

struct FlashFp8Args {
    const unsigned char *q_row, *k_row, *v_tr;   // [BH][Lp][hd], [BH][Sp][hd], [BH][hd][Sp] (e4m3)
    const float *q_us, *k_us, *v_us, *bias;
    float *out, *lse;
    int H, L, Lp, S, Sp;
    int64_t ld_out;
    int off_out;
    float scale_log2e;
    unsigned thr16;
    float inv_keep;
    unsigned seed_lo, seed_hi;
    const unsigned* salt;       // optional device word XORed into the seed when the kernel runs (ix_set_dropout_salt)
};

template <int HD, bool DROP>
__global__ __launch_bounds__(256, 2) void flash_fwd_fp8_kernel(FlashFp8Args p) {
    if (DROP && p.salt) { p.seed_lo ^= p.salt[0]; p.seed_hi ^= p.salt[1]; }
    constexpr int NKS = HD / 16, NDB = HD / 32;
    constexpr int KROW = HD + 8, VROW = 32 + 8, OFF_VT = 32 * KROW, BYTES = OFF_VT + HD * VROW;   // 8-byte padded rows
    constexpr int NCH = 32 * HD / 8;   // 8-byte chunks of the K tile (= of the V tile): 256 at hd 64, 128 at hd 32
    __shared__ __attribute__((aligned(16))) unsigned char ldsb[2][BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane & 31, a = lane >> 5;
    const int bh = blockIdx.y, b = bh / p.H, h = bh % p.H;
    const int q0 = blockIdx.x * 128 + wave * 32;
    const int ntiles = (p.S + 31) / 32;
    const unsigned char* krow = p.k_row + (int64_t)bh * p.Sp * HD;
    const unsigned char* vtr = p.v_tr + (int64_t)bh * HD * p.Sp;
    const float* bias = p.bias + (int64_t)b * p.Sp;
    const float* kus = p.k_us + (int64_t)bh * (p.Sp / 32);
    const float* vus = p.v_us + (int64_t)bh * (p.Sp / 32);

    long qf[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
        qf[ks] = *reinterpret_cast<const long*>(p.q_row + ((int64_t)bh * p.Lp + q0 + lq) * HD + ks * 16 + 8 * a);
    const float c2 = p.scale_log2e * p.q_us[(int64_t)bh * (p.Lp / 32) + q0 / 32];
    f32x16 o[NDB];
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
    float m = -1e30f, l = 0.f;
    const unsigned rid = (unsigned)(bh * p.L + q0 + lq);

    // staging: one 8-byte chunk of the K tile and one of the V tile per thread (hd 64); hd 32: threads 0..127 K, 128..255 V
    uint2 s0, s1;
    const int c = HD == 64 ? tid : (tid & 127);
    const bool doK = HD == 64 || tid < 128, doV = HD == 64 || tid >= 128;
#define FL8_LOAD(T0)                                                                                                   \
    if (doK) s0 = *reinterpret_cast<const uint2*>(krow + (int64_t)(T0) * HD + c * 8);                                  \
    if (doV) s1 = *reinterpret_cast<const uint2*>(vtr + (int64_t)(c >> 2) * p.Sp + (T0) + (c & 3) * 8);
#define FL8_STORE(BUF)                                                                                                 \
    if (doK) *reinterpret_cast<uint2*>((BUF) + (c / (HD / 8)) * KROW + (c % (HD / 8)) * 8) = s0;                       \
    if (doV) *reinterpret_cast<uint2*>((BUF) + OFF_VT + (c >> 2) * VROW + (c & 3) * 8) = s1;
    (void)NCH;
    s0 = make_uint2(0, 0); s1 = make_uint2(0, 0);
    FL8_LOAD(0)
    FL8_STORE(ldsb[0])
    __syncthreads();

    for (int t = 0; t < ntiles; ++t) {
        const unsigned char* lds = ldsb[t & 1];
        const int t0 = t * 32;
        f32x4 kb[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) kb[g] = *reinterpret_cast<const f32x4*>(bias + t0 + 8 * g + 4 * a);
        const float cs = c2 * kus[t], cv = vus[t] * (1.f / 256.f);
        FL8_LOAD(min(t0 + 32, ntiles * 32 - 32))
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const long kf = *reinterpret_cast<const long*>(lds + lq * KROW + ks * 16 + 8 * a);
            s = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(kf, qf[ks], s, 0, 0, 0);
        }
        float x[16], tmax = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            x[r] = s[r] * cs + kb[r >> 2][r & 3];
            tmax = fmaxf(tmax, x[r]);
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float mn = fmaxf(m, tmax);
        const float alpha = fl_exp2(m - mn);
        m = mn;
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            x[r] = fl_exp2(x[r] - mn);
            ps += x[r];
        }
        l = l * alpha + ps;
        if (__builtin_amdgcn_ballot_w64(alpha != 1.f)) {
#pragma unroll
            for (int db = 0; db < NDB; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
        }
        if (DROP) {
            bool kp[16];
            FL_MASK_KEYS_IN_REGS(kp, rid, t0)
#pragma unroll
            for (int r = 0; r < 16; ++r) x[r] = kp[r] ? x[r] : 0.f;
        }
        // P^T x 256 -> e4m3: accumulator registers 8 s2 + j are the B fragment of key slice s2
        long pf[2];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const unsigned lo = fl_pack_fp8x4(x[8 * s2] * 256.f, x[8 * s2 + 1] * 256.f, x[8 * s2 + 2] * 256.f, x[8 * s2 + 3] * 256.f);
            const unsigned hi = fl_pack_fp8x4(x[8 * s2 + 4] * 256.f, x[8 * s2 + 5] * 256.f, x[8 * s2 + 6] * 256.f, x[8 * s2 + 7] * 256.f);
            pf[s2] = (long)(((unsigned long)hi << 32) | lo);
        }
        // O^T += (V^T . P^T) x (V block unscale / 256): the V tile carries its own block scale, so it goes through a
        // zero-initialised accumulator and one multiply-add per output register
#pragma unroll
        for (int db = 0; db < NDB; ++db) {
            f32x16 tmp;
#pragma unroll
            for (int r = 0; r < 16; ++r) tmp[r] = 0.f;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const long vf = *reinterpret_cast<const long*>(lds + OFF_VT + (db * 32 + lq) * VROW + s2 * 16 + 8 * a);
                tmp = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(vf, pf[s2], tmp, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) o[db][r] += tmp[r] * cv;
        }
        FL8_STORE(ldsb[(t + 1) & 1])
        __syncthreads();
    }
#undef FL8_LOAD
#undef FL8_STORE
    l += __shfl_xor(l, 32, 64);
    const int q = q0 + lq;
    if (q < p.L) {
        const float inv = p.inv_keep / l;
        float* dst = p.out + ((int64_t)b * p.L + q) * p.ld_out + p.off_out + h * HD + 4 * a;
        FL_STORE_ROWS(o, dst, inv)
        if (a == 0) p.lse[(int64_t)bh * p.Lp + q] = (m + log2f(l)) * FL_LN2;
    } else if (a == 0) {
        p.lse[(int64_t)bh * p.Lp + q] = INFINITY;
    }
}

extern "C" int ix_flash_fwd_fp8_f32(const void* q_row8, const float* q_unscale, const void* k_row8, const float* k_unscale,
                                    const void* v_tr8, const float* v_unscale, const float* bias, float* out, float* lse, int n,
                                    int H, int L, int Lp, int S, int Sp, int hd, int64_t ld_out, int off_out, float scale,
                                    float p_drop, uint64_t seed, hipStream_t stream) {
    if (n <= 0 || L <= 0) return IX_OK;
    IX_CHECK_ARG(q_row8 && q_unscale && k_row8 && k_unscale && v_tr8 && v_unscale && bias && out && lse, "ix_flash_fwd_fp8_f32: null pointer");
    IX_CHECK_ARG(hd == 32 || hd == 64, "ix_flash_fwd_fp8_f32: head dim %d (32 or 64)", hd);
    IX_CHECK_ARG(S > 0 && Lp % 128 == 0 && Sp % 128 == 0 && Lp >= L && Sp >= S, "ix_flash_fwd_fp8_f32: bad padded sizes");
    IX_CHECK_ARG(FL_OUT_OK(ld_out, off_out) && ((uintptr_t)out & 15) == 0, "ix_flash_fwd_fp8_f32: output rows must be 16-byte aligned");
    IX_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f, "ix_flash_fwd_fp8_f32: p=%f outside [0,1)", p_drop);
    IX_CHECK_ARG((int64_t)n * H * L < ((int64_t)1 << 32) && n * H <= 65535, "ix_flash_fwd_fp8_f32: too many rows");
    FlashFp8Args a;
    a.q_row = (const unsigned char*)q_row8; a.k_row = (const unsigned char*)k_row8; a.v_tr = (const unsigned char*)v_tr8;
    a.q_us = q_unscale; a.k_us = k_unscale; a.v_us = v_unscale; a.bias = bias; a.out = out; a.lse = lse;
    a.H = H; a.L = L; a.Lp = Lp; a.S = S; a.Sp = Sp; a.ld_out = ld_out; a.off_out = off_out;
    a.scale_log2e = scale * FL_LOG2E;
    a.thr16 = (unsigned)((double)p_drop * 65536.0 + 0.5);
    a.inv_keep = a.thr16 ? 65536.f / (float)(65536u - a.thr16) : 1.f;
    a.seed_lo = (unsigned)seed; a.seed_hi = (unsigned)(seed >> 32);
    a.salt = reinterpret_cast<const unsigned*>(ix_g_salt);
    dim3 grid((L + 127) / 128, n * H);
    ix_prof_begin(stream, 2, 2.0 * FL_PRODUCT_FLOPS, 2.0 * FL_PRODUCT_FLOPS, 1);
    if (hd == 64) {
        if (a.thr16) hipLaunchKernelGGL((flash_fwd_fp8_kernel<64, true>), grid, dim3(256), 0, stream, a);
        else hipLaunchKernelGGL((flash_fwd_fp8_kernel<64, false>), grid, dim3(256), 0, stream, a);
    } else {
        if (a.thr16) hipLaunchKernelGGL((flash_fwd_fp8_kernel<32, true>), grid, dim3(256), 0, stream, a);
        else hipLaunchKernelGGL((flash_fwd_fp8_kernel<32, false>), grid, dim3(256), 0, stream, a);
    }
    ix_prof_end(stream);
    IX_CHECK_LAUNCH("ix_flash_fwd_fp8_f32");
    return IX_OK;
}
