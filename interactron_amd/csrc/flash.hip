// Flash-style attention on the bf16 matrix cores with fp32-grade arithmetic: forward, backward and double backward of
//     O = dropout(softmax(scale Q K^T + key bias)) V        per (batch, head)
// without ever writing an [L, S] tensor to HBM.  Replaces the attention core of reference models/gpt.py:39-57
// (att = softmax(q k^T / sqrt(hd)); att = drop(att); y = att v) and of nn.MultiheadAttention as called from
// models/detr_models/transformer.py:148-161,211-232, plus the autograd derivatives the MAML meta-gradient takes of it
// (models/interactron.py:99-123: grad(create_graph=True) then backward).
//
// Arithmetic.  Every fp32 operand element is split EXACTLY into three bf16 values x = h + m + l (24 significant
// bits), a product of two operands is the six MFMA terms  l.h + h.l + m.m + m.h + h.m + h.h  on
// v_mfma_f32_32x32x16_bf16 with fp32 accumulation -- the same scheme, and the same accuracy class, as the bf16x6
// contraction kernel in gemm.hip.  Operands that come from HBM (q, k, v, dO and the second-order cotangents) are split
// ONCE by ix_attn_split_f32 into bf16 planes in the two layouts the kernels consume; [L, S]-shaped intermediates
// (probabilities, score cotangents) are split in registers, straight out of the MFMA accumulators.
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
//   row layout  [plane][batch*head][Rp][hd]   bf16   fragment of 8 consecutive d of one row     (contraction over d)
//   tr  layout  [plane][batch*head][hd][Rp]   bf16   rows permuted within 16-groups             (contraction over rows)
//   Rp = R rounded up to 128 (zero rows); keys are walked in tiles of 32 up to ceil(S / 32) * 32.
#include "common.h"
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define FL_LOG2E 1.4426950408889634f
#define FL_LN2 0.6931471805599453f

__device__ __forceinline__ unsigned fl_pack(float a, float b) {   // v_cvt_pk_bf16_f32 (RNE), a in the low half
    f32x2 v;
    v.x = a;
    v.y = b;
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

// (x0, x1) -> packed bf16 pairs of the three planes, exact: x = h + m + l
__device__ __forceinline__ void fl_split3(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    h = fl_pack(x0, x1);
    const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);
    m = fl_pack(r0, r1);
    const float s0 = r0 - __uint_as_float(m << 16), s1 = r1 - __uint_as_float(m & 0xffff0000u);
    l = fl_pack(s0, s1);
}

__device__ __forceinline__ f32x16 fl_mfma(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// position of row r (0..15) of a 16-group in the tr layout: bits 2 and 3 swapped
__device__ __host__ __forceinline__ int fl_perm16(int r) { return (r & 3) | ((r & 4) << 1) | ((r & 8) >> 1); }

// ------------------------------------------------------------------------------------------------------------
// split kernel: fp32 [n][R][ld] (head h at columns off + h*hd) -> bf16 planes in row and tr layout
// ------------------------------------------------------------------------------------------------------------
template <int HD>
__global__ __launch_bounds__(256) void attn_split_kernel(const float* __restrict__ X, unsigned short* __restrict__ rowp,
                                                         unsigned short* __restrict__ trp, int R, int Rp, int64_t ld,
                                                         int off, int H, int64_t plane_elems) {
    constexpr int EPT = 32 * HD / 256;          // elements per thread: 8 (hd 64) / 4 (hd 32)
    constexpr int TPR = HD / EPT;               // threads per row: 8
    __shared__ __attribute__((aligned(16))) unsigned short lt[3][HD][32 + 8];   // [plane][d][permuted row], 80-byte rows
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
    unsigned ph[EPT / 2], pm[EPT / 2], pl[EPT / 2];
#pragma unroll
    for (int i = 0; i < EPT / 2; ++i) fl_split3(v[2 * i], v[2 * i + 1], ph[i], pm[i], pl[i]);
    // row layout: EPT consecutive bf16 of one row
    if (rowp) {
        unsigned short* dst = rowp + ((int64_t)bh * Rp + r0 + row) * HD + c0;
#pragma unroll
        for (int i = 0; i < EPT / 2; ++i) {
            reinterpret_cast<unsigned*>(dst)[i] = ph[i];
            reinterpret_cast<unsigned*>(dst + plane_elems)[i] = pm[i];
            reinterpret_cast<unsigned*>(dst + 2 * plane_elems)[i] = pl[i];
        }
    }
    if (!trp) return;
    const int prow = (row & 16) | fl_perm16(row & 15);
#pragma unroll
    for (int i = 0; i < EPT / 2; ++i) {
        lt[0][c0 + 2 * i][prow] = (unsigned short)(ph[i] & 0xffff); lt[0][c0 + 2 * i + 1][prow] = (unsigned short)(ph[i] >> 16);
        lt[1][c0 + 2 * i][prow] = (unsigned short)(pm[i] & 0xffff); lt[1][c0 + 2 * i + 1][prow] = (unsigned short)(pm[i] >> 16);
        lt[2][c0 + 2 * i][prow] = (unsigned short)(pl[i] & 0xffff); lt[2][c0 + 2 * i + 1][prow] = (unsigned short)(pl[i] >> 16);
    }
    __syncthreads();
    // tr layout: per plane HD rows of 32 bf16 (64 bytes = four 16-byte chunks)
    for (int c = tid; c < 3 * HD * 4; c += 256) {
        const int pl_ = c / (HD * 4), d = (c / 4) % HD, ch = c % 4;
        const uint4 val = *reinterpret_cast<const uint4*>(&lt[pl_][d][ch * 8]);
        *reinterpret_cast<uint4*>(trp + pl_ * plane_elems + ((int64_t)bh * HD + d) * Rp + r0 + ch * 8) = val;
    }
}

extern "C" int ix_attn_split_f32(const float* x, void* row_planes, void* tr_planes, int n, int R, int Rp, int64_t ld, int off,
                                 int H, int hd, hipStream_t stream) {
    if (n <= 0 || R <= 0) return IX_OK;
    IX_CHECK_ARG(x && (row_planes || tr_planes), "ix_attn_split_f32: null pointer");
    IX_CHECK_ARG(hd == 32 || hd == 64, "ix_attn_split_f32: head dim %d (32 or 64)", hd);
    IX_CHECK_ARG(Rp % 128 == 0 && Rp >= R, "ix_attn_split_f32: Rp=%d must be R=%d rounded up to 128", Rp, R);
    IX_CHECK_ARG(ld % 4 == 0 && off % 4 == 0 && ((uintptr_t)x & 15) == 0, "ix_attn_split_f32: rows must be 16-byte aligned");
    const int64_t plane = (int64_t)n * H * Rp * hd;
    dim3 grid(Rp / 32, n * H);
    if (hd == 64)
        hipLaunchKernelGGL(attn_split_kernel<64>, grid, dim3(256), 0, stream, x, (unsigned short*)row_planes,
                           (unsigned short*)tr_planes, R, Rp, ld, off, H, plane);
    else
        hipLaunchKernelGGL(attn_split_kernel<32>, grid, dim3(256), 0, stream, x, (unsigned short*)row_planes,
                           (unsigned short*)tr_planes, R, Rp, ld, off, H, plane);
    IX_CHECK_LAUNCH("ix_attn_split_f32");
    return IX_OK;
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
// dropout mask of the flash kernels: a pure function of (seed, row id = (batch*head)*L + query, key), one 32-bit hash
// per PAIR of neighbouring keys (two 16-bit draws), so forward, backward and double backward regenerate the same mask
// whichever way their tiles are oriented.  keep <=> draw >= thr16, thr16 = round(p * 65536).
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned fl_hash(unsigned seed_lo, unsigned seed_hi, unsigned rid, unsigned kpair) {
    unsigned x = seed_lo ^ (rid * 0x9E3779B1u) ^ (kpair * 0x85EBCA77u);
    x ^= x >> 16;
    x *= 0x7FEB352Du;
    x ^= seed_hi;
    x ^= x >> 15;
    x *= 0x846CA68Bu;
    x ^= x >> 16;
    return x;
}

struct FlashArgs {
    const unsigned short* q_row;   // query-side row planes  [3][BH][Lp][hd]
    const unsigned short* k_row;   // key-side row planes    [3][BH][Sp][hd]
    const unsigned short* v_tr;    // value tr planes        [3][BH][hd][Sp]
    const float* bias;             // [n][Sp] additive key bias (0 / -inf)
    float* out;                    // [n][L][ld_out], head h at off_out + h*hd
    float* lse;                    // [BH][Lp] natural-log row normalisers
    int H, L, Lp, S, Sp, Sb;
    int64_t ld_out;
    int off_out;
    int64_t q_plane, k_plane;      // elements per plane
    float scale_log2e;             // softmax scale * log2(e)
    unsigned thr16;                // dropout threshold (0 = no dropout)
    float inv_keep;
    unsigned seed_lo, seed_hi;
};

// LDS image of one key tile (32 keys): K row planes [3][32][HD + 8] then V tr planes [3][HD][32 + 8] (bf16); the 16-byte
// row padding keeps ds_read_b128 fragment reads conflict-free (row pitch / 16 odd).
template <int HD>
struct FlTile {
    static constexpr int KROW = (HD + 8) * 2;            // bytes per key row
    static constexpr int KPLANE = 32 * KROW;
    static constexpr int VROW = (32 + 8) * 2;            // bytes per d row
    static constexpr int VPLANE = HD * VROW;
    static constexpr int BYTES = 3 * (KPLANE + VPLANE);
    static constexpr int JOBS = 8 * HD / 256;            // 16-byte chunks per thread per plane (K: 4*HD chunks, V: 4*HD)
};

// global -> registers (issued early) and registers -> LDS (after the tile that is being computed): T14 async stage
template <int HD>
struct FlStage {
    uint4 v[3 * FlTile<HD>::JOBS];
    __device__ __forceinline__ void load(const unsigned short* krow, const unsigned short* vtr, int64_t kplane, int Sp,
                                         int t0, int tid) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int i = 0; i < FlTile<HD>::JOBS; ++i) {
                const int j = tid + 256 * i;
                const unsigned short* src;
                if (j < 4 * HD) {   // K: 32 rows of HD bf16, contiguous
                    src = krow + pl * kplane + (int64_t)t0 * HD + j * 8;
                } else {            // V tr: HD rows, 32 keys (64 bytes) each
                    const int jj = j - 4 * HD;
                    src = vtr + pl * kplane + (int64_t)(jj >> 2) * Sp + t0 + (jj & 3) * 8;
                }
                v[pl * FlTile<HD>::JOBS + i] = *reinterpret_cast<const uint4*>(src);
            }
    }
    __device__ __forceinline__ void store(unsigned char* lds, int tid) const {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int i = 0; i < FlTile<HD>::JOBS; ++i) {
                const int j = tid + 256 * i;
                unsigned char* dst;
                if (j < 4 * HD) {
                    constexpr int CPR = HD / 8;   // chunks per key row
                    dst = lds + pl * FlTile<HD>::KPLANE + (j / CPR) * FlTile<HD>::KROW + (j % CPR) * 16;
                } else {
                    const int jj = j - 4 * HD;
                    dst = lds + 3 * FlTile<HD>::KPLANE + pl * FlTile<HD>::VPLANE + (jj >> 2) * FlTile<HD>::VROW + (jj & 3) * 16;
                }
                *reinterpret_cast<uint4*>(dst) = v[pl * FlTile<HD>::JOBS + i];
            }
    }
};

// six-term product of one k-slice: acc += A(3 planes) . B(3 planes), smallest terms first
#define FL_MMA6(ACC, A, B)                                                             \
    ACC = fl_mfma(A[2], B[0], ACC); ACC = fl_mfma(A[0], B[2], ACC); ACC = fl_mfma(A[1], B[1], ACC); \
    ACC = fl_mfma(A[1], B[0], ACC); ACC = fl_mfma(A[0], B[1], ACC); ACC = fl_mfma(A[0], B[0], ACC);

template <int HD, bool DROP>
__global__ __launch_bounds__(256, 2) void flash_fwd_kernel(FlashArgs p) {
    constexpr int NKS = HD / 16, NDB = HD / 32;
    typedef FlTile<HD> T;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2][T::BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane & 31, a = lane >> 5;
    const int bh = blockIdx.y, b = bh / p.H, h = bh % p.H;
    const int q0 = blockIdx.x * 128 + wave * 32;
    const int ntiles = p.Sb / 32;

    const unsigned short* krow = p.k_row + (int64_t)bh * p.Sp * HD;
    const unsigned short* vtr = p.v_tr + (int64_t)bh * HD * p.Sp;
    const float* bias = p.bias + (int64_t)b * p.Sp;

    // query fragments (B operand: column = query lane & 31, 8 consecutive d per lane), all three planes, all k-slices
    u32x4 qf[NKS][3];
    {
        const unsigned short* qsrc = p.q_row + ((int64_t)bh * p.Lp + q0 + lq) * HD + 8 * a;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                qf[ks][pl] = *reinterpret_cast<const u32x4*>(qsrc + pl * p.q_plane + ks * 16);
    }
    f32x16 o[NDB];
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
    float m = -1e30f, l = 0.f;
    const unsigned rid = (unsigned)(bh * p.L + q0 + lq);

    FlStage<HD> st;
    st.load(krow, vtr, p.k_plane, p.Sp, 0, tid);
    st.store(lds[0], tid);
    __syncthreads();

    for (int t = 0; t < ntiles; ++t) {
        const unsigned char* cur = lds[t & 1];
        const int t0 = t * 32;
        // next tile's operands: requested now, written to the other LDS buffer after this tile's products (the last
        // iteration re-requests its own tile: unconditional code keeps the staging registers out of scratch memory)
        // key bias of this lane's 16 keys: register r <-> key t0 + (r & 3) + 8 (r >> 2) + 4 a.  Requested BEFORE the
        // prefetch: vmcnt retires in order, so a wait for these (L1 hits) must not sit behind the next tile's HBM loads
        f32x4 kb[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) kb[g] = *reinterpret_cast<const f32x4*>(bias + t0 + 8 * g + 4 * a);
                st.load(krow, vtr, p.k_plane, p.Sp, min(t0 + 32, p.Sb - 32), tid);
        
        // ---- S^T[key, query] = K . Q^T ----
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            u32x4 kf[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                kf[pl] = *reinterpret_cast<const u32x4*>(cur + pl * T::KPLANE + lq * T::KROW + (ks * 16 + 8 * a) * 2);
            FL_MMA6(s, kf, qf[ks])
        }
        // ---- online softmax (base 2), one query per lane ----
        float x[16], tmax = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            x[r] = s[r] * p.scale_log2e + kb[r >> 2][r & 3];
            tmax = fmaxf(tmax, x[r]);
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float mn = fmaxf(m, tmax);
        const float alpha = exp2f(m - mn);
        m = mn;
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            x[r] = exp2f(x[r] - mn);
            ps += x[r];
        }
        l = l * alpha + ps;
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
        if (DROP) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {   // pair i: registers 2i, 2i+1 = keys k, k+1 (k even)
                const int key = t0 + ((2 * i) & 3) + 8 * ((2 * i) >> 2) + 4 * a;
                const unsigned hsh = fl_hash(p.seed_lo, p.seed_hi, rid, (unsigned)key >> 1);
                x[2 * i] = (hsh & 0xffffu) >= p.thr16 ? x[2 * i] : 0.f;
                x[2 * i + 1] = (hsh >> 16) >= p.thr16 ? x[2 * i + 1] : 0.f;
            }
        }
        // ---- P^T planes: accumulator registers 8 s2 + j are the B fragment of key slice s2 ----
        u32x4 pp[2][3];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                unsigned hh, mm, ll;
                fl_split3(x[8 * s2 + 2 * j], x[8 * s2 + 2 * j + 1], hh, mm, ll);
                pp[s2][0][j] = hh; pp[s2][1][j] = mm; pp[s2][2][j] = ll;
            }
        // ---- O^T[d, query] += V^T[d, key] . P^T[key, query] ----
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int db = 0; db < NDB; ++db) {
                u32x4 vf[3];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    vf[pl] = *reinterpret_cast<const u32x4*>(cur + 3 * T::KPLANE + pl * T::VPLANE + (db * 32 + lq) * T::VROW +
                                                             (s2 * 16 + 8 * a) * 2);
                FL_MMA6(o[db], vf, pp[s2])
            }
        st.store(lds[(t + 1) & 1], tid);
        __syncthreads();
    }
    // ---- epilogue ----
    l += __shfl_xor(l, 32, 64);
    const int q = q0 + lq;
    if (q < p.L) {
        const float inv = p.inv_keep / l;
        float* dst = p.out + ((int64_t)b * p.L + q) * p.ld_out + p.off_out + h * HD + 4 * a;
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v;
                v.x = o[db][4 * g] * inv; v.y = o[db][4 * g + 1] * inv; v.z = o[db][4 * g + 2] * inv; v.w = o[db][4 * g + 3] * inv;
                *reinterpret_cast<f32x4*>(dst + db * 32 + 8 * g) = v;
            }
        if (a == 0) p.lse[(int64_t)bh * p.Lp + q] = (m + log2f(l)) * FL_LN2;
    }
}

extern "C" int ix_flash_fwd_f32(const void* q_row, const void* k_row, const void* v_tr, const float* bias, float* out,
                                float* lse, int n, int H, int L, int Lp, int S, int Sp, int hd, int64_t ld_out, int off_out,
                                float scale, float p_drop, uint64_t seed, hipStream_t stream) {
    if (n <= 0 || L <= 0) return IX_OK;
    IX_CHECK_ARG(q_row && k_row && v_tr && bias && out && lse, "ix_flash_fwd_f32: null pointer");
    IX_CHECK_ARG(hd == 32 || hd == 64, "ix_flash_fwd_f32: head dim %d (32 or 64)", hd);
    IX_CHECK_ARG(S > 0 && Lp % 128 == 0 && Sp % 128 == 0 && Lp >= L && Sp >= S, "ix_flash_fwd_f32: bad padded sizes");
    IX_CHECK_ARG(ld_out % 4 == 0 && off_out % 4 == 0 && ((uintptr_t)out & 15) == 0, "ix_flash_fwd_f32: output rows must be 16-byte aligned");
    IX_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f, "ix_flash_fwd_f32: p=%f outside [0,1)", p_drop);
    IX_CHECK_ARG((int64_t)n * H * L < ((int64_t)1 << 32) && n * H <= 65535, "ix_flash_fwd_f32: too many rows");
    FlashArgs a;
    a.q_row = (const unsigned short*)q_row; a.k_row = (const unsigned short*)k_row; a.v_tr = (const unsigned short*)v_tr;
    a.bias = bias; a.out = out; a.lse = lse;
    a.H = H; a.L = L; a.Lp = Lp; a.S = S; a.Sp = Sp; a.Sb = (S + 31) / 32 * 32;
    a.ld_out = ld_out; a.off_out = off_out;
    a.q_plane = (int64_t)n * H * Lp * hd; a.k_plane = (int64_t)n * H * Sp * hd;
    a.scale_log2e = scale * FL_LOG2E;
    a.thr16 = (unsigned)((double)p_drop * 65536.0 + 0.5);
    a.inv_keep = a.thr16 ? 65536.f / (float)(65536u - a.thr16) : 1.f;
    a.seed_lo = (unsigned)seed; a.seed_hi = (unsigned)(seed >> 32);
    dim3 grid((L + 127) / 128, n * H);
    if (hd == 64) {
        if (a.thr16) hipLaunchKernelGGL((flash_fwd_kernel<64, true>), grid, dim3(256), 0, stream, a);
        else hipLaunchKernelGGL((flash_fwd_kernel<64, false>), grid, dim3(256), 0, stream, a);
    } else {
        if (a.thr16) hipLaunchKernelGGL((flash_fwd_kernel<32, true>), grid, dim3(256), 0, stream, a);
        else hipLaunchKernelGGL((flash_fwd_kernel<32, false>), grid, dim3(256), 0, stream, a);
    }
    IX_CHECK_LAUNCH("ix_flash_fwd_f32");
    return IX_OK;
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

// A tile of 32 rows as staged in LDS: "row" segments [3][32][HD + 8] (fragments along d) and "tr" segments
// [3][HD][32 + 8] (fragments along the 32 rows), bf16.  One unit = one plane of one segment = 4 * HD 16-byte chunks.
template <int HD>
struct FlSeg {
    static constexpr int RROW = (HD + 8) * 2, RPLANE = 32 * RROW, RBYTES = 3 * RPLANE;
    static constexpr int TROW = (32 + 8) * 2, TPLANE = HD * TROW, TBYTES = 3 * TPLANE;
    static constexpr int CH = 4 * HD;   // chunks per unit
    // chunk c of a row-layout unit: global element offset from the tile's first row / LDS byte offset inside the plane
    static __device__ __forceinline__ int64_t row_src(int c) { return (int64_t)c * 8; }
    static __device__ __forceinline__ int row_dst(int c) { return (c / (HD / 8)) * RROW + (c % (HD / 8)) * 16; }
    static __device__ __forceinline__ int64_t tr_src(int c, int Rp) { return (int64_t)(c >> 2) * Rp + (c & 3) * 8; }
    static __device__ __forceinline__ int tr_dst(int c) { return (c >> 2) * TROW + (c & 3) * 16; }
};

struct FlashBwdArgs {
    // row / tr planes of the query side (q, dO) and the key side (k, v); [3][BH][Rp][hd] / [3][BH][hd][Rp]
    const unsigned short *q_row, *q_tr, *do_row, *do_tr, *k_row, *k_tr, *v_row;
    const float* bias;     // [n][Sp]
    const float* lse;      // [BH][Lp]   (+inf beyond L)
    const float* delta;    // [BH][Lp]   t_i = dO_i . O_i
    float *gq, *gk, *gv;   // [n][L][ld_q] at off_q + h*hd;  [n][S][ld_k] at off_k + h*hd;  [n][S][ld_v] at off_v + h*hd
    int64_t ld_q, ld_k, ld_v;
    int off_q, off_k, off_v;
    int H, L, Lp, S, Sp;
    int64_t q_plane, k_plane;
    float scale, scale_log2e;
    unsigned thr16;
    float inv_keep;
    unsigned seed_lo, seed_hi;
};

// ---- query-owning workgroup: gQ ------------------------------------------------------------------------------------
template <int HD, bool DROP>
__global__ __launch_bounds__(256, 1) void flash_bwd_q_kernel(FlashBwdArgs p) {
    constexpr int NKS = HD / 16, NDB = HD / 32;
    typedef FlSeg<HD> G;
    constexpr int OFF_K = 0, OFF_V = G::RBYTES, OFF_KT = 2 * G::RBYTES, BYTES = 2 * G::RBYTES + G::TBYTES;
    __shared__ __attribute__((aligned(16))) unsigned char ldsq[2][BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane & 31, a = lane >> 5;
    const int bh = blockIdx.y, b = bh / p.H, h = bh % p.H;
    const int q0 = blockIdx.x * 128 + wave * 32;
    const int ntiles = (p.S + 31) / 32;
    const unsigned short* krow = p.k_row + (int64_t)bh * p.Sp * HD;
    const unsigned short* vrow = p.v_row + (int64_t)bh * p.Sp * HD;
    const unsigned short* ktr = p.k_tr + (int64_t)bh * HD * p.Sp;
    const float* bias = p.bias + (int64_t)b * p.Sp;

    u32x4 qf[NKS][3], df[NKS][3];
    {
        const int64_t o = ((int64_t)bh * p.Lp + q0 + lq) * HD + 8 * a;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                qf[ks][pl] = *reinterpret_cast<const u32x4*>(p.q_row + o + pl * p.q_plane + ks * 16);
                df[ks][pl] = *reinterpret_cast<const u32x4*>(p.do_row + o + pl * p.q_plane + ks * 16);
            }
    }
    const float lse2 = p.lse[(int64_t)bh * p.Lp + q0 + lq] * FL_LOG2E;
    const float dl = p.delta[(int64_t)bh * p.Lp + q0 + lq];
    const unsigned rid = (unsigned)(bh * p.L + q0 + lq);
    f32x16 gq[NDB];
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) gq[db][r] = 0.f;

    // staging: 9 units (K row, V row, K tr x 3 planes); HD 64: one chunk per thread per unit, HD 32: two units per pass
    constexpr int NU = 9, NJ = HD == 64 ? NU : (NU + 1) / 2;
    uint4 sv[NJ];
#define FLQ_UNIT(I) (HD == 64 ? (I) : min(2 * (I) + (tid >> 7), NU - 1))   /* (9 units: the odd one out is copied twice) */
#define FLQ_LOAD(T0)                                                                                                   \
    _Pragma("unroll") for (int i = 0; i < NJ; ++i) {                                                                   \
        const int u = FLQ_UNIT(i), c = HD == 64 ? tid : (tid & 127);                                                   \
        const int seg = u / 3, pl = u % 3;                                                                             \
        const unsigned short* src = seg == 0 ? krow + pl * p.k_plane + (int64_t)(T0) * HD + G::row_src(c)             \
                                  : seg == 1 ? vrow + pl * p.k_plane + (int64_t)(T0) * HD + G::row_src(c)             \
                                             : ktr + pl * p.k_plane + (T0) + G::tr_src(c, p.Sp);                       \
        sv[i] = *reinterpret_cast<const uint4*>(src);                                                                  \
    }
#define FLQ_STORE(BUF)                                                                                                 \
    _Pragma("unroll") for (int i = 0; i < NJ; ++i) {                                                                   \
        const int u = FLQ_UNIT(i), c = HD == 64 ? tid : (tid & 127);                                                   \
        const int seg = u / 3, pl = u % 3;                                                                             \
        unsigned char* dst = seg == 0 ? (BUF) + OFF_K + pl * G::RPLANE + G::row_dst(c)                                 \
                           : seg == 1 ? (BUF) + OFF_V + pl * G::RPLANE + G::row_dst(c)                                 \
                                      : (BUF) + OFF_KT + pl * G::TPLANE + G::tr_dst(c);                                \
        *reinterpret_cast<uint4*>(dst) = sv[i];                                                                        \
    }
    FLQ_LOAD(0)
    FLQ_STORE(ldsq[0])
    __syncthreads();

    for (int t = 0; t < ntiles; ++t) {
        const unsigned char* lds = ldsq[t & 1];
        const int t0 = t * 32;
        f32x4 kb[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) kb[g] = *reinterpret_cast<const f32x4*>(bias + t0 + 8 * g + 4 * a);
                FLQ_LOAD(min(t0 + 32, ntiles * 32 - 32))
                // ---- S^T = K Q^T and gd^T = V dO^T (two independent accumulator chains) ----
        f32x16 s, gd;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; gd[r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            u32x4 kf[3], vf[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                kf[pl] = *reinterpret_cast<const u32x4*>(lds + OFF_K + pl * G::RPLANE + lq * G::RROW + (ks * 16 + 8 * a) * 2);
                vf[pl] = *reinterpret_cast<const u32x4*>(lds + OFF_V + pl * G::RPLANE + lq * G::RROW + (ks * 16 + 8 * a) * 2);
            }
            FL_MMA6(s, kf, qf[ks])
            FL_MMA6(gd, vf, df[ks])
        }
        // ---- gs = P o (M o gd - t) ----
        float x[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float pr = exp2f(s[r] * p.scale_log2e + kb[r >> 2][r & 3] - lse2);
            x[r] = gd[r];
            s[r] = pr;
        }
        if (DROP) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int key = t0 + ((2 * i) & 3) + 8 * ((2 * i) >> 2) + 4 * a;
                const unsigned hsh = fl_hash(p.seed_lo, p.seed_hi, rid, (unsigned)key >> 1);
                x[2 * i] = (hsh & 0xffffu) >= p.thr16 ? x[2 * i] * p.inv_keep : 0.f;
                x[2 * i + 1] = (hsh >> 16) >= p.thr16 ? x[2 * i + 1] * p.inv_keep : 0.f;
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) x[r] = s[r] * (x[r] - dl);
        u32x4 pp[2][3];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                unsigned hh, mm, ll;
                fl_split3(x[8 * s2 + 2 * j], x[8 * s2 + 2 * j + 1], hh, mm, ll);
                pp[s2][0][j] = hh; pp[s2][1][j] = mm; pp[s2][2][j] = ll;
            }
        // ---- gQ^T[d, query] += K^T[d, key] gs^T[key, query] ----
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int db = 0; db < NDB; ++db) {
                u32x4 tf[3];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    tf[pl] = *reinterpret_cast<const u32x4*>(lds + OFF_KT + pl * G::TPLANE + (db * 32 + lq) * G::TROW + (s2 * 16 + 8 * a) * 2);
                FL_MMA6(gq[db], tf, pp[s2])
            }
        FLQ_STORE(ldsq[(t + 1) & 1])
        __syncthreads();
    }
#undef FLQ_UNIT
#undef FLQ_LOAD
#undef FLQ_STORE
    const int q = q0 + lq;
    if (q < p.L) {
        float* dst = p.gq + ((int64_t)b * p.L + q) * p.ld_q + p.off_q + h * HD + 4 * a;
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v;
                v.x = gq[db][4 * g] * p.scale; v.y = gq[db][4 * g + 1] * p.scale; v.z = gq[db][4 * g + 2] * p.scale; v.w = gq[db][4 * g + 3] * p.scale;
                *reinterpret_cast<f32x4*>(dst + db * 32 + 8 * g) = v;
            }
    }
}

// ---- key-owning workgroup: gK, gV (tiles oriented [query, key]: lane = key, registers = queries) ---------------------
template <int HD, bool DROP>
__global__ __launch_bounds__(256, 1) void flash_bwd_kv_kernel(FlashBwdArgs p) {
    constexpr int NKS = HD / 16, NDB = HD / 32;
    typedef FlSeg<HD> G;
    constexpr int OFF_Q = 0, OFF_D = G::RBYTES, OFF_QT = 2 * G::RBYTES, OFF_DT = 2 * G::RBYTES + G::TBYTES;
    constexpr int OFF_ST = 2 * G::RBYTES + 2 * G::TBYTES, BYTES = OFF_ST + 256;   // + lse[32], delta[32]
    __shared__ __attribute__((aligned(16))) unsigned char ldsb[2][BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lk = lane & 31, a = lane >> 5;
    const int bh = blockIdx.y, b = bh / p.H, h = bh % p.H;
    const int k0 = blockIdx.x * 128 + wave * 32;
    const int ntiles = (p.L + 31) / 32;
    const unsigned short* qrow = p.q_row + (int64_t)bh * p.Lp * HD;
    const unsigned short* drow = p.do_row + (int64_t)bh * p.Lp * HD;
    const unsigned short* qtr = p.q_tr + (int64_t)bh * HD * p.Lp;
    const unsigned short* dtr = p.do_tr + (int64_t)bh * HD * p.Lp;
    const float* lse = p.lse + (int64_t)bh * p.Lp;
    const float* delta = p.delta + (int64_t)bh * p.Lp;

    u32x4 kf[NKS][3], vf[NKS][3];
    {
        const int64_t o = ((int64_t)bh * p.Sp + k0 + lk) * HD + 8 * a;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                kf[ks][pl] = *reinterpret_cast<const u32x4*>(p.k_row + o + pl * p.k_plane + ks * 16);
                vf[ks][pl] = *reinterpret_cast<const u32x4*>(p.v_row + o + pl * p.k_plane + ks * 16);
            }
    }
    const float kbias = p.bias[(int64_t)b * p.Sp + k0 + lk] * 1.0f;
    const int key = k0 + lk;
    f32x16 gk[NDB], gv[NDB];
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) { gk[db][r] = 0.f; gv[db][r] = 0.f; }

    constexpr int NU = 12, NJ = HD == 64 ? NU : NU / 2;
    uint4 sv[NJ];
    float sst = 0.f;   // staged statistic: threads 0..31 carry lse, 32..63 delta of the next tile
#define FLK_UNIT(I) (HD == 64 ? (I) : 2 * (I) + (tid >> 7))
#define FLK_LOAD(T0)                                                                                                   \
    _Pragma("unroll") for (int i = 0; i < NJ; ++i) {                                                                   \
        const int u = FLK_UNIT(i), c = HD == 64 ? tid : (tid & 127);                                                   \
        const int seg = u / 3, pl = u % 3;                                                                             \
        const unsigned short* src = seg == 0 ? qrow + pl * p.q_plane + (int64_t)(T0) * HD + G::row_src(c)             \
                                  : seg == 1 ? drow + pl * p.q_plane + (int64_t)(T0) * HD + G::row_src(c)             \
                                  : seg == 2 ? qtr + pl * p.q_plane + (T0) + G::tr_src(c, p.Lp)                        \
                                             : dtr + pl * p.q_plane + (T0) + G::tr_src(c, p.Lp);                       \
        sv[i] = *reinterpret_cast<const uint4*>(src);                                                                  \
    }                                                                                                                  \
    if (tid < 64) sst = tid < 32 ? lse[(T0) + tid] : delta[(T0) + tid - 32];
#define FLK_STORE(BUF)                                                                                                 \
    _Pragma("unroll") for (int i = 0; i < NJ; ++i) {                                                                   \
        const int u = FLK_UNIT(i), c = HD == 64 ? tid : (tid & 127);                                                   \
        const int seg = u / 3, pl = u % 3;                                                                             \
        unsigned char* dst = seg == 0 ? (BUF) + OFF_Q + pl * G::RPLANE + G::row_dst(c)                                 \
                           : seg == 1 ? (BUF) + OFF_D + pl * G::RPLANE + G::row_dst(c)                                 \
                           : seg == 2 ? (BUF) + OFF_QT + pl * G::TPLANE + G::tr_dst(c)                                 \
                                      : (BUF) + OFF_DT + pl * G::TPLANE + G::tr_dst(c);                                \
        *reinterpret_cast<uint4*>(dst) = sv[i];                                                                        \
    }                                                                                                                  \
    if (tid < 64) reinterpret_cast<float*>((BUF) + OFF_ST)[tid] = sst;
    FLK_LOAD(0)
    FLK_STORE(ldsb[0])
    __syncthreads();

    for (int t = 0; t < ntiles; ++t) {
        const unsigned char* cur = ldsb[t & 1];
        const int t0 = t * 32;
        FLK_LOAD(min(t0 + 32, ntiles * 32 - 32))
                // statistics of this lane's 16 queries: register r <-> query t0 + (r & 3) + 8 (r >> 2) + 4 a
        f32x4 ls[4], dl[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            ls[g] = *reinterpret_cast<const f32x4*>(cur + OFF_ST + (8 * g + 4 * a) * 4);
            dl[g] = *reinterpret_cast<const f32x4*>(cur + OFF_ST + 128 + (8 * g + 4 * a) * 4);
        }
        // ---- S[query, key] = Q K^T and gd = dO V^T ----
        f32x16 s, gd;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; gd[r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            u32x4 qa[3], da[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                qa[pl] = *reinterpret_cast<const u32x4*>(cur + OFF_Q + pl * G::RPLANE + lk * G::RROW + (ks * 16 + 8 * a) * 2);
                da[pl] = *reinterpret_cast<const u32x4*>(cur + OFF_D + pl * G::RPLANE + lk * G::RROW + (ks * 16 + 8 * a) * 2);
            }
            FL_MMA6(s, qa, kf[ks])
            FL_MMA6(gd, da, vf[ks])
        }
        float pd[16], gs[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float pr = exp2f(s[r] * p.scale_log2e + kbias - ls[r >> 2][r & 3] * FL_LOG2E);
            float keep = 1.f;
            if (DROP) {
                const int q = t0 + (r & 3) + 8 * (r >> 2) + 4 * a;
                const unsigned hsh = fl_hash(p.seed_lo, p.seed_hi, (unsigned)(bh * p.L + q), (unsigned)key >> 1);
                keep = ((key & 1) ? (hsh >> 16) : (hsh & 0xffffu)) >= p.thr16 ? 1.f : 0.f;
            }
            pd[r] = pr * keep;                                        // (x 1/keep at the end, on gV)
            gs[r] = pr * (gd[r] * keep * p.inv_keep - dl[r >> 2][r & 3]);
        }
        u32x4 pp[2][3], gp[2][3];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                unsigned hh, mm, ll;
                fl_split3(pd[8 * s2 + 2 * j], pd[8 * s2 + 2 * j + 1], hh, mm, ll);
                pp[s2][0][j] = hh; pp[s2][1][j] = mm; pp[s2][2][j] = ll;
                fl_split3(gs[8 * s2 + 2 * j], gs[8 * s2 + 2 * j + 1], hh, mm, ll);
                gp[s2][0][j] = hh; gp[s2][1][j] = mm; gp[s2][2][j] = ll;
            }
        // ---- gV^T[d, key] += dO^T[d, query] Pd[query, key];  gK^T[d, key] += Q^T[d, query] gs[query, key] ----
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int db = 0; db < NDB; ++db) {
                u32x4 qt[3], dt[3];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    qt[pl] = *reinterpret_cast<const u32x4*>(cur + OFF_QT + pl * G::TPLANE + (db * 32 + lk) * G::TROW + (s2 * 16 + 8 * a) * 2);
                    dt[pl] = *reinterpret_cast<const u32x4*>(cur + OFF_DT + pl * G::TPLANE + (db * 32 + lk) * G::TROW + (s2 * 16 + 8 * a) * 2);
                }
                FL_MMA6(gv[db], dt, pp[s2])
                FL_MMA6(gk[db], qt, gp[s2])
            }
        FLK_STORE(ldsb[(t + 1) & 1])
        __syncthreads();
    }
#undef FLK_UNIT
#undef FLK_LOAD
#undef FLK_STORE
    if (key < p.S) {
        float* dk = p.gk + ((int64_t)b * p.S + key) * p.ld_k + p.off_k + h * HD + 4 * a;
        float* dv = p.gv + ((int64_t)b * p.S + key) * p.ld_v + p.off_v + h * HD + 4 * a;
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v;
                v.x = gk[db][4 * g] * p.scale; v.y = gk[db][4 * g + 1] * p.scale; v.z = gk[db][4 * g + 2] * p.scale; v.w = gk[db][4 * g + 3] * p.scale;
                *reinterpret_cast<f32x4*>(dk + db * 32 + 8 * g) = v;
                v.x = gv[db][4 * g] * p.inv_keep; v.y = gv[db][4 * g + 1] * p.inv_keep; v.z = gv[db][4 * g + 2] * p.inv_keep; v.w = gv[db][4 * g + 3] * p.inv_keep;
                *reinterpret_cast<f32x4*>(dv + db * 32 + 8 * g) = v;
            }
    }
}

extern "C" int ix_flash_bwd_f32(const void* q_row, const void* q_tr, const void* do_row, const void* do_tr, const void* k_row,
                                const void* k_tr, const void* v_row, const float* bias, const float* lse, const float* delta,
                                float* gq, float* gk, float* gv, int n, int H, int L, int Lp, int S, int Sp, int hd,
                                int64_t ld_q, int off_q, int64_t ld_k, int off_k, int64_t ld_v, int off_v, float scale,
                                float p_drop, uint64_t seed, hipStream_t stream) {
    if (n <= 0 || L <= 0 || S <= 0) return IX_OK;
    IX_CHECK_ARG(q_row && q_tr && do_row && do_tr && k_row && k_tr && v_row && bias && lse && delta, "ix_flash_bwd_f32: null operand");
    IX_CHECK_ARG(gq || (gk && gv), "ix_flash_bwd_f32: no output requested");
    IX_CHECK_ARG(hd == 32 || hd == 64, "ix_flash_bwd_f32: head dim %d (32 or 64)", hd);
    IX_CHECK_ARG(Lp % 128 == 0 && Sp % 128 == 0 && Lp >= L && Sp >= S, "ix_flash_bwd_f32: bad padded sizes");
    IX_CHECK_ARG(ld_q % 4 == 0 && ld_k % 4 == 0 && ld_v % 4 == 0 && off_q % 4 == 0 && off_k % 4 == 0 && off_v % 4 == 0,
                 "ix_flash_bwd_f32: output rows must be 16-byte aligned");
    IX_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f, "ix_flash_bwd_f32: p=%f outside [0,1)", p_drop);
    IX_CHECK_ARG((int64_t)n * H * L < ((int64_t)1 << 32) && n * H <= 65535, "ix_flash_bwd_f32: too many rows");
    FlashBwdArgs a;
    a.q_row = (const unsigned short*)q_row; a.q_tr = (const unsigned short*)q_tr;
    a.do_row = (const unsigned short*)do_row; a.do_tr = (const unsigned short*)do_tr;
    a.k_row = (const unsigned short*)k_row; a.k_tr = (const unsigned short*)k_tr; a.v_row = (const unsigned short*)v_row;
    a.bias = bias; a.lse = lse; a.delta = delta; a.gq = gq; a.gk = gk; a.gv = gv;
    a.ld_q = ld_q; a.ld_k = ld_k; a.ld_v = ld_v; a.off_q = off_q; a.off_k = off_k; a.off_v = off_v;
    a.H = H; a.L = L; a.Lp = Lp; a.S = S; a.Sp = Sp;
    a.q_plane = (int64_t)n * H * Lp * hd; a.k_plane = (int64_t)n * H * Sp * hd;
    a.scale = scale; a.scale_log2e = scale * FL_LOG2E;
    a.thr16 = (unsigned)((double)p_drop * 65536.0 + 0.5);
    a.inv_keep = a.thr16 ? 65536.f / (float)(65536u - a.thr16) : 1.f;
    a.seed_lo = (unsigned)seed; a.seed_hi = (unsigned)(seed >> 32);
    if (gq) {
        dim3 grid((L + 127) / 128, n * H);
        if (hd == 64) {
            if (a.thr16) hipLaunchKernelGGL((flash_bwd_q_kernel<64, true>), grid, dim3(256), 0, stream, a);
            else hipLaunchKernelGGL((flash_bwd_q_kernel<64, false>), grid, dim3(256), 0, stream, a);
        } else {
            if (a.thr16) hipLaunchKernelGGL((flash_bwd_q_kernel<32, true>), grid, dim3(256), 0, stream, a);
            else hipLaunchKernelGGL((flash_bwd_q_kernel<32, false>), grid, dim3(256), 0, stream, a);
        }
    }
    if (gk && gv) {
        dim3 grid((S + 127) / 128, n * H);
        if (hd == 64) {
            if (a.thr16) hipLaunchKernelGGL((flash_bwd_kv_kernel<64, true>), grid, dim3(256), 0, stream, a);
            else hipLaunchKernelGGL((flash_bwd_kv_kernel<64, false>), grid, dim3(256), 0, stream, a);
        } else {
            if (a.thr16) hipLaunchKernelGGL((flash_bwd_kv_kernel<32, true>), grid, dim3(256), 0, stream, a);
            else hipLaunchKernelGGL((flash_bwd_kv_kernel<32, false>), grid, dim3(256), 0, stream, a);
        }
    }
    IX_CHECK_LAUNCH("ix_flash_bwd_f32");
    return IX_OK;
}

// The flash kernels' dropout mask as a tensor (tests only): m[bh][q][key] = 1/keep where kept, 0 where dropped.
__global__ void flash_dropmask_kernel(float* __restrict__ m, int L, int S, unsigned thr16, float inv_keep, unsigned seed_lo,
                                      unsigned seed_hi) {
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
                       (unsigned)seed, (unsigned)(seed >> 32));
    IX_CHECK_LAUNCH("ix_flash_dropmask_f32");
    return IX_OK;
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
struct FlashBBArgs {
    const unsigned short *q_row, *q_tr, *hq_row, *hq_tr, *do_row, *do_tr;                    // query side
    const unsigned short *k_row, *k_tr, *hk_row, *hk_tr, *v_row, *v_tr, *hv_row, *hv_tr;    // key side
    const float *bias, *lse, *delta;   // [n][Sp], [BH][Lp], [BH][Lp]
    float *u, *w;                      // [BH][Lp] each (workspace)
    float *dq, *ddo, *dk, *dv;
    int64_t ld_q, ld_k, ld_v, ld_do;
    int off_q, off_k, off_v, off_do;
    int H, L, Lp, S, Sp;
    int64_t q_plane, k_plane;
    float scale, scale_log2e;
    unsigned thr16;
    float inv_keep;
    unsigned seed_lo, seed_hi;
};

#define FL_ROWFRAG(DST, BASE, OFF, ROWIDX, KS)                                                                         \
    _Pragma("unroll") for (int pl_ = 0; pl_ < 3; ++pl_) DST[pl_] = *reinterpret_cast<const u32x4*>(                    \
        (BASE) + (OFF) + pl_ * G::RPLANE + (ROWIDX) * G::RROW + ((KS) * 16 + 8 * a) * 2);
#define FL_SPLIT16(PLANES, X)                                                                                          \
    _Pragma("unroll") for (int s2_ = 0; s2_ < 2; ++s2_) _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) {              \
        unsigned hh_, mm_, ll_;                                                                                        \
        fl_split3(X[8 * s2_ + 2 * j_], X[8 * s2_ + 2 * j_ + 1], hh_, mm_, ll_);                                        \
        PLANES[s2_][0][j_] = hh_; PLANES[s2_][1][j_] = mm_; PLANES[s2_][2][j_] = ll_;                                  \
    }
// ACC[db] += X^T[d, row] . PLANES[row, col]   for the tr segment at OFF
#define FL_STAGE2(ACC, BASE, OFF, ROWIDX, PLANES)                                                                      \
    _Pragma("unroll") for (int s2_ = 0; s2_ < 2; ++s2_) _Pragma("unroll") for (int db_ = 0; db_ < NDB; ++db_) {         \
        u32x4 tf_[3];                                                                                                  \
        _Pragma("unroll") for (int pl_ = 0; pl_ < 3; ++pl_) tf_[pl_] = *reinterpret_cast<const u32x4*>(                \
            (BASE) + (OFF) + pl_ * G::TPLANE + (db_ * 32 + (ROWIDX)) * G::TROW + (s2_ * 16 + 8 * a) * 2);              \
        FL_MMA6(ACC[db_], tf_, PLANES[s2_])                                                                            \
    }
// Staging registers as a compile-time list (no arrays: an indexed array of in-flight loads is easily demoted to scratch
// memory or to an LDS copy by the compiler), visited with compile-time indices.
template <int N>
struct FlRegs {
    uint4 v;
    FlRegs<N - 1> r;
};
template <>
struct FlRegs<0> {};
template <int I, int N, class F>
__device__ __forceinline__ void fl_each(FlRegs<N>& s, F&& f) {
    f(std::integral_constant<int, I>(), s.v);
    if constexpr (N > 1) fl_each<I + 1>(s.r, f);
}
// global -> staging registers -> LDS for the units FIRST .. FIRST + COUNT - 1 (unit = one plane of one segment = 4 HD
// 16-byte chunks: one chunk per thread at HD 64, two units per pass at HD 32 -- an odd unit out is copied twice).
// SRC / DST are expressions in seg_ (segment), pl_ (plane), c_ (chunk).
#define FL_NREGS(COUNT) (HD == 64 ? (COUNT) : ((COUNT) + 1) / 2)
#define FL_STAGE_LOAD(REGS, FIRST, COUNT, SRC)                                                                         \
    fl_each<0>(REGS, [&](auto I_, uint4& v_) {                                                                          \
        constexpr int i_ = decltype(I_)::value;                                                                        \
        const int u_ = (FIRST) + (HD == 64 ? i_ : min(2 * i_ + (tid >> 7), (COUNT) - 1));                              \
        const int c_ = HD == 64 ? tid : (tid & 127);                                                                   \
        const int seg_ = u_ / 3, pl_ = u_ % 3;                                                                         \
        v_ = *reinterpret_cast<const uint4*>(SRC);                                                                     \
    });
#define FL_STAGE_STORE(REGS, FIRST, COUNT, DST)                                                                        \
    fl_each<0>(REGS, [&](auto I_, uint4& v_) {                                                                          \
        constexpr int i_ = decltype(I_)::value;                                                                        \
        const int u_ = (FIRST) + (HD == 64 ? i_ : min(2 * i_ + (tid >> 7), (COUNT) - 1));                              \
        const int c_ = HD == 64 ? tid : (tid & 127);                                                                   \
        const int seg_ = u_ / 3, pl_ = u_ % 3;                                                                         \
        *reinterpret_cast<uint4*>(DST) = v_;                                                                           \
    });

// passes 1 and 2 (query-owning).
// STATS (pass 1): only u, w are produced; four row segments, double-buffered LDS, one barrier per tile.
// pass 2: row + tr segments fill the LDS once, so a tile is two phases around two barriers -- phase 1 forms the four
// [key, query] tiles out of the ROW region while the next tile's rows are in flight, phase 2 runs the four output
// products out of the TR region while the next tile's tr operands are in flight; each region is refilled right after
// the barrier that ends its phase, from one shared set of staging registers.
template <int HD, bool DROP, bool STATS>
__global__ __launch_bounds__(256, 1) void flash_bb_q_kernel(FlashBBArgs p) {
    constexpr int NKS = HD / 16, NDB = HD / 32;
    typedef FlSeg<HD> G;
    // row segments k, hk, v, hv; then (pass 2) tr segments hk, k, hv, v
    constexpr int OFF_K = 0, OFF_HK = G::RBYTES, OFF_V = 2 * G::RBYTES, OFF_HV = 3 * G::RBYTES;
    constexpr int OFF_HKT = 4 * G::RBYTES, OFF_KT = OFF_HKT + G::TBYTES, OFF_HVT = OFF_KT + G::TBYTES, OFF_VT = OFF_HVT + G::TBYTES;
    constexpr int BYTES = STATS ? 4 * G::RBYTES : 4 * G::RBYTES + 4 * G::TBYTES;
    constexpr int NBUF = STATS ? 2 : 1;
    __shared__ __attribute__((aligned(16))) unsigned char ldsq[NBUF][BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane & 31, a = lane >> 5;
    const int bh = blockIdx.y, b = bh / p.H, h = bh % p.H;
    const int q0 = blockIdx.x * 128 + wave * 32;
    const int ntiles = (p.S + 31) / 32;
    const int64_t kro = (int64_t)bh * p.Sp * HD, kto = (int64_t)bh * HD * p.Sp;
    const float* bias = p.bias + (int64_t)b * p.Sp;

    u32x4 qf[NKS][3], hqf[NKS][3], df[NKS][3];
    {
        const int64_t o = ((int64_t)bh * p.Lp + q0 + lq) * HD + 8 * a;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                qf[ks][pl] = *reinterpret_cast<const u32x4*>(p.q_row + o + pl * p.q_plane + ks * 16);
                hqf[ks][pl] = *reinterpret_cast<const u32x4*>(p.hq_row + o + pl * p.q_plane + ks * 16);
                df[ks][pl] = *reinterpret_cast<const u32x4*>(p.do_row + o + pl * p.q_plane + ks * 16);
            }
    }
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

    FlRegs<FL_NREGS(12)> sv;
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
    FL_STAGE_LOAD(sv, 0, 12, FLB_SRC(0))
    FL_STAGE_STORE(sv, 0, 12, FLB_DST(ldsq[0]))
    if (!STATS) {
        FL_STAGE_LOAD(sv, 12, 12, FLB_SRC(0))
        FL_STAGE_STORE(sv, 12, 12, FLB_DST(ldsq[0]))
    }
    __syncthreads();

    for (int t = 0; t < ntiles; ++t) {
        unsigned char* lds = ldsq[STATS ? (t & 1) : 0];
        const int t0 = t * 32, tn = min(t0 + 32, ntiles * 32 - 32);
        f32x4 kb[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) kb[g] = *reinterpret_cast<const f32x4*>(bias + t0 + 8 * g + 4 * a);
        FL_STAGE_LOAD(sv, 0, 12, FLB_SRC(tn))
        // ---- the four [key, query] tiles ----
        f32x16 s, gd, gg, hd_;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; gd[r] = 0.f; gg[r] = 0.f; hd_[r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            u32x4 kf[3], hkf[3], vf[3], hvf[3];
            FL_ROWFRAG(kf, lds, OFF_K, lq, ks)
            FL_ROWFRAG(hkf, lds, OFF_HK, lq, ks)
            FL_ROWFRAG(vf, lds, OFF_V, lq, ks)
            FL_ROWFRAG(hvf, lds, OFF_HV, lq, ks)
            FL_MMA6(s, kf, qf[ks])
            FL_MMA6(gd, vf, df[ks])
            FL_MMA6(gg, kf, hqf[ks])
            FL_MMA6(hd_, hvf, df[ks])
            FL_MMA6(gg, hkf, qf[ks])
        }
        float pr[16], mk[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            pr[r] = exp2f(s[r] * p.scale_log2e + kb[r >> 2][r & 3] - lse2);
            mk[r] = 1.f;
        }
        if (DROP) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int key = t0 + ((2 * i) & 3) + 8 * ((2 * i) >> 2) + 4 * a;
                const unsigned hsh = fl_hash(p.seed_lo, p.seed_hi, rid, (unsigned)key >> 1);
                mk[2 * i] = (hsh & 0xffffu) >= p.thr16 ? p.inv_keep : 0.f;
                mk[2 * i + 1] = (hsh >> 16) >= p.thr16 ? p.inv_keep : 0.f;
            }
        }
        if (STATS) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float G_ = gg[r] * p.scale, gy = mk[r] * gd[r], pg = pr[r] * G_;
                uu += pg;
                aa += pg * gy;
                bq += pr[r] * mk[r] * hd_[r];
            }
            FL_STAGE_STORE(sv, 0, 12, FLB_DST(ldsq[(t + 1) & 1]))
            __syncthreads();
        } else {
            __syncthreads();   // (A) every wave has formed its tiles: the ROW region is free, the TR region is complete
            FL_STAGE_STORE(sv, 0, 12, FLB_DST(lds))
            FL_STAGE_LOAD(sv, 12, 12, FLB_SRC(tn))
            float x[16];
            u32x4 pp[2][3];
            // gs = P (gy - t)                                   dq += hk^T gs
#pragma unroll
            for (int r = 0; r < 16; ++r) x[r] = pr[r] * (mk[r] * gd[r] - dl);
            FL_SPLIT16(pp, x)
            FL_STAGE2(dq, lds, OFF_HKT, lq, pp)
            // HS = P (G (gy - t) - gy u + M HD - w)             dq += k^T HS
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float G_ = gg[r] * p.scale, gy = mk[r] * gd[r];
                x[r] = pr[r] * (G_ * (gy - dl) - gy * uu + mk[r] * hd_[r] - ww);
            }
            FL_SPLIT16(pp, x)
            FL_STAGE2(dq, lds, OFF_KT, lq, pp)
            // Pd = M P                                          ddO += hv^T Pd
#pragma unroll
            for (int r = 0; r < 16; ++r) x[r] = mk[r] * pr[r];
            FL_SPLIT16(pp, x)
            FL_STAGE2(ddo, lds, OFF_HVT, lq, pp)
            // HgD = M P (G - u)                                 ddO += v^T HgD
#pragma unroll
            for (int r = 0; r < 16; ++r) x[r] = mk[r] * pr[r] * (gg[r] * p.scale - uu);
            FL_SPLIT16(pp, x)
            FL_STAGE2(ddo, lds, OFF_VT, lq, pp)
            __syncthreads();   // (B) every wave is done with the TR region; the next tile's rows are visible
            FL_STAGE_STORE(sv, 12, 12, FLB_DST(lds))
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
        float* d1 = p.dq + ((int64_t)b * p.L + q) * p.ld_q + p.off_q + h * HD + 4 * a;
        float* d2 = p.ddo + ((int64_t)b * p.L + q) * p.ld_do + p.off_do + h * HD + 4 * a;
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v;
                v.x = dq[db][4 * g] * p.scale; v.y = dq[db][4 * g + 1] * p.scale; v.z = dq[db][4 * g + 2] * p.scale; v.w = dq[db][4 * g + 3] * p.scale;
                *reinterpret_cast<f32x4*>(d1 + db * 32 + 8 * g) = v;
                v.x = ddo[db][4 * g]; v.y = ddo[db][4 * g + 1]; v.z = ddo[db][4 * g + 2]; v.w = ddo[db][4 * g + 3];
                *reinterpret_cast<f32x4*>(d2 + db * 32 + 8 * g) = v;
            }
    }
}

// pass 3 (key-owning): dk, dv.  Tiles [query, key]: lane = key, registers = queries.  Same two-phase tile as pass 2.
template <int HD, bool DROP>
__global__ __launch_bounds__(256, 1) void flash_bb_kv_kernel(FlashBBArgs p) {
    constexpr int NKS = HD / 16, NDB = HD / 32;
    typedef FlSeg<HD> G;
    // row segments q, hq, dO; tr segments hq, q, dO; statistics lse, delta, u, w [32] each (part of the TR phase)
    constexpr int OFF_Q = 0, OFF_HQ = G::RBYTES, OFF_D = 2 * G::RBYTES;
    constexpr int OFF_HQT = 3 * G::RBYTES, OFF_QT = OFF_HQT + G::TBYTES, OFF_DT = OFF_QT + G::TBYTES, OFF_ST = OFF_DT + G::TBYTES;
    constexpr int BYTES = OFF_ST + 512;
    __shared__ __attribute__((aligned(16))) unsigned char lds[BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lk = lane & 31, a = lane >> 5;
    const int bh = blockIdx.y, b = bh / p.H, h = bh % p.H;
    const int k0 = blockIdx.x * 128 + wave * 32;
    const int ntiles = (p.L + 31) / 32;
    const int64_t qro = (int64_t)bh * p.Lp * HD, qto = (int64_t)bh * HD * p.Lp, sto = (int64_t)bh * p.Lp;

    u32x4 kf[NKS][3], hkf[NKS][3], vf[NKS][3], hvf[NKS][3];
    {
        const int64_t o = ((int64_t)bh * p.Sp + k0 + lk) * HD + 8 * a;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                kf[ks][pl] = *reinterpret_cast<const u32x4*>(p.k_row + o + pl * p.k_plane + ks * 16);
                hkf[ks][pl] = *reinterpret_cast<const u32x4*>(p.hk_row + o + pl * p.k_plane + ks * 16);
                vf[ks][pl] = *reinterpret_cast<const u32x4*>(p.v_row + o + pl * p.k_plane + ks * 16);
                hvf[ks][pl] = *reinterpret_cast<const u32x4*>(p.hv_row + o + pl * p.k_plane + ks * 16);
            }
    }
    const float kbias = p.bias[(int64_t)b * p.Sp + k0 + lk];
    const int key = k0 + lk;
    f32x16 dk[NDB], dv[NDB];
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[db][r] = 0.f; dv[db][r] = 0.f; }

    FlRegs<FL_NREGS(9)> sv;
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
    FL_STAGE_LOAD(sv, 0, 9, FLC_SRC(0))
    FL_STAGE_STORE(sv, 0, 9, FLC_DST)
    FL_STAGE_LOAD(sv, 9, 9, FLC_SRC(0))
    FLC_STAT_LOAD(0)
    FL_STAGE_STORE(sv, 9, 9, FLC_DST)
    if (tid < 128) reinterpret_cast<float*>(lds + OFF_ST)[tid] = sst;
    __syncthreads();

    for (int t = 0; t < ntiles; ++t) {
        const int t0 = t * 32, tn = min(t0 + 32, ntiles * 32 - 32);
        FL_STAGE_LOAD(sv, 0, 9, FLC_SRC(tn))
        f32x16 s, gd, gg, hd_;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; gd[r] = 0.f; gg[r] = 0.f; hd_[r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            u32x4 qa[3], hqa[3], da[3];
            FL_ROWFRAG(qa, lds, OFF_Q, lk, ks)
            FL_ROWFRAG(hqa, lds, OFF_HQ, lk, ks)
            FL_ROWFRAG(da, lds, OFF_D, lk, ks)
            FL_MMA6(s, qa, kf[ks])
            FL_MMA6(gd, da, vf[ks])
            FL_MMA6(gg, hqa, kf[ks])
            FL_MMA6(hd_, da, hvf[ks])
            FL_MMA6(gg, qa, hkf[ks])
        }
        __syncthreads();   // (A) the ROW region is free, the TR region (and the statistics) are complete
        FL_STAGE_STORE(sv, 0, 9, FLC_DST)
        FL_STAGE_LOAD(sv, 9, 9, FLC_SRC(tn))
        FLC_STAT_LOAD(tn)
        // statistics of this lane's 16 queries: register r <-> query t0 + (r & 3) + 8 (r >> 2) + 4 a; read per use
#define FLC_ST(WHICH, R) (reinterpret_cast<const float*>(lds + OFF_ST + (WHICH) * 128)[((R) & 3) + 8 * ((R) >> 2) + 4 * a])
        float pr[16], mk[16], x[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            pr[r] = exp2f(s[r] * p.scale_log2e + kbias - FLC_ST(0, r) * FL_LOG2E);
            mk[r] = 1.f;
            if (DROP) {
                const int q = t0 + (r & 3) + 8 * (r >> 2) + 4 * a;
                const unsigned hsh = fl_hash(p.seed_lo, p.seed_hi, (unsigned)(bh * p.L + q), (unsigned)key >> 1);
                mk[r] = ((key & 1) ? (hsh >> 16) : (hsh & 0xffffu)) >= p.thr16 ? p.inv_keep : 0.f;
            }
        }
        u32x4 pp[2][3];
        // gs                                                     dk += hq^T gs
#pragma unroll
        for (int r = 0; r < 16; ++r) x[r] = pr[r] * (mk[r] * gd[r] - FLC_ST(1, r));
        FL_SPLIT16(pp, x)
        FL_STAGE2(dk, lds, OFF_HQT, lk, pp)
        // HS                                                     dk += q^T HS
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float G_ = gg[r] * p.scale, gy = mk[r] * gd[r];
            x[r] = pr[r] * (G_ * (gy - FLC_ST(1, r)) - gy * FLC_ST(2, r) + mk[r] * hd_[r] - FLC_ST(3, r));
        }
        FL_SPLIT16(pp, x)
        FL_STAGE2(dk, lds, OFF_QT, lk, pp)
        // HgD                                                    dv += dO^T HgD
#pragma unroll
        for (int r = 0; r < 16; ++r) x[r] = mk[r] * pr[r] * (gg[r] * p.scale - FLC_ST(2, r));
        FL_SPLIT16(pp, x)
        FL_STAGE2(dv, lds, OFF_DT, lk, pp)
#undef FLC_ST
        __syncthreads();   // (B) every wave is done with the TR region; the next tile's rows are visible
        FL_STAGE_STORE(sv, 9, 9, FLC_DST)
        if (tid < 128) reinterpret_cast<float*>(lds + OFF_ST)[tid] = sst;
    }
#undef FLC_SRC
#undef FLC_DST
#undef FLC_STAT_LOAD
    if (key < p.S) {
        float* d1 = p.dk + ((int64_t)b * p.S + key) * p.ld_k + p.off_k + h * HD + 4 * a;
        float* d2 = p.dv + ((int64_t)b * p.S + key) * p.ld_v + p.off_v + h * HD + 4 * a;
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v;
                v.x = dk[db][4 * g] * p.scale; v.y = dk[db][4 * g + 1] * p.scale; v.z = dk[db][4 * g + 2] * p.scale; v.w = dk[db][4 * g + 3] * p.scale;
                *reinterpret_cast<f32x4*>(d1 + db * 32 + 8 * g) = v;
                v.x = dv[db][4 * g]; v.y = dv[db][4 * g + 1]; v.z = dv[db][4 * g + 2]; v.w = dv[db][4 * g + 3];
                *reinterpret_cast<f32x4*>(d2 + db * 32 + 8 * g) = v;
            }
    }
}

// planes: HOST array of 14 device pointers in the order
//   q_row q_tr hq_row hq_tr do_row do_tr | k_row k_tr hk_row hk_tr v_row v_tr hv_row hv_tr
extern "C" int ix_flash_bwd_bwd_f32(const void* const* planes, const float* bias, const float* lse, const float* delta,
                                    float* dq, float* dk, float* dv, float* ddo, int n, int H, int L, int Lp, int S, int Sp,
                                    int hd, int64_t ld_q, int off_q, int64_t ld_k, int off_k, int64_t ld_v, int off_v,
                                    int64_t ld_do, int off_do, float scale, float p_drop, uint64_t seed, void* workspace,
                                    size_t workspace_bytes, hipStream_t stream) {
    if (n <= 0 || L <= 0 || S <= 0) return IX_OK;
    IX_CHECK_ARG(planes && bias && lse && delta && dq && dk && dv && ddo, "ix_flash_bwd_bwd_f32: null pointer");
    for (int i = 0; i < 14; ++i) IX_CHECK_ARG(planes[i] != nullptr, "ix_flash_bwd_bwd_f32: operand plane %d is null", i);
    IX_CHECK_ARG(hd == 32 || hd == 64, "ix_flash_bwd_bwd_f32: head dim %d (32 or 64)", hd);
    IX_CHECK_ARG(Lp % 128 == 0 && Sp % 128 == 0 && Lp >= L && Sp >= S, "ix_flash_bwd_bwd_f32: bad padded sizes");
    IX_CHECK_ARG(ld_q % 4 == 0 && ld_k % 4 == 0 && ld_v % 4 == 0 && ld_do % 4 == 0 && off_q % 4 == 0 && off_k % 4 == 0 &&
                 off_v % 4 == 0 && off_do % 4 == 0, "ix_flash_bwd_bwd_f32: output rows must be 16-byte aligned");
    IX_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f, "ix_flash_bwd_bwd_f32: p=%f outside [0,1)", p_drop);
    IX_CHECK_ARG((int64_t)n * H * L < ((int64_t)1 << 32) && n * H <= 65535, "ix_flash_bwd_bwd_f32: too many rows");
    const size_t need = (size_t)2 * n * H * Lp * sizeof(float);
    if (!workspace || workspace_bytes < need) {
        ix_set_error("ix_flash_bwd_bwd_f32: workspace of %zu bytes needed, %zu given", need, workspace_bytes);
        return IX_ERR_WORKSPACE;
    }
    FlashBBArgs a;
    const unsigned short* const* pp = reinterpret_cast<const unsigned short* const*>(planes);
    a.q_row = pp[0]; a.q_tr = pp[1]; a.hq_row = pp[2]; a.hq_tr = pp[3]; a.do_row = pp[4]; a.do_tr = pp[5];
    a.k_row = pp[6]; a.k_tr = pp[7]; a.hk_row = pp[8]; a.hk_tr = pp[9]; a.v_row = pp[10]; a.v_tr = pp[11];
    a.hv_row = pp[12]; a.hv_tr = pp[13];
    a.bias = bias; a.lse = lse; a.delta = delta;
    a.u = (float*)workspace; a.w = a.u + (size_t)n * H * Lp;
    a.dq = dq; a.ddo = ddo; a.dk = dk; a.dv = dv;
    a.ld_q = ld_q; a.ld_k = ld_k; a.ld_v = ld_v; a.ld_do = ld_do;
    a.off_q = off_q; a.off_k = off_k; a.off_v = off_v; a.off_do = off_do;
    a.H = H; a.L = L; a.Lp = Lp; a.S = S; a.Sp = Sp;
    a.q_plane = (int64_t)n * H * Lp * hd; a.k_plane = (int64_t)n * H * Sp * hd;
    a.scale = scale; a.scale_log2e = scale * FL_LOG2E;
    a.thr16 = (unsigned)((double)p_drop * 65536.0 + 0.5);
    a.inv_keep = a.thr16 ? 65536.f / (float)(65536u - a.thr16) : 1.f;
    a.seed_lo = (unsigned)seed; a.seed_hi = (unsigned)(seed >> 32);
    const dim3 gq((L + 127) / 128, n * H), gk((S + 127) / 128, n * H), blk(256);
#define FL_BB_LAUNCH(HD_, DR_)                                                                         \
    hipLaunchKernelGGL((flash_bb_q_kernel<HD_, DR_, true>), gq, blk, 0, stream, a);                    \
    hipLaunchKernelGGL((flash_bb_q_kernel<HD_, DR_, false>), gq, blk, 0, stream, a);                   \
    hipLaunchKernelGGL((flash_bb_kv_kernel<HD_, DR_>), gk, blk, 0, stream, a);
    if (hd == 64) {
        if (a.thr16) { FL_BB_LAUNCH(64, true) } else { FL_BB_LAUNCH(64, false) }
    } else {
        if (a.thr16) { FL_BB_LAUNCH(32, true) } else { FL_BB_LAUNCH(32, false) }
    }
#undef FL_BB_LAUNCH
    IX_CHECK_LAUNCH("ix_flash_bwd_bwd_f32");
    return IX_OK;
}

extern "C" int ix_workspace_bytes_flash_bwd_bwd(int n, int H, int L, size_t* out) {
    IX_CHECK_ARG(out && n >= 0 && H >= 0 && L >= 0, "ix_workspace_bytes_flash_bwd_bwd: bad args");
    *out = (size_t)2 * n * H * ((L + 127) / 128 * 128) * sizeof(float);
    return IX_OK;
}
