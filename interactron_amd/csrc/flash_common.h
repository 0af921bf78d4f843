// Shared declarations of the attention kernels: csrc/flash.hip (32x32x16 tiles, both tr forms, head dims 32 / 64) and
// csrc/flash16.hip (16x16x32 tiles, head dim 64, fp16 form).  See the header of flash.hip for arithmetic and layouts.
#pragma once
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// v_exp_f32 directly: exp2f() wraps it in denormal-range handling (4 more instructions per element), and every use here
// either has a non-positive argument or is multiplied into a sum where a flushed 2^-126 does not matter
#define fl_exp2(X) __builtin_amdgcn_exp2f(X)
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
__device__ __forceinline__ f32x16 fl_mfma_h(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// (x0, x1), already scaled into fp16 range -> packed fp16 pairs of the two planes: x = h + l up to 2^-22 relative
__device__ __forceinline__ void fl_split2h(float x0, float x1, unsigned& h, unsigned& l) {
    f32x2 v;
    v.x = x0; v.y = x1;
    const f16x2 hh = __builtin_convertvector(v, f16x2);
    const f32x2 back = __builtin_convertvector(hh, f32x2);
    f32x2 r;
    r.x = x0 - back.x; r.y = x1 - back.y;
    h = __builtin_bit_cast(unsigned, hh);
    l = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2));
}

// position of row r (0..15) of a 16-group in the tr layout: bits 2 and 3 swapped
__device__ __host__ __forceinline__ int fl_perm16(int r) { return (r & 3) | ((r & 4) << 1) | ((r & 8) >> 1); }

// ------------------------------------------------------------------------------------------------------------
// dropout mask of the flash kernels: a pure function of (seed, row id = (batch*head)*L + query, key), one 32-bit hash
// per PAIR of neighbouring keys (two 16-bit draws), so forward, backward and double backward regenerate the same mask
// whichever way their tiles are oriented.  keep <=> draw >= thr16, thr16 = round(p * 65536).
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned fl_hash(unsigned seed_lo, unsigned seed_hi, unsigned rid, unsigned kpair) {
    // one multiply-fold round on (row term) ^ (key-pair term): in the kernels one of the two terms is invariant per lane
    // and the other advances by a wave-uniform amount per tile, so a draw costs an add, an xor, the 32 x 32 -> 64-bit
    // product and a fold (the earlier two-multiply xorshift mix was a third of the forward kernel's vector instructions)
    const unsigned a = (rid * 0x9E3779B1u) ^ seed_lo;
    const unsigned b = kpair * 0x85EBCA77u + seed_hi;
    const unsigned long long m = (unsigned long long)(a ^ b) * 0xD6E8FEB9ull;
    return (unsigned)m ^ (unsigned)(m >> 32);
}

#define FL_F0 1152921504606846976.f   // 2^60: where a running factor starts

// One argument block for all kernels.  Operand planes: *_row = fp16 row planes [2][BH][Rp][hd], *_us = their block
// unscale factors [BH][Rp / 32], *_tr = bf16 tr planes [3][BH][hd][Rp].  Query side: q, dO (do_), hq; key side: k, v, hk, hv.
struct FlashArgs {
    const unsigned short *q_row, *do_row, *hq_row, *q_tr, *do_tr, *hq_tr;
    const float *q_us, *do_us, *hq_us;
    const unsigned short *k_row, *v_row, *hk_row, *hv_row, *k_tr, *v_tr, *hk_tr, *hv_tr;
    const float *k_us, *v_us, *hk_us, *hv_us;
    const float* bias;     // [n][Sp] additive key bias (0 / -inf)
    float* lse;            // [BH][Lp] natural-log row normalisers (+inf beyond L)
    const float* delta;    // [BH][Lp] t_i = dO_i . O_i
    float *u, *w;          // [BH][Lp] second-order row statistics (workspace)
    float *o1, *o2, *o3, *o4;   // outputs: fwd out | gq gk gv | dq dk dv ddo; [n][L|S][ld] with head h at off + h*hd
    int64_t ld1, ld2, ld3, ld4;
    int off1, off2, off3, off4;
    int H, L, Lp, S, Sp;
    int64_t q_plane, k_plane;   // elements per plane
    float scale, scale_log2e;
    unsigned thr16;             // dropout threshold (0 = no dropout)
    float inv_keep;
    unsigned seed_lo, seed_hi;
    const unsigned* salt;       // optional device word XORed into the seed when the kernel runs (ix_set_dropout_salt)
};
