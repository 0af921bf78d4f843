// Head-dim-64 attention passes on v_mfma_f32_16x16x32_f16: forward, backward and double backward of
//     O = dropout(softmax(scale Q K^T + key bias)) V
// for the fp16 form of csrc/flash.hip (same algebra, same operand planes, same dropout mask, same accuracy class; reference
// models/gpt.py:39-57 and its autograd derivatives as taken by models/interactron.py:99-123).  Why a second set of kernels:
// the 32x32x16 passes of flash.hip need 370-430 registers and 115-148 KB of LDS at head dim 64 -- ONE wave per SIMD, whose
// matrix and vector work (96-108 matrix instructions against ~1 000 vector instructions per 32 x 32 tile) then simply add up.
// Here a workgroup is EIGHT waves (two per SIMD, <= 256 registers), each owning 16 rows of the owner side:
//
//   * tiles are [32 streamed rows] x [16 owner rows] = two 16 x 16 accumulator blocks (4 registers each); lane
//     (n = lane & 15, g = lane >> 4) holds owner row n and streamed rows 16 blk + 4 g + r: eight elements per lane per tile
//     instead of sixteen, a quarter of the accumulator registers per tile kind;
//   * all four owner-side operands stay in registers as B fragments (16 registers each);
//   * the streamed side is staged ONCE, as rows ([2 planes][4 panels of 16 d][32 rows][32 B]): products that contract over
//     the head dim read it with ds_read_b128 (A fragment = 8 consecutive d of one row), products that contract over the
//     streamed rows read THE SAME image with ds_read_b64_tr_b16 (the hardware transpose: lane (d, g) receives rows 4 g ..
//     4 g + 3 and 16 + 4 g .. of column d), so no tr planes are read, staged or even written for head dim 64 -- half the LDS
//     footprint, half the staging traffic.  With 32-byte rows inside a panel both access kinds are conflict-free without
//     any swizzle (16-lane groups of ds_read_b128 as listed in MI355X_MICROARCH "LDS"; 32-lane halves of the transpose read
//     cover 256 contiguous bytes) and every k-slice / d-block / row-block / plane / operand step is an immediate offset;
//     tools/micro/m16_layout.hip checks the layout facts on the device;
//   * the accumulator registers of a tile ARE the B fragment (k = 8 g + e <-> streamed rows 4 g + e | 16 + 4 g + e - 4) of
//     the products that contract over the streamed rows: no [L, S] value passes through LDS;
//   * two LDS buffers, ONE barrier per tile; the next tile's rows are requested before the tile's arithmetic and written to
//     the other buffer after it (forward, gQ, row statistics: two workgroups per CU);
//   * the passes that run one workgroup per CU (gK / gV, dq / ddO, dk / dv) separate vector and matrix work in time and run
//     the two halves of the workgroup in opposite segments over a ring of four tiles -- see "split-phase passes" below.
#include "flash_common.h"
#include <type_traits>

// Second build of this file (-DM16_ONE, Makefile: flash16_one): ONE matrix instruction per k-slice -- the h planes only -- for operands
// that ARE 16-bit values (the 16-bit activation mode, b16.py: bf16 q / k / v / dO fit the h plane exactly, the l plane is zero), and the
// register operands ([L, S] intermediates) rounded once to fp16.  Same kernels under the suffix _one, selected by
// ix_flash_set_single_term (csrc/flash.hip).  M16_L(...) marks what only serves the l planes.
#ifdef M16_ONE
#define M16_L(...)
#define M16_N(X) X##_one
#else
#define M16_L(...) __VA_ARGS__
#define M16_N(X) X
#endif

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// experiment switch (-DM16_PRIO_LEVEL=n): waves 0..3 (one per SIMD) at priority n, their partners at 0
#ifdef M16_PRIO_LEVEL
#define M16_PRIO if (__builtin_amdgcn_readfirstlane(threadIdx.x) < 256) __builtin_amdgcn_s_setprio(M16_PRIO_LEVEL);
#else
#define M16_PRIO
#endif
// diagnostic build (-DM16_DIAG, never loaded by the package): s_memtime stamps of waves 0 and 4 of workgroup (0, 0) over tiles
// 8 .. 15 of the double-backward passes, read back through ix_diag_m16_read (tools/m16_timeline.py)
#ifdef M16_DIAG
__device__ unsigned long long m16_diag[2 * 8 * 16];
#define M16_STAMP(K)                                                                                         \
    { __builtin_amdgcn_sched_barrier(0);                                                                     \
      if (blockIdx.x == 0 && blockIdx.y == 0 && (wave & 3) == 0 && t >= 8 && t < 16 && lane == 0)            \
          m16_diag[((wave >> 2) * 8 + (t - 8)) * 16 + (K)] = __builtin_amdgcn_s_memtime();                   \
      __builtin_amdgcn_sched_barrier(0); }
extern "C" int ix_diag_m16_read(unsigned long long* host) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(m16_diag), sizeof(m16_diag)) == hipSuccess ? 0 : -1;
}
#else
#define M16_STAMP(K)
#endif
#define M16_OPB 8192   // one streamed operand tile in LDS: 2 planes x 4 panels (16 d each) x 32 rows x 32 bytes
#define M16_PLB 4096
#define M16_PANEL 1024
#define M16_HALF 512   // rows 16 .. 31 of a panel

__device__ __forceinline__ f32x4 m16_mma(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ u32x4 m16_ld128(const unsigned char* p) { return *reinterpret_cast<const u32x4*>(p); }
__device__ __forceinline__ u32x2 m16_tr(const unsigned char* p) {
    return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p)));
}

// A fragments (planes h, l) of k-slice offset OA, row block BLK of the operand tile at OP
#define M16_AFRAG(DST, OP, OA, BLK)                                            \
    DST[0] = m16_ld128((OP) + (BLK) * M16_HALF + (OA));                        \
    DST[1] = m16_ld128((OP) + M16_PLB + (BLK) * M16_HALF + (OA));

// (x0, x1) * sc -> packed fp16 pair h; then l = fp16(x * sc - h): one rounding each (v_fma_mix*_f16 take fp32 and fp16 sources)
__device__ __forceinline__ unsigned m16_cvt_h(float x0, float x1, float sc) {
    unsigned d;
    asm("v_fma_mixlo_f16 %0, %1, %3, 0 op_sel_hi:[0,0,0]\n\t"
        "v_fma_mixhi_f16 %0, %2, %3, 0 op_sel_hi:[0,0,0]"
        : "=&v"(d) : "v"(x0), "v"(x1), "v"(sc));
    return d;
}
__device__ __forceinline__ unsigned m16_cvt_l(float x0, float x1, float sc, unsigned h) {
    unsigned d;
    asm("v_fma_mixlo_f16 %0, %1, %3, -%4 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %0, %2, %3, -%4 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
        : "=&v"(d) : "v"(x0), "v"(x1), "v"(sc), "v"(h));
    return d;
}

// The running power-of-two factor F of an output accumulator (tr form 1 of flash.hip: F only ever decreases; when the
// tile's magnitude bound MXUS = max|x| . unscale times F would reach 2^15, F drops to put it into [2^14, 2^15) and the
// accumulator is rescaled exactly).  The four lanes (g = 0..3) of an owner row feed the same output elements and must agree
// on F: the common path decides on each lane's OWN eight values (no exchange: if nobody's product leaves the range nothing
// changes), the rare path takes the maximum over the four lanes first.
template <bool EXCHANGE>
__device__ __forceinline__ void m16_fit(float mxus, float& F, f32x4 (&acc)[4]) {
    if (__builtin_amdgcn_ballot_w64(mxus * F >= 32768.f)) {
        if (EXCHANGE) {
            mxus = fmaxf(mxus, __shfl_xor(mxus, 16, 64));
            mxus = fmaxf(mxus, __shfl_xor(mxus, 32, 64));
        }
        const unsigned e = (__float_as_uint(mxus) >> 23) & 0xffu;
        const float fn = mxus * F >= 32768.f ? __uint_as_float(min(268u - e, 187u) << 23) : F;
        const int d = (int)(__float_as_uint(fn) >> 23) - (int)(__float_as_uint(F) >> 23) + 127;
        const float rs = d > 0 ? __uint_as_float((unsigned)d << 23) : 0.f;
#pragma unroll
        for (int db = 0; db < 4; ++db) acc[db] *= rs;
        F = fn;
    }
}
// x (this lane's 2 x 4 values of one tile column) . c -> B fragments (h, l); the s_nop statement keeps the matrix
// instructions that read them behind the conversions' wait states (not volatile: it must not pin LDS reads below itself)
__device__ __forceinline__ void m16_split(u32x4 (&bf)[2], const f32x4 (&x)[2], float c) {
    unsigned h0 = m16_cvt_h(x[0][0], x[0][1], c), h1 = m16_cvt_h(x[0][2], x[0][3], c);
    unsigned h2 = m16_cvt_h(x[1][0], x[1][1], c), h3 = m16_cvt_h(x[1][2], x[1][3], c);
    unsigned l0 = m16_cvt_l(x[0][0], x[0][1], c, h0), l1 = m16_cvt_l(x[0][2], x[0][3], c, h1);
    unsigned l2 = m16_cvt_l(x[1][0], x[1][1], c, h2), l3 = m16_cvt_l(x[1][2], x[1][3], c, h3);
    asm("s_nop 1" : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3), "+v"(l0), "+v"(l1), "+v"(l2), "+v"(l3));
    bf[0][0] = h0; bf[0][1] = h1; bf[0][2] = h2; bf[0][3] = h3;
    bf[1][0] = l0; bf[1][1] = l1; bf[1][2] = l2; bf[1][3] = l3;
}
__device__ __forceinline__ float m16_absmax(const f32x4 (&x)[2]) {   // four instructions for eight values
    float m;
    asm("v_max3_f32 %0, |%1|, |%2|, |%3|" : "=v"(m) : "v"(x[0][0]), "v"(x[0][1]), "v"(x[0][2]));
    asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(m) : "v"(x[0][3]), "v"(x[1][0]));
    asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(m) : "v"(x[1][1]), "v"(x[1][2]));
    asm("v_max_f32 %0, %0, |%1|" : "+v"(m) : "v"(x[1][3]));
    return m;
}
// values of unknown range / values in [0, bound]
#define M16_FIT_SPLIT(BF, X, US, F, ACC)                                       \
    { m16_fit<true>(m16_absmax(X) * (US), F, ACC); m16_split(BF, X, (US) * F); }
#define M16_FIT_SPLIT_BOUNDED(BF, X, BOUND, US, F, ACC)                        \
    { m16_fit<false>((BOUND) * (US), F, ACC); m16_split(BF, X, (US) * F); }
// one second-stage product ACC += OP^T . X: the operand's transposed fragments are requested first, in program order, so that
// they sit above the range check's branch and fly during the conversions
#define M16_PRODUCT(ACC, OP, X, US, F)                                         \
    { M16Tr tr_; u32x4 bf_[2]; m16_stage2_load(tr_, OP, offT);                 \
      M16_FIT_SPLIT(bf_, X, US, F, ACC) m16_stage2_mma(ACC, tr_, bf_); }
#define M16_PRODUCT_BOUNDED(ACC, OP, X, BOUND, US, F)                          \
    { M16Tr tr_; u32x4 bf_[2]; m16_stage2_load(tr_, OP, offT);                 \
      M16_FIT_SPLIT_BOUNDED(bf_, X, BOUND, US, F, ACC) m16_stage2_mma(ACC, tr_, bf_); }

// acc[db] (16 d x 16 owner rows) += X^T[d, streamed rows] . bf[streamed rows, owner] for the operand tile at OP: the A
// fragment of d block db comes out of the row-layout image by two transpose reads per plane (rows 4 g + e, 16 + 4 g + e)
struct M16Tr {   // A fragments (planes h, l) of the four d blocks of one streamed operand, transposed
    u32x4 h[4], l[4];
};
__device__ __forceinline__ void m16_stage2_load(M16Tr& a, const unsigned char* op, int offT) {
    u32x4 (&ah)[4] = a.h;
    u32x4 (&al)[4] = a.l;
#pragma unroll
    for (int db = 0; db < 4; ++db) {
        const int o = offT + db * M16_PANEL;
        const u32x2 h0 = m16_tr(op + o), h1 = m16_tr(op + M16_HALF + o);
        const u32x2 l0 = m16_tr(op + M16_PLB + o), l1 = m16_tr(op + M16_PLB + M16_HALF + o);
        ah[db][0] = h0[0]; ah[db][1] = h0[1]; ah[db][2] = h1[0]; ah[db][3] = h1[1];
        al[db][0] = l0[0]; al[db][1] = l0[1]; al[db][2] = l1[0]; al[db][3] = l1[1];
    }
}
__device__ __forceinline__ void m16_stage2_mma(f32x4 (&acc)[4], const M16Tr& a, const u32x4 (&bf)[2]) {
    const u32x4 (&ah)[4] = a.h;
    const u32x4 (&al)[4] = a.l;
    M16_L(
    _Pragma("unroll") for (int db = 0; db < 4; ++db) acc[db] = m16_mma(al[db], bf[0], acc[db]);
    _Pragma("unroll") for (int db = 0; db < 4; ++db) acc[db] = m16_mma(ah[db], bf[1], acc[db]);
    )
#pragma unroll
    for (int db = 0; db < 4; ++db) acc[db] = m16_mma(ah[db], bf[0], acc[db]);
    (void)al;
}

// per-thread staging geometry of one operand tile (512 threads: one 16-byte chunk per plane-row-chunk)
// (thread bits: 0 = 16-byte half of a panel row, 1-2 = row & 3, 3-4 = panel, 5-7 = row >> 2: eight consecutive lanes write 128
// contiguous bytes of LDS, a wave reads 1 KiB of contiguous global memory)
#define M16_STAGE_GEOMETRY                                                                                   \
    const int st_pl = tid >> 8, st_idx = tid & 255;                                                          \
    const int st_row = ((st_idx >> 5) << 2) | ((st_idx >> 1) & 3), st_c = (((st_idx >> 3) & 3) << 1) | (st_idx & 1); \
    const int st_dst = st_pl * M16_PLB + (st_c >> 1) * M16_PANEL + st_row * 32 + ((st_c & 1) << 4);          \
    const int st_off = st_row * 64 + st_c * 8;   /* element offset of the chunk inside the tile's 32 x 64 rows */
// per-lane fragment offsets: A fragment of k-slice 0 / 1 (row n of a block), transpose-read base (rows 4 g + (n >> 2))
#define M16_LANE_GEOMETRY                                                                                    \
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), n = lane & 15, g = lane >> 4; \
    const int offA0 = (g >> 1) * M16_PANEL + n * 32 + ((g & 1) << 4), offA1 = offA0 + 2 * M16_PANEL;        \
    const int offT = (4 * g + (n >> 2)) * 32 + ((n & 3) << 3);                                               \
    (void)offA1; (void)offT;                                                                                 \
    M16_PRIO
// resident B fragments of owner row ROW (element offset of its first d) from row planes PTR
#define M16_BFRAGS(DST, PTR, ELEM, PLANE)                                                                    \
    _Pragma("unroll") for (int ks_ = 0; ks_ < 2; ++ks_) _Pragma("unroll") for (int pl_ = 0; pl_ < 2; ++pl_)   \
        DST[ks_][pl_] = *reinterpret_cast<const u32x4*>((PTR) + (ELEM) + pl_ * (PLANE) + ks_ * 32);
#define M16_ZERO4(A) _Pragma("unroll") for (int db_ = 0; db_ < 4; ++db_) A[db_] = f32x4{0.f, 0.f, 0.f, 0.f};
#define M16_STORE_ROW(ACC, DST, MUL)                                                                         \
    _Pragma("unroll") for (int db_ = 0; db_ < 4; ++db_) *reinterpret_cast<f32x4*>((DST) + 16 * db_) = ACC[db_] * (MUL);

// dropout keep flags (bool: lane masks in scalar registers).  Query-owning: the lane's query RID, keys in registers (pairs
// r = 2 i, 2 i + 1 share a hash); key-owning: the lane's key, queries in registers.
#define M16_MASK_KEYS(KP, RID, T0)                                                                           \
    _Pragma("unroll") for (int blk_ = 0; blk_ < 2; ++blk_) _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) { \
        const int key_ = (T0) + 16 * blk_ + 4 * g + 2 * i_;                                                  \
        const unsigned hsh_ = fl_hash(p.seed_lo, p.seed_hi, (RID), (unsigned)key_ >> 1);                     \
        KP[blk_][2 * i_] = (hsh_ & 0xffffu) >= p.thr16;                                                      \
        KP[blk_][2 * i_ + 1] = (hsh_ >> 16) >= p.thr16;                                                      \
    }
// Key-owning: lanes n and n ^ 1 hold keys 2 j and 2 j + 1 -- the same hash for every query row, different halves of it.  Each of
// the two computes the hashes of HALF of the tile's query rows (rows 2 (n & 1) + {0, 1} of every four) and reads the other half
// out of its partner's registers (v_mov_b32_dpp quad_perm); the lane's half is masked in place and compared there (odd keys: the
// high half against thr16 << 16).  Same mask as fl_hash taken per element, half the hashes (gK / gV: 366 -> 342 instructions per tile).
#define M16_MASK_QUERIES(KP, KEY, T0)                                                                        \
    {                                                                                                        \
        const unsigned odd_ = (unsigned)(KEY) & 1u, half_ = odd_ ? 0xffff0000u : 0xffffu;                    \
        const unsigned thr_ = odd_ ? p.thr16 << 16 : p.thr16;                                                \
        unsigned hq_[2][2];                                                                                  \
        _Pragma("unroll") for (int blk_ = 0; blk_ < 2; ++blk_) _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_) { \
            const int q_ = (T0) + 16 * blk_ + 4 * g + 2 * (int)odd_ + j_;                                    \
            hq_[blk_][j_] = fl_hash(p.seed_lo, p.seed_hi, (unsigned)(bh * p.L + q_), (unsigned)(KEY) >> 1);  \
        }                                                                                                    \
        _Pragma("unroll") for (int blk_ = 0; blk_ < 2; ++blk_) _Pragma("unroll") for (int r_ = 0; r_ < 4; ++r_) { \
            const unsigned own_ = hq_[blk_][r_ & 1];   /* rows 0, 1: the even lane's registers; rows 2, 3: the odd lane's */ \
            const unsigned h_ = (r_ >> 1) ? (unsigned)__builtin_amdgcn_mov_dpp((int)own_, 0xF5, 0xf, 0xf, true)   /* quad_perm [1,1,3,3] */ \
                                          : (unsigned)__builtin_amdgcn_mov_dpp((int)own_, 0xA0, 0xf, 0xf, true);  /* quad_perm [0,0,2,2] */ \
            KP[blk_][r_] = (h_ & half_) >= thr_;                                                             \
        }                                                                                                    \
    }
// Dropout enters every formula through pm = keep ? P : 0 (one select per element); the 1 / keep of M = keep / (1 - p) is folded
// into wave-uniform constants: gy = M o gd -> P o gy = pm (gd cg / keep), M o HD likewise, and products whose whole [L, S]
// operand carries an M (Pd, HgD) take 1 / keep in their conversion scale.
#define M16_PM(B, R, X) (DROP ? (kp[B][R] ? (X) : 0.f) : (X))
// Key bias.  BIAS: the additive bias row of the batch entry ([n][Sp]: 0 / -inf, padded keys and the tail).  !BIAS (no bias
// pointer: no key is masked): nothing is loaded or added -- a query-owning pass sends the scores of keys >= S of its LAST tile
// to -inf itself (P = 0 there; the other tile kinds are finite, the planes are zero beyond S), a key-owning pass needs nothing:
// owner keys >= S are never stored.
#define M16_KB_LOAD(KB, T0)                                                                                  \
    f32x4 KB[2];                                                                                             \
    if (BIAS) {                                                                                              \
        KB[0] = *reinterpret_cast<const f32x4*>(bias + (T0) + 4 * g);                                        \
        KB[1] = *reinterpret_cast<const f32x4*>(bias + (T0) + 16 + 4 * g);                                   \
    }
#define M16_TAIL_KEYS(SC, T0)                                                                                \
    if (!BIAS && LAST) {                                                                                     \
        _Pragma("unroll") for (int blk_ = 0; blk_ < 2; ++blk_) _Pragma("unroll") for (int r_ = 0; r_ < 4; ++r_) \
            if ((T0) + 16 * blk_ + 4 * g + r_ >= p.S) SC[blk_][r_] = -INFINITY;                              \
    }
// The tile loop of a query-owning pass: with a bias every tile is the same; without one the LAST tile is peeled off (its copy
// of the body carries the blanking of keys >= S; a branch inside the loop body would cut the block the compiler schedules the
// matrix and the element-wise instructions in).  BODY(t, slot) is a generic lambda taking the tile index, the ring slot and
// std::bool_constant<LAST>.
#define M16_TILE_LOOP(BODY, NTILES)                                                                          \
    if (BIAS) {                                                                                              \
        for (int t_ = 0; t_ < (NTILES); ++t_) BODY(t_, t_ & 3, std::false_type{});                           \
    } else {                                                                                                 \
        for (int t_ = 0; t_ + 1 < (NTILES); ++t_) BODY(t_, t_ & 3, std::false_type{});                       \
        BODY((NTILES) - 1, ((NTILES) - 1) & 3, std::true_type{});                                            \
    }
// log2 P of element (B, R): S cs + (bias - lse2)
#define M16_ARG(B, R, LSE2) (BIAS ? s[B][R] * cs + (kb[B][R] - (LSE2)) : s[B][R] * cs - (LSE2))

// ============================================================================================================
// forward: query-owning, streams k (scores) and v (P v, transposed reads); online softmax, one query per lane quadruple
// ============================================================================================================
template <bool DROP, bool BIAS>
__global__ __launch_bounds__(512, 2) void M16_N(flash16_fwd_kernel)(FlashArgs p) {
    if (DROP && p.salt) { p.seed_lo ^= p.salt[0]; p.seed_hi ^= p.salt[1]; }
    constexpr int BUFB = 2 * M16_OPB;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * BUFB];
    M16_LANE_GEOMETRY
    M16_STAGE_GEOMETRY
    const int bh = blockIdx.y, b = bh / p.H, h = bh % p.H;
    const int q0 = blockIdx.x * 128 + wave * 16, q = q0 + n;
    const int ntiles = (p.S + 31) / 32;
    const int64_t kro = (int64_t)bh * p.Sp * 64, kbo = (int64_t)bh * (p.Sp / 32);
    const float* bias = p.bias + (int64_t)b * p.Sp;
    u32x4 qf[2][2];
    M16_BFRAGS(qf, p.q_row, ((int64_t)bh * p.Lp + q) * 64 + 8 * g, p.q_plane)
    const float c2 = p.scale_log2e * p.q_us[(int64_t)bh * (p.Lp / 32) + q0 / 32];
    const unsigned rid = (unsigned)(bh * p.L + q);
    f32x4 o[4];
    M16_ZERO4(o)
    float fo = FL_F0, mrun = -1e30f, lrun = 0.f;

    uint4 sk, sv;
    const int64_t st_src = kro + st_pl * p.k_plane + st_off;
#define F16_LOAD(T0)                                                                                         \
    sk = *reinterpret_cast<const uint4*>(p.k_row + st_src + (int64_t)(T0) * 64);                             \
    sv = *reinterpret_cast<const uint4*>(p.v_row + st_src + (int64_t)(T0) * 64);
#define F16_STORE(BUF)                                                                                       \
    *reinterpret_cast<uint4*>((BUF) + st_dst) = sk; *reinterpret_cast<uint4*>((BUF) + M16_OPB + st_dst) = sv;
    F16_LOAD(0)
    F16_STORE(lds)
    __syncthreads();

    auto tile = [&](const int t, const int, auto last_tile) {
        constexpr bool LAST = decltype(last_tile)::value;
        const unsigned char* cur = lds + (t & 1) * BUFB;
        const int t0 = t * 32;
        M16_KB_LOAD(kb, t0)
        const float cs = c2 * p.k_us[kbo + t], usv = p.v_us[kbo + t];
        F16_LOAD(min(t0 + 32, ntiles * 32 - 32))
        f32x4 s[2];
        s[0] = f32x4{0.f, 0.f, 0.f, 0.f}; s[1] = s[0];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int oa = ks ? offA1 : offA0;
            u32x4 k0f[2], k1f[2];
            M16_AFRAG(k0f, cur, oa, 0)
            M16_AFRAG(k1f, cur, oa, 1)
            M16_L(s[0] = m16_mma(k0f[1], qf[ks][0], s[0]); s[1] = m16_mma(k1f[1], qf[ks][0], s[1]);
                  s[0] = m16_mma(k0f[0], qf[ks][1], s[0]); s[1] = m16_mma(k1f[0], qf[ks][1], s[1]);)
            s[0] = m16_mma(k0f[0], qf[ks][0], s[0]); s[1] = m16_mma(k1f[0], qf[ks][0], s[1]);
        }
        M16_TAIL_KEYS(s, t0)
        f32x4 x[2];
        float tmax = -INFINITY;
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                x[blk][r] = BIAS ? s[blk][r] * cs + kb[blk][r] : s[blk][r] * cs;
                tmax = fmaxf(tmax, x[blk][r]);
            }
        {   // the query's maximum over the four lanes that share it
            auto w16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(tmax), __float_as_uint(tmax), false, false);
            tmax = fmaxf(__uint_as_float(w16[0]), __uint_as_float(w16[1]));
            auto w32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(tmax), __float_as_uint(tmax), false, false);
            tmax = fmaxf(__uint_as_float(w32[0]), __uint_as_float(w32[1]));
        }
        const float mn = fmaxf(mrun, tmax);
        const float alpha = fl_exp2(mrun - mn);
        mrun = mn;
        float ps = 0.f;
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                x[blk][r] = fl_exp2(x[blk][r] - mn);
                ps += x[blk][r];
            }
        lrun = lrun * alpha + ps;
        if (__builtin_amdgcn_ballot_w64(alpha != 1.f)) {
#pragma unroll
            for (int db = 0; db < 4; ++db) o[db] *= alpha;
        }
        if (DROP) {
            bool kp[2][4];
            M16_MASK_KEYS(kp, rid, t0)
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int r = 0; r < 4; ++r) x[blk][r] = kp[blk][r] ? x[blk][r] : 0.f;   // (1 / keep is applied once, to O)
        }
        M16_PRODUCT_BOUNDED(o, cur + M16_OPB, x, 1.f, usv, fo)
        F16_STORE(lds + ((t + 1) & 1) * BUFB)
        __syncthreads();
    };
    M16_TILE_LOOP(tile, ntiles)
#undef F16_LOAD
#undef F16_STORE
    lrun += __shfl_xor(lrun, 16, 64);
    lrun += __shfl_xor(lrun, 32, 64);
    if (q < p.L) {
        const float inv = p.inv_keep / (lrun * fo);
        float* dst = p.o1 + ((int64_t)b * p.L + q) * p.ld1 + p.off1 + h * 64 + 4 * g;
        M16_STORE_ROW(o, dst, inv)
        if (g == 0) p.lse[(int64_t)bh * p.Lp + q] = (mrun + log2f(lrun)) * FL_LN2;
    } else if (g == 0) {
        p.lse[(int64_t)bh * p.Lp + q] = INFINITY;   // padded query rows: the derivative kernels then see P = 0 there
    }
}

// ============================================================================================================
// backward (algebra in flash.hip): gQ from query-owning workgroups, gK and gV from key-owning ones
// ============================================================================================================
template <bool DROP, bool BIAS>
__global__ __launch_bounds__(512, 4) void M16_N(flash16_bwd_q_kernel)(FlashArgs p) {
    if (DROP && p.salt) { p.seed_lo ^= p.salt[0]; p.seed_hi ^= p.salt[1]; }
    constexpr int BUFB = 2 * M16_OPB;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * BUFB];
    M16_LANE_GEOMETRY
    M16_STAGE_GEOMETRY
    const int bh = blockIdx.y, b = bh / p.H, h = bh % p.H;
    const int q0 = blockIdx.x * 128 + wave * 16, q = q0 + n;
    const int ntiles = (p.S + 31) / 32;
    const int64_t kro = (int64_t)bh * p.Sp * 64, kbo = (int64_t)bh * (p.Sp / 32);
    const float* bias = p.bias + (int64_t)b * p.Sp;
    u32x4 qf[2][2], df[2][2];
    M16_BFRAGS(qf, p.q_row, ((int64_t)bh * p.Lp + q) * 64 + 8 * g, p.q_plane)
    M16_BFRAGS(df, p.do_row, ((int64_t)bh * p.Lp + q) * 64 + 8 * g, p.q_plane)
    const int64_t qb = (int64_t)bh * (p.Lp / 32) + q0 / 32;
    const float c2 = p.scale_log2e * p.q_us[qb], usd = p.do_us[qb];
    const float lse2 = p.lse[(int64_t)bh * p.Lp + q] * FL_LOG2E, dl = p.delta[(int64_t)bh * p.Lp + q];
    const unsigned rid = (unsigned)(bh * p.L + q);
    f32x4 gq[4];
    M16_ZERO4(gq)
    float fq = FL_F0;

    uint4 sk, sv;
    const int64_t st_src = kro + st_pl * p.k_plane + st_off;
#define Q16_LOAD(T0)                                                                                         \
    sk = *reinterpret_cast<const uint4*>(p.k_row + st_src + (int64_t)(T0) * 64);                             \
    sv = *reinterpret_cast<const uint4*>(p.v_row + st_src + (int64_t)(T0) * 64);
#define Q16_STORE(BUF)                                                                                       \
    *reinterpret_cast<uint4*>((BUF) + st_dst) = sk; *reinterpret_cast<uint4*>((BUF) + M16_OPB + st_dst) = sv;
    Q16_LOAD(0)
    Q16_STORE(lds)
    __syncthreads();

    auto tile = [&](const int t, const int, auto last_tile) {
        constexpr bool LAST = decltype(last_tile)::value;
        const unsigned char* cur = lds + (t & 1) * BUFB;
        const int t0 = t * 32;
        M16_KB_LOAD(kb, t0)
        const float usk = p.k_us[kbo + t];
        const float cs = c2 * usk, cgk = usd * p.v_us[kbo + t] * p.inv_keep;
        Q16_LOAD(min(t0 + 32, ntiles * 32 - 32))
        f32x4 s[2], gd[2];
        s[0] = f32x4{0.f, 0.f, 0.f, 0.f}; s[1] = s[0]; gd[0] = s[0]; gd[1] = s[0];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int oa = ks ? offA1 : offA0;
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                u32x4 kf[2], vf[2];
                M16_AFRAG(kf, cur, oa, blk)
                M16_AFRAG(vf, cur + M16_OPB, oa, blk)
                M16_L(s[blk] = m16_mma(kf[1], qf[ks][0], s[blk]); gd[blk] = m16_mma(vf[1], df[ks][0], gd[blk]);
                      s[blk] = m16_mma(kf[0], qf[ks][1], s[blk]); gd[blk] = m16_mma(vf[0], df[ks][1], gd[blk]);)
                s[blk] = m16_mma(kf[0], qf[ks][0], s[blk]); gd[blk] = m16_mma(vf[0], df[ks][0], gd[blk]);
            }
        }
        bool kp[2][4];
        if (DROP) { M16_MASK_KEYS(kp, rid, t0) }
        M16_TAIL_KEYS(s, t0)
        f32x4 x[2];
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int r = 0; r < 4; ++r) {   // gs = P o (M o gd - t) = pm gd' - P t
                const float pr = fl_exp2(M16_ARG(blk, r, lse2));
                x[blk][r] = M16_PM(blk, r, pr) * (gd[blk][r] * cgk) - pr * dl;
            }
        M16_PRODUCT(gq, cur, x, usk, fq)   // gQ^T[d, query] += K^T[d, key] gs^T[key, query]
        Q16_STORE(lds + ((t + 1) & 1) * BUFB)
        __syncthreads();
    };
    M16_TILE_LOOP(tile, ntiles)
#undef Q16_LOAD
#undef Q16_STORE
    if (q < p.L) {
        float* dst = p.o1 + ((int64_t)b * p.L + q) * p.ld1 + p.off1 + h * 64 + 4 * g;
        const float mq = p.scale / fq;
        M16_STORE_ROW(gq, dst, mq)
    }
}

// ---- split-phase passes (one workgroup per CU: gK / gV, dq / ddO, dk / dv) ---------------------------------------------
// A wave issues in order: a matrix instruction that finds the SIMD's matrix pipe taken by its partner wave blocks every
// vector instruction behind it, so two waves whose streams both mix matrix and vector work take about the SUM of their
// times (measured: lock-stepped phases and de-phased ones alike, profiles/r5*_m16_*.txt).  These passes therefore
// separate the two kinds of work in time: a tile is a V segment -- element-wise arithmetic, range checks and fp16
// conversions of tile t, no matrix instruction -- and an M segment -- the second-stage products of tile t out of the converted
// fragments, then the first-stage products of tile t + 1, (almost) no vector instruction -- with a barrier after each, and
// waves 4..7 run one segment behind waves 0..3 (one extra barrier before their loop, the others one after theirs): at any
// time one wave of a SIMD streams matrix instructions while its partner issues vector instructions.  Products that go
// into the same accumulator share one range check (their fragments are converted before either product runs).  The
// staging ring is four tiles deep: tile t is read in M(t) (first stage) and M(t + 1) (second stage) by both halves, and
// tile t + 2 is written at the end of V(t).
// Waves whose 16 owner rows all lie beyond the tensor (the tail of the last workgroup of a row: 2 060 rows = 16 x 128 + 12 leaves
// seven of its eight waves empty) stage their share of every tile and keep the barriers, nothing else (`live`): the last
// workgroup of a row then costs a fraction of a full one -- it is the whole second round of the grid at two episodes (17 x 16
// workgroups on 256 CUs, one per CU).  (The passes that run two workgroups per CU have room for that round anyway.)
#define M16_LAG_IF(COND) if (COND) __syncthreads();
#define M16_UPPER_HALF (__builtin_amdgcn_readfirstlane(threadIdx.x) >= 256)
#define M16_NSLOT 4

template <bool DROP, bool BIAS>
__global__ __launch_bounds__(512, 2) void M16_N(flash16_bwd_kv_kernel)(FlashArgs p) {
    if (DROP && p.salt) { p.seed_lo ^= p.salt[0]; p.seed_hi ^= p.salt[1]; }
    constexpr int OFF_ST = 2 * M16_OPB, BUFB = OFF_ST + 256;   // q, dO rows + lse[32], delta[32]
    __shared__ __attribute__((aligned(16))) unsigned char lds[M16_NSLOT * BUFB];
    M16_LANE_GEOMETRY
    M16_STAGE_GEOMETRY
    const int bh = blockIdx.y, b = bh / p.H, h = bh % p.H;
    const int k0 = blockIdx.x * 128 + wave * 16, key = k0 + n;
    const int ntiles = (p.L + 31) / 32, last = ntiles * 32 - 32;
    const int64_t qro = (int64_t)bh * p.Lp * 64, sto = (int64_t)bh * p.Lp, qbo = (int64_t)bh * (p.Lp / 32);
    u32x4 kf[2][2], vf[2][2];
    M16_BFRAGS(kf, p.k_row, ((int64_t)bh * p.Sp + key) * 64 + 8 * g, p.k_plane)
    M16_BFRAGS(vf, p.v_row, ((int64_t)bh * p.Sp + key) * 64 + 8 * g, p.k_plane)
    const int64_t kbk = (int64_t)bh * (p.Sp / 32) + k0 / 32;
    const float c2 = p.scale_log2e * p.k_us[kbk], usv = p.v_us[kbk];
    const float kbias = BIAS ? p.bias[(int64_t)b * p.Sp + key] : 0.f;
    f32x4 gk[4], gv[4];
    M16_ZERO4(gk)
    M16_ZERO4(gv)
    float fk = FL_F0, fv = FL_F0;

    uint4 sq, sd;
    float sst = 0.f;   // threads 0..31: lse, 32..63: delta of the staged tile
    const int64_t st_src = qro + st_pl * p.q_plane + st_off;
#define K16_LOAD(T0)                                                                                         \
    sq = *reinterpret_cast<const uint4*>(p.q_row + st_src + (int64_t)(T0) * 64);                             \
    sd = *reinterpret_cast<const uint4*>(p.do_row + st_src + (int64_t)(T0) * 64);                            \
    if (tid < 64) sst = tid < 32 ? p.lse[sto + (T0) + tid] : p.delta[sto + (T0) + tid - 32];
#define K16_STORE(BUF)                                                                                       \
    *reinterpret_cast<uint4*>((BUF) + st_dst) = sq; *reinterpret_cast<uint4*>((BUF) + M16_OPB + st_dst) = sd; \
    if (tid < 64) reinterpret_cast<float*>((BUF) + OFF_ST)[tid] = tid < 32 ? sst * FL_LOG2E : sst;
    // first stage of the tile at BUF: S[query, key] = Q K^T, gd = dO V^T
#define K16_PHASE1(BUF)                                                                                      \
    s[0] = f32x4{0.f, 0.f, 0.f, 0.f}; s[1] = s[0]; gd[0] = s[0]; gd[1] = s[0];                               \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                       \
        const int oa = ks ? offA1 : offA0;                                                                   \
        _Pragma("unroll") for (int blk = 0; blk < 2; ++blk) {                                                \
            u32x4 qa[2], da[2];                                                                              \
            M16_AFRAG(qa, (BUF), oa, blk)                                                                    \
            M16_AFRAG(da, (BUF) + M16_OPB, oa, blk)                                                          \
            M16_L(s[blk] = m16_mma(qa[1], kf[ks][0], s[blk]); gd[blk] = m16_mma(da[1], vf[ks][0], gd[blk]);  \
                  s[blk] = m16_mma(qa[0], kf[ks][1], s[blk]); gd[blk] = m16_mma(da[0], vf[ks][1], gd[blk]);) \
            s[blk] = m16_mma(qa[0], kf[ks][0], s[blk]); gd[blk] = m16_mma(da[0], vf[ks][0], gd[blk]);       \
        }                                                                                                    \
    }
    K16_LOAD(0)
    K16_STORE(lds)
    K16_LOAD(min(32, last))
    K16_STORE(lds + BUFB)
    __syncthreads();
    M16_LAG_IF(M16_UPPER_HALF)
    const bool live = k0 < p.S;   // (wave-uniform)
    f32x4 s[2], gd[2];
    u32x4 bfv[2], bfk[2];
    if (live) { K16_PHASE1(lds) }
    __syncthreads();

    for (int t = 0, slot = 0; t < ntiles; ++t, slot = (slot + 1) & 3) {
        const unsigned char* cur = lds + slot * BUFB;
        const int t0 = t * 32;
        // ---- V(t) ----
        const float usq = p.q_us[qbo + t], usd = p.do_us[qbo + t];
        const float cs = c2 * usq, cgk = usv * usd * p.inv_keep;
        K16_LOAD(min(t0 + 64, last))
        if (live) {
        const float* st = reinterpret_cast<const float*>(cur + OFF_ST);
        f32x4 lse2[2], dl[2];
        lse2[0] = *reinterpret_cast<const f32x4*>(st + 4 * g); lse2[1] = *reinterpret_cast<const f32x4*>(st + 16 + 4 * g);
        dl[0] = *reinterpret_cast<const f32x4*>(st + 32 + 4 * g); dl[1] = *reinterpret_cast<const f32x4*>(st + 48 + 4 * g);
        bool kp[2][4];
        if (DROP) { M16_MASK_QUERIES(kp, key, t0) }
        f32x4 pd[2], gs[2];
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pr = fl_exp2(BIAS ? s[blk][r] * cs + (kbias - lse2[blk][r]) : s[blk][r] * cs - lse2[blk][r]);
                pd[blk][r] = M16_PM(blk, r, pr);                                  // (x 1/keep at the end, on gV)
                gs[blk][r] = pd[blk][r] * (gd[blk][r] * cgk) - pr * dl[blk][r];   // gs = P o (M o gd - t)
            }
        M16_FIT_SPLIT_BOUNDED(bfv, pd, 1.f, usd, fv, gv)
        M16_FIT_SPLIT(bfk, gs, usq, fk, gk)
        }
        K16_STORE(lds + ((slot + 2) & 3) * BUFB)
        __syncthreads();
        // ---- M(t + 1) ----
        if (live) {
            M16Tr ta, tb;
            m16_stage2_load(ta, cur + M16_OPB, offT);
            m16_stage2_load(tb, cur, offT);
            m16_stage2_mma(gv, ta, bfv);   // gV^T[d, key] += dO^T[d, query] Pd[query, key]
            m16_stage2_mma(gk, tb, bfk);   // gK^T[d, key] += Q^T[d, query] gs[query, key]
            if (t + 1 < ntiles) { K16_PHASE1(lds + ((slot + 1) & 3) * BUFB) }
        }
        __syncthreads();
    }
    M16_LAG_IF(!M16_UPPER_HALF)
#undef K16_LOAD
#undef K16_STORE
#undef K16_PHASE1
    if (key < p.S) {
        float* dk = p.o2 + ((int64_t)b * p.S + key) * p.ld2 + p.off2 + h * 64 + 4 * g;
        float* dv = p.o3 + ((int64_t)b * p.S + key) * p.ld3 + p.off3 + h * 64 + 4 * g;
        const float mk = p.scale / fk, mv = p.inv_keep / fv;
        M16_STORE_ROW(gk, dk, mk)
        M16_STORE_ROW(gv, dv, mv)
    }
}

// ============================================================================================================
// double backward (algebra in flash.hip): (1) row statistics u, w; (2) dq, ddO (query-owning); (3) dk, dv (key-owning)
// ============================================================================================================
// common to (1) and (2): the query-owning setup and the first stage of a key tile
#define B16_SETUP                                                                                            \
    M16_LANE_GEOMETRY                                                                                        \
    M16_STAGE_GEOMETRY                                                                                       \
    const int bh = blockIdx.y, b = bh / p.H, h = bh % p.H;                                                   \
    const int q0 = blockIdx.x * 128 + wave * 16, q = q0 + n;                                                 \
    const int ntiles = (p.S + 31) / 32, last = ntiles * 32 - 32;                                             \
    const int64_t kro = (int64_t)bh * p.Sp * 64, kbo = (int64_t)bh * (p.Sp / 32);                            \
    const float* bias = p.bias + (int64_t)b * p.Sp;                                                          \
    u32x4 qf[2][2], hqf[2][2], df[2][2];                                                                     \
    M16_BFRAGS(qf, p.q_row, ((int64_t)bh * p.Lp + q) * 64 + 8 * g, p.q_plane)                                \
    M16_BFRAGS(hqf, p.hq_row, ((int64_t)bh * p.Lp + q) * 64 + 8 * g, p.q_plane)                              \
    M16_BFRAGS(df, p.do_row, ((int64_t)bh * p.Lp + q) * 64 + 8 * g, p.q_plane)                               \
    const int64_t qb = (int64_t)bh * (p.Lp / 32) + q0 / 32;                                                  \
    const float usq = p.q_us[qb], usd = p.do_us[qb], ushq = p.hq_us[qb];                                     \
    const int64_t so = (int64_t)bh * p.Lp + q;                                                               \
    const float lse2 = p.lse[so] * FL_LOG2E, dl = p.delta[so];                                               \
    const unsigned rid = (unsigned)(bh * p.L + q);                                                           \
    (void)b; (void)h; (void)last;                                                                            \
    uint4 sk, shk, sv, shv;                                                                                  \
    const int64_t st_src = kro + st_pl * p.k_plane + st_off;
#define B16_LOAD(T0)                                                                                         \
    sk = *reinterpret_cast<const uint4*>(p.k_row + st_src + (int64_t)(T0) * 64);                             \
    shk = *reinterpret_cast<const uint4*>(p.hk_row + st_src + (int64_t)(T0) * 64);                           \
    sv = *reinterpret_cast<const uint4*>(p.v_row + st_src + (int64_t)(T0) * 64);                             \
    shv = *reinterpret_cast<const uint4*>(p.hv_row + st_src + (int64_t)(T0) * 64);
#define B16_STORE(BUF)                                                                                       \
    *reinterpret_cast<uint4*>((BUF) + st_dst) = sk; *reinterpret_cast<uint4*>((BUF) + M16_OPB + st_dst) = shk; \
    *reinterpret_cast<uint4*>((BUF) + 2 * M16_OPB + st_dst) = sv; *reinterpret_cast<uint4*>((BUF) + 3 * M16_OPB + st_dst) = shv;
// the [key, query] tiles: S, gd, G (two chains: hq.k and q.hk carry different block scales), HD
#define B16_PHASE1(BUF)                                                                                      \
    _Pragma("unroll") for (int blk = 0; blk < 2; ++blk) { s[blk] = f32x4{0.f, 0.f, 0.f, 0.f}; gd[blk] = s[blk]; g1[blk] = s[blk]; g2[blk] = s[blk]; hd_[blk] = s[blk]; } \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                       \
        const int oa = ks ? offA1 : offA0;                                                                   \
        _Pragma("unroll") for (int blk = 0; blk < 2; ++blk) {                                                \
            u32x4 kf[2], hkf[2], vf[2], hvf[2];                                                              \
            M16_AFRAG(kf, (BUF), oa, blk)                                                                    \
            M16_AFRAG(hkf, (BUF) + M16_OPB, oa, blk)                                                         \
            M16_AFRAG(vf, (BUF) + 2 * M16_OPB, oa, blk)                                                      \
            M16_AFRAG(hvf, (BUF) + 3 * M16_OPB, oa, blk)                                                     \
            M16_L(B16_TERM(1, 0) B16_TERM(0, 1)) B16_TERM(0, 0)                                                     \
        }                                                                                                    \
    }
#define B16_TERM(I, J)                                                                                       \
    s[blk] = m16_mma(kf[I], qf[ks][J], s[blk]); gd[blk] = m16_mma(vf[I], df[ks][J], gd[blk]);               \
    g1[blk] = m16_mma(kf[I], hqf[ks][J], g1[blk]); hd_[blk] = m16_mma(hvf[I], df[ks][J], hd_[blk]);         \
    g2[blk] = m16_mma(hkf[I], qf[ks][J], g2[blk]);
// element-wise part common to (1) and (2): pr = P, pm = keep ? P : 0, g1 = G, gd = gd cg / keep, hd_ = HD ch / keep
// (P o gy = pm gd, P o M o HD = pm hd_)
#define B16_ELEMENTWISE                                                                                      \
    M16_KB_LOAD(kb, t0)                                                                                      \
    M16_TAIL_KEYS(s, t0)                                                                                     \
    const float usk = p.k_us[kbo + t], ushk = p.hk_us[kbo + t], usv = p.v_us[kbo + t], ushv = p.hv_us[kbo + t]; \
    const float cs = p.scale_log2e * usq * usk, cgk = usd * usv * p.inv_keep, c1 = p.scale * ushq * usk,     \
                c3 = p.scale * usq * ushk, chk = usd * ushv * p.inv_keep;                                    \
    bool kp[2][4];                                                                                           \
    if (DROP) { M16_MASK_KEYS(kp, rid, t0) }                                                                 \
    f32x4 pr[2], pm[2];                                                                                      \
    _Pragma("unroll") for (int blk = 0; blk < 2; ++blk) _Pragma("unroll") for (int r = 0; r < 4; ++r) {       \
        pr[blk][r] = fl_exp2(M16_ARG(blk, r, lse2));                                                         \
        pm[blk][r] = M16_PM(blk, r, pr[blk][r]);                                                             \
        g1[blk][r] = g1[blk][r] * c1 + g2[blk][r] * c3;                                                      \
        gd[blk][r] *= cgk;                                                                                   \
        hd_[blk][r] *= chk;                                                                                  \
    }

// (1) row statistics: two workgroups per CU, double-buffered, one barrier per tile.  u = sum P G and w = sum pm (G gd' + hd') - 2 t u:
// the two masked sums of w share their multiply by pm.  (Four waves per SIMD asked for explicitly: at two the compiler's schedule
// of this pass sits at 127-129 registers, and 129 would leave one workgroup per CU.)
template <bool DROP, bool BIAS>
__global__ __launch_bounds__(512, 4) void M16_N(flash16_bb_stats_kernel)(FlashArgs p) {
    if (DROP && p.salt) { p.seed_lo ^= p.salt[0]; p.seed_hi ^= p.salt[1]; }
    constexpr int BUFB = 4 * M16_OPB;   // k, hk, v, hv rows
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * BUFB];
    B16_SETUP
    float uu = 0.f, ab = 0.f;
    B16_LOAD(0)
    B16_STORE(lds)
    __syncthreads();
    auto tile = [&](const int t, const int, auto last_tile) {
        constexpr bool LAST = decltype(last_tile)::value;
        const unsigned char* cur = lds + (t & 1) * BUFB;
        const int t0 = t * 32;
        B16_LOAD(min(t0 + 32, last))
        f32x4 s[2], gd[2], g1[2], g2[2], hd_[2];
        B16_PHASE1(cur)
        M16_KB_LOAD(kb, t0)
        const float usk = p.k_us[kbo + t], ushk = p.hk_us[kbo + t], usv = p.v_us[kbo + t], ushv = p.hv_us[kbo + t];
        const float cs = p.scale_log2e * usq * usk, cgk = usd * usv * p.inv_keep, c1 = p.scale * ushq * usk,
                    c3 = p.scale * usq * ushk, chk = usd * ushv * p.inv_keep;
        bool kp[2][4];
        if (DROP) { M16_MASK_KEYS(kp, rid, t0) }
        M16_TAIL_KEYS(s, t0)
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int r = 0; r < 4; ++r) {   // w = sum pm (G gd' + hd') - 2 t u: the two masked sums as one
                const float pr = fl_exp2(M16_ARG(blk, r, lse2));
                const float G = g1[blk][r] * c1 + g2[blk][r] * c3;
                uu += pr * G;
                ab += M16_PM(blk, r, pr) * (G * (gd[blk][r] * cgk) + hd_[blk][r] * chk);
            }
        B16_STORE(lds + ((t + 1) & 1) * BUFB)
        __syncthreads();
    };
    M16_TILE_LOOP(tile, ntiles)
    uu += __shfl_xor(uu, 16, 64); uu += __shfl_xor(uu, 32, 64);
    ab += __shfl_xor(ab, 16, 64); ab += __shfl_xor(ab, 32, 64);
    if (g == 0) {   // (padded queries: P = 0 -> zeros; the whole [BH][Lp] workspace is written)
        p.u[so] = uu;
        p.w[so] = ab - 2.f * dl * uu;
    }
}

// (2) dq (o1), ddO (o4): split-phase
template <bool DROP, bool BIAS>
__global__ __launch_bounds__(512, 2) void M16_N(flash16_bb_q_kernel)(FlashArgs p) {
    if (DROP && p.salt) { p.seed_lo ^= p.salt[0]; p.seed_hi ^= p.salt[1]; }
    constexpr int BUFB = 4 * M16_OPB;   // k, hk, v, hv rows
    __shared__ __attribute__((aligned(16))) unsigned char lds[M16_NSLOT * BUFB];
    B16_SETUP
    const float uu = p.u[so], ww = p.w[so];
    f32x4 dq[4], ddo[4];
    M16_ZERO4(dq)
    M16_ZERO4(ddo)
    float fdq = FL_F0, fddo = FL_F0;
    B16_LOAD(0)
    B16_STORE(lds)
    B16_LOAD(min(32, last))
    B16_STORE(lds + BUFB)
    __syncthreads();
    M16_LAG_IF(M16_UPPER_HALF)
    const bool live = q0 < p.L;   // (wave-uniform)
    f32x4 s[2], gd[2], g1[2], g2[2], hd_[2];
    u32x4 bf1[2], bf2[2], bf3[2], bf4[2];
    if (live) { B16_PHASE1(lds) }
    __syncthreads();

    auto tile = [&](const int t, const int slot, auto last_tile) {
        constexpr bool LAST = decltype(last_tile)::value;
        const unsigned char* cur = lds + slot * BUFB;
        const int t0 = t * 32;
        M16_STAMP(0)
        // ---- V(t) ----
        B16_LOAD(min(t0 + 64, last))
        if (live) {
        B16_ELEMENTWISE
        f32x4 x1[2], x2[2], x4[2];
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                x1[blk][r] = pm[blk][r] * gd[blk][r] - pr[blk][r] * dl;                                     // gs = P (gy - t)
                x2[blk][r] = g1[blk][r] * x1[blk][r] + pm[blk][r] * (hd_[blk][r] - gd[blk][r] * uu) - pr[blk][r] * ww;   // HS
                x4[blk][r] = pm[blk][r] * (g1[blk][r] - uu);                                                 // HgD keep (the 1 / keep rides on its scale)
            }
        // dq += hk^T gs + k^T HS: one range check for both;  ddO += hv^T Pd + v^T HgD likewise (Pd = pm / keep <= 1 / keep)
        m16_fit<true>(fmaxf(m16_absmax(x1) * ushk, m16_absmax(x2) * usk), fdq, dq);
        m16_split(bf1, x1, ushk * fdq);
        m16_split(bf2, x2, usk * fdq);
        const float usvk = usv * p.inv_keep, ushvk = ushv * p.inv_keep;
        m16_fit<true>(fmaxf(ushvk, m16_absmax(x4) * usvk), fddo, ddo);
        m16_split(bf3, pm, ushvk * fddo);
        m16_split(bf4, x4, usvk * fddo);
        }
        B16_STORE(lds + ((slot + 2) & 3) * BUFB)
        M16_STAMP(1)
        __syncthreads();
        M16_STAMP(2)
        // ---- M(t + 1) ----
        if (live) {
            M16Tr ta, tb;   // (two sets: the next operand's transposed fragments fly during a product)
            m16_stage2_load(ta, cur + M16_OPB, offT);
            m16_stage2_load(tb, cur + 3 * M16_OPB, offT);
            m16_stage2_mma(dq, ta, bf1);
            m16_stage2_load(ta, cur, offT);
            m16_stage2_mma(ddo, tb, bf3);
            m16_stage2_load(tb, cur + 2 * M16_OPB, offT);
            m16_stage2_mma(dq, ta, bf2);
            m16_stage2_mma(ddo, tb, bf4);
            if (t + 1 < ntiles) { B16_PHASE1(lds + ((slot + 1) & 3) * BUFB) }
        }
        M16_STAMP(3)
        __syncthreads();
        M16_STAMP(4)
    };
    M16_TILE_LOOP(tile, ntiles)
    M16_LAG_IF(!M16_UPPER_HALF)
    if (q < p.L) {
        float* d1 = p.o1 + ((int64_t)b * p.L + q) * p.ld1 + p.off1 + h * 64 + 4 * g;
        float* d4 = p.o4 + ((int64_t)b * p.L + q) * p.ld4 + p.off4 + h * 64 + 4 * g;
        const float m1 = p.scale / fdq, m4 = 1.f / fddo;
        M16_STORE_ROW(dq, d1, m1)
        M16_STORE_ROW(ddo, d4, m4)
    }
}
#undef B16_SETUP
#undef B16_LOAD
#undef B16_STORE
#undef B16_PHASE1
#undef B16_TERM
#undef B16_ELEMENTWISE

// (3) dk (o2), dv (o3): split-phase, key-owning
template <bool DROP, bool BIAS>
__global__ __launch_bounds__(512, 2) void M16_N(flash16_bb_kv_kernel)(FlashArgs p) {
    if (DROP && p.salt) { p.seed_lo ^= p.salt[0]; p.seed_hi ^= p.salt[1]; }
    constexpr int OFF_ST = 3 * M16_OPB, BUFB = OFF_ST + 512;   // q, hq, dO rows + lse, delta, u, w [32] each
    __shared__ __attribute__((aligned(16))) unsigned char lds[M16_NSLOT * BUFB];
    M16_LANE_GEOMETRY
    M16_STAGE_GEOMETRY
    const int bh = blockIdx.y, b = bh / p.H, h = bh % p.H;
    const int k0 = blockIdx.x * 128 + wave * 16, key = k0 + n;
    const int ntiles = (p.L + 31) / 32, last = ntiles * 32 - 32;
    const int64_t qro = (int64_t)bh * p.Lp * 64, sto = (int64_t)bh * p.Lp, qbo = (int64_t)bh * (p.Lp / 32);
    u32x4 kf[2][2], vf[2][2], hkf[2][2], hvf[2][2];
    M16_BFRAGS(kf, p.k_row, ((int64_t)bh * p.Sp + key) * 64 + 8 * g, p.k_plane)
    M16_BFRAGS(vf, p.v_row, ((int64_t)bh * p.Sp + key) * 64 + 8 * g, p.k_plane)
    M16_BFRAGS(hkf, p.hk_row, ((int64_t)bh * p.Sp + key) * 64 + 8 * g, p.k_plane)
    M16_BFRAGS(hvf, p.hv_row, ((int64_t)bh * p.Sp + key) * 64 + 8 * g, p.k_plane)
    const int64_t kbk = (int64_t)bh * (p.Sp / 32) + k0 / 32;
    const float usk = p.k_us[kbk], ushk = p.hk_us[kbk], usv = p.v_us[kbk], ushv = p.hv_us[kbk];
    const float kbias = BIAS ? p.bias[(int64_t)b * p.Sp + key] : 0.f;
    f32x4 dk[4], dv[4];
    M16_ZERO4(dk)
    M16_ZERO4(dv)
    float fdk = FL_F0, fdv = FL_F0;

    uint4 sq, shq, sd;
    float sst = 0.f;   // threads 0..127 carry lse | delta | u | w of the staged tile
    const int64_t st_src = qro + st_pl * p.q_plane + st_off;
#define C16_LOAD(T0)                                                                                         \
    sq = *reinterpret_cast<const uint4*>(p.q_row + st_src + (int64_t)(T0) * 64);                             \
    shq = *reinterpret_cast<const uint4*>(p.hq_row + st_src + (int64_t)(T0) * 64);                           \
    sd = *reinterpret_cast<const uint4*>(p.do_row + st_src + (int64_t)(T0) * 64);                            \
    if (tid < 128) {                                                                                         \
        const float* sp_ = tid < 32 ? p.lse : tid < 64 ? p.delta : tid < 96 ? p.u : p.w;                     \
        sst = sp_[sto + (T0) + (tid & 31)];                                                                  \
    }
#define C16_STORE(BUF)                                                                                       \
    *reinterpret_cast<uint4*>((BUF) + st_dst) = sq; *reinterpret_cast<uint4*>((BUF) + M16_OPB + st_dst) = shq; \
    *reinterpret_cast<uint4*>((BUF) + 2 * M16_OPB + st_dst) = sd;                                            \
    if (tid < 128) reinterpret_cast<float*>((BUF) + OFF_ST)[tid] = tid < 32 ? sst * FL_LOG2E : sst;
#define C16_TERM(I, J)                                                                                       \
    s[blk] = m16_mma(qa[I], kf[ks][J], s[blk]); gd[blk] = m16_mma(da[I], vf[ks][J], gd[blk]);               \
    g1[blk] = m16_mma(hqa[I], kf[ks][J], g1[blk]); hd_[blk] = m16_mma(da[I], hvf[ks][J], hd_[blk]);         \
    g2[blk] = m16_mma(qa[I], hkf[ks][J], g2[blk]);
#define C16_PHASE1(BUF)                                                                                      \
    _Pragma("unroll") for (int blk = 0; blk < 2; ++blk) { s[blk] = f32x4{0.f, 0.f, 0.f, 0.f}; gd[blk] = s[blk]; g1[blk] = s[blk]; g2[blk] = s[blk]; hd_[blk] = s[blk]; } \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                       \
        const int oa = ks ? offA1 : offA0;                                                                   \
        _Pragma("unroll") for (int blk = 0; blk < 2; ++blk) {                                                \
            u32x4 qa[2], hqa[2], da[2];                                                                      \
            M16_AFRAG(qa, (BUF), oa, blk)                                                                    \
            M16_AFRAG(hqa, (BUF) + M16_OPB, oa, blk)                                                         \
            M16_AFRAG(da, (BUF) + 2 * M16_OPB, oa, blk)                                                      \
            M16_L(C16_TERM(1, 0) C16_TERM(0, 1)) C16_TERM(0, 0)                                                     \
        }                                                                                                    \
    }
    C16_LOAD(0)
    C16_STORE(lds)
    C16_LOAD(min(32, last))
    C16_STORE(lds + BUFB)
    __syncthreads();
    M16_LAG_IF(M16_UPPER_HALF)
    const bool live = k0 < p.S;   // (wave-uniform)
    f32x4 s[2], gd[2], g1[2], g2[2], hd_[2];
    u32x4 bf1[2], bf2[2], bf3[2];
    if (live) { C16_PHASE1(lds) }
    __syncthreads();

    for (int t = 0, slot = 0; t < ntiles; ++t, slot = (slot + 1) & 3) {
        const unsigned char* cur = lds + slot * BUFB;
        const int t0 = t * 32;
        // ---- V(t) ----
        const float usq = p.q_us[qbo + t], ushq = p.hq_us[qbo + t], usd = p.do_us[qbo + t];
        const float cs = p.scale_log2e * usq * usk, cgk = usd * usv * p.inv_keep, c1 = p.scale * ushq * usk, c3 = p.scale * usq * ushk,
                    chk = usd * ushv * p.inv_keep;
        C16_LOAD(min(t0 + 64, last))
        if (live) {
        // statistics of this lane's 2 x 4 queries (rows 16 blk + 4 g + r of the tile)
        const float* st = reinterpret_cast<const float*>(cur + OFF_ST);
        f32x4 lse2[2], dl[2], uu[2], ww[2];
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            lse2[blk] = *reinterpret_cast<const f32x4*>(st + 16 * blk + 4 * g);
            dl[blk] = *reinterpret_cast<const f32x4*>(st + 32 + 16 * blk + 4 * g);
            uu[blk] = *reinterpret_cast<const f32x4*>(st + 64 + 16 * blk + 4 * g);
            ww[blk] = *reinterpret_cast<const f32x4*>(st + 96 + 16 * blk + 4 * g);
        }
        bool kp[2][4];
        if (DROP) { M16_MASK_QUERIES(kp, key, t0) }
        // pr = P, pm = keep ? P : 0, g1 = G, gd = gd cg / keep, hd_ = HD ch / keep   (P o gy = pm gd, P o M o HD = pm hd_)
        f32x4 x1[2], x2[2], x3[2];
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pr = fl_exp2(BIAS ? s[blk][r] * cs + (kbias - lse2[blk][r]) : s[blk][r] * cs - lse2[blk][r]);
                const float pm = M16_PM(blk, r, pr);
                const float G = g1[blk][r] * c1 + g2[blk][r] * c3, gdk = gd[blk][r] * cgk, hdk = hd_[blk][r] * chk;
                x1[blk][r] = pm * gdk - pr * dl[blk][r];                                              // gs
                x2[blk][r] = G * x1[blk][r] + pm * (hdk - gdk * uu[blk][r]) - pr * ww[blk][r];        // HS
                x3[blk][r] = pm * (G - uu[blk][r]);                                                    // HgD keep
            }
        // dk += hq^T gs + q^T HS: one range check for both;  dv += dO^T HgD
        m16_fit<true>(fmaxf(m16_absmax(x1) * ushq, m16_absmax(x2) * usq), fdk, dk);
        m16_split(bf1, x1, ushq * fdk);
        m16_split(bf2, x2, usq * fdk);
        const float usdk = usd * p.inv_keep;
        m16_fit<true>(m16_absmax(x3) * usdk, fdv, dv);
        m16_split(bf3, x3, usdk * fdv);
        }
        C16_STORE(lds + ((slot + 2) & 3) * BUFB)
        __syncthreads();
        // ---- M(t + 1) ----
        if (live) {
            M16Tr ta, tb;   // (two sets: the next operand's transposed fragments fly during a product)
            m16_stage2_load(ta, cur + M16_OPB, offT);
            m16_stage2_load(tb, cur + 2 * M16_OPB, offT);
            m16_stage2_mma(dk, ta, bf1);
            m16_stage2_load(ta, cur, offT);
            m16_stage2_mma(dv, tb, bf3);
            m16_stage2_mma(dk, ta, bf2);
            if (t + 1 < ntiles) { C16_PHASE1(lds + ((slot + 1) & 3) * BUFB) }
        }
        __syncthreads();
    }
    M16_LAG_IF(!M16_UPPER_HALF)
#undef C16_LOAD
#undef C16_STORE
#undef C16_TERM
#undef C16_PHASE1
    if (key < p.S) {
        float* d2 = p.o2 + ((int64_t)b * p.S + key) * p.ld2 + p.off2 + h * 64 + 4 * g;
        float* d3 = p.o3 + ((int64_t)b * p.S + key) * p.ld3 + p.off3 + h * 64 + 4 * g;
        const float m2 = p.scale / fdk, m3 = 1.f / fdv;
        M16_STORE_ROW(dk, d2, m2)
        M16_STORE_ROW(dv, d3, m3)
    }
}

// ------------------------------------------------------------------------------------------------------------
// launchers (called from the C-ABI entry points of flash.hip when hd == 64 and the operands carry the fp16 form)
// ------------------------------------------------------------------------------------------------------------
#define M16_LAUNCH(KERNEL, GRID)                                                                             \
    if (a.bias) {                                                                                            \
        if (a.thr16) hipLaunchKernelGGL((KERNEL<true, true>), GRID, dim3(512), 0, stream, a);                \
        else hipLaunchKernelGGL((KERNEL<false, true>), GRID, dim3(512), 0, stream, a);                       \
    } else {                                                                                                 \
        if (a.thr16) hipLaunchKernelGGL((KERNEL<true, false>), GRID, dim3(512), 0, stream, a);               \
        else hipLaunchKernelGGL((KERNEL<false, false>), GRID, dim3(512), 0, stream, a);                      \
    }

void M16_N(fl16_launch_fwd)(const FlashArgs& a, dim3 grid, hipStream_t stream) { M16_LAUNCH(M16_N(flash16_fwd_kernel), grid) }
void M16_N(fl16_launch_bwd_q)(const FlashArgs& a, dim3 grid, hipStream_t stream) { M16_LAUNCH(M16_N(flash16_bwd_q_kernel), grid) }
void M16_N(fl16_launch_bwd_kv)(const FlashArgs& a, dim3 grid, hipStream_t stream) { M16_LAUNCH(M16_N(flash16_bwd_kv_kernel), grid) }
void M16_N(fl16_launch_bb_stats)(const FlashArgs& a, dim3 grid, hipStream_t stream) { M16_LAUNCH(M16_N(flash16_bb_stats_kernel), grid) }
void M16_N(fl16_launch_bb_q)(const FlashArgs& a, dim3 grid, hipStream_t stream) { M16_LAUNCH(M16_N(flash16_bb_q_kernel), grid) }
void M16_N(fl16_launch_bb_kv)(const FlashArgs& a, dim3 grid, hipStream_t stream) { M16_LAUNCH(M16_N(flash16_bb_kv_kernel), grid) }
