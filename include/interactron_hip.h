/* C-ABI of libinteractron_hip.so -- the gfx950 (MI355X) kernel library behind the Interactron
 * adaptive-detection hot path.
 *
 * The reference (allenai/interactron) has no native layer: every FLOP goes through ATen ops dispatched from
 * Python.  This header is therefore the boundary SURVEY.md 8(b) defines: each entry point names the reference
 * call site (path:line under the reference checkout) whose ATen/scipy work it replaces.
 *
 * Conventions
 *   - plain C types only; every pointer except the explicitly marked host arrays is a DEVICE pointer to f32
 *     (or int64 / uint8 where stated) owned by the caller; the library never allocates or frees device memory;
 *   - all work is enqueued on `stream` (pass torch.cuda.current_stream().cuda_stream); no internal syncs;
 *   - return 0 on success, negative on error; ix_last_error() returns a thread-local message;
 *   - compute entry points are re-entrant; the only global mutable state is the ix_gemm_stats / ix_flash_stats / prof
 *     counters and the test hook ix_gemm_set_mode (kernel choice for cross-checks; never called by the product).
 */
#ifndef INTERACTRON_HIP_H
#define INTERACTRON_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* ix_stream_t; /* == hipStream_t */

int ix_version(void);
const char* ix_last_error(void);

/* ---- contractions ------------------------------------------------------------------------------------------
 * C[b] (MxN, row-major, ldc) = alpha * A[b] (MxK) * B[b] (KxN) (+ bias[n]); b = bo*batch_inner + bi.
 * a_kcontig: A(m,k)=A[m*lda+k] else A[k*lda+m]; b_kcontig: B(k,n)=B[n*ldb+k] else B[k*ldb+n].
 * bias_stride_outer: element stride of bias per OUTER batch index (0 = one bias shared by all batches; N = a bias per
 * episode for episode-batched fast weights).  tile_hint in {0,64,128,1128 (= 128 on the bf16x6 kernel)},
 * split_k_hint 0 = auto.  f32-in/f32-acc MFMA (v_mfma_f32_32x32x2_f32).
 * Replaces: nn.Linear / F.linear in models/detr_models/transformer.py:148-232, models/gpt.py:39-78,
 * models/transformer.py:49-60, models/detr_models/detr.py:69-72,299-311; torch.bmm inside nn.MultiheadAttention
 * and `q @ k.transpose` / `att @ v` in models/gpt.py:48-53; conv2d of torchvision resnet50 + input_proj
 * (models/detr_models/backbone.py:88-90, detr.py:40,68) after ix_im2col_f32; and all their autograd derivatives. */
int ix_gemm_f32(const float* A, const float* B, float* C, const float* bias, int M, int N, int K, int a_kcontig,
                int b_kcontig, int64_t lda, int64_t ldb, int64_t ldc, int batch_outer, int batch_inner, int64_t sAo,
                int64_t sAi, int64_t sBo, int64_t sBi, int64_t sCo, int64_t sCi, int64_t bias_stride_outer, float alpha,
                int tile_hint, int split_k_hint, ix_stream_t stream);

/* ix_gemm_f32_ws: the same contraction with caller-provided scratch memory (the section-8b workspace convention).
 * `workspace`: device memory, 16-byte aligned, at least ix_workspace_bytes_gemm_f32(...) bytes (0 for calls that need none).
 * What it is used for:
 *   - split-K (skinny outputs with a long contracted extent; the plan is a pure function of the shapes): every split writes
 *     its partial sums into its own plane of the workspace with plain stores and one reduction launch adds the planes IN
 *     ORDER into C -- two runs give the same bits.  ix_gemm_f32 (no workspace) keeps the older scheme, fp32 atomics onto a
 *     zero-filled C, whose rounding depends on arrival order; the Python package always passes a workspace.
 *   - after ix_gemm_presplit_enable(1) (opt-in, off by default): eligible contractions run on the pre-split fp16x3 kernel
 *     (csrc/gemm_x3.hip): each operand is converted ONCE into two fp16 planes with a power-of-two scale per block of 32 rows,
 *     three v_mfma_f32_32x32x16_f16 terms per 16 contracted elements.
 * The workspace may be reused by the next call on the same stream. */
int ix_gemm_f32_ws(const float* A, const float* B, float* C, const float* bias, int M, int N, int K, int a_kcontig,
                   int b_kcontig, int64_t lda, int64_t ldb, int64_t ldc, int batch_outer, int batch_inner, int64_t sAo,
                   int64_t sAi, int64_t sBo, int64_t sBi, int64_t sCo, int64_t sCi, int64_t bias_stride_outer, float alpha,
                   int tile_hint, int split_k_hint, void* workspace, size_t workspace_bytes, ix_stream_t stream);
int ix_workspace_bytes_gemm_f32(int M, int N, int K, int a_kcontig, int b_kcontig, int64_t lda, int64_t ldb, int batch_outer,
                                int batch_inner, int64_t sAo, int64_t sBo, const float* A, const float* B, int tile_hint,
                                int split_k_hint, size_t* out_host);
int ix_gemm_presplit_enable(int on);
/* Single-pass 16-bit contraction mode (MODEL.COMPUTE_DTYPE: bf16 / fp16 -- BASELINE.json configs[1], reference arithmetic
 * models/gpt.py:39-57, models/detr_multiframe.py:55-109 under autocast-like 16-bit matmuls): every contraction the fp16x3 form
 * takes runs on its h plane alone (one fp16 value of x * 2^-E per element, block exponent per 32 x 32 sub-block, ONE MFMA per
 * k-slice, fp32 accumulation).  NOT fp32-grade; process-global; never the parity path.  Returns the previous setting. */
int ix_gemm_set_single_pass(int on);
/* 256 x 128 x 32 tiles of the fp16x3 form (gemm_f32_f16x3_w256_kernel, csrc/gemm.hip: eight waves, 128 x 64 outputs per consumer
 * wave) for plain contractions -- same reference operators as ix_gemm_f32_ws, bit-identical results to its 128 x 128 tiles.
 * mode 0: never, 1 (default): where the cost model prefers it, 2: every eligible contraction.  Returns the previous mode.
 * ix_gemm_w256_launches: launches taken so far (tests / bench). */
int ix_gemm_set_w256(int mode);
/* Measurement aid (bench.py roofline): the dense fp16 MFMA rate this GPU sustains under an all-CU matrix load (TFLOP/s; the
 * data-sheet 2 500 assumes 2.4 GHz).  scratch4: >= 4 bytes of device memory.  Synchronises `stream`. */
int ix_diag_mfma_rate_f16(double* tflops, void* scratch4, ix_stream_t stream);
int ix_gemm_w256_launches(int64_t* out);
int ix_prof_x3(double* ms, double* flops, int64_t* calls);
int ix_prof_contractions(double* ms3, double* flops3, double* mfma_flops3, int64_t* launches3); /* by form: fp32 / bf16x6 / fp16x3 */ /* profiled ix_gemm_f32_ws calls on the fp16x3 path */
/* ALGORITHMIC HBM bytes of the profiled contraction launches by the same three forms: 4 (M K + K N + M N) per batch slice -- each
 * operand read once, the result written once (an operand shared by all slices is counted per slice: an upper bound).  The
 * figure a PMC traffic measurement of the same launches is to be compared with (SURVEY 8d / bench.py roofline). */
int ix_prof_contraction_bytes(double* bytes3);

/* ---- "activation x weight planes" contraction (csrc/gemm_wp.hip): C[b] = alpha A[b] W[b]^T (+ bias) with the WEIGHT converted
 * once into the kernel's own LDS image (two fp16 planes of w 2^-E, one exponent per 32 output rows) and streamed by the
 * LDS-DMA path, the activation split in the consumers' registers -- what nn.Linear's forward and input gradient are
 * (reference models/detr_models/transformer.py:148-232, models/gpt.py:39-78).  fp32-grade like the fp16x3 form of ix_gemm_f32.
 *   ix_wp_planes_bytes : sizes of `planes` / `unscale` for a weight of N rows x K (nb batch slices)
 *   ix_wp_split_f32    : W(n, k) = k_contig ? W[n ld + k] : W[k ld + n]  ->  planes, unscale (caller-owned device buffers)
 *   ix_gemm_wp_f32     : A(m, k) = A[bo sAo + bi sAi + m lda + k], K % 32 == 0, 16-byte aligned rows; b_shared: one weight for
 *                        every slice, else slice bo of the planes; C row-major at ldc; bias [N] per outer slice (sBias) or null */
int ix_wp_planes_bytes(int N, int K, int nb, size_t* planes_bytes_host, size_t* unscale_bytes_host);
int ix_wp_split_f32(const float* W, int64_t ld, int64_t batch_stride, int N, int K, int k_contig, int nb, void* planes,
                    float* unscale, ix_stream_t stream);
int ix_prof_wp(double* ms, double* flops, int64_t* launches); /* profiled launches of the weight-planes kernel (also in slot [2] of ix_prof_contractions) */
int ix_gemm_wp_debug(int flags);
int ix_gemm_wp_debug_stamps(int64_t* device_buffer); /* 8 x int64 per workgroup, filled when flag 16 is set (tools/wp_timeline.py) */ /* diagnostic switches of tools/wp_bench.py (0 = off; the product never sets them) */
int ix_gemm_wp_f32(const float* A, int64_t lda, int64_t sAo, int64_t sAi, const void* planes, const float* unscale, int b_shared,
                   float* C, int64_t ldc, int64_t sCo, int64_t sCi, const float* bias, int64_t sBias, int M, int N, int K,
                   int batch_outer, int batch_inner, float alpha, ix_stream_t stream);

/* ix_gemm_rowsum_f32: C = alpha A B and, from the same launch, rowsum[bo * rowsum_stride + m] = sum_k A(m, k).  With A
 * stored m-contiguous (a_kcontig = 0) the bf16x6 kernel's A-producer waves accumulate the sums from the tiles they stream
 * anyway: the bias gradient colsum(dy) of a Linear rides on its weight-gradient contraction dW = dy^T x (reference:
 * F.linear under autograd -- models/gpt.py:27-32,79-84, detr_models/transformer.py FFN / projections).  Shapes the bf16x6
 * kernel does not take fall back to ix_gemm_f32 + ix_colsum_f32 inside the call (A must then be contiguous). */
int ix_gemm_rowsum_f32(const float* A, const float* B, float* C, int M, int N, int K, int a_kcontig, int b_kcontig, int64_t lda,
                       int64_t ldb, int64_t ldc, int batch_outer, int64_t sAo, int64_t sBo, int64_t sCo, float alpha,
                       float* rowsum, int64_t rowsum_stride, void* workspace, size_t workspace_bytes, ix_stream_t stream);
/* (workspace: as ix_gemm_f32_ws, sized by ix_workspace_bytes_gemm_f32 of the same contraction; split-K partial row sums go
 *  through it as well) */

/* Test hook like ix_gemm_set_mode: 1 = 128-wide tiles of eligible contractions run the fp16x3 form of the 12-wave kernel
 * (two fp16 planes + one exponent per 32 x 32 sub-block found by the producer waves, three v_mfma_f32_32x32x16_f16 per
 * k-slice; opt-in, also IX_GEMM_KERNEL=x3 in the environment), 0 (default) = the bf16x6 form everywhere.  Returns the
 * previous setting. */
int ix_gemm_set_x3(int on);

/* Launch statistics of ix_gemm_f32 (HOST pointers; process-global, single host thread): executed FLOPs
 * (2*M*N*K*batch) and launch count since the last reset; with ix_gemm_prof_enable(1) every launch is bracketed by a
 * hipEvent pair on its stream and ix_gemm_prof_read returns the summed kernel time (it waits for the events). */
int ix_gemm_set_mode(int mode); /* TEST HOOK (process-global, not part of the re-entrant compute surface; the product never
                                   calls it): 0 = every contraction on the exact-fp32 MFMA kernel, 3 (default) = eligible
                                   contractions on the persistent 12-wave bf16x6 kernel.  Returns the previous mode. */
int ix_gemm_stats(double* flops, int64_t* launches, int reset);
int ix_gemm_prof_enable(int on);
int ix_gemm_prof_kinds(double* ms2, double* flops2, int64_t* launches2); /* [0] fp32-MFMA kernel, [1] bf16x6 kernel */
int ix_prof_kinds3(double* ms3, double* flops3, int64_t* launches3); /* [2] = flash attention kernels (ix_flash_*) */
int ix_flash_stats(double* flops, int64_t* launches, int reset);     /* algorithmic FLOPs / launches of ix_flash_* */
int ix_prof_flash(double* ms7, double* flops7, double* mfma_flops7, int64_t* launches7); /* profiled flash launches by
                                   kernel: [0] all, [1] forward, [2] backward-q, [3] backward-kv, [4] statistics,
                                   [5] second-order q, [6] second-order kv; mfma_flops = FLOPs of the issued matrix
                                   instructions (3 fp16 / 6 bf16 terms per product) */
int ix_gemm_prof_read(double* total_ms, int64_t* pairs);
int ix_gemm_prof_dump(const char* path_host); /* per-launch CSV (shape, tile, split, ms); call before ix_gemm_prof_read */

/* ---- convolution gather / scatter (NHWC) -- torchvision resnet50 convs, backbone.py:88-90 ------------------- */
/* Implicit-GEMM convolution on the bf16x6 contraction kernel -- no patch matrix in HBM: the kernel's producer waves gather
 * the NHWC taps themselves.  Bias-free convolution with weights stored [out][kh][kw][in] (nn.Conv2dNHWC's layout):
 *   kind 0: src = x [groups*imgs, H, W, Cin],   other = w [groups, Cout, KH, KW, Cin] -> out = y  [groups*imgs, OH, OW, Cout]
 *   kind 1: src = dy [groups*imgs, OH, OW, Cout], other = w                           -> out = dx [groups*imgs, H, W, Cin]
 *   kind 2: src = dy,                            other = x                            -> out = dw [groups, Cout, KH, KW, Cin]
 * groups = episodes with their own fast weights (shared weights: groups = 1, imgs = every image).  The three kinds are each
 * other's derivatives (reference: torch.nn.functional.conv2d under autograd, models/detr_models/backbone.py:88-90 layer2-4
 * 3x3 convolutions, incl. the dilated stage).  ix_conv_gemm_supported: Cin, Cout multiples of 64, stride 1 / 2 / 4, each
 * group's tensors below 2 GiB; otherwise callers use ix_im2col_f32 + ix_gemm_f32 (+ ix_col2im_f32). */
int ix_conv_gemm_supported(int groups, int imgs, int H, int W, int Cin, int OH, int OW, int Cout, int KH, int KW, int stride,
                           int pad, int dil);
int ix_conv_gemm_f32(int kind, const float* src, const float* other, float* out, int groups, int imgs, int H, int W, int Cin,
                     int OH, int OW, int Cout, int KH, int KW, int stride, int pad, int dil, void* workspace,
                     size_t workspace_bytes, ix_stream_t stream);
/* scratch of one convolution kind: split-K planes (deterministic ordered reduction, as ix_gemm_f32_ws; workspace NULL =
 * atomics) and, for the stride-2 data gradient, the regrouped weights + per-parity-class outputs of its stride-1 form
 * (csrc/gemm.hip conv_bwd_data_s2: dx of a stride-2 convolution as four stride-1 convolutions of dy, 9 taps instead of 36 for
 * a 3x3; without a workspace, below the size where it pays, or after ix_conv_set_s2_split(0) / IX_CONV_S2_SPLIT=0 the one-launch
 * gather runs) */
int ix_workspace_bytes_conv_gemm_f32(int kind, int groups, int imgs, int H, int W, int Cin, int OH, int OW, int Cout, int KH,
                                     int KW, int stride, int pad, int dil, size_t* out_host);
int ix_conv_set_s2_split(int mode); /* 0 never, 1 where it pays (default: >= 40 GFLOP executed by the one-launch form, 4 for a 1x1), 2 always;
                                         | 4: class outputs through scratch + the interleaving pass even where a class can write dx in place */

int ix_im2col_f32(const float* x, float* cols, int n, int H, int W, int C, int64_t sxn, int64_t sxh, int64_t sxw,
                  int64_t sxc, int KH, int KW, int stride, int pad, int dil, int Kp, ix_stream_t stream);
int ix_col2im_f32(const float* cols, float* dx, int n, int H, int W, int C, int KH, int KW, int stride, int pad,
                  int dil, int Kp, ix_stream_t stream);
int ix_maxpool_nhwc_f32(const float* x, float* y, int n, int H, int W, int C, int k, int stride, int pad,
                        ix_stream_t stream);
int ix_nhwc_to_nchw_f32(const float* x, float* y, int n, int64_t HW, int C, ix_stream_t stream);

/* ---- FrozenBatchNorm2d (backbone.py:44-54) ---------------------------------------------------------------- */
int ix_bn_fold_f32(const float* w, const float* b, const float* rm, const float* rv, float* scale, float* shift,
                   int C, float eps, ix_stream_t stream);
/* Contraction + frozen-BN affine (+ residual) (+ ReLU) as one launch where the kernel can take it:
 *   C[b][m][n] = [relu]((A B)[b][m][n] * scale[n] + shift[n] (+ residual[b][m][n])),  C dense [batch_outer][M][N].
 * Reference: FrozenBatchNorm2d behind every backbone convolution (models/detr_models/backbone.py:19-54) and torchvision's
 * Bottleneck tail `out += identity; out = relu(out)`.  A split-K launch applies it in its ordered reduction (one launch
 * fewer); after an unsplit launch the library runs ix_channel_affine_f32 on C -- the same result.
 * ix_gemm_bn_act_f32: operands as ix_gemm_f32_ws (one batch level).  ix_conv_gemm_bn_act_f32: kind 0 of ix_conv_gemm_f32.
 * Workspace sizes: ix_workspace_bytes_gemm_f32 / ix_workspace_bytes_conv_gemm_f32 of the same problem. */
int ix_gemm_epilogue_stats(int64_t* in_reduction, int64_t* separate, int reset); /* where the affine of the fused calls ran */
/* ... and the third place: in the store of the contraction kernel itself (unsplit forward contractions on the fp16x3 form, a
 * separate kernel instance); ix_gemm_set_epilogue_in_store(0) / IX_GEMM_EPI_IN_STORE=0 sends those back to the separate launch */
int ix_gemm_epilogue_in_store(int64_t* count, int reset);
int ix_gemm_set_epilogue_in_store(int on);
int ix_gemm_bn_act_f32(const float* A, const float* B, float* C, int M, int N, int K, int a_kcontig, int b_kcontig, int64_t lda,
                       int64_t ldb, int batch_outer, int64_t sAo, int64_t sBo, const float* scale, const float* shift,
                       const float* residual, int relu, void* workspace, size_t workspace_bytes, ix_stream_t stream);
int ix_conv_gemm_bn_act_f32(const float* x, const float* w, float* y, int groups, int imgs, int H, int W, int Cin, int OH, int OW,
                            int Cout, int KH, int KW, int stride, int pad, int dil, const float* scale, const float* shift,
                            const float* residual, int relu, void* workspace, size_t workspace_bytes, ix_stream_t stream);
/* out = (a + b) * [y > 0]: the ReLU derivative of an activation with two consumers, their gradients summed in the same pass
 * (torchvision Bottleneck: `out = relu(out + identity)` feeding the next block's conv1 AND its identity branch) */
int ix_relu_bwd_sum_f32(const float* a, const float* b, const float* y, float* out, int64_t n, ix_stream_t stream);
/* out[o][n][r] = x[o][n][r] * scale[n] (R % 4 == 0): the frozen-BN scale applied to a weight tensor [(E,) N, R] or to its gradient --
 * what the backward of convolution + FrozenBatchNorm2d needs when the scale sits on the weight side (dx = g (W o scale),
 * dW = scale o (g^T x)) instead of on the activation-sized gradient (models/detr_models/backbone.py:44-54 under autograd) */
int ix_row_scale_f32(const float* x, const float* scale, float* out, int64_t outer, int N, int64_t R, ix_stream_t stream);
int ix_channel_affine_f32(const float* x, const float* scale, const float* shift, const float* residual, float* out,
                          int64_t n, int C, int relu, ix_stream_t stream);

/* ---- elementwise (transformer.py:148-232 residuals/ReLU/dropout, gpt.py:66-78 GELU, detr.py:72 sigmoid) ---- */
int ix_axpby_f32(const float* a, const float* b, float* out, int64_t n, float alpha, float beta, ix_stream_t stream);
/* out = ((srcs[0] + srcs[1]) + ...) over 2 <= n <= 8 device tensors of `count` floats; `srcs` is a HOST array of device pointers.
 * The gradient of a tensor with several consumers in one pass (what torch.autograd does with n - 1 aten::add launches). */
int ix_sum_n_f32(const float* const* srcs, int n, float* out, int64_t count, ix_stream_t stream);
int ix_mul_f32(const float* a, const float* b, float* out, int64_t n, ix_stream_t stream);
int ix_scale_f32(const float* x, float* out, int64_t n, float alpha, ix_stream_t stream);
int ix_scale_dev_f32(const float* x, const float* s, float* out, int64_t n, ix_stream_t stream);
/* backward of relu(FrozenBN(x)) w.r.t. x in one pass: out = [y > 0] * g * scale[c] (models/detr_models/backbone.py:44-54) */
int ix_relu_bwd_channel_scale_f32(const float* g, const float* y, const float* scale, float* out, int64_t n, int C,
                                  ix_stream_t stream);
int ix_relu_f32(const float* x, float* out, int64_t n, ix_stream_t stream);
int ix_relu_bwd_f32(const float* dy, const float* y, float* dx, int64_t n, ix_stream_t stream);
/* dropout(relu(x)) in one pass (FFN of models/detr_models/transformer.py:158,229) and its backward
   dx = dy * [y > 0] * scale, scale = 1 / (1 - p) */
int ix_relu_dropout_f32(const float* x, float* out, int64_t n, float p, uint64_t seed, ix_stream_t stream);
/* out = x + dropout(a): residual adds of the transformer blocks (transformer.py:157-160,222-231; gpt.py:75-77) */
int ix_add_dropout_f32(const float* x, const float* a, float* out, int64_t n, float p, uint64_t seed, ix_stream_t stream);
int ix_relu_bwd_scaled_f32(const float* dy, const float* y, float* dx, int64_t n, float scale, ix_stream_t stream);
int ix_gelu_f32(const float* x, float* out, int64_t n, ix_stream_t stream);
int ix_gelu_bwd_f32(const float* dy, const float* x, float* dx, int64_t n, ix_stream_t stream);
int ix_gelu_bwd_bwd_f32(const float* G, const float* dy, const float* x, float* grad_dy, float* grad_x, int64_t n,
                        ix_stream_t stream);
int ix_sigmoid_f32(const float* x, float* out, int64_t n, ix_stream_t stream);
int ix_sigmoid_bwd_f32(const float* dy, const float* y, float* dx, int64_t n, ix_stream_t stream);
int ix_sigmoid_bwd_bwd_f32(const float* G, const float* dy, const float* y, float* grad_dy, float* grad_y, int64_t n,
                           ix_stream_t stream);
int ix_dropout_f32(const float* x, float* out, int64_t n, float p, uint64_t seed, ix_stream_t stream);
/* Dropout masks are pure functions of (seed, element index) and the seed is a launch argument: a captured HIP graph would
 * replay the same masks for ever.  ix_set_dropout_salt(p): every dropout-carrying launch issued from now on (ix_dropout_f32,
 * ix_relu_dropout_f32, ix_add_dropout_f32, ix_attn_prob_*, ix_flash_*) XORs the 64-bit word at DEVICE address p into its seed
 * when the kernel runs; the caller changes that word between replays.  NULL (the default) switches it off.  Process-global,
 * one host thread; the word must stay allocated while such launches (or graphs holding them) can run.
 * (reference: nn.Dropout draws from torch's device RNG at run time -- models/gpt.py:33-34,52,56, transformer.py:143-146) */
int ix_set_dropout_salt(const void* device_word);
/* grouped forms: a/out [groups, rows, C], v [groups, C] (groups = episodes processed together; 1 = plain) */
int ix_add_rowvec_f32(const float* a, const float* v, float* out, int64_t rows, int C, int groups, ix_stream_t stream);
int ix_bcast_rows_f32(const float* v, float* out, int64_t rows, int C, int groups, ix_stream_t stream);
/* Reductions that span several workgroups (column sums, dot, LayerNorm parameter gradients, weighted-CE sums, the clip norm)
 * take the section-8b workspace: [IX_TICKET_BYTES = 65536 bytes of tickets, ZERO on first use and left zero by every call]
 * [partials].  Each workgroup stores one partial, the last one to arrive adds them in index order: one launch, no zero-fill,
 * and the same bits on every run (they replaced fp32 atomics, whose rounding depended on arrival order -- autograd's bias /
 * LayerNorm gradients, reference transformer.py / gpt.py under torch.autograd).  Pass the same zero-initialised buffer to
 * every call of a stream; ix_colsum_f32 alone accepts NULL (atomics onto a zero-filled output). */
int ix_colsum_f32(const float* x, float* out, int64_t rows, int C, int groups, void* workspace, size_t workspace_bytes,
                  ix_stream_t stream);
int ix_workspace_bytes_colsum_f32(int64_t rows, int C, int groups, size_t* out_host);
/* total = sum_e ||x_e||_2 over the rows of x [E, n] (E <= 1024) in one launch, norms[e] beside it; its backward
 * gx = g x_e / ||x_e|| (g: device scalar) and that kernel's own backward (cotangent H -> Gx, Gg): the learned loss of a chunk
 * of episodes, reference models/interactron.py:96 `torch.norm(fusion_out["loss"])` per task, differentiated twice by MAML. */
int ix_rownorm_sum_f32(const float* x, float* norms, float* total, int E, int n, ix_stream_t stream);
int ix_rownorm_sum_bwd_f32(const float* x, const float* norms, const float* g, float* gx, int E, int n, ix_stream_t stream);
int ix_rownorm_sum_bwd_bwd_f32(const float* x, const float* norms, const float* g, const float* H, float* Gx, float* Gg, int E,
                               int n, ix_stream_t stream);
int ix_dot_f32(const float* a, const float* b, float* out, int64_t n, void* workspace, size_t workspace_bytes,
               ix_stream_t stream); /* workspace: IX_TICKET_BYTES + 1024 bytes */

/* ---- wavefront-reduction kernels: softmax (transformer.py MHA, gpt.py:50) and LayerNorm ---------------------
 * LayerNorm: x [groups, rows, D]; gamma/beta (and dgamma/dbeta/grad_gamma, Gg/Gb) [groups, D]; groups = 1 is the plain
 * op, groups > 1 = per-episode LayerNorm weights of the episode-batched MAML fast weights. */
int ix_softmax_fwd_f32(const float* x, float* y, int64_t rows, int len, int64_t ld, const uint8_t* mask,
                       int rows_per_mask, int64_t mask_ld, ix_stream_t stream);
int ix_softmax_bwd_f32(const float* y, const float* dy, float* dx, int64_t rows, int len, int64_t ld,
                       ix_stream_t stream);
int ix_softmax_bwd_bwd_f32(const float* G, const float* y, const float* dy, float* grad_y, float* grad_dy,
                           int64_t rows, int len, int64_t ld, ix_stream_t stream);
/* Attention probabilities as one node: y = softmax(x [+ key mask]) (y may alias x), d = dropout(y) (d may be NULL);
   its backward gs = softmax_bwd(y, dropout_bwd(gd)); and the double backward of that pair (HgD = dL/d gd, HS = dL/dx
   given G1 + G2 = dL/d gs and HD = dL/d d).  Replaces the softmax -> dropout node pairs of
   models/gpt.py:48-52 and nn.MultiheadAttention (models/detr_models/transformer.py:153,216,219) on the [L, S] tensors. */
int ix_attn_prob_fwd_f32(const float* x, float* y, float* d, int64_t rows, int len, int64_t ld, const uint8_t* mask,
                         int rows_per_mask, int64_t mask_ld, float p, uint64_t seed, ix_stream_t stream);
int ix_attn_prob_bwd_f32(const float* y, const float* gd, float* gs, int64_t rows, int len, int64_t ld, float p,
                         uint64_t seed, ix_stream_t stream);
int ix_attn_prob_bwd_bwd_f32(const float* G1, const float* G2, const float* y, const float* gd, const float* HD,
                             float* HgD, float* HS, int64_t rows, int len, int64_t ld, float p, uint64_t seed,
                             ix_stream_t stream);
int ix_layernorm_fwd_f32(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                         int64_t rows, int D, float eps, int groups, ix_stream_t stream);
int ix_layernorm_bwd_f32(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                         float* dx, float* dgamma, float* dbeta, int64_t rows, int D, int groups, void* workspace,
                         size_t workspace_bytes, ix_stream_t stream);
int ix_layernorm_bwd_bwd_f32(const float* Gx, const float* Gg, const float* Gb, const float* dy, const float* x,
                             const float* gamma, const float* mean, const float* rstd, float* grad_dy, float* grad_x,
                             float* grad_gamma, int64_t rows, int D, int groups, void* workspace, size_t workspace_bytes,
                             ix_stream_t stream);
/* scratch of both (ticket layout, see ix_colsum_f32); rows = rows per group */
int ix_workspace_bytes_layernorm_bwd(int64_t rows, int D, int groups, size_t* out_host);

/* ---- flash-style attention (no [L, S] tensor in HBM) ------------------------------------------------------------
 * O = dropout(softmax(scale Q K^T + key bias)) V per (batch, head), and its first and second derivative, in fp32-grade
 * arithmetic on the 16-bit matrix cores (exact operand splits: two fp16 planes with a per-32-row power-of-two scale for
 * the products that contract over the head dim, three bf16 planes for the products that contract over tokens).
 * Replaces the same reference lines as the ix_attn_prob_* family: `att = softmax(q k^T / sqrt(hd)); att = drop(att);
 * y = att v` of models/gpt.py:39-57 and nn.MultiheadAttention's core in models/detr_models/transformer.py:148-161,
 * 211-232, plus autograd's first / second derivative of them (models/interactron.py:99-123).
 *
 * ix_attn_split_f32: fp32 activations x [n][R][ld] (head h = columns off + h*hd .. + hd) ->
 *   row_planes  [2][n*H][Rp][hd] fp16 (h, l of x * 2^e) + row_unscale [n*H][Rp/32] f32 (2^-e per block of 32 rows), and/or
 *   tr_planes   tr_form 0: [3][n*H][hd][Rp] bf16 (h, m, l of x; rows permuted within 16-groups to the MFMA accumulator
 *               order) -- the products that contract over tokens then take six matrix instructions per k-slice;
 *               tr_form 1: [2][n*H][hd][Rp] fp16, the row planes' scaled (h, l) transposed and permuted likewise (reads
 *               row_unscale too) -- three matrix instructions per k-slice, the [L, S] intermediates are brought into fp16
 *               range in registers by a running power-of-two factor per output row (csrc/flash.hip).  The caller may
 *               always allocate 3 planes.
 *   Rp = R rounded up to 128, padded rows are zero.  row_planes or tr_planes may be null; row_unscale is needed with
 *   row_planes and with tr_planes of form 1.
 * struct ix_attn_planes: the pointers of one operand and the tr form they were split with, as the entry points below take
 *   them (members an entry point does not read may be null: forward reads q.row, k.row, v.tr); all tr operands of one
 *   call must share one form.
 * ix_attn_bias_f32: key_padding_mask uint8 [n][mask_ld] (nonzero = ignore; null = none) -> additive bias [n][Sb],
 *   Sb >= S (the kernels want S rounded up to 128): 0 for valid keys, -inf for masked keys and the tail.
 *   bias == NULL in ix_flash_fwd / bwd / bwd_bwd_f32 means "no key is masked" and is taken by the head-dim-64 fp16-form
 *   passes (csrc/flash16.hip) only: they skip the bias loads and adds and blank the keys >= S of the last tile themselves;
 *   every other kernel family needs the tensor (IX_ERR_ARG otherwise).
 * ix_flash_fwd_f32: out [n][L][ld_out] (head h at off_out + h*hd), lse [n*H][Lp] (natural-log softmax normalisers; rows
 *   L..Lp are written as +inf so that padded query rows count as P = 0 in the derivative kernels).
 *   p_drop / seed: dropout on the probabilities, mask = pure function of (seed, batch*head, query, key).
 *   out == null: only lse is produced (q k^T + softmax statistics) -- used after the fp8 forward, whose own normalisers
 *   belong to fp8 scores and not to the fp16 scores the derivative kernels recompute.
 * ix_attn_rowdot_f32: t[bh][q] = sum_d a[q, h, d] b[q, h, d] over heads of two [n][L][ld] activations (delta = dO . O).
 * ix_flash_bwd_f32: (gq, gk, gv) from the planes of (q, k, v, dO) + lse + delta; probabilities are recomputed tile by
 *   tile.  gq (may be null) is written by query-owning workgroups, gk + gv (both or neither) by key-owning ones; outputs
 *   are [n][L|S][ld_*] with head h at off_* + h*hd (so gq and gk may share one packed buffer).
 * ix_flash_bwd_bwd_f32: double backward -- gradients (dq, dk, dv, ddo) of sum(hq.gq + hk.gk + hv.gv) w.r.t. (q, k, v, dO)
 *   for the cotangents (hq, hk, hv) of ix_flash_bwd_f32's outputs (all seven operands need row + tr planes).
 *   `workspace` (device, caller-owned): ix_workspace_bytes_flash_bwd_bwd(n, H, L) bytes for the per-row statistics.
 * ix_flash_dropmask_f32: the kernels' dropout mask as a tensor m[BH][L][S] (1/keep or 0) -- tests only. */
struct ix_attn_planes {
    const void* row;
    const float* unscale;
    const void* tr;
    int tr_form;
};
int ix_attn_split_f32(const float* x, void* row_planes, float* row_unscale, void* tr_planes, int tr_form, int n, int R, int Rp,
                      int64_t ld, int off, int H, int hd, ix_stream_t stream);
/* ix_attn_split_f32 of x with t[bh][row] = sum_d x[row, h, d] y[row, h, d] riding along (t [n*H][Rp], rows R..Rp written as 0): the
 * backward pass's planes of dO and delta = dO . O from ONE read of dO (instead of ix_attn_split_f32 + ix_attn_rowdot_f32) */
int ix_attn_split_dot_f32(const float* x, void* row_planes, float* row_unscale, void* tr_planes, int tr_form, int n, int R, int Rp,
                          int64_t ld, int off, int H, int hd, const float* y, int64_t ldy, int offy, float* t, ix_stream_t stream);
/* up to three operands of one attention call (q, k, v / hq, hk, hv) in ONE launch; arrays of `count` entries, null entries
 * of row_planes / tr_planes as in ix_attn_split_f32 */
int ix_attn_split_multi_f32(int count, const float* const* x, void* const* row_planes, float* const* row_unscale,
                            void* const* tr_planes, int tr_form, int n, const int* R, const int* Rp, const int64_t* ld,
                            const int* off, int H, int hd, ix_stream_t stream);
int ix_attn_bias_f32(const uint8_t* mask, float* bias, int n, int S, int Sb, int64_t mask_ld, ix_stream_t stream);
int ix_flash_fwd_f32(const struct ix_attn_planes* q, const struct ix_attn_planes* k, const struct ix_attn_planes* v,
                     const float* bias, float* out, float* lse, int n, int H, int L, int Lp, int S, int Sp, int hd,
                     int64_t ld_out, int off_out, float scale, float p_drop, uint64_t seed, ix_stream_t stream);
int ix_attn_rowdot_f32(const float* a, const float* b, float* t, int n, int H, int L, int Lp, int hd, int64_t lda, int offa,
                       int64_t ldb, int offb, ix_stream_t stream);
int ix_flash_bwd_f32(const struct ix_attn_planes* q, const struct ix_attn_planes* k, const struct ix_attn_planes* v,
                     const struct ix_attn_planes* d_out, const float* bias, const float* lse, const float* delta, float* gq,
                     float* gk, float* gv, int n, int H, int L, int Lp, int S, int Sp, int hd, int64_t ld_q, int off_q,
                     int64_t ld_k, int off_k, int64_t ld_v, int off_v, float scale, float p_drop, uint64_t seed,
                     ix_stream_t stream);
int ix_flash_bwd_bwd_f32(const struct ix_attn_planes* q, const struct ix_attn_planes* k, const struct ix_attn_planes* v,
                         const struct ix_attn_planes* d_out, const struct ix_attn_planes* hq, const struct ix_attn_planes* hk,
                         const struct ix_attn_planes* hv, const float* bias, const float* lse, const float* delta, float* dq,
                         float* dk, float* dv, float* ddo, int n, int H, int L, int Lp, int S, int Sp, int hd, int64_t ld_q,
                         int off_q, int64_t ld_k, int off_k, int64_t ld_v, int off_v, int64_t ld_do, int off_do, float scale,
                         float p_drop, uint64_t seed, void* workspace, size_t workspace_bytes, ix_stream_t stream);
int ix_workspace_bytes_flash_bwd_bwd(int n, int H, int L, size_t* out_host);
int ix_flash_dropmask_f32(float* m, int BH, int L, int S, float p_drop, uint64_t seed, ix_stream_t stream);
/* Head dim 64 in tr form 1 runs the 16x16x32 passes of csrc/flash16.hip: eight waves per workgroup, 16 owner rows per wave, the
 * streamed operand staged once in row layout and read both ways (ds_read_b128 / ds_read_b64_tr_b16) -- such calls need ROW
 * planes only (v included, in the forward pass too) and never read tr planes.  ix_flash_set_m16(0 / 1) switches the family
 * off / on for the process (any other value: query), returns the previous setting; IX_FLASH_M16=0 in the environment starts
 * with it off. */
int ix_flash_set_m16(int on);

/* fp8 forward (opt-in, BASELINE.json configs[4] "fp8 MFMA attention"): q k^T and P v of the forward pass on
 * v_mfma_f32_32x32x16_fp8_fp8 (OCP e4m3 operands, fp32 accumulate, fp32 softmax / output / lse).
 * ix_attn_split_fp8_f32: x -> one e4m3 plane in row layout [n*H][Rp][hd] and / or tr layout [n*H][hd][Rp] (either may be
 *   null) + unscale [n*H][Rp/32] (block maximum scaled into [128, 256)).
 * ix_flash_fwd_fp8_f32: as ix_flash_fwd_f32 on those planes (q row, k row, v tr).  The derivative entry points are not
 *   affected: they recompute the probabilities from the fp16 planes. */
int ix_attn_split_fp8_f32(const float* x, void* row_plane, void* tr_plane, float* unscale, int n, int R, int Rp, int64_t ld,
                          int off, int H, int hd, ix_stream_t stream);
int ix_flash_fwd_fp8_f32(const void* q_row8, const float* q_unscale, const void* k_row8, const float* k_unscale,
                         const void* v_tr8, const float* v_unscale, const float* bias, float* out, float* lse, int n, int H,
                         int L, int Lp, int S, int Sp, int hd, int64_t ld_out, int off_out, float scale, float p_drop,
                         uint64_t seed, ix_stream_t stream);

/* ---- set criterion ---------------------------------------------------------------------------------------
 * ix_match_cost_f32: HungarianMatcher cost matrix (matcher.py:54-73); ix_lsap_f32 (HOST pointers): the
 * scipy.optimize.linear_sum_assignment call of matcher.py:76; weighted CE: detr.py:111-132; box losses:
 * detr.py:148-167 with box_ops.py:8-58; sine position embedding: position_encoding.py:28-48; mask resize:
 * backbone.py:77. */
int ix_match_cost_f32(const float* logits, const float* boxes, const int64_t* tgt_ids, const float* tgt_boxes,
                      float* cost, int rows, int C, int T, float w_class, float w_bbox, float w_giou,
                      ix_stream_t stream);
int ix_lsap_f32(const float* cost_host, int64_t nr, int64_t nc, int64_t* row_idx_host, int64_t* col_idx_host);

/* Device-resident matcher + set criterion of a whole chunk of images (no host round trip, no per-image launches).
 * Targets of all images as one CSR list: tgt_ids int64 [T], tgt_boxes [T, 4] cxcywh, off int32 [I + 1] (image i owns targets
 * off[i] .. off[i + 1]).
 *   ix_match_cost_csr_f32   cost [I, Q, ldn] = the matrices of matcher.py:54-73 for every image, column t = the image's t-th
 *                           target (same arithmetic as ix_match_cost_f32)
 *   ix_lsap_device_f32      scipy.optimize.linear_sum_assignment per image (matcher.py:76) on the GPU, one wavefront per image,
 *                           the algorithm, double arithmetic and tie-breaking of ix_lsap_f32: identical assignments.
 *                           tgt_of_q int32 [I, Q]: image-local target matched to query q or -1; q_of_tgt int32 [T]: query
 *                           matched to target t or -1.  Sides (Q, targets per image, ldn) up to 256.
 *   ix_set_loss_rows_f32    per prediction row: rowstat [I*Q, 4] = (w nll, w, L1, 1 - GIoU), lse [I*Q], flags int32 [I*Q]
 *                           (detr.py:111-167; w = 1, w_noobj for the no-object class = `background_c` of interactron.py:103)
 *   ix_set_loss_groups_f32  group g = images g*stride .. g*stride + len - 1: out[g] = (loss_ce, class_error, loss_bbox,
 *                           loss_giou, cardinality_error) with the group's own normalisers norm[g] = (sum w, max(#targets, 1))
 *                           (detr.py:220-265 evaluated once per group: the 5 frames of an episode, frame 0 alone for the policy
 *                           reward, one frame for the detector loss -- interactron.py:101-108,128-131); ordered sums
 *   ix_set_loss_bwd_f32     d(sum_g gout[g] . out[g]) / d(logits, boxes); images outside every group get zeros */
int ix_match_cost_csr_f32(const float* logits, const float* boxes, const int64_t* tgt_ids, const float* tgt_boxes, const int* off,
                          float* cost, int I, int Q, int C, int ldn, float w_class, float w_bbox, float w_giou,
                          ix_stream_t stream);
int ix_lsap_device_f32(const float* cost, const int* off, int I, int Q, int ldn, int* tgt_of_q, int* q_of_tgt,
                       ix_stream_t stream);
int ix_set_loss_rows_f32(const float* logits, const float* boxes, const int64_t* tgt_ids, const float* tgt_boxes, const int* off,
                         const int* tgt_of_q, float* rowstat, float* lse, int* flags, int I, int Q, int C, float w_noobj,
                         ix_stream_t stream);
int ix_set_loss_groups_f32(const float* rowstat, const int* flags, const int* off, int stride, int len, int G, int Q, float* out,
                           float* norm, ix_stream_t stream);
int ix_set_loss_bwd_f32(const float* logits, const float* boxes, const int64_t* tgt_ids, const float* tgt_boxes, const int* off,
                        const int* tgt_of_q, const float* lse, const float* gout, const float* norm, int stride, int len, int I,
                        int Q, int C, float w_noobj, float* dlogits, float* dboxes, ix_stream_t stream);
int ix_weighted_ce_fwd_f32(const float* logits, const int64_t* target, const float* weight, float* lse,
                           int64_t* argmax, float* sums, int rows, int C, void* workspace, size_t workspace_bytes,
                           ix_stream_t stream);
int ix_workspace_bytes_weighted_ce(int rows, size_t* out_host); /* ticket layout, see ix_colsum_f32 */
int ix_weighted_ce_bwd_f32(const float* logits, const int64_t* target, const float* weight, const float* lse,
                           const float* sums, const float* gout, float* dlogits, int rows, int C, ix_stream_t stream);
int ix_box_loss_fwd_f32(const float* pred, const int64_t* src_idx, const float* tgt, float* out, int K,
                        ix_stream_t stream);
int ix_box_loss_bwd_f32(const float* pred, const int64_t* src_idx, const float* tgt, const float* gout, float* dpred,
                        int R, int K, ix_stream_t stream);
int ix_sine_pos_f32(const uint8_t* mask, float* pos, int n, int h, int w, int num_pos_feats, float temperature,
                    float scale, ix_stream_t stream);
int ix_mask_nearest_u8(const uint8_t* in, uint8_t* out, int n, int H, int W, int h, int w, ix_stream_t stream);

/* ---- The 16-bit activation mode (MODEL.COMPUTE_DTYPE: bf16 -- BASELINE.json configs[1] "multi_frame_baseline ... bf16"; the reference
 * computes all of it in fp32: models/gpt.py:39-78, models/detr_models/transformer.py:148-232, backbone.py:88-90).  Activations live in
 * HBM as bf16 (device pointers to 2-byte elements, passed as void*); statistics, accumulation and parameters stay fp32.
 *   ix_gemm_b16   : C[b] = act((alpha A[b] B[b] + bias[n]) scale[n] + shift[n] + residual) with bf16 A, B (csrc/gemm16.hip: both
 *                   operand tiles HBM -> LDS by LDS-DMA, one v_mfma_f32_16x16x32_bf16 per k-slice, fp32 accumulation).  Layouts and
 *                   batch strides as ix_gemm_f32 (a_kcontig: A(m, k) = A[m lda + k], else A[k lda + m]; likewise B(k, n)); C and
 *                   `residual` (same indexing as C) are bf16, or fp32 when c_f32; bias [N] per outer slice (bias_stride), scale /
 *                   shift [N] (a frozen-BN affine, backbone.py:19-54) or null; act 0 none, 1 ReLU, 2 GELU (erf form).  Contractions
 *                   with few output tiles and a long K are cut along K into fp32 planes in `workspace` (ticket layout, see
 *                   ix_colsum_f32) and added in order.  Requirements: ix_gemm_b16_supported (16-byte aligned rows).
 *   ix_cast_*     : elementwise fp32 <-> bf16 (round to nearest even), n elements, 16-byte aligned pointers.
 *   ix_prof_b16   : profiled ix_gemm_b16 launches (summed event ms, FLOPs, algorithmic HBM bytes, launches). */
int ix_gemm_b16(const void* A, const void* B, void* C, const float* bias, int M, int N, int K, int a_kcontig, int b_kcontig,
                int64_t lda, int64_t ldb, int64_t ldc, int batch_outer, int batch_inner, int64_t sAo, int64_t sAi, int64_t sBo,
                int64_t sBi, int64_t sCo, int64_t sCi, int64_t bias_stride, float alpha, int c_f32, const float* scale,
                const float* shift, const void* residual, int act, void* workspace, size_t workspace_bytes, ix_stream_t stream);
int ix_gemm_b16_supported(const void* A, const void* B, const void* C, int M, int N, int K, int a_kcontig, int b_kcontig, int64_t lda,
                          int64_t ldb, int64_t ldc, int64_t sAo, int64_t sAi, int64_t sBo, int64_t sBi, int64_t sCo, int64_t sCi);
int ix_workspace_bytes_gemm_b16(int M, int N, int K, int nbatch, size_t* out_host);
int ix_gemm_rowsum_b16(const void* A, const void* B, float* C, float* rowsum, int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc,
                       int batch_outer, int64_t sAo, int64_t sBo, int64_t sCo, int64_t rowsum_stride, float alpha, void* workspace,
                       size_t workspace_bytes, ix_stream_t stream); /* C = alpha A B (fp32) and rowsum[b][m] = sum_k A(m, k), A m-contiguous,
                       B n-contiguous: a Linear's weight gradient dY^T x with its bias gradient riding on it (the fp32 twin: ix_gemm_rowsum_f32) */
int ix_gemm_b16_set_big(int on);  /* the 256 x 256-tile form of the kernel (one eight-wave workgroup per CU, two 64 KB LDS stages): 1 (default)
                                     where the launch still fills the chip | 0 never | 2 for every plain contraction (tests) | < 0 query;
                                     returns the previous setting (IX_GEMM16_BIG) */
int ix_gemm_b16_set_stages(int stages); /* 1: one LDS stage, four workgroups per CU | 2: two stages, the next K step's DMA under this one's
                                           matrix instructions; returns the previous setting (IX_GEMM16_STAGES) */
/*   ix_conv_gemm_b16 : the three implicit-GEMM convolution kinds of ix_conv_gemm_f32 (forward, data gradient, weight gradient of a
 *                   bias-free NHWC convolution with [out][kh][kw][in] weights: backbone.py:88-90) on bf16 tensors -- no patch matrix in
 *                   HBM, the gather is the per-lane SOURCE address of the LDS-DMA.  kind 0 takes the frozen-BN affine (+ residual)
 *                   (+ ReLU) epilogue; c_f32: fp32 result (the weight gradient).  Geometry limits: ix_conv_gemm_b16_supported
 *                   (channels % 64 / % 128 so that a K step / an N tile lies inside one tap; stride 1, 2 or 4). */
int ix_conv_gemm_b16(int kind, const void* src, const void* other, void* out, int groups, int imgs, int H, int W, int Cin, int OH, int OW,
                     int Cout, int KH, int KW, int stride, int pad, int dil, int c_f32, const float* scale, const float* shift,
                     const void* residual, int relu, void* workspace, size_t workspace_bytes, ix_stream_t stream);
int ix_conv_gemm_b16_supported(int kind, int groups, int imgs, int H, int W, int Cin, int OH, int OW, int Cout, int KH, int KW, int stride,
                               int pad, int dil);
int ix_workspace_bytes_conv_gemm_b16(int kind, int groups, int imgs, int H, int W, int Cin, int OH, int OW, int Cout, int KH, int KW,
                                     size_t* out_host);
/*   ix_map_b16    : out[k] = f_op(a[k], b[k], c[k]) on bf16 tensors (csrc/ew16.hip), the arithmetic of the fp32 kernels of the same name,
 *                   rounded once: op 0 a + b | 1 p0 a + p1 b | 2 p0 a | 3 relu(a) | 4 a [b > 0] p0 | 5 (a + b) [c > 0] | 6 gelu(a) |
 *                   7 a gelu'(b) | 8 dropout(a; p0, seed) | 9 dropout(relu(a)) | 10 a + dropout(b) (the dropout hash of ix_dropout_f32:
 *                   a pure function of (seed ^ salt, element index)); unused operands null
 *   ix_sum_n_b16  : ((a0 + a1) + ...) over 2..8 bf16 tensors, left to right in fp32, one rounding
 *   ix_channel_b16: [rows, C] bf16 with per-channel fp32 vectors: op 0 [relu](x scale[c] + shift[c] (+ y)) | 1 x [y > 0] scale[c] |
 *                   2 x scale[c] | 3 x + scale[g][c] (a row vector per slab of rows / groups rows); C % 8 == 0
 *   ix_layernorm_*_b16 : nn.LayerNorm forward / first derivative on bf16 rows (D % 4 == 0, D <= 1024), fp32 statistics, gamma / beta and
 *                   parameter gradients; the backward's workspace (ix_workspace_bytes_layernorm_bwd_b16, ticket layout) holds the
 *                   per-workgroup partial parameter gradients, added in order */
int ix_map_b16(int op, const void* a, const void* b, const void* c, void* out, int64_t n, float p0, float p1, uint64_t seed,
               ix_stream_t stream);
int ix_sum_n_b16(const void* const* srcs_host_array, int n, void* out, int64_t count, ix_stream_t stream);
int ix_channel_b16(int op, const void* x, const void* y, const float* scale, const float* shift, void* out, int64_t rows, int C, int relu,
                   int groups, ix_stream_t stream);
int ix_layernorm_fwd_b16(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, int64_t rows, int D,
                         float eps, ix_stream_t stream);
int ix_workspace_bytes_layernorm_bwd_b16(int64_t rows, int D, size_t* out_host);
int ix_layernorm_bwd_b16(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd, void* dx, float* dgamma,
                         float* dbeta, int64_t rows, int D, void* workspace, size_t workspace_bytes, ix_stream_t stream);
int ix_workspace_bytes_colsum_b16(int64_t rows, int C, int groups, size_t* out_host);
int ix_colsum_b16(const void* x, float* out, int64_t rows, int C, int groups, void* workspace, size_t workspace_bytes,
                  ix_stream_t stream); /* [G][rows][C] bf16 -> [G][C] fp32 (bias gradients), ordered partial sums; C % 8 == 0 */
/*   attention in the mode: ix_attn_split_multi_b16 / ix_attn_split_dot_b16 write the kernels' operand planes from bf16 q / k / v / dO
 *                   (every bf16 value fits the h plane; dot operand y fp32), and ix_flash_set_single_term(1) sends the head-dim-64
 *                   passes of ix_flash_fwd_f32 / ix_flash_bwd_f32 / ix_flash_bwd_bwd_f32 to their single-term build (one matrix
 *                   instruction per k-slice instead of three; [L, S] intermediates rounded once to fp16); returns the old setting */
int ix_attn_split_multi_b16(int count, const void* const* x, void* const* row_planes, float* const* row_unscale, void* const* tr_planes,
                            int tr_form, int n, const int* R, const int* Rp, const int64_t* ld, const int* off, int H, int hd,
                            ix_stream_t stream);
int ix_attn_split_dot_b16(const void* x, void* row_planes, float* row_unscale, void* tr_planes, int tr_form, int n, int R, int Rp,
                          int64_t ld, int off, int H, int hd, const float* y, int64_t ldy, int offy, float* t, ix_stream_t stream);
int ix_flash_set_single_term(int on);
int ix_cast_f32_b16(const float* x, void* y, int64_t n, ix_stream_t stream);
int ix_cast_b16_f32(const void* x, float* y, int64_t n, ix_stream_t stream);
int ix_prof_b16(double* ms, double* flops, double* bytes, int64_t* launches);

/* Episode expansion of a parameter list and its adjoint, all tensors in one launch set (HOST arrays of device pointers;
 * sizes[i] = elements of one copy).  expand: out_i[E][n_i] = E copies of src_i[n_i] -- the per-episode copies of theta the
 * episode-batched step differentiates (reference models/interactron.py:86-90, clone_parameters per task);  reduce:
 * out_i[n_i] = sum over the E copies of src_i[E][n_i], in a fixed order -- the reference's gradient accumulation over tasks. */
int ix_expand_multi_f32(const float* const* src, float* const* out, const int64_t* sizes, int ntensors, int E, ix_stream_t stream);
int ix_reduce_multi_f32(const float* const* src, float* const* out, const int64_t* sizes, int ntensors, int E, ix_stream_t stream);

/* ---- MAML fast weights (utils/meta_utils.py:135-142) and outer step (engine/interactron_trainer.py:107-111) --
 * p / g / out / G are HOST arrays of `ntensors` device pointers, sizes a HOST array of element counts. */
int ix_sgd_clip_multi_f32(const float* const* p, const float* const* g, float* const* out, const int64_t* sizes,
                          int ntensors, float lr, float clip, ix_stream_t stream);
int ix_sgd_clip_bwd_multi_f32(const float* const* G, const float* const* g, float* const* out, const int64_t* sizes,
                              int ntensors, float lr, float clip, ix_stream_t stream);
int ix_sumsq_accum_f32(const float* x, int64_t n, float* out, void* workspace, size_t workspace_bytes,
                       ix_stream_t stream); /* workspace: IX_TICKET_BYTES + 4096 bytes (ticket layout, see ix_colsum_f32) */
int ix_adam_step_f32(float* p, float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                     int step, const float* sumsq, float max_norm, int zero_grad, ix_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* INTERACTRON_HIP_H */
