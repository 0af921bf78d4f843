"""The 16-bit activation mode's kernels (csrc/gemm16.hip ...) against float64 on the SAME bf16-rounded inputs: the arithmetic under test
is "bf16 operands, exact products, fp32 accumulation, one rounding of the result" -- so with inputs that already are bf16 values the
only error left is the accumulation order and the final rounding (2^-9 relative for a bf16 result, fp32-grade for an fp32 result).
``pytest -m gpu``."""
import ctypes
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from interactron_amd import _lib
    return _lib.load()


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(torch.bfloat16)


def gemm_b16(lib, A, B, M, N, K, a_kc, b_kc, lda, ldb, bo=1, bi=1, sA=(0, 0), sB=(0, 0), bias=None, alpha=1.0, c_f32=False,
             scale=None, shift=None, res=None, act=0, ws=True, a_off=0, b_off=0):
    C = torch.full((bo, bi, M, N), float("nan"), dtype=torch.float32 if c_f32 else torch.bfloat16, device="cuda")
    need = ctypes.c_size_t()
    assert lib.ix_workspace_bytes_gemm_b16(M, N, K, bo * bi, ctypes.byref(need)) == 0
    w = torch.zeros(max(need.value, 16) if ws else 16, dtype=torch.uint8, device="cuda")
    p = lambda t: None if t is None else t.data_ptr()
    rc = lib.ix_gemm_b16(A.data_ptr() + 2 * a_off, B.data_ptr() + 2 * b_off, C.data_ptr(), p(bias), M, N, K, int(a_kc), int(b_kc), lda, ldb, N,
                         bo, bi, sA[0], sA[1], sB[0], sB[1], bi * M * N, M * N, N if (bias is not None and bias.dim() == 2) else 0,
                         alpha, int(c_f32), p(scale), p(shift), p(res), act, w.data_ptr() if ws else None, w.numel() if ws else 0,
                         torch.cuda.current_stream().cuda_stream)
    assert rc == 0, lib.ix_last_error()
    torch.cuda.synchronize()
    return C, need.value


@pytest.mark.parametrize("a_kc,b_kc", [(True, True), (True, False), (False, True), (False, False)])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 72, 136), (1000, 260, 1496), (64, 2048, 256), (333 * 8, 132, 72)])
def test_bf16_gemm_layouts_and_tails_against_float64(lib, a_kc, b_kc, M, N, K):
    """All four operand layouts (what forward, input gradient and weight gradient of a Linear present), ragged M / N / K (K tails
    inside and across 16-byte chunks of the m-contiguous layouts, N = 4 (mod 8)), leading dimensions larger than the rows."""
    lda = (K if a_kc else M) + 8
    ldb = (K if b_kc else N) + 16
    ldb += (-ldb) % 8
    lda += (-lda) % 8
    A = rnd(M if a_kc else K, lda, seed=1).cuda()
    B = rnd(N if b_kc else K, ldb, seed=2).cuda()
    C, _ = gemm_b16(lib, A, B, M, N, K, a_kc, b_kc, lda, ldb, c_f32=True)
    a = A.double().cpu()[:, :K] if a_kc else A.double().cpu()[:K, :M].t()
    b = B.double().cpu()[:, :K].t() if b_kc else B.double().cpu()[:K, :N]
    ref = a @ b
    bound = (a.abs() @ b.abs())
    err = (C[0, 0].double().cpu() - ref).abs()
    assert bool(torch.isfinite(C).all())
    assert float((err / bound.clamp_min(1e-30)).max()) <= 4e-7, float((err / bound.clamp_min(1e-30)).max())   # fp32 accumulation of exact products
    Cb, _ = gemm_b16(lib, A, B, M, N, K, a_kc, b_kc, lda, ldb, c_f32=False)
    errb = (Cb[0, 0].double().cpu() - ref).abs()
    assert float((errb / (ref.abs() * 2.0 ** -8 + bound * 4e-7)).max()) <= 1.0   # one bf16 rounding of the result


def test_bf16_gemm_batches_bias_affine_residual_and_activations(lib):
    """Outer / inner batch strides (episode-batched weights: one B per outer slice), per-slice bias, the frozen-BN affine + residual +
    ReLU epilogue of the backbone, and GELU (models/gpt.py:70-75) -- against float64 of the same formula."""
    bo, bi, M, N, K = 3, 2, 150, 136, 200
    A = rnd(bo, bi, M, K, seed=3).cuda()
    B = rnd(bo, N, K, seed=4, scale=0.2).cuda()
    bias = (torch.randn(bo, N, generator=torch.Generator().manual_seed(5))).cuda()
    scale = (torch.rand(N, generator=torch.Generator().manual_seed(6)) + 0.5).cuda()
    shift = torch.randn(N, generator=torch.Generator().manual_seed(7)).cuda()
    res = rnd(bo, bi, M, N, seed=8).cuda()
    ref0 = torch.einsum("oimk,onk->oimn", A.double().cpu(), B.double().cpu()) * 0.5 + bias.double().cpu()[:, None, None, :]
    for act, fn in ((0, lambda v: v), (1, torch.relu), (2, lambda v: torch.nn.functional.gelu(v))):
        for use_res in (False, True):
            for c_f32 in (True, False):
                r = res.float().contiguous() if c_f32 else res
                C, _ = gemm_b16(lib, A, B, M, N, K, True, True, K, K, bo, bi, (bi * M * K, M * K), (N * K, 0), bias=bias, alpha=0.5,
                                c_f32=c_f32, scale=scale, shift=shift, res=r if use_res else None, act=act)
                ref = ref0 * scale.double().cpu() + shift.double().cpu()
                if use_res:
                    ref = ref + res.double().cpu()
                ref = fn(ref)
                tol = 1e-5 * float(ref.abs().max()) if c_f32 else 2.0 ** -8 * ref.abs() + 1e-5 * float(ref.abs().max())
                assert bool(((C.double().cpu() - ref).abs() <= tol).all()), (act, use_res, c_f32, float((C.double().cpu() - ref).abs().max()))


def test_bf16_gemm_split_k_is_ordered_and_equals_one_pass(lib):
    """The weight-gradient shape: a small output with a long K (dW = dY^T X, both operands m / n-contiguous).  With a workspace the
    launch is cut along K into fp32 planes that are added in order (two runs: identical bits); without one it runs as a single pass --
    the same fp32 sums in another order."""
    M, N, K = 256, 512, 28880
    A = rnd(K, M, seed=9).cuda()      # dY: [rows, out]
    B = rnd(K, N, seed=10).cuda()     # X:  [rows, in]
    C1, need = gemm_b16(lib, A, B, M, N, K, False, False, M, N, c_f32=True)
    assert need > 65536, "this shape is expected to be cut along K"
    C2, _ = gemm_b16(lib, A, B, M, N, K, False, False, M, N, c_f32=True)
    assert torch.equal(C1, C2)
    C3, _ = gemm_b16(lib, A, B, M, N, K, False, False, M, N, c_f32=True, ws=False)
    ref = A.double().cpu().t() @ B.double().cpu()
    bound = A.double().cpu().abs().t() @ B.double().cpu().abs()
    for C in (C1, C3):
        assert float(((C[0, 0].double().cpu() - ref).abs() / bound).max()) <= 4e-7


def test_bf16_gemm_refuses_unaligned_operands(lib):
    A, B = rnd(64, 100, seed=1).cuda(), rnd(64, 100, seed=2).cuda()
    assert lib.ix_gemm_b16_supported(A.data_ptr(), B.data_ptr(), A.data_ptr(), 64, 64, 100, 1, 1, 100, 100, 64, 0, 0, 0, 0, 0, 0) == 0
    assert lib.ix_gemm_b16_supported(A.data_ptr(), B.data_ptr(), A.data_ptr(), 64, 64, 96, 1, 1, 104, 104, 64, 0, 0, 0, 0, 0, 0) == 1
    C = torch.empty(64, 64, dtype=torch.bfloat16, device="cuda")
    rc = lib.ix_gemm_b16(A.data_ptr(), B.data_ptr(), C.data_ptr(), None, 64, 64, 100, 1, 1, 100, 100, 64, 1, 1, 0, 0, 0, 0, 0, 0, 0, 1.0, 0,
                         None, None, None, 0, None, 0, torch.cuda.current_stream().cuda_stream)
    assert rc != 0 and b"16-byte" in lib.ix_last_error()


def test_casts_round_to_nearest_even(lib):
    x = torch.randn(100003, generator=torch.Generator().manual_seed(1)).cuda() * 37.0
    y = torch.empty(100003, dtype=torch.bfloat16, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    assert lib.ix_cast_f32_b16(x.data_ptr(), y.data_ptr(), x.numel(), st) == 0
    assert torch.equal(y, x.to(torch.bfloat16))
    z = torch.empty(100003, device="cuda")
    assert lib.ix_cast_b16_f32(y.data_ptr(), z.data_ptr(), y.numel(), st) == 0
    assert torch.equal(z, y.float())
