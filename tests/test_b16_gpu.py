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


@pytest.fixture(params=[1, 2, "big"])
def stages(request, lib):
    """every bf16 GEMM test runs on the three forms of the kernel: 128 x 128 tiles with one LDS stage / four workgroups per CU, with two
    stages and the next K step's DMA in flight (ix_gemm_b16_set_stages), and the 256 x 256-tile form forced onto every plain contraction
    (ix_gemm_b16_set_big(2); the convolution gathers have no such form and run the default one)"""
    old = lib.ix_gemm_b16_set_stages(1 if request.param == "big" else request.param)
    old_big = lib.ix_gemm_b16_set_big(2 if request.param == "big" else 0)
    yield request.param
    lib.ix_gemm_b16_set_stages(old)
    lib.ix_gemm_b16_set_big(old_big)


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
def test_bf16_gemm_layouts_and_tails_against_float64(lib, stages, a_kc, b_kc, M, N, K):
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


def test_bf16_gemm_batches_bias_affine_residual_and_activations(lib, stages):
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


@pytest.mark.parametrize("a_kc,b_kc", [(True, True), (True, False), (False, False)])
def test_bf16_gemm_256_tile_form_is_what_large_launches_run_and_equals_the_128_tile_form(lib, a_kc, b_kc):
    """A launch of >= ~200 tiles of 256 x 256 takes the eight-wave form by itself (ix_gemm_b16_set_big(1), the default): same k order,
    same fp32 accumulation -- bit for bit the 128 x 128 kernel's result (no split-K in either plan at this shape), bf16 with bias + GELU
    through the LDS store pass and fp32; ragged M (not a multiple of 256) and N = 1000 (a partial last tile, N = 8 (mod 16))."""
    M, N, K = 16384 + 72, 1000, 520
    A = rnd(M if a_kc else K, K if a_kc else M, seed=31).cuda()
    B = rnd(N if b_kc else K, K if b_kc else N, seed=32, scale=0.1).cuda()
    bias = torch.randn(N, generator=torch.Generator().manual_seed(33)).cuda()
    res = rnd(M, N, seed=34).cuda()
    outs = {}
    for big in (0, 1):
        old = lib.ix_gemm_b16_set_big(big)
        try:
            outs[big] = [gemm_b16(lib, A, B, M, N, K, a_kc, b_kc, K if a_kc else M, K if b_kc else N, bias=bias, c_f32=f, act=2, res=r)[0]
                         for f, r in ((False, None), (False, res), (True, None))]
        finally:
            lib.ix_gemm_b16_set_big(old)
    for a, b in zip(outs[0], outs[1]):
        assert bool(torch.isfinite(b).all())
        assert torch.equal(a, b)
    a = A.double().cpu() if a_kc else A.double().cpu().t()
    b = B.double().cpu().t() if b_kc else B.double().cpu()
    ref = torch.nn.functional.gelu(a[:512] @ b + bias.double().cpu())
    assert float((outs[1][2][0, 0, :512].double().cpu() - ref).abs().max()) <= 1e-4 * float(ref.abs().max())


def test_bf16_gemm_split_k_is_ordered_and_equals_one_pass(lib, stages):
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


# ---- the elementwise / channel / LayerNorm twins (csrc/ew16.hip through b16.py) against the fp32 Functions on the same values --------
def _close16(got, ref32, what, extra=0.0):
    """a bf16 result against the fp32 kernel's result on the same (bf16-valued) inputs: one rounding apart"""
    assert got.dtype == torch.bfloat16, what
    err = (got.float() - ref32).abs()
    tol = ref32.abs() * 2.0 ** -8 + 1e-30 + extra
    assert bool((err <= tol).all()), (what, float((err - tol).max()))


def test_elementwise_twins_equal_the_fp32_functions_rounded_once(lib):
    from interactron_amd import b16, hipops as ops
    n = 3 * 1000 * 256 + 5    # (a ragged tail behind the 16-byte bulk)
    x, y, z = rnd(n, seed=1).cuda(), rnd(n, seed=2).cuda(), rnd(n, seed=3).cuda()
    f = lambda t: t.float()
    seed = 0x1234567
    cases = [
        ("add", ops.Axpby, (x, y, 1.0, 1.0)), ("axpby", ops.Axpby, (x, y, 0.5, -2.0)), ("scale", ops.Scale, (x, 0.37)),
        ("relu", ops.Relu, (x,)), ("relu_bwd", ops.ReluBwd, (x, y)), ("relu_bwd_sum", ops.ReluBwdSum, (x, y, z)),
        ("relu_bwd_scaled", ops.ReluBwdScaled, (x, y, 1.0 / 0.9)), ("gelu", ops.Gelu, (x,)), ("gelu_bwd", ops.GeluBwd, (x, y)),
        ("dropout", ops._Dropout, (x, 0.1, seed)), ("relu_dropout", ops.ReluDropout, (x, 0.1, seed)),
        ("add_dropout", ops.AddDropout, (x, y, 0.1, seed)), ("sum_n", ops.SumN, (x, y, z, x, y)),
    ]
    with torch.no_grad():
        for name, fn, args in cases:
            got = fn.call(*args)
            ref = fn.call(*[f(a) if torch.is_tensor(a) else a for a in args])
            # (FMA contraction may differ by an fp32 ulp before the rounding: 2e-7 of the value on top of the bf16 half-ulp)
            _close16(got, ref, name, extra=4e-7 * float(ref.abs().max()))
            if "dropout" in name:   # the SAME mask: zeros in the same places
                assert torch.equal(got == 0, ref == 0), name
    assert ops.Relu.b16_twin is b16.Relu16


def test_channel_twins_and_bias_gradient_column_sums(lib):
    from interactron_amd import hipops as ops
    rows, C = 3000, 256
    x, y = rnd(rows, C, seed=4).cuda(), rnd(rows, C, seed=5).cuda()
    g = torch.Generator().manual_seed(6)
    scale, shift = (torch.rand(C, generator=g) + 0.5).cuda(), torch.randn(C, generator=g).cuda()
    with torch.no_grad():
        for relu in (False, True):
            for res in (None, y):
                got = ops.BnAct.call(x, scale, shift, res, relu)
                ref = ops.BnAct.call(x.float(), scale, shift, None if res is None else res.float(), relu)
                _close16(got, ref, "bn_act", extra=4e-7 * float(ref.abs().max()))
        _close16(ops.ChannelScale.call(x, scale), ops.ChannelScale.call(x.float(), scale), "channel_scale")
        _close16(ops.ReluBwdChannelScale.call(x, y, scale), ops.ReluBwdChannelScale.call(x.float(), y.float(), scale), "relu_bwd_channel_scale")
        v = torch.randn(4, C * 2, generator=g).cuda()   # one row vector per slab of 25 rows
        a = rnd(4 * 25, C * 2, seed=7).cuda()
        _close16(ops.AddRowVec.call(a, v, 4), ops.AddRowVec.call(a.float(), v, 4), "add_rowvec", extra=1e-6)
        cs = ops.ColSum.call(x)
        assert cs.dtype == torch.float32
        ref = x.double().sum(0).cpu()
        assert float((cs.double().cpu() - ref).abs().max()) <= 1e-5 * float(x.double().abs().sum(0).max())
        cs3 = ops.ColSum.call(x.reshape(3, 1000, C))
        assert float((cs3.double().cpu() - x.double().reshape(3, 1000, C).sum(1).cpu()).abs().max()) <= 1e-5 * float(x.double().abs().sum(0).max())
        assert torch.equal(cs, ops.ColSum.call(x))   # ordered partial sums: the same bits every time


@pytest.mark.parametrize("D", [256, 512])
def test_layernorm_twin_forward_and_first_derivative(lib, D):
    """nn.LayerNorm (models/gpt.py:60-78, models/detr_models/transformer.py:148-232) on bf16 rows: output and dx one rounding from the
    fp32 kernels' on the same values, fp32 parameter gradients to 1e-4 of their scale, statistics fp32."""
    from interactron_amd import hipops as ops
    rows = 4133
    x = (rnd(rows, D, seed=8) * 3 + 0.5).cuda()
    g = torch.Generator().manual_seed(9)
    gamma, beta = (torch.rand(D, generator=g) + 0.5).cuda(), torch.randn(D, generator=g).cuda()
    dy = rnd(rows, D, seed=10).cuda()

    def run(xx, dd):
        xx = xx.clone().requires_grad_(True)
        ga, be = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
        yy = ops.layer_norm(xx, ga, be)
        yy.backward(dd)
        return yy.detach(), xx.grad, ga.grad, be.grad

    y16, dx16, dg16, db16 = run(x, dy)
    y32, dx32, dg32, db32 = run(x.float(), dy.float())
    assert dx16.dtype == torch.bfloat16 and dg16.dtype == torch.float32 and db16.dtype == torch.float32
    _close16(y16, y32, "ln forward", extra=2e-6 * float(y32.abs().max()))
    _close16(dx16, dx32, "ln dx", extra=2e-6 * float(dx32.abs().max()))
    for a, b, nm in ((dg16, dg32, "dgamma"), (db16, db32, "dbeta")):
        assert float((a - b).abs().max()) <= 1e-4 * float(b.abs().max()), nm


@pytest.mark.parametrize("geom", [  # (n, H, W, Cin, Cout, k, stride, pad, dil, residual)
    (3, 19, 19, 256, 256, 3, 1, 1, 1, False),     # layer3 conv2
    (2, 38, 37, 128, 128, 3, 2, 1, 1, False),     # layer2.0 conv2: stride on the 3 x 3, odd width
    (2, 19, 19, 512, 512, 3, 1, 2, 2, False),     # layer4 conv2: dilation 2
    (2, 38, 38, 256, 512, 1, 2, 0, 1, False),     # strided 1 x 1 downsample
    (5, 10, 10, 128, 256, 1, 1, 0, 1, True),      # plain 1 x 1 + residual (the bf16 GEMM's own epilogue)
])
def test_bf16_convolutions_forward_and_both_gradients(lib, stages, geom):
    """The implicit-GEMM convolution kinds of the 16-bit mode (ix_conv_gemm_b16: forward with the frozen-BN affine + ReLU in the store,
    data gradient, weight gradient; reference backbone.py:88-90 bottlenecks) against the fp32 path on the SAME bf16-valued inputs:
    outputs and dx one bf16 rounding (of an accumulated value) apart, the fp32 weight gradient within 5e-3 of its scale."""
    from interactron_amd import b16, hipops as ops
    n, H, W, Cin, Cout, k, stride, pad, dil, use_res = geom
    g = torch.Generator().manual_seed(11)
    x = rnd(n, H, W, Cin, seed=12).cuda()
    w = (torch.randn(Cout, k, k, Cin, generator=g) * (2.0 / (k * k * Cin)) ** 0.5).cuda()
    scale, shift = (torch.rand(Cout, generator=g) + 0.5).cuda(), (torch.randn(Cout, generator=g) * 0.1).cuda()
    OH, OW = (H + 2 * pad - dil * (k - 1) - 1) // stride + 1, (W + 2 * pad - dil * (k - 1) - 1) // stride + 1
    res = rnd(n, OH, OW, Cout, seed=13).cuda() if use_res else None
    dy = rnd(n, OH, OW, Cout, seed=14).cuda()
    before = b16._stats["native_gemms"]

    def run(xx, rr, dd, ww):
        xx = xx.clone().requires_grad_(True)
        ww = ww.clone().requires_grad_(True)
        yy = ops.conv2d_nhwc_bn_act(xx, ww, scale, shift, rr, True, stride, pad, dil)
        yy.backward(dd)
        return yy.detach(), xx.grad, ww.grad

    y16, dx16, dw16 = run(x, res, dy, w)
    assert b16._stats["native_gemms"] >= before + 3, "the bf16 kernels did not run"
    # the fp32 reference multiplies the SAME bf16-rounded weights
    y32, dx32, dw32 = run(x.float(), None if res is None else res.float(), dy.float(), w.to(torch.bfloat16).float())
    assert y16.dtype == torch.bfloat16 and dx16.dtype == torch.bfloat16 and dw16.dtype == torch.float32
    # (dx additionally carries the bf16 rounding of the ReLU-masked gradient and, behind a residual, of the BN-scaled weights)
    for a, b, what in ((y16, y32, "y"), (dx16, dx32, "dx")):
        err = (a.float() - b).abs()
        tol = b.abs() * 2.0 ** -7 + (2e-3 if what == "y" else 6e-3) * float(b.abs().max())
        assert bool((err <= tol).all()), (what, float(err.max()), float(b.abs().max()))
    assert float((dw16 - dw32).abs().max()) <= 5e-3 * float(dw32.abs().max()), float((dw16 - dw32).abs().max()) / float(dw32.abs().max())


@pytest.mark.parametrize("hd,masked,pdrop", [(64, False, 0.1), (64, True, 0.0), (32, True, 0.1)])
def test_attention_twin_forward_and_first_derivative(lib, hd, masked, pdrop):
    """Attention on bf16 q / k / v (packed [k | q | v] of the fusion blocks at head dim 64: the single-term passes of csrc/flash16.hip;
    head dim 32 of the detector: the 32 x 32 family on the same planes) against float64 autograd on the same bf16 values with the
    kernels' own dropout mask: output and gradients within 1 % of their scale (the [L, S] intermediates are rounded once to fp16 in the
    single-term passes; results are stored as bf16)."""
    import math
    from interactron_amd import b16, hipops as ops
    from tests.test_ops_gpu import _ref_attention_drop
    n, H, L = 2, 4, 300
    E = H * hd
    kqv = rnd(n, L, 3 * E, seed=31).cuda()
    scale = 1.0 / math.sqrt(hd)
    g = ops.AttnGeom(n, H, L, L, hd, 3 * E, 3 * E, E, 0, 3 * E, 2 * E, scale)
    mask = None
    if masked:
        mask = torch.zeros(n, L, dtype=torch.uint8, device="cuda")
        mask[1, L - 40:] = 1
    seed = 0x7654321
    drop = ops.flash_dropmask(n * H, L, L, pdrop, seed).cpu().double() if pdrop > 0 else None
    gy = rnd(n, L, E, seed=32).cuda()
    a = kqv.clone().requires_grad_(True)
    out = ops.FlashAttention.apply(a, a, a, g, mask, pdrop, seed)
    assert out.dtype == torch.bfloat16
    (ga,) = torch.autograd.grad(out, [a], gy)
    assert ga.dtype == torch.bfloat16
    r = kqv.double().cpu().requires_grad_(True)
    ro = _ref_attention_drop(r[..., E:2 * E], r[..., :E], r[..., 2 * E:], H, scale, None if mask is None else mask.cpu(), drop)
    (rg,) = torch.autograd.grad(ro, [r], gy.double().cpu())
    for got, ref, what in ((out, ro, "output"), (ga, rg, "gradient")):
        err = float((got.double().cpu() - ref.detach()).abs().max())
        assert err <= 1e-2 * float(ref.abs().max()), (what, err, float(ref.abs().max()))


def test_bias_gradient_rides_on_the_weight_gradient_contraction(lib):
    """ix_gemm_rowsum_b16: dW = dY^T x (fp32) and dbias = colsum(dY) from ONE launch -- a Linear's weight and bias gradient in the
    16-bit mode (with and without the split along K) against float64."""
    from interactron_amd import hipops as ops
    for rows, out, inn in ((28880, 256, 512), (700, 264, 136), (4000, 2048, 256)):
        x, w = rnd(rows, inn, seed=41).cuda(), rnd(out, inn, seed=42, scale=0.1).cuda().float().requires_grad_(True)
        bias = torch.zeros(out, device="cuda", requires_grad=True)
        dy = rnd(rows, out, seed=43).cuda()
        xg = x.clone().requires_grad_(True)
        y = ops.linear(xg, w, bias)
        assert y.dtype == torch.bfloat16
        y.backward(dy)
        ref_w = dy.double().cpu().t() @ x.double().cpu()
        ref_b = dy.double().cpu().sum(0)
        assert w.grad.dtype == torch.float32 and bias.grad.dtype == torch.float32
        assert float((w.grad.double().cpu() - ref_w).abs().max()) <= 1e-5 * float((dy.double().cpu().abs().t() @ x.double().cpu().abs()).max())
        assert float((bias.grad.double().cpu() - ref_b).abs().max()) <= 1e-5 * float(dy.double().cpu().abs().sum(0).max())
