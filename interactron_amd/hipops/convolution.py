"""hipops.convolution -- frozen-BN affine Functions, fused contraction + BN, implicit-GEMM convolutions."""
import ctypes
import gc as _gc
import os
import os as _os
from collections import namedtuple

import torch
from torch.autograd import Function as _TorchFunction
from torch.autograd.function import once_differentiable

from .. import _lib
from . import core
from .core import (Function, _L, _chk, _is_unwanted, _numel, _param_key, _req, _stream, _workspace, mark_weight, weight_view)
from . import elementwise
from .elementwise import (ReluBwd, _two_gradients, fanout)
from . import contraction
from .contraction import (GemmSpec, View, _gemm_backward, _gemm_workspace_bytes, linear)


# ---------------------------------------------------------------------------------------------------------
# FrozenBatchNorm2d affine (+ residual, + ReLU), NHWC
# ---------------------------------------------------------------------------------------------------------
def bn_fold(weight, bias, running_mean, running_var, eps=1e-5):
    C = weight.numel()
    scale = torch.empty(C, device=weight.device, dtype=torch.float32)
    shift = torch.empty_like(scale)
    _chk(_L().ix_bn_fold_f32(_req(weight).data_ptr(), _req(bias).data_ptr(), _req(running_mean).data_ptr(),
                             _req(running_var).data_ptr(), scale.data_ptr(), shift.data_ptr(), C, eps, _stream()),
         "ix_bn_fold_f32")
    return scale, shift


def _channel_affine(x, scale, shift, residual, relu):
    out = torch.empty_like(x)
    _chk(_L().ix_channel_affine_f32(x.data_ptr(), scale.data_ptr(), shift.data_ptr() if shift is not None else None,
                                    residual.data_ptr() if residual is not None else None, out.data_ptr(), x.numel(),
                                    scale.numel(), 1 if relu else 0, _stream()), "ix_channel_affine_f32")
    return out


class ChannelScale(Function):
    """x * scale[c] (channel = last dim); scale is a constant buffer."""

    @staticmethod
    def forward(ctx, x, scale):
        x = _req(x)
        ctx.save_for_backward(scale)
        return _channel_affine(x, scale, None, None, False)

    @staticmethod
    def backward(ctx, g):
        (scale,) = ctx.saved_tensors
        return ChannelScale.call(g, scale), None


class RowScale(Function):
    """w * scale[n] along the output-channel dim of a weight tensor -- linear [(E,) N, K] (tail = 1), convolution
    [(E,) Cout, KH, KW, Cin] (tail = 3) -- or of its gradient; scale is a constant buffer.  The result keeps standing for the
    Parameter in skip_param_grads."""

    @staticmethod
    def forward(ctx, w, scale, tail):
        key = _param_key(w)
        w = _req(w, "weight")
        N = w.shape[-(tail + 1)]
        R = _numel(w.shape[-tail:])
        out = torch.empty_like(w)
        _chk(_L().ix_row_scale_f32(w.data_ptr(), scale.data_ptr(), out.data_ptr(), w.numel() // (N * R), N, R, _stream()),
             "ix_row_scale_f32")
        ctx.tail = tail
        ctx.save_for_backward(scale)
        # the scaled copy keeps standing for its Parameter in skip_param_grads, but it is NOT marked for the weight-planes route:
        # it is fresh in every backward and read by exactly one contraction, so its planes could never be reused -- an eager
        # ix_wp_split_f32 launch per call for nothing, and inside a capture (planes, copy) pinned in the graph's pool for good
        out._ix_of_param = key
        return out

    @staticmethod
    def backward(ctx, g):
        (scale,) = ctx.saved_tensors
        return RowScale.call(g.contiguous(), scale, ctx.tail), None, None


# IX_BN_SCALE_ON_WEIGHTS: "1" (default) = the backward of a fused contraction + frozen-BN node applies the BN scale to the weights
# (dx = g (W o scale), dW = scale o (g^T x): two passes over a weight tensor) where it would otherwise run a pass over the
# activation-sized gradient (no ReLU, or ReLU behind a residual: the bottleneck tails and the downsample branches); "0" = g o scale
BN_SCALE_ON_WEIGHTS = os.environ.get("IX_BN_SCALE_ON_WEIGHTS", "1") == "1"


def _bn_scale_on_weights(relu, has_res, w, tail):
    return BN_SCALE_ON_WEIGHTS and (has_res or not relu) and _numel(w.shape[-tail:]) % 4 == 0


class ReluBwdChannelScale(Function):
    """[y > 0] * g * scale[c]: linear in g, so it is its own second-order form."""

    @staticmethod
    def forward(ctx, g, y, scale):
        g, y = _req(g), _req(y)
        ctx.save_for_backward(y, scale)
        out = torch.empty_like(g)
        _chk(_L().ix_relu_bwd_channel_scale_f32(g.data_ptr(), y.data_ptr(), scale.data_ptr(), out.data_ptr(), g.numel(),
                                                g.shape[-1], _stream()), "ix_relu_bwd_channel_scale_f32")
        return out

    @staticmethod
    def backward(ctx, G):
        y, scale = ctx.saved_tensors
        return ReluBwdChannelScale.call(G, y, scale), None, None


class BnAct(Function):
    """y = [relu](x*scale[c] + shift[c] (+ residual)) on NHWC activations (FrozenBatchNorm2d folded)."""

    @staticmethod
    def forward(ctx, x, scale, shift, residual, relu):
        x = _req(x)
        if residual is not None:
            residual = _req(residual)
        y = _channel_affine(x, scale, shift, residual, relu)
        ctx.relu = relu
        ctx.has_res = residual is not None
        ctx.save_for_backward(scale, y if relu else None)
        return y

    @staticmethod
    def backward(ctx, g):
        scale, y = ctx.saved_tensors
        gx, gres = _bn_act_backward(g, y, scale, ctx.relu, ctx.has_res, ctx.needs_input_grad[0], ctx.needs_input_grad[3])
        return gx, None, None, gres, None


def _bn_act_backward(g, y, scale, relu, has_res, need_x, need_res):
    """(gradient of the pre-affine tensor, gradient of the residual) of y = [relu](z * scale + shift (+ res)) -- BnAct.backward's
    arithmetic on differentiable nodes, shared with the fused contraction + affine Functions"""
    g = g.contiguous()
    if relu and not has_res:   # one pass instead of relu-backward + channel scale
        return (ReluBwdChannelScale.call(g, y, scale) if need_x else None), None
    if relu:
        g = ReluBwd.call(g, y)
    return (ChannelScale.call(g, scale) if need_x else None), (g if (has_res and need_res) else None)


# IX_FUSE_CONV_BN: "1" = backbone convolutions carry their frozen-BN affine (+ residual) (+ ReLU) on the contraction call
# (ix_gemm_bn_act_f32 / ix_conv_gemm_bn_act_f32) instead of a separate elementwise launch; "0" = separate launches
FUSE_CONV_BN = os.environ.get("IX_FUSE_CONV_BN", "1") == "1"


class GemmBnAct(Function):
    """y = [relu]((A B) * scale[n] + shift[n] (+ residual)) for the plain row-major product of Gemm (1 x 1 convolutions);
    backward = BnAct's backward followed by Gemm's (all differentiable nodes: closed under the MAML double backward)."""

    b16 = "native"

    @staticmethod
    def forward(ctx, a, b, scale, shift, residual, relu, sp, fan=1):
        ctx.set_materialize_grads(False)
        ctx.a_key, ctx.b_key = _param_key(a), _param_key(b)
        assert sp.bi == 1 and sp.alpha == 1.0 and sp.C.offset == 0 and sp.C.ld == sp.N and not sp.C.trans
        assert sp.A.offset == 0 and sp.B.offset == 0 and (sp.bo == 1 or sp.C.so == sp.M * sp.N)
        if a.dtype == torch.bfloat16:   # 16-bit mode: the affine (+ residual) (+ ReLU) in the bf16 GEMM's own store (csrc/gemm16.hip)
            from .. import b16
            wb = getattr(b, "_ix_weight", False)
            a, b, scale, shift = b16._reqd(a, "gemm A"), b16._reqd(b, "gemm B"), _req(scale), _req(shift)
            if wb:
                mark_weight(b)
            if residual is not None:
                residual = b16._reqd(residual)
                assert residual.dtype == torch.bfloat16
            out = b16.run_gemm(a, b, None, sp, scale=scale, shift=shift, residual=residual, act=1 if relu else 0)
            ctx.sp, ctx.relu, ctx.has_res = sp, relu, residual is not None
            ctx.a_shape, ctx.b_shape = tuple(a.shape), tuple(b.shape)
            ctx.save_for_backward(a, b, scale, out if relu else None)
            return out if fan == 1 else (out, out.view_as(out))
        a, b, scale, shift = _req(a, "gemm A"), _req(b, "gemm B"), _req(scale), _req(shift)
        if residual is not None:
            residual = _req(residual)
        out = torch.empty(sp.out_shape, device=a.device, dtype=torch.float32)
        nws, _ = _gemm_workspace_bytes(a.data_ptr(), b.data_ptr(), sp, presplit=False)
        ws = _workspace(nws, a.device) if nws else None
        _chk(_L().ix_gemm_bn_act_f32(a.data_ptr(), b.data_ptr(), out.data_ptr(), sp.M, sp.N, sp.K, 0 if sp.A.trans else 1,
                                     1 if sp.B.trans else 0, sp.A.ld, sp.B.ld, sp.bo, sp.A.so, sp.B.so, scale.data_ptr(),
                                     shift.data_ptr(), residual.data_ptr() if residual is not None else None,
                                     1 if relu else 0, ws.data_ptr() if nws else None, nws, _stream()), "ix_gemm_bn_act_f32")
        ctx.sp, ctx.relu, ctx.has_res = sp, relu, residual is not None
        ctx.a_shape, ctx.b_shape = tuple(a.shape), tuple(b.shape)
        ctx.save_for_backward(a, b, scale, out if relu else None)
        return out if fan == 1 else (out, out.view_as(out))   # (fan = 2: two aliases for two consumers, see _two_gradients)

    @staticmethod
    def backward(ctx, *gs):
        a, b, scale, y = ctx.saved_tensors
        if all(g is None for g in gs):
            return (None,) * 8
        need_a = ctx.needs_input_grad[0] and not _is_unwanted(ctx.a_key, core._unwanted)
        need_b = ctx.needs_input_grad[1] and not _is_unwanted(ctx.b_key, core._unwanted)
        da = db = None
        if _bn_scale_on_weights(ctx.relu, ctx.has_res, b, 1):
            g, masked = _two_gradients(gs, y, ctx.relu)
            g1 = ReluBwd.call(g, y) if (ctx.relu and not masked) else g
            want_res = ctx.has_res and ctx.needs_input_grad[4]
            gs = list(fanout(g1, int(need_a) + int(need_b) + int(want_res)))   # (one sum of its consumers' gradients in the outer backward)
            gres = gs.pop() if want_res else None
            if need_a:
                da = _gemm_backward(ctx.sp, a, RowScale.call(b, scale, 1), ctx.a_shape, ctx.b_shape, gs.pop(), True, False)[0]
            if need_b:
                db = RowScale.call(_gemm_backward(ctx.sp, a, b, ctx.a_shape, ctx.b_shape, gs.pop(), False, True)[1], scale, 1)
            return da, db, None, None, gres, None, None, None
        g, _ = _two_gradients(gs, None, False)
        gz, gres = _bn_act_backward(g, y, scale, ctx.relu, ctx.has_res, need_a or need_b, ctx.needs_input_grad[4])
        if gz is not None:
            da, db = _gemm_backward(ctx.sp, a, b, ctx.a_shape, ctx.b_shape, gz, need_a, need_b)
        return da, db, None, None, gres, None, None, None


def linear_bn_act(x, weight, scale, shift, residual, relu, fan=1):
    """[relu](linear(x, weight) * scale + shift (+ residual)) -- `linear` without bias, episode-batched weights included"""
    if weight.dim() == 3:
        E, N, K = weight.shape
        R = x.numel() // (E * K)
        sp = GemmSpec(R, N, K, E, 1, View(0, K, False, R * K, 0), View(0, K, True, N * K, 0), View(0, N, False, R * N, 0),
                      tuple(x.shape[:-1]) + (N,), 1.0)
    else:
        K, N = x.shape[-1], weight.shape[0]
        sp = GemmSpec(x.numel() // K, N, K, 1, 1, View(0, K, False, 0, 0), View(0, K, True, 0, 0), View(0, N, False, 0, 0),
                      tuple(x.shape[:-1]) + (N,), 1.0)
    return GemmBnAct.call(x, weight, scale, shift, residual, relu, sp, fan)


# ---------------------------------------------------------------------------------------------------------
# convolution pieces (NHWC)
# ---------------------------------------------------------------------------------------------------------
ConvGeom = namedtuple("ConvGeom", "n H W C KH KW stride pad dil OH OW Kp")


def conv_geom(n, H, W, C, KH, KW, stride, pad, dil):
    OH = (H + 2 * pad - dil * (KH - 1) - 1) // stride + 1
    OW = (W + 2 * pad - dil * (KW - 1) - 1) // stride + 1
    K = KH * KW * C
    return ConvGeom(n, H, W, C, KH, KW, stride, pad, dil, OH, OW, (K + 3) // 4 * 4)


def _im2col(x, g, strides):
    cols = torch.empty(g.n * g.OH * g.OW, g.Kp, device=x.device, dtype=torch.float32)
    sxn, sxh, sxw, sxc = strides
    _chk(_L().ix_im2col_f32(x.data_ptr(), cols.data_ptr(), g.n, g.H, g.W, g.C, sxn, sxh, sxw, sxc, g.KH, g.KW, g.stride,
                            g.pad, g.dil, g.Kp, _stream()), "ix_im2col_f32")
    return cols


def im2col_any_layout(x_nchw_or_nhwc, g, channels_last):
    """Non-differentiable patch extraction for the frozen stem; accepts the NCHW input frames directly."""
    x = _req(x_nchw_or_nhwc)
    if channels_last:
        strides = (g.H * g.W * g.C, g.W * g.C, g.C, 1)
    else:
        strides = (g.C * g.H * g.W, g.W, 1, g.H * g.W)
    return _im2col(x, g, strides)


class Im2Col(Function):
    @staticmethod
    def forward(ctx, x, g):
        x = _req(x)
        ctx.g = g
        return _im2col(x, g, (g.H * g.W * g.C, g.W * g.C, g.C, 1))

    @staticmethod
    def backward(ctx, dcols):
        return Col2Im.call(dcols, ctx.g), None


class Col2Im(Function):
    @staticmethod
    def forward(ctx, cols, g):
        cols = _req(cols)
        ctx.g = g
        dx = torch.empty(g.n, g.H, g.W, g.C, device=cols.device, dtype=torch.float32)
        _chk(_L().ix_col2im_f32(cols.data_ptr(), dx.data_ptr(), g.n, g.H, g.W, g.C, g.KH, g.KW, g.stride, g.pad, g.dil,
                                g.Kp, _stream()), "ix_col2im_f32")
        return dx

    @staticmethod
    def backward(ctx, G):
        return Im2Col.call(G, ctx.g), None


def maxpool_nhwc(x, k, stride, pad):
    x = _req(x)
    n, H, W, C = x.shape
    OH, OW = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    y = torch.empty(n, OH, OW, C, device=x.device, dtype=torch.float32)
    _chk(_L().ix_maxpool_nhwc_f32(x.data_ptr(), y.data_ptr(), n, H, W, C, k, stride, pad, _stream()), "ix_maxpool_nhwc_f32")
    return y


# ---- implicit-GEMM convolution (csrc/gemm.hip ix_conv_gemm_f32): no patch matrix in HBM ----------------------------------
ConvGemmGeom = namedtuple("ConvGemmGeom", "E imgs H W Cin OH OW Cout KH KW stride pad dil")
CONV_IMPL = os.environ.get("IX_CONV", "implicit")   # "im2col": keep every convolution on the patch-matrix path (A/B runs)
_conv_ok = {}


def conv_gemm_supported(cg):
    ok = _conv_ok.get(cg)
    if ok is None:
        ok = _conv_ok[cg] = CONV_IMPL == "implicit" and bool(_L().ix_conv_gemm_supported(
            cg.E, cg.imgs, cg.H, cg.W, cg.Cin, cg.OH, cg.OW, cg.Cout, cg.KH, cg.KW, cg.stride, cg.pad, cg.dil))
    return ok


_conv_ws = {}


def _conv_gemm(kind, src, other, out_shape, cg):
    out = torch.empty(out_shape, device=src.device, dtype=torch.float32)
    nws = _conv_ws.get((kind, cg))
    if nws is None:
        n = ctypes.c_size_t(0)
        _chk(_L().ix_workspace_bytes_conv_gemm_f32(kind, cg.E, cg.imgs, cg.H, cg.W, cg.Cin, cg.OH, cg.OW, cg.Cout, cg.KH, cg.KW,
                                                   cg.stride, cg.pad, cg.dil, ctypes.byref(n)), "ix_workspace_bytes_conv_gemm_f32")
        nws = _conv_ws[(kind, cg)] = n.value
    ws = _workspace(nws, src.device) if nws else None
    _chk(_L().ix_conv_gemm_f32(kind, src.data_ptr(), other.data_ptr(), out.data_ptr(), cg.E, cg.imgs, cg.H, cg.W, cg.Cin,
                               cg.OH, cg.OW, cg.Cout, cg.KH, cg.KW, cg.stride, cg.pad, cg.dil, ws.data_ptr() if nws else None,
                               nws, _stream()), "ix_conv_gemm_f32")
    return out


class ConvFwd(Function):
    """y = conv(x, w): x [E*imgs, H, W, Cin], w [(E,) Cout, KH, KW, Cin] -> [E*imgs, OH, OW, Cout].  With ConvBwdData and
    ConvBwdWeight the three implicit-GEMM kinds are closed under differentiation (each one's backward is the other two)."""

    @staticmethod
    def forward(ctx, x, w, cg):
        ctx.w_key = _param_key(w)
        x, w = _req(x, "conv x"), _req(w, "conv weight")
        ctx.cg = cg
        ctx.save_for_backward(x, w)
        return _conv_gemm(0, x, w, (cg.E * cg.imgs, cg.OH, cg.OW, cg.Cout), cg)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        need_w = ctx.needs_input_grad[1] and not _is_unwanted(ctx.w_key, core._unwanted)
        dx = ConvBwdData.call(dy, w, ctx.cg) if ctx.needs_input_grad[0] else None
        dw = ConvBwdWeight.call(dy, x, ctx.cg, tuple(w.shape)) if need_w else None
        return dx, dw, None


class ConvBwdData(Function):
    @staticmethod
    def forward(ctx, dy, w, cg):
        ctx.w_key = _param_key(w)
        dy, w = _req(dy, "conv dy"), _req(w, "conv weight")
        ctx.cg = cg
        ctx.save_for_backward(dy, w)
        return _conv_gemm(1, dy, w, (cg.E * cg.imgs, cg.H, cg.W, cg.Cin), cg)

    @staticmethod
    def backward(ctx, g):
        dy, w = ctx.saved_tensors
        g = g.contiguous()
        need_w = ctx.needs_input_grad[1] and not _is_unwanted(ctx.w_key, core._unwanted)
        ddy = ConvFwd.call(g, w, ctx.cg) if ctx.needs_input_grad[0] else None
        dw = ConvBwdWeight.call(dy, g, ctx.cg, tuple(w.shape)) if need_w else None
        return ddy, dw, None


class ConvBwdWeight(Function):
    @staticmethod
    def forward(ctx, dy, x, cg, w_shape):
        dy, x = _req(dy, "conv dy"), _req(x, "conv x")
        ctx.cg = cg
        ctx.save_for_backward(dy, x)
        return _conv_gemm(2, dy, x, w_shape, cg)

    @staticmethod
    def backward(ctx, g):
        dy, x = ctx.saved_tensors
        g = g.contiguous()
        ddy = ConvFwd.call(x, g, ctx.cg) if ctx.needs_input_grad[0] else None
        dx = ConvBwdData.call(dy, g, ctx.cg) if ctx.needs_input_grad[1] else None
        return ddy, dx, None, None


class ConvFwdBnAct(Function):
    """y = [relu](conv(x, w) * scale[c] + shift[c] (+ residual)): ConvFwd with the frozen-BN affine in the contraction's store"""

    @staticmethod
    def forward(ctx, x, w, scale, shift, residual, relu, cg, fan=1):
        ctx.set_materialize_grads(False)
        ctx.w_key = _param_key(w)
        x, w, scale, shift = _req(x, "conv x"), _req(w, "conv weight"), _req(scale), _req(shift)
        if residual is not None:
            residual = _req(residual)
        out = torch.empty((cg.E * cg.imgs, cg.OH, cg.OW, cg.Cout), device=x.device, dtype=torch.float32)
        nws = _conv_ws.get((0, cg))
        if nws is None:
            n = ctypes.c_size_t(0)
            _chk(_L().ix_workspace_bytes_conv_gemm_f32(0, cg.E, cg.imgs, cg.H, cg.W, cg.Cin, cg.OH, cg.OW, cg.Cout, cg.KH, cg.KW,
                                                       cg.stride, cg.pad, cg.dil, ctypes.byref(n)), "ix_workspace_bytes_conv_gemm_f32")
            nws = _conv_ws[(0, cg)] = n.value
        ws = _workspace(nws, x.device) if nws else None
        _chk(_L().ix_conv_gemm_bn_act_f32(x.data_ptr(), w.data_ptr(), out.data_ptr(), cg.E, cg.imgs, cg.H, cg.W, cg.Cin, cg.OH,
                                          cg.OW, cg.Cout, cg.KH, cg.KW, cg.stride, cg.pad, cg.dil, scale.data_ptr(),
                                          shift.data_ptr(), residual.data_ptr() if residual is not None else None,
                                          1 if relu else 0, ws.data_ptr() if nws else None, nws, _stream()),
             "ix_conv_gemm_bn_act_f32")
        ctx.cg, ctx.relu, ctx.has_res = cg, relu, residual is not None
        ctx.save_for_backward(x, w, scale, out if relu else None)
        return out if fan == 1 else (out, out.view_as(out))

    @staticmethod
    def backward(ctx, *gs):
        x, w, scale, y = ctx.saved_tensors
        if all(g is None for g in gs):
            return (None,) * 8
        need_w = ctx.needs_input_grad[1] and not _is_unwanted(ctx.w_key, core._unwanted)
        dx = dw = None
        if _bn_scale_on_weights(ctx.relu, ctx.has_res, w, 3):
            g, masked = _two_gradients(gs, y, ctx.relu)
            g1 = ReluBwd.call(g, y) if (ctx.relu and not masked) else g
            want_res = ctx.has_res and ctx.needs_input_grad[4]
            gs = list(fanout(g1, int(bool(ctx.needs_input_grad[0])) + int(need_w) + int(want_res)))
            gres = gs.pop() if want_res else None
            if ctx.needs_input_grad[0]:
                dx = ConvBwdData.call(gs.pop(), RowScale.call(w, scale, 3), ctx.cg)
            if need_w:
                dw = RowScale.call(ConvBwdWeight.call(gs.pop(), x, ctx.cg, tuple(w.shape)), scale, 3)
            return dx, dw, None, None, gres, None, None, None
        g, _ = _two_gradients(gs, None, False)
        gz, gres = _bn_act_backward(g, y, scale, ctx.relu, ctx.has_res, ctx.needs_input_grad[0] or need_w, ctx.needs_input_grad[4])
        if gz is not None:
            dx = ConvBwdData.call(gz, w, ctx.cg) if ctx.needs_input_grad[0] else None
            dw = ConvBwdWeight.call(gz, x, ctx.cg, tuple(w.shape)) if need_w else None
        return dx, dw, None, None, gres, None, None, None


def conv2d_nhwc_bn_act(x, weight, scale, shift, residual=None, relu=False, stride=1, pad=0, dil=1, fan=1):
    """[relu](conv2d_nhwc(x, weight) * scale + shift (+ residual)): one launch where the contraction kernel takes the affine
    (1 x 1 / stride 1 and implicit-GEMM geometries, output channels % 4 == 0), else the two separate nodes"""
    n, H, W, C = x.shape
    batched = weight.dim() == 5
    Cout, KH, KW = weight.shape[-4], weight.shape[-3], weight.shape[-2]
    E = weight.shape[0] if batched else 1
    if FUSE_CONV_BN and Cout % 4 == 0:
        if KH == 1 and KW == 1 and stride == 1 and pad == 0:
            wv = weight_view(weight, E, Cout, C) if batched else weight_view(weight, Cout, C)
            return linear_bn_act(x, wv, scale, shift, residual, relu, fan)
        g = conv_geom(n, H, W, C, KH, KW, stride, pad, dil)
        cg = ConvGemmGeom(E, n // E, H, W, C, g.OH, g.OW, Cout, KH, KW, stride, pad, dil)
        if conv_gemm_supported(cg):
            return ConvFwdBnAct.call(x, weight, scale, shift, residual, relu, cg, fan)
    y = BnAct.apply(conv2d_nhwc(x, weight, stride, pad, dil), scale, shift, residual, relu)
    return y if fan == 1 else fanout(y, fan)


def conv2d_nhwc(x, weight, stride=1, pad=0, dil=1):
    """x [n,H,W,Cin] NHWC, weight [Cout,KH,KW,Cin] (nn.Conv2dNHWC's storage layout: the patch-matrix column order
    (kh, kw, cin), so it is the contraction's k-contiguous operand as stored) -> [n,OH,OW,Cout]."""
    n, H, W, C = x.shape
    if weight.dim() == 5:   # episode-batched fast weights [E, Cout, KH, KW, Cin]; frames of episode e are x[e*n/E:(e+1)*n/E]
        E, Cout, KH, KW, Cin = weight.shape
        assert Cin == C and n % E == 0
        if KH == 1 and KW == 1 and stride == 1 and pad == 0:
            return linear(x, weight_view(weight, E, Cout, Cin))
        g = conv_geom(n, H, W, C, KH, KW, stride, pad, dil)
        cg = ConvGemmGeom(E, n // E, H, W, Cin, g.OH, g.OW, Cout, KH, KW, stride, pad, dil)
        if conv_gemm_supported(cg):
            return ConvFwd.call(x, weight, cg)
        cols = Im2Col.call(x, g)
        assert g.Kp == KH * KW * Cin, "episode-batched convs need KH*KW*Cin % 4 == 0"
        return linear(cols.reshape(E, -1, g.Kp), weight_view(weight, E, Cout, KH * KW * Cin)).reshape(n, g.OH, g.OW, Cout)
    Cout, KH, KW, Cin = weight.shape
    assert Cin == C
    if KH == 1 and KW == 1 and stride == 1 and pad == 0:
        return linear(x, weight_view(weight, Cout, Cin))
    g = conv_geom(n, H, W, C, KH, KW, stride, pad, dil)
    cg = ConvGemmGeom(1, n, H, W, Cin, g.OH, g.OW, Cout, KH, KW, stride, pad, dil)
    if conv_gemm_supported(cg):
        return ConvFwd.call(x, weight, cg)
    cols = Im2Col.call(x, g)
    return linear(cols, weight_view(weight, Cout, KH * KW * Cin)).reshape(n, g.OH, g.OW, Cout)
