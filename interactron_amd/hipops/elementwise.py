"""hipops.elementwise -- elementwise / broadcast / reduction Functions, dropout, softmax and LayerNorm."""
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
from .core import (Function, _L, _NullCtx, _chk, _numel, _reduce_ws, _req, _stream)


# ---------------------------------------------------------------------------------------------------------
# elementwise / broadcast
# ---------------------------------------------------------------------------------------------------------
class ColSum(Function):
    """[rows, C] -> [C], or grouped [G, rows, C] -> [G, C]."""
    b16_out = False   # (a bias gradient: fp32)

    @staticmethod
    def forward(ctx, x):
        x = _req(x)
        if x.dim() == 3:
            G, rows, C = x.shape
            out = torch.empty(G, C, device=x.device, dtype=torch.float32)
        else:
            (rows, C), G = x.shape, 1
            out = torch.empty(C, device=x.device, dtype=torch.float32)
        ctx.rows = rows
        wp, wn = _reduce_ws("colsum", rows, C, G, x.device)
        _chk(_L().ix_colsum_f32(x.data_ptr(), out.data_ptr(), rows, C, G, wp, wn, _stream()), "ix_colsum_f32")
        return out

    @staticmethod
    def backward(ctx, g):
        return BcastRows.call(g, ctx.rows)


class BcastRows(Function):
    """[C] -> [rows, C], or grouped [G, C] -> [G, rows, C]."""

    @staticmethod
    def forward(ctx, v, rows):
        v = _req(v)
        if v.dim() == 2:
            G, C = v.shape
            out = torch.empty(G, rows, C, device=v.device, dtype=torch.float32)
        else:
            G, C = 1, v.numel()
            out = torch.empty(rows, C, device=v.device, dtype=torch.float32)
        _chk(_L().ix_bcast_rows_f32(v.data_ptr(), out.data_ptr(), rows, C, G, _stream()), "ix_bcast_rows_f32")
        return out

    @staticmethod
    def backward(ctx, g):
        return ColSum.call(g), None


class Axpby(Function):
    """alpha*a + beta*b (same shapes)."""

    @staticmethod
    def forward(ctx, a, b, alpha, beta):
        a, b = _req(a), _req(b)
        assert a.shape == b.shape, (a.shape, b.shape)
        ctx.alpha, ctx.beta = alpha, beta
        out = torch.empty_like(a)
        _chk(_L().ix_axpby_f32(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), alpha, beta, _stream()),
             "ix_axpby_f32")
        return out

    @staticmethod
    def backward(ctx, g):
        ga = g if ctx.alpha == 1.0 else Scale.call(g, ctx.alpha)
        gb = g if ctx.beta == 1.0 else Scale.call(g, ctx.beta)
        return (ga if ctx.needs_input_grad[0] else None), (gb if ctx.needs_input_grad[1] else None), None, None


def add(a, b):
    return Axpby.call(a, b, 1.0, 1.0)


# ---- tensors with several consumers --------------------------------------------------------------------------------------
# The autograd engine sums the gradients of a tensor that feeds n nodes with n - 1 two-operand aten::add launches (each
# reads two tensors and writes one): 800 launches / 13 ms of a 16-episode step were the last stock kernels on the path.
# `fanout(x, n)` hands out n aliases of x whose gradients come back TOGETHER and are summed by one hand-written pass
# (ix_sum_n_f32: n reads, one write, left to right).  Closed under differentiation: the sum's own backward hands its
# cotangent to every operand, no kernel.
FANOUT = os.environ.get("IX_FANOUT", "1") == "1"   # "0": plain aliases, autograd sums (A/B runs)


def sum_n(tensors):
    """((t0 + t1) + t2) + ... over 2..8 tensors of one shape, one launch; longer lists in groups of 8"""
    ts = [_req(t) for t in tensors]
    while len(ts) > 1:
        head, ts = ts[:8], ts[8:]
        if len(head) == 1:
            ts.insert(0, head[0])
            break
        out = torch.empty_like(head[0])
        arr = (ctypes.c_void_p * len(head))(*[t.data_ptr() for t in head])
        _chk(_L().ix_sum_n_f32(arr, len(head), out.data_ptr(), out.numel(), _stream()), "ix_sum_n_f32")
        ts.insert(0, out)
    return ts[0]


class SumN(Function):
    @staticmethod
    def forward(ctx, *xs):
        assert all(x.shape == xs[0].shape for x in xs), [tuple(x.shape) for x in xs]
        return sum_n(xs)

    @staticmethod
    def backward(ctx, g):
        return tuple(g if need else None for need in ctx.needs_input_grad)


class Fanout(Function):
    b16 = "native"   # (aliases: no arithmetic)
    @staticmethod
    def forward(ctx, x, n):
        ctx.set_materialize_grads(False)
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *gs):
        gs = [g for g in gs if g is not None]
        if not gs:
            return None, None
        return (gs[0] if len(gs) == 1 else SumN.call(*gs)), None


def fanout(x, n):
    """n aliases of x for n consumers (x itself n times when nothing is recorded or x needs no gradient)"""
    if n <= 1 or not FANOUT or not torch.is_grad_enabled() or not x.requires_grad:
        return (x,) * n
    return Fanout.apply(x, n)


class Scale(Function):
    @staticmethod
    def forward(ctx, x, alpha):
        x = _req(x)
        ctx.alpha = alpha
        out = torch.empty_like(x)
        _chk(_L().ix_scale_f32(x.data_ptr(), out.data_ptr(), x.numel(), alpha, _stream()), "ix_scale_f32")
        return out

    @staticmethod
    def backward(ctx, g):
        return Scale.call(g, ctx.alpha), None


class AddRowVec(Function):
    """a [R, C] + v [C] broadcast over rows (learned query / position tables shared by all frames)."""

    @staticmethod
    def forward(ctx, a, v, groups=1):
        """groups > 1: v is [groups, C] (one vector per episode) and a is [groups, rows, C] flattened any way."""
        a, v = _req(a), _req(v)
        C = v.numel() // groups
        assert a.numel() % (C * groups) == 0
        out = torch.empty_like(a)
        _chk(_L().ix_add_rowvec_f32(a.data_ptr(), v.data_ptr(), out.data_ptr(), a.numel() // (C * groups), C, groups,
                                    _stream()), "ix_add_rowvec_f32")
        ctx.vshape, ctx.groups = tuple(v.shape), groups
        return out

    @staticmethod
    def backward(ctx, g):
        gv = None
        if ctx.needs_input_grad[1]:
            C = _numel(ctx.vshape) // ctx.groups
            gg = g.reshape(ctx.groups, -1, C) if ctx.groups > 1 else g.reshape(-1, C)
            gv = ColSum.call(gg).reshape(ctx.vshape)
        return g, gv, None


class Dot(Function):
    """sum(a*b) -> 0-d tensor."""
    b16_out = False   # (a scalar)

    @staticmethod
    def forward(ctx, a, b):
        a, b = _req(a), _req(b)
        ctx.save_for_backward(a, b)
        out = torch.empty((), device=a.device, dtype=torch.float32)
        wp, wn = _reduce_ws("scalar", 0, 0, 0, a.device)
        _chk(_L().ix_dot_f32(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), wp, wn, _stream()), "ix_dot_f32")
        return out

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        return ScaleDev.call(b, g), ScaleDev.call(a, g)


class ScaleDev(Function):
    """x * s with s a 0-d device tensor."""

    @staticmethod
    def forward(ctx, x, s):
        x, s = _req(x), _req(s)
        ctx.save_for_backward(x, s)
        out = torch.empty_like(x)
        _chk(_L().ix_scale_dev_f32(x.data_ptr(), s.data_ptr(), out.data_ptr(), x.numel(), _stream()), "ix_scale_dev_f32")
        return out

    @staticmethod
    def backward(ctx, g):
        x, s = ctx.saved_tensors
        return ScaleDev.call(g, s), Dot.call(g, x)


def l2_norm(x):
    """torch.norm(x): sqrt(sum x^2).  The 1-element sqrt stays a torch scalar op (plumbing)."""
    return torch.sqrt(Dot.call(x, x))


class RowNormSum(Function):
    """sum_e ||x_e||_2 over the rows of x [E, n] -> 0-d tensor, ONE launch (the learned loss of a chunk of episodes: reference
    models/interactron.py:96 per task).  Closed under the differentiation MAML needs: its backward is RowNormSumBwd, whose
    own backward is one more kernel."""
    b16_out = False   # (a scalar)

    @staticmethod
    def forward(ctx, x):
        x = _req(x)
        E, n = x.shape
        norms = torch.empty(E, device=x.device, dtype=torch.float32)
        total = torch.empty((), device=x.device, dtype=torch.float32)
        _chk(_L().ix_rownorm_sum_f32(x.data_ptr(), norms.data_ptr(), total.data_ptr(), E, n, _stream()), "ix_rownorm_sum_f32")
        ctx.save_for_backward(x, norms)
        return total

    @staticmethod
    def backward(ctx, g):
        x, norms = ctx.saved_tensors
        return RowNormSumBwd.call(x, norms, g)


class RowNormSumBwd(Function):
    """y = g x_e / ||x_e|| (g a 0-d tensor); norms are a function of x kept as a constant operand: the backward below carries
    their derivative (the - x <H, x> / n^3 term)."""

    @staticmethod
    def forward(ctx, x, norms, g):
        x, g = _req(x), _req(g)
        ctx.save_for_backward(x, norms, g)
        out = torch.empty_like(x)
        _chk(_L().ix_rownorm_sum_bwd_f32(x.data_ptr(), norms.data_ptr(), g.data_ptr(), out.data_ptr(), x.shape[0], x.shape[1],
                                         _stream()), "ix_rownorm_sum_bwd_f32")
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, H):
        x, norms, g = ctx.saved_tensors
        H = _req(H)
        Gx, Gg = torch.empty_like(x), torch.empty((), device=x.device, dtype=torch.float32)
        _chk(_L().ix_rownorm_sum_bwd_bwd_f32(x.data_ptr(), norms.data_ptr(), g.data_ptr(), H.data_ptr(), Gx.data_ptr(),
                                             Gg.data_ptr(), x.shape[0], x.shape[1], _stream()), "ix_rownorm_sum_bwd_bwd_f32")
        return Gx, None, Gg


def rownorm_sum(x):
    """sum of the L2 norms of the rows of x [E, n]"""
    return RowNormSum.call(x)


class Relu(Function):
    @staticmethod
    def forward(ctx, x):
        x = _req(x)
        out = torch.empty_like(x)
        _chk(_L().ix_relu_f32(x.data_ptr(), out.data_ptr(), x.numel(), _stream()), "ix_relu_f32")
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        return ReluBwd.call(g, y)


class ReluBwd(Function):
    """dy * [y > 0]; linear in dy, piecewise constant in y."""

    @staticmethod
    def forward(ctx, dy, y):
        dy, y = _req(dy), _req(y)
        ctx.save_for_backward(y)
        out = torch.empty_like(dy)
        _chk(_L().ix_relu_bwd_f32(dy.data_ptr(), y.data_ptr(), out.data_ptr(), dy.numel(), _stream()), "ix_relu_bwd_f32")
        return out

    @staticmethod
    def backward(ctx, G):
        (y,) = ctx.saved_tensors
        return ReluBwd.call(G, y), None


class ReluBwdSum(Function):
    """(ga + gb) * [y > 0]: ReluBwd of an activation with two consumers, the sum of their gradients in the same pass."""

    @staticmethod
    def forward(ctx, ga, gb, y):
        ga, gb, y = _req(ga), _req(gb), _req(y)
        ctx.save_for_backward(y)
        out = torch.empty_like(ga)
        _chk(_L().ix_relu_bwd_sum_f32(ga.data_ptr(), gb.data_ptr(), y.data_ptr(), out.data_ptr(), ga.numel(), _stream()),
             "ix_relu_bwd_sum_f32")
        return out

    @staticmethod
    def backward(ctx, G):
        (y,) = ctx.saved_tensors
        t = ReluBwd.call(G, y)
        return t, t, None


def _two_gradients(gs, y, relu):
    """the gradient(s) of a fused contraction + BN node's output(s) -> the ReLU-masked (if relu) single gradient.  Two outputs
    (fan = 2: the node handed out two aliases of its result) with two gradients: summed inside the ReLU derivative's pass."""
    gs = [g for g in gs if g is not None]
    if len(gs) == 2:
        if relu:
            return ReluBwdSum.call(gs[0].contiguous(), gs[1].contiguous(), y), True
        return SumN.call(gs[0], gs[1]), False
    return gs[0].contiguous(), False


class ReluBwdScaled(Function):
    """dy * [y > 0] * scale; linear in dy."""

    @staticmethod
    def forward(ctx, dy, y, scale):
        dy, y = _req(dy), _req(y)
        ctx.save_for_backward(y)
        ctx.scale = scale
        out = torch.empty_like(dy)
        _chk(_L().ix_relu_bwd_scaled_f32(dy.data_ptr(), y.data_ptr(), out.data_ptr(), dy.numel(), scale, _stream()),
             "ix_relu_bwd_scaled_f32")
        return out

    @staticmethod
    def backward(ctx, G):
        (y,) = ctx.saved_tensors
        return ReluBwdScaled.call(G, y, ctx.scale), None, None


class ReluDropout(Function):
    """dropout(relu(x)) as one pass; y > 0 exactly where the relu and the mask both pass, so the backward is one pass
    over (dy, y) with neither the mask hash nor x."""

    @staticmethod
    def forward(ctx, x, p, seed):
        x = _req(x)
        out = torch.empty_like(x)
        _chk(_L().ix_relu_dropout_f32(x.data_ptr(), out.data_ptr(), x.numel(), p, seed, _stream()), "ix_relu_dropout_f32")
        ctx.save_for_backward(out)
        ctx.scale = 1.0 / (1.0 - p)
        return out

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        return ReluBwdScaled.call(g, y, ctx.scale), None, None


class AddDropout(Function):
    """x + dropout(a) as one pass (residual connections)."""

    @staticmethod
    def forward(ctx, x, a, p, seed):
        x, a = _req(x), _req(a)
        assert x.shape == a.shape, (x.shape, a.shape)
        ctx.p, ctx.seed = p, seed
        out = torch.empty_like(x)
        _chk(_L().ix_add_dropout_f32(x.data_ptr(), a.data_ptr(), out.data_ptr(), x.numel(), p, seed, _stream()),
             "ix_add_dropout_f32")
        return out

    @staticmethod
    def backward(ctx, g):
        if ctx.needs_input_grad[0] and ctx.needs_input_grad[1] and torch.is_grad_enabled() and g.requires_grad:
            gx, ga = _AddDropoutBwd.apply(g, ctx.p, ctx.seed)   # (recorded: its own backward is ONE add_dropout pass)
            return gx, ga, None, None
        return (g if ctx.needs_input_grad[0] else None), \
            (_Dropout.call(g, ctx.p, ctx.seed) if ctx.needs_input_grad[1] else None), None, None


class _AddDropoutBwd(Function):
    """g -> (g, dropout(g)): AddDropout's backward as one node, so that the gradient of g in the outer backward is
    G_x + dropout(G_a) in one pass (add_dropout) instead of a dropout pass and an autograd sum."""

    @staticmethod
    def forward(ctx, g, p, seed):
        ctx.set_materialize_grads(False)
        ctx.p, ctx.seed = p, seed
        g = _req(g)
        return g.view_as(g), _Dropout.forward(_NullCtx(), g, p, seed)

    @staticmethod
    def backward(ctx, Gx, Ga):
        if Gx is None and Ga is None:
            return None, None, None
        if Ga is None:
            return Gx, None, None
        if Gx is None:
            return _Dropout.call(Ga.contiguous(), ctx.p, ctx.seed), None, None
        return AddDropout.call(Gx.contiguous(), Ga.contiguous(), ctx.p, ctx.seed), None, None


def add_dropout(x, a, p, training):
    if not training or p <= 0.0:
        return add(x, a)
    return AddDropout.call(x, a, float(p), core._next_seed())


def relu_dropout(x, p, training):
    if not training or p <= 0.0:
        return Relu.call(x)
    return ReluDropout.call(x, float(p), core._next_seed())


class Gelu(Function):
    @staticmethod
    def forward(ctx, x):
        x = _req(x)
        ctx.save_for_backward(x)
        out = torch.empty_like(x)
        _chk(_L().ix_gelu_f32(x.data_ptr(), out.data_ptr(), x.numel(), _stream()), "ix_gelu_f32")
        return out

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return GeluBwd.call(g, x)


class GeluBwd(Function):
    @staticmethod
    def forward(ctx, dy, x):
        dy, x = _req(dy), _req(x)
        ctx.save_for_backward(dy, x)
        out = torch.empty_like(dy)
        _chk(_L().ix_gelu_bwd_f32(dy.data_ptr(), x.data_ptr(), out.data_ptr(), dy.numel(), _stream()), "ix_gelu_bwd_f32")
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, G):
        dy, x = ctx.saved_tensors
        G = _req(G)
        gdy, gx = torch.empty_like(dy), torch.empty_like(x)
        _chk(_L().ix_gelu_bwd_bwd_f32(G.data_ptr(), dy.data_ptr(), x.data_ptr(), gdy.data_ptr(), gx.data_ptr(),
                                      dy.numel(), _stream()), "ix_gelu_bwd_bwd_f32")
        return gdy, gx


class Sigmoid(Function):
    @staticmethod
    def forward(ctx, x):
        x = _req(x)
        out = torch.empty_like(x)
        _chk(_L().ix_sigmoid_f32(x.data_ptr(), out.data_ptr(), x.numel(), _stream()), "ix_sigmoid_f32")
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        return SigmoidBwd.call(g, y)


class SigmoidBwd(Function):
    @staticmethod
    def forward(ctx, dy, y):
        dy, y = _req(dy), _req(y)
        ctx.save_for_backward(dy, y)
        out = torch.empty_like(dy)
        _chk(_L().ix_sigmoid_bwd_f32(dy.data_ptr(), y.data_ptr(), out.data_ptr(), dy.numel(), _stream()),
             "ix_sigmoid_bwd_f32")
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, G):
        dy, y = ctx.saved_tensors
        G = _req(G)
        gdy, gy = torch.empty_like(dy), torch.empty_like(y)
        _chk(_L().ix_sigmoid_bwd_bwd_f32(G.data_ptr(), dy.data_ptr(), y.data_ptr(), gdy.data_ptr(), gy.data_ptr(),
                                         dy.numel(), _stream()), "ix_sigmoid_bwd_bwd_f32")
        return gdy, gy


class _Dropout(Function):
    @staticmethod
    def forward(ctx, x, p, seed):
        x = _req(x)
        ctx.p, ctx.seed = p, seed
        out = torch.empty_like(x)
        _chk(_L().ix_dropout_f32(x.data_ptr(), out.data_ptr(), x.numel(), p, seed, _stream()), "ix_dropout_f32")
        return out

    @staticmethod
    def backward(ctx, g):
        return _Dropout.call(g, ctx.p, ctx.seed), None, None


def dropout(x, p, training):
    if not training or p <= 0.0:
        return x
    return _Dropout.call(x, float(p), core._next_seed())


# ---------------------------------------------------------------------------------------------------------
# softmax / LayerNorm
# ---------------------------------------------------------------------------------------------------------
class Softmax(Function):
    """softmax over the first `length` entries of the last dim (row pitch = last dim size); optional uint8
    key-padding mask [nmask, length] with `rows_per_mask` consecutive rows sharing one mask row."""

    @staticmethod
    def forward(ctx, x, length, mask, rows_per_mask):
        x = _req(x)
        ld = x.shape[-1]
        rows = x.numel() // ld
        y = torch.empty_like(x) if ld == length else torch.zeros_like(x)
        _chk(_L().ix_softmax_fwd_f32(x.data_ptr(), y.data_ptr(), rows, length, ld,
                                     mask.data_ptr() if mask is not None else None, rows_per_mask,
                                     mask.shape[-1] if mask is not None else 0, _stream()), "ix_softmax_fwd_f32")
        ctx.length = length
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        return SoftmaxBwd.call(y, g, ctx.length), None, None, None


class SoftmaxBwd(Function):
    @staticmethod
    def forward(ctx, y, dy, length):
        y, dy = _req(y), _req(dy)
        ld = y.shape[-1]
        dx = torch.empty_like(y) if ld == length else torch.zeros_like(y)
        _chk(_L().ix_softmax_bwd_f32(y.data_ptr(), dy.data_ptr(), dx.data_ptr(), y.numel() // ld, length, ld, _stream()),
             "ix_softmax_bwd_f32")
        ctx.length = length
        ctx.save_for_backward(y, dy)
        return dx

    @staticmethod
    @once_differentiable
    def backward(ctx, G):
        y, dy = ctx.saved_tensors
        G = _req(G)
        ld = y.shape[-1]
        alloc = torch.empty_like if ld == ctx.length else torch.zeros_like
        gy, gdy = alloc(y), alloc(y)
        _chk(_L().ix_softmax_bwd_bwd_f32(G.data_ptr(), y.data_ptr(), dy.data_ptr(), gy.data_ptr(), gdy.data_ptr(),
                                         y.numel() // ld, ctx.length, ld, _stream()), "ix_softmax_bwd_bwd_f32")
        return gy, gdy, None


class LayerNorm(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        x, gamma, beta = _req(x), _req(gamma), _req(beta)
        D = x.shape[-1]
        G = gamma.shape[0] if gamma.dim() == 2 else 1      # per-episode affine [G, D]: x is [G * rows, D]
        rows = x.numel() // D
        assert rows % G == 0
        y = torch.empty_like(x)
        mean = torch.empty(rows, device=x.device, dtype=torch.float32)
        rstd = torch.empty_like(mean)
        _chk(_L().ix_layernorm_fwd_f32(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), mean.data_ptr(),
                                       rstd.data_ptr(), rows // G, D, eps, G, _stream()), "ix_layernorm_fwd_f32")
        ctx.save_for_backward(x, gamma, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, g):
        x, gamma, mean, rstd = ctx.saved_tensors
        dx, dgamma, dbeta = LayerNormBwd.call(g, x, gamma, mean, rstd)
        return dx, dgamma, dbeta, None


class LayerNormBwd(Function):
    """(dy, x, gamma) -> (dx, dgamma, dbeta); mean/rstd are recomputable statistics of x (handled analytically)."""
    b16_out = (True, False, False)   # (dx is an activation; the parameter gradients stay fp32)

    @staticmethod
    def forward(ctx, dy, x, gamma, mean, rstd):
        dy = _req(dy)
        D = x.shape[-1]
        G = gamma.shape[0] if gamma.dim() == 2 else 1
        rows = x.numel() // D
        dx = torch.empty_like(x)
        both = torch.empty((2,) + tuple(gamma.shape), device=gamma.device, dtype=torch.float32)   # (one fill for the two)
        dgamma, dbeta = both[0], both[1]
        wp, wn = _reduce_ws("ln", rows // G, D, G, x.device)
        _chk(_L().ix_layernorm_bwd_f32(dy.data_ptr(), x.data_ptr(), gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                       dx.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), rows // G, D, G, wp, wn, _stream()),
             "ix_layernorm_bwd_f32")
        ctx.save_for_backward(dy, x, gamma, mean, rstd)
        return dx, dgamma, dbeta

    @staticmethod
    @once_differentiable
    def backward(ctx, Gx, Gg, Gb):
        dy, x, gamma, mean, rstd = ctx.saved_tensors
        D = x.shape[-1]
        G = gamma.shape[0] if gamma.dim() == 2 else 1
        rows = x.numel() // D
        Gx = _req(Gx) if Gx is not None else None
        Gg = _req(Gg) if Gg is not None else None
        Gb = _req(Gb) if Gb is not None else None
        gdy, gx = torch.empty_like(x), torch.empty_like(x)
        ggamma = torch.empty_like(gamma)
        wp, wn = _reduce_ws("ln", rows // G, D, G, x.device)
        _chk(_L().ix_layernorm_bwd_bwd_f32(Gx.data_ptr() if Gx is not None else None,
                                           Gg.data_ptr() if Gg is not None else None,
                                           Gb.data_ptr() if Gb is not None else None, dy.data_ptr(), x.data_ptr(),
                                           gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gdy.data_ptr(),
                                           gx.data_ptr(), ggamma.data_ptr(), rows // G, D, G, wp, wn, _stream()),
             "ix_layernorm_bwd_bwd_f32")
        return gdy, gx, ggamma, None, None


def layer_norm(x, gamma, beta, eps=1e-5):
    return LayerNorm.call(x, gamma, beta, eps)
