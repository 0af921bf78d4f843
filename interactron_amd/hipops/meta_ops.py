"""hipops.meta_ops -- multi-tensor MAML plumbing (episode expansion, clipped SGD) and the fused outer step."""
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
from .core import (Function, _L, _chk, _reduce_ws, _req, _stream, weights_changed)


# ---------------------------------------------------------------------------------------------------------
# MAML fast weights and the outer step
# ---------------------------------------------------------------------------------------------------------
def _ptr_array(tensors):
    arr = (ctypes.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr() if t is not None else None
    return arr


def _size_array(tensors):
    arr = (ctypes.c_int64 * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = t.numel() if t is not None else 0
    return arr


def copy_multi(srcs, dsts):
    """dst_i = src_i for a list of fp32 tensor pairs in one multi-tensor launch set (ix_reduce_multi_f32 over ONE copy): the gradients
    autograd left on the parameters into their slots of the flat gradient buffer (trainer.FlatOuterStep, models without step graphs)"""
    if not srcs:
        return
    srcs = [_req(t) for t in srcs]
    assert all(a.numel() == b.numel() and a.is_contiguous() and b.is_contiguous() and b.dtype == torch.float32 for a, b in zip(srcs, dsts))
    _chk(_L().ix_reduce_multi_f32(_ptr_array(srcs), _ptr_array(dsts), _size_array(dsts), len(srcs), 1, _stream()), "ix_reduce_multi_f32")


class ExpandEpisodes(Function):
    """apply(E, p_1..p_n) -> ([E, *p_1.shape], ..): the per-episode copies of a parameter list in one multi-tensor launch
    set (199 single-tensor launches before).  Backward = ReduceEpisodes: the sum over the E copies, i.e. the reference's
    gradient accumulation over the tasks of a batch, in a fixed order (no atomics)."""

    @staticmethod
    def forward(ctx, E, *ps):
        pc = [_req(p) for p in ps]
        outs = [torch.empty((E,) + tuple(p.shape), device=p.device, dtype=torch.float32) for p in pc]
        _chk(_L().ix_expand_multi_f32(_ptr_array(pc), _ptr_array(outs), _size_array(pc), len(pc), E, _stream()),
             "ix_expand_multi_f32")
        ctx.E = E
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        idx = [i for i, g in enumerate(gs) if g is not None and ctx.needs_input_grad[1 + i]]
        res = [None] * len(gs)
        if idx:
            for i, r in zip(idx, ReduceEpisodes.call(ctx.E, *[gs[i] for i in idx])):
                res[i] = r
        return (None,) + tuple(res)


def expand_episodes(E, tensors, groups=8):
    """ExpandEpisodes over consecutive groups of the parameter list (the detector's stages in module order) instead of one
    node for all ~200 tensors: a group's per-episode gradients are reduced and RELEASED as soon as the backward has passed
    its stage, instead of all E-copy gradients staying alive until the end of the backward (E x |theta| x 4 bytes of peak
    memory at 800x800)."""
    tensors = list(tensors)
    n = max(1, (len(tensors) + groups - 1) // groups)
    out = []
    for i in range(0, len(tensors), n):
        out.extend(ExpandEpisodes.apply(E, *tensors[i:i + n]))
    return out


class ReduceEpisodes(Function):
    """apply(E, g_1..g_n) with g_i [E, ...] -> (sum over the leading dim, ..) in one multi-tensor launch set."""

    @staticmethod
    def forward(ctx, E, *gs):
        gc = [_req(g) for g in gs]
        outs = [torch.empty(tuple(g.shape[1:]), device=g.device, dtype=torch.float32) for g in gc]
        _chk(_L().ix_reduce_multi_f32(_ptr_array(gc), _ptr_array(outs), _size_array(outs), len(gc), E, _stream()),
             "ix_reduce_multi_f32")
        ctx.E = E
        return tuple(outs)

    @staticmethod
    def backward(ctx, *hs):
        idx = [i for i, h in enumerate(hs) if h is not None and ctx.needs_input_grad[1 + i]]
        res = [None] * len(hs)
        if idx:
            for i, r in zip(idx, ExpandEpisodes.call(ctx.E, *[hs[i] for i in idx])):
                res[i] = r
        return (None,) + tuple(res)


class ClippedSGD(Function):
    """fast_i = p_i - clamp(lr*g_i, +-clip) for all tensors in one multi-tensor launch set.

    apply(lr, clip, n, p_1..p_n, g_1..g_n) -> (fast_1..fast_n).  g_i may be None (tensor passes through)."""

    @staticmethod
    def forward(ctx, lr, clip, n, *tensors):
        ps, gs = tensors[:n], tensors[n:]
        idx = [i for i in range(n) if gs[i] is not None]
        pc = [_req(ps[i]) for i in idx]
        gc = [_req(gs[i]) for i in idx]
        outs = [torch.empty_like(p) for p in pc]
        if idx:
            _chk(_L().ix_sgd_clip_multi_f32(_ptr_array(pc), _ptr_array(gc), _ptr_array(outs), _size_array(pc), len(pc),
                                            lr, clip, _stream()), "ix_sgd_clip_multi_f32")
        ctx.lr, ctx.clip, ctx.n, ctx.idx = lr, clip, n, idx
        ctx.save_for_backward(*gc)
        res = list(ps)
        for j, i in enumerate(idx):
            res[i] = outs[j]
        # pass-through tensors must not alias the inputs for autograd
        return tuple(r if i in set(idx) else r.view_as(r) for i, r in enumerate(res))

    @staticmethod
    def backward(ctx, *G):
        gs = ctx.saved_tensors
        n, idx = ctx.n, ctx.idx
        grad_p = [G[i] if ctx.needs_input_grad[3 + i] else None for i in range(n)]
        grad_g = [None] * n
        need = [j for j, i in enumerate(idx) if ctx.needs_input_grad[3 + n + i] and G[i] is not None]
        if need:
            res = _ClippedSGDBwd.call(ctx.lr, ctx.clip, len(need), *([G[idx[j]] for j in need] + [gs[j] for j in need]))
            for j, r in zip(need, res):
                grad_g[idx[j]] = r
        return (None, None, None) + tuple(grad_p) + tuple(grad_g)


class _ClippedSGDBwd(Function):
    """out_i = -lr * G_i * [|lr*g_i| <= clip]."""

    @staticmethod
    def forward(ctx, lr, clip, n, *tensors):
        Gs = [_req(t) for t in tensors[:n]]
        gs = [_req(t) for t in tensors[n:]]
        outs = [torch.empty_like(t) for t in Gs]
        _chk(_L().ix_sgd_clip_bwd_multi_f32(_ptr_array(Gs), _ptr_array(gs), _ptr_array(outs), _size_array(Gs), n, lr,
                                            clip, _stream()), "ix_sgd_clip_bwd_multi_f32")
        ctx.lr, ctx.clip, ctx.n = lr, clip, n
        ctx.save_for_backward(*gs)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *GG):
        # linear in G (the indicator is piecewise constant in g)
        gs = ctx.saved_tensors
        n = ctx.n
        res = _ClippedSGDBwd.call(ctx.lr, ctx.clip, n, *(list(GG) + list(gs)))
        return (None, None, None) + tuple(res) + (None,) * n


def accumulate_multi(dst, src):
    """dst_i += src_i for two lists of tensors in ONE multi-tensor launch (the clipped-SGD kernel with lr = -1 and no clip:
    p - clamp(-g) = p + g)."""
    dst, src = [_req(d) for d in dst], [_req(s) for s in src]
    if dst:
        _chk(_L().ix_sgd_clip_multi_f32(_ptr_array(dst), _ptr_array(src), _ptr_array(dst), _size_array(dst), len(dst), -1.0, 3.0e38,
                                        _stream()), "ix_sgd_clip_multi_f32")


def sumsq_accum(x_flat, out_scalar):
    wp, wn = _reduce_ws("scalar", 0, 0, 0, x_flat.device)
    _chk(_L().ix_sumsq_accum_f32(x_flat.data_ptr(), x_flat.numel(), out_scalar.data_ptr(), wp, wn, _stream()), "ix_sumsq_accum_f32")


def adam_step(p, g, m, v, lr, beta1, beta2, eps, step, sumsq=None, max_norm=0.0, zero_grad=False):
    _chk(_L().ix_adam_step_f32(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), lr, beta1, beta2, eps,
                               step, sumsq.data_ptr() if sumsq is not None else None, max_norm, 1 if zero_grad else 0,
                               _stream()), "ix_adam_step_f32")
    weights_changed()   # (raw-pointer update: no autograd version moves; cached weight planes are void)
