"""The 16-bit activation mode (``MODEL.COMPUTE_DTYPE: bf16`` -- BASELINE.json configs[1] "multi_frame_baseline ... bf16").

Activations live in HBM as ``torch.bfloat16`` tensors; parameters, their gradients, optimiser state, LayerNorm / softmax statistics
and every accumulation stay fp32 (the reference computes everything in fp32: models/gpt.py:39-78, models/detr_models/
transformer.py:148-232, backbone.py:88-90 -- this mode is checked at SURVEY 8d's bf16 tolerances, never at the fp32 ones).

How an op meets a bf16 tensor (``hipops.Function.b16``):
  * ``"native"``  -- its forward takes bf16 tensors and launches the 16-bit kernels (csrc/gemm16.hip: both operand tiles HBM -> LDS
                    by LDS-DMA, one bf16 matrix instruction per k-slice; the elementwise / row kernels' ``_b16`` entry points);
  * ``"adapt"``   -- (default) the op is computed by its fp32 kernels between two conversion passes, ``ToF32`` on the way in and
                    ``ToB16`` on the way out.  Both are autograd Functions whose backward is the other one, so an adapted op stays
                    closed under differentiation, and "compute in fp32, store as bf16" is at least as accurate as a native kernel.
                    It costs two extra passes: every op that matters to the step time is native, the adapter is the net under the
                    rest.
Nothing here falls back to ATen or to the CPU: the conversions are HIP kernels of this library.
"""
import ctypes

import torch
from torch.autograd.function import once_differentiable

from . import hipops as ops
from ._lib import HipLibraryError

import os

B16 = torch.bfloat16
STEM_LAYER1 = os.environ.get("IX_B16_STEM", "1") == "1"   # "0": the frozen layer1 stays on the fp32 kernels, the cast follows it (A/B runs)
_seen = ops._b16_seen   # [False] until this module is imported -- by the first model that runs in the mode, or by a test: the fp32 path's
_seen[0] = True         # per-call cost stays at one list read in processes that never use 16-bit activations


def is_b16(t):
    return torch.is_tensor(t) and t.dtype == B16


def _reqd(t, name="tensor"):
    """like hipops._req, for a bf16 OR fp32 tensor"""
    if t.dtype == torch.float32:
        return ops._req(t, name)
    if t.dtype != B16:
        raise TypeError("%s must be float32 or bfloat16, got %s" % (name, t.dtype))
    if not t.is_cuda:
        raise HipLibraryError("%s must live on the GPU: the HIP path has no CPU fallback" % name)
    return t if t.is_contiguous() else t.contiguous()


def cast_b16(x):
    """fp32 -> bf16 (round to nearest even), one HIP pass"""
    x = ops._req(x)
    y = torch.empty(x.shape, dtype=B16, device=x.device)
    ops._chk(ops._L().ix_cast_f32_b16(x.data_ptr(), y.data_ptr(), x.numel(), ops._stream()), "ix_cast_f32_b16")
    _seen[0] = True
    return y


def cast_b16_into(x, y):
    """fp32 -> bf16 into an existing tensor of the same element count (the flat parameter shadow, trainer.FlatBuffers.sync_b16)"""
    x = ops._req(x)
    assert y.dtype == B16 and y.numel() == x.numel() and y.is_contiguous() and x.is_contiguous()
    ops._chk(ops._L().ix_cast_f32_b16(x.data_ptr(), y.data_ptr(), x.numel(), ops._stream()), "ix_cast_f32_b16")
    _seen[0] = True
    return y


def cast_f32(x):
    x = _reqd(x)
    if x.dtype == torch.float32:
        return x
    y = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    ops._chk(ops._L().ix_cast_b16_f32(x.data_ptr(), y.data_ptr(), x.numel(), ops._stream()), "ix_cast_b16_f32")
    return y


class ToB16(ops.Function):
    """fp32 -> bf16; the gradient comes back as fp32"""
    b16 = "native"

    @staticmethod
    def forward(ctx, x):
        return cast_b16(x) if x.dtype == torch.float32 else x

    @staticmethod
    def backward(ctx, g):
        return ToF32.call(g)


class ToF32(ops.Function):
    """bf16 -> fp32; the gradient goes back as bf16"""
    b16 = "native"

    @staticmethod
    def forward(ctx, x):
        return cast_f32(x)

    @staticmethod
    def backward(ctx, g):
        return ToB16.call(g)


def to_b16(x):
    return x if x.dtype == B16 else ToB16.call(x)


def to_f32(x):
    return x if x.dtype == torch.float32 else ToF32.call(x)


def adapt(cls, args, run):
    """An op without 16-bit kernels on bf16 inputs: fp32 kernels between conversion passes (see the module docstring).
    `run(args)` applies the op (recorded or not, the caller decides)."""
    done = {}   # (one conversion per distinct tensor: the packed [k | q | v] buffer is passed three times)
    up = []
    for a in args:
        if is_b16(a):
            if id(a) not in done:
                done[id(a)] = ToF32.call(a)
            a = done[id(a)]
        up.append(a)
    out = run(tuple(up))
    keep = cls.b16_out
    if keep is False:
        return out
    if torch.is_tensor(out):
        return ToB16.call(out) if out.dtype == torch.float32 and out.is_floating_point() else out
    res = []
    for i, o in enumerate(out):
        down = torch.is_tensor(o) and o.dtype == torch.float32 and (keep is True or (i < len(keep) and keep[i]))
        res.append(ToB16.call(o) if down else o)
    return tuple(res)


# ---- weights as bf16 -------------------------------------------------------------------------------------------------------------
# A parameter stays fp32 (master copy, optimiser state, gradient); the contraction reads a bf16 copy made once per (tensor, version,
# address, epoch) -- the key of the weight-planes cache of hipops, bumped by everything that rewrites parameters behind autograd's
# back (fused Adam, FlatBuffers, load_state_dict).  Inside a HIP-graph capture the cache is the capture's own: the cast is part of the
# graph, replays redo it (weights change between replays while their addresses stay).
_stats = {"weight_casts": 0, "native_gemms": 0, "fallback_gemms": 0}


def weight_b16(w):
    if w.dtype == B16:
        return w
    fl = w.__dict__.get("_ix_b16_flat")   # a trainable parameter's view of the flat bf16 shadow (trainer.FlatBuffers.sync_b16)
    if fl is not None and fl[1] == ops._wp_epoch[0] and fl[2] == w._version and fl[3] == w.data_ptr():
        return fl[0]
    cap = ops._capture[0]
    if cap is not None:
        store, tag = cap, ("b16", ops._scratch_slot[0], id(w), ops._wp_epoch[0])
    else:
        # (the epoch moves with every raw-pointer update of the TRAINABLE parameters -- the optimiser step; a frozen parameter changes only
        #  through copy_ / load_state_dict, which its version counter sees: it is not converted again after every step)
        frozen = isinstance(w, torch.nn.Parameter) and not w.requires_grad
        store, tag = w.__dict__.setdefault("_ix_b16", {}), (-1 if frozen else ops._wp_epoch[0])
        if len(store) > 2:
            store.clear()
    hit = store.get(tag)
    if hit is not None and hit[1] == w._version and hit[2] == w.data_ptr():
        return hit[0]
    with torch.no_grad():
        c = cast_b16(w.detach())
    store[tag] = (c, w._version, w.data_ptr(), w if cap is not None else None)
    _stats["weight_casts"] += 1
    return c


def _as_b16(t):
    """operand of a 16-bit contraction: bf16 as it is; an fp32 weight through the cache; any other fp32 tensor by a cast pass"""
    if t.dtype == B16:
        return t
    if getattr(t, "_ix_weight", False) or isinstance(t, torch.nn.Parameter):
        return weight_b16(t)
    return cast_b16(t)


def run_gemm(a, b, bias, sp, fill=True, scale=None, shift=None, residual=None, act=0):
    """hipops._run_gemm for a contraction with at least one bf16 operand: C in sp.odt (default bf16).  Operands whose rows are not
    16-byte aligned (the 1236-class head's gradient: ld 1236) take the fp32 kernels between conversion passes."""
    out_f32 = sp.odt == torch.float32
    covered = not fill or sp.bo * sp.bi * sp.M * sp.N == ops._numel(sp.out_shape)
    L = ops._L()
    a_kc, b_kc = (0 if sp.A.trans else 1), (1 if sp.B.trans else 0)
    ok = not sp.C.trans and L.ix_gemm_b16_supported(
        (a.data_ptr() if a.dtype == B16 else 0) + 2 * sp.A.offset, (b.data_ptr() if b.dtype == B16 else 0) + 2 * sp.B.offset,
        (4 if out_f32 else 2) * sp.C.offset, sp.M, sp.N, sp.K, a_kc, b_kc, sp.A.ld, sp.B.ld, sp.C.ld, sp.A.so, sp.A.si, sp.B.so, sp.B.si,
        sp.C.so, sp.C.si) == 1
    if bias is not None and (bias.data_ptr() & 15 or (bias.dim() == 2 and sp.N % 4)):
        ok = False
    if not ok:
        _stats["fallback_gemms"] += 1
        o = ops._run_gemm(cast_f32(a), cast_f32(b), bias, sp._replace(odt=None), fill)
        if scale is not None or residual is not None or act:
            assert scale is not None and act in (0, 1), "only the frozen-BN affine (+ residual) (+ ReLU) has an fp32 twin here"
            o = ops._channel_affine(o, scale, shift, cast_f32(residual) if residual is not None else None, act == 1)
        return o if out_f32 else cast_b16(o)
    a16, b16_ = _as_b16(a), _as_b16(b)
    out = (torch.empty if covered else torch.zeros)(sp.out_shape, device=a.device, dtype=torch.float32 if out_f32 else B16)
    nb = sp.bo * sp.bi
    key = ("b16", sp.M, sp.N, sp.K, nb)
    nws = ops._ws_bytes.get(key)
    if nws is None:
        n = ctypes.c_size_t()
        ops._chk(L.ix_workspace_bytes_gemm_b16(sp.M, sp.N, sp.K, nb, ctypes.byref(n)), "ix_workspace_bytes_gemm_b16")
        nws = ops._ws_bytes[key] = n.value
    ws = ops._workspace(nws, a.device) if nws else None
    esz = 4 if out_f32 else 2
    ops._chk(L.ix_gemm_b16(a16.data_ptr() + 2 * sp.A.offset, b16_.data_ptr() + 2 * sp.B.offset, out.data_ptr() + esz * sp.C.offset,
                           bias.data_ptr() if bias is not None else None, sp.M, sp.N, sp.K, a_kc, b_kc, sp.A.ld, sp.B.ld, sp.C.ld,
                           sp.bo, sp.bi, sp.A.so, sp.A.si, sp.B.so, sp.B.si, sp.C.so, sp.C.si,
                           sp.N if (bias is not None and bias.dim() == 2) else 0, sp.alpha, 1 if out_f32 else 0,
                           scale.data_ptr() if scale is not None else None, shift.data_ptr() if shift is not None else None,
                           residual.data_ptr() if residual is not None else None, act,
                           ws.data_ptr() if nws else None, nws, ops._stream()), "ix_gemm_b16")
    _stats["native_gemms"] += 1
    _seen[0] = True
    return out


# ---- native 16-bit forwards ("twins") --------------------------------------------------------------------------------------------
# A twin subclasses the fp32 Function, inherits its backward (which is written with Functions that dispatch on dtype again) and
# replaces the forward by the bf16 kernel of csrc/ew16.hip.  hipops.Function.apply / call send calls with bf16 tensors here.
def _c16(t):
    """an operand of a 16-bit elementwise op: bf16 as it is, an fp32 tensor (a position table, a cotangent from an fp32 island) cast"""
    return _reqd(t) if t.dtype == B16 else cast_b16(t)


def _ptr(t):
    return t.data_ptr() if t is not None else None


def _map(op, a, b=None, c=None, p0=0.0, p1=0.0, seed=0):
    out = torch.empty(a.shape, dtype=B16, device=a.device)
    ops._chk(ops._L().ix_map_b16(op, a.data_ptr(), _ptr(b), _ptr(c), out.data_ptr(), a.numel(), p0, p1, seed, ops._stream()), "ix_map_b16")
    return out


def _channel(op, x, y, scale, shift, relu=False, groups=1):
    C = scale.numel() // groups
    out = torch.empty(x.shape, dtype=B16, device=x.device)
    ops._chk(ops._L().ix_channel_b16(op, x.data_ptr(), _ptr(y), scale.data_ptr(), _ptr(shift), out.data_ptr(), x.numel() // C, C,
                                     1 if relu else 0, groups, ops._stream()), "ix_channel_b16")
    return out


def _twin(base):
    def deco(cls):
        cls.b16 = "native"
        base.b16_twin = cls
        return cls
    return deco


@_twin(ops.Axpby)
class Axpby16(ops.Axpby):
    @staticmethod
    def forward(ctx, a, b, alpha, beta):
        a, b = _c16(a), _c16(b)
        assert a.shape == b.shape, (a.shape, b.shape)
        ctx.alpha, ctx.beta = alpha, beta
        return _map(0, a, b) if alpha == 1.0 and beta == 1.0 else _map(1, a, b, None, alpha, beta)


@_twin(ops.Scale)
class Scale16(ops.Scale):
    @staticmethod
    def forward(ctx, x, alpha):
        ctx.alpha = alpha
        return _map(2, _c16(x), None, None, alpha)


@_twin(ops.Relu)
class Relu16(ops.Relu):
    @staticmethod
    def forward(ctx, x):
        out = _map(3, _c16(x))
        ctx.save_for_backward(out)
        return out


@_twin(ops.ReluBwd)
class ReluBwd16(ops.ReluBwd):
    @staticmethod
    def forward(ctx, dy, y):
        dy, y = _c16(dy), _c16(y)
        ctx.save_for_backward(y)
        return _map(4, dy, y, None, 1.0)


@_twin(ops.ReluBwdSum)
class ReluBwdSum16(ops.ReluBwdSum):
    @staticmethod
    def forward(ctx, ga, gb, y):
        ga, gb, y = _c16(ga), _c16(gb), _c16(y)
        ctx.save_for_backward(y)
        return _map(5, ga, gb, y)


@_twin(ops.ReluBwdScaled)
class ReluBwdScaled16(ops.ReluBwdScaled):
    @staticmethod
    def forward(ctx, dy, y, scale):
        dy, y = _c16(dy), _c16(y)
        ctx.save_for_backward(y)
        ctx.scale = scale
        return _map(4, dy, y, None, scale)


@_twin(ops.ReluDropout)
class ReluDropout16(ops.ReluDropout):
    @staticmethod
    def forward(ctx, x, p, seed):
        out = _map(9, _c16(x), None, None, p, 0.0, seed)
        ctx.save_for_backward(out)
        ctx.scale = 1.0 / (1.0 - p)
        return out


@_twin(ops.AddDropout)
class AddDropout16(ops.AddDropout):
    @staticmethod
    def forward(ctx, x, a, p, seed):
        x, a = _c16(x), _c16(a)
        assert x.shape == a.shape, (x.shape, a.shape)
        ctx.p, ctx.seed = p, seed
        return _map(10, x, a, None, p, 0.0, seed)


@_twin(ops._Dropout)
class Dropout16(ops._Dropout):
    @staticmethod
    def forward(ctx, x, p, seed):
        ctx.p, ctx.seed = p, seed
        return _map(8, _c16(x), None, None, p, 0.0, seed)


@_twin(ops._AddDropoutBwd)
class AddDropoutBwd16(ops._AddDropoutBwd):
    @staticmethod
    def forward(ctx, g, p, seed):
        ctx.set_materialize_grads(False)
        ctx.p, ctx.seed = p, seed
        g = _c16(g)
        return g.view_as(g), _map(8, g, None, None, p, 0.0, seed)


@_twin(ops.Gelu)
class Gelu16(ops.Gelu):
    @staticmethod
    def forward(ctx, x):
        x = _c16(x)
        ctx.save_for_backward(x)
        return _map(6, x)


@_twin(ops.GeluBwd)
class GeluBwd16(ops.GeluBwd):
    @staticmethod
    def forward(ctx, dy, x):
        dy, x = _c16(dy), _c16(x)
        ctx.save_for_backward(dy, x)
        return _map(7, dy, x)

    @staticmethod
    def backward(ctx, G):   # second order: the fp32 kernel between conversion passes
        dy, x = ctx.saved_tensors
        gdy, gx = ops.GeluBwd.backward(_Saved(cast_f32(dy), cast_f32(x)), cast_f32(G))
        return cast_b16(gdy), cast_b16(gx)


class _Saved:
    """stand-in context carrying `saved_tensors` for a parent-class backward run on converted tensors"""

    def __init__(self, *tensors, **attrs):
        self.saved_tensors = tensors
        self.__dict__.update(attrs)


@_twin(ops.SumN)
class SumN16(ops.SumN):
    @staticmethod
    def forward(ctx, *xs):
        assert all(x.shape == xs[0].shape for x in xs), [tuple(x.shape) for x in xs]
        ts = [_c16(x) for x in xs]
        L = ops._L()
        while len(ts) > 1:
            head, ts = ts[:8], ts[8:]
            if len(head) == 1:
                ts.insert(0, head[0])
                break
            out = torch.empty(head[0].shape, dtype=B16, device=head[0].device)
            arr = (ctypes.c_void_p * len(head))(*[t.data_ptr() for t in head])
            ops._chk(L.ix_sum_n_b16(arr, len(head), out.data_ptr(), out.numel(), ops._stream()), "ix_sum_n_b16")
            ts.insert(0, out)
        return ts[0]


@_twin(ops.ChannelScale)
class ChannelScale16(ops.ChannelScale):
    @staticmethod
    def forward(ctx, x, scale):
        x = _c16(x)
        if scale.numel() % 8:
            return cast_b16(ops.ChannelScale.forward(ctx, cast_f32(x), scale))
        ctx.save_for_backward(scale)
        return _channel(2, x, None, scale, None)


@_twin(ops.ReluBwdChannelScale)
class ReluBwdChannelScale16(ops.ReluBwdChannelScale):
    @staticmethod
    def forward(ctx, g, y, scale):
        g, y = _c16(g), _c16(y)
        ctx.save_for_backward(y, scale)
        return _channel(1, g, y, scale, None)


@_twin(ops.BnAct)
class BnAct16(ops.BnAct):
    @staticmethod
    def forward(ctx, x, scale, shift, residual, relu):
        x = _c16(x)
        residual = _c16(residual) if residual is not None else None
        y = _channel(0, x, residual, scale, shift, relu)
        ctx.relu = relu
        ctx.has_res = residual is not None
        ctx.save_for_backward(scale, y if relu else None)
        return y


@_twin(ops.AddRowVec)
class AddRowVec16(ops.AddRowVec):
    @staticmethod
    def forward(ctx, a, v, groups=1):
        a, v = _c16(a), ops._req(cast_f32(v))
        C = v.numel() // groups
        assert a.numel() % (C * groups) == 0 and C % 8 == 0
        ctx.vshape, ctx.groups = tuple(v.shape), groups
        return _channel(3, a, None, v, None, False, groups)


@_twin(ops.ColSum)
class ColSum16(ops.ColSum):
    b16_out = False

    @staticmethod
    def forward(ctx, x):
        x = _c16(x)
        if x.dim() == 3:
            G, rows, C = x.shape
            out = torch.empty(G, C, device=x.device, dtype=torch.float32)
        else:
            (rows, C), G = x.shape, 1
            out = torch.empty(C, device=x.device, dtype=torch.float32)
        ctx.rows = rows
        if C % 8:
            return ops.ColSum.forward(ctx, cast_f32(x))
        n = ctypes.c_size_t()
        ops._chk(ops._L().ix_workspace_bytes_colsum_b16(rows, C, G, ctypes.byref(n)), "ix_workspace_bytes_colsum_b16")
        ws = ops._workspace(n.value, x.device)
        ops._chk(ops._L().ix_colsum_b16(x.data_ptr(), out.data_ptr(), rows, C, G, ws.data_ptr(), n.value, ops._stream()), "ix_colsum_b16")
        return out


@_twin(ops.LayerNorm)
class LayerNorm16(ops.LayerNorm):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        x, gamma, beta = _c16(x), ops._req(gamma), ops._req(beta)
        D = x.shape[-1]
        rows = x.numel() // D
        if gamma.dim() == 2 or D % 4 or D > 1024:   # per-episode affine / odd widths: the fp32 kernel between conversion passes
            G = gamma.shape[0] if gamma.dim() == 2 else 1
            x32 = cast_f32(x)
            y32 = torch.empty_like(x32)
            mean = torch.empty(rows, device=x.device, dtype=torch.float32)
            rstd = torch.empty_like(mean)
            ops._chk(ops._L().ix_layernorm_fwd_f32(x32.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y32.data_ptr(), mean.data_ptr(),
                                                   rstd.data_ptr(), rows // G, D, eps, G, ops._stream()), "ix_layernorm_fwd_f32")
            ctx.save_for_backward(x, gamma, mean, rstd)
            return cast_b16(y32)
        y = torch.empty(x.shape, dtype=B16, device=x.device)
        mean = torch.empty(rows, device=x.device, dtype=torch.float32)
        rstd = torch.empty_like(mean)
        ops._chk(ops._L().ix_layernorm_fwd_b16(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), mean.data_ptr(),
                                               rstd.data_ptr(), rows, D, eps, ops._stream()), "ix_layernorm_fwd_b16")
        ctx.save_for_backward(x, gamma, mean, rstd)
        return y


@_twin(ops.LayerNormBwd)
class LayerNormBwd16(ops.LayerNormBwd):
    @staticmethod
    def forward(ctx, dy, x, gamma, mean, rstd):
        dy, x = _c16(dy), _c16(x)
        D = x.shape[-1]
        if gamma.dim() == 2 or D % 4 or D > 1024:
            dx, dg, db = ops.LayerNormBwd.forward(ops._NullCtx(), cast_f32(dy), cast_f32(x), gamma, mean, rstd)
            ctx.save_for_backward(dy, x, gamma, mean, rstd)
            return cast_b16(dx), dg, db
        rows = x.numel() // D
        dx = torch.empty(x.shape, dtype=B16, device=x.device)
        both = torch.empty((2,) + tuple(gamma.shape), device=gamma.device, dtype=torch.float32)
        n = ctypes.c_size_t()
        ops._chk(ops._L().ix_workspace_bytes_layernorm_bwd_b16(rows, D, ctypes.byref(n)), "ix_workspace_bytes_layernorm_bwd_b16")
        ws = ops._workspace(n.value, x.device)
        ops._chk(ops._L().ix_layernorm_bwd_b16(dy.data_ptr(), x.data_ptr(), gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(), dx.data_ptr(),
                                               both[0].data_ptr(), both[1].data_ptr(), rows, D, ws.data_ptr(), n.value, ops._stream()),
                 "ix_layernorm_bwd_b16")
        ctx.save_for_backward(dy, x, gamma, mean, rstd)
        return dx, both[0], both[1]

    @staticmethod
    def backward(ctx, Gx, Gg, Gb):   # second order: the fp32 kernel between conversion passes
        dy, x, gamma, mean, rstd = ctx.saved_tensors
        up = lambda t: cast_f32(t) if t is not None else None
        gdy, gx, gg, _, _ = ops.LayerNormBwd.backward(_Saved(cast_f32(dy), cast_f32(x), gamma, mean, rstd), up(Gx), Gg, Gb)
        return cast_b16(gdy), cast_b16(gx), gg, None, None


# ---- implicit-GEMM convolutions on bf16 NHWC activations (csrc/gemm16.hip, ix_conv_gemm_b16) -------------------------------------------
_conv16_ok, _conv16_ws = {}, {}


def conv16_supported(kind, cg):
    ok = _conv16_ok.get((kind, cg))
    if ok is None:
        ok = _conv16_ok[(kind, cg)] = bool(ops._L().ix_conv_gemm_b16_supported(
            kind, cg.E, cg.imgs, cg.H, cg.W, cg.Cin, cg.OH, cg.OW, cg.Cout, cg.KH, cg.KW, cg.stride, cg.pad, cg.dil))
    return ok


def _conv16(kind, src, other, out_shape, cg, out_f32=False, scale=None, shift=None, residual=None, relu=False):
    out = torch.empty(out_shape, device=src.device, dtype=torch.float32 if out_f32 else B16)
    nws = _conv16_ws.get((kind, cg))
    if nws is None:
        n = ctypes.c_size_t(0)
        ops._chk(ops._L().ix_workspace_bytes_conv_gemm_b16(kind, cg.E, cg.imgs, cg.H, cg.W, cg.Cin, cg.OH, cg.OW, cg.Cout, cg.KH, cg.KW,
                                                           ctypes.byref(n)), "ix_workspace_bytes_conv_gemm_b16")
        nws = _conv16_ws[(kind, cg)] = n.value
    ws = ops._workspace(nws, src.device) if nws else None
    ops._chk(ops._L().ix_conv_gemm_b16(kind, src.data_ptr(), other.data_ptr(), out.data_ptr(), cg.E, cg.imgs, cg.H, cg.W, cg.Cin, cg.OH,
                                       cg.OW, cg.Cout, cg.KH, cg.KW, cg.stride, cg.pad, cg.dil, 1 if out_f32 else 0, _ptr(scale),
                                       _ptr(shift), _ptr(residual), 1 if relu else 0, ws.data_ptr() if nws else None, nws,
                                       ops._stream()), "ix_conv_gemm_b16")
    _stats["native_gemms"] += 1
    return out


@_twin(ops.ConvFwd)
class ConvFwd16(ops.ConvFwd):
    @staticmethod
    def forward(ctx, x, w, cg):
        ctx.w_key = ops._param_key(w)
        ctx.cg = cg
        x = _c16(x)
        w = _reqd(w, "conv weight")
        ctx.save_for_backward(x, w)
        if not conv16_supported(0, cg):
            return cast_b16(ops._conv_gemm(0, cast_f32(x), cast_f32(w), (cg.E * cg.imgs, cg.OH, cg.OW, cg.Cout), cg))
        return _conv16(0, x, _as_b16(w), (cg.E * cg.imgs, cg.OH, cg.OW, cg.Cout), cg)


@_twin(ops.ConvBwdData)
class ConvBwdData16(ops.ConvBwdData):
    @staticmethod
    def forward(ctx, dy, w, cg):
        ctx.w_key = ops._param_key(w)
        ctx.cg = cg
        dy = _c16(dy)
        w = _reqd(w, "conv weight")
        ctx.save_for_backward(dy, w)
        if not conv16_supported(1, cg):
            return cast_b16(ops._conv_gemm(1, cast_f32(dy), cast_f32(w), (cg.E * cg.imgs, cg.H, cg.W, cg.Cin), cg))
        return _conv16(1, dy, _as_b16(w), (cg.E * cg.imgs, cg.H, cg.W, cg.Cin), cg)


@_twin(ops.ConvBwdWeight)
class ConvBwdWeight16(ops.ConvBwdWeight):
    b16_out = False

    @staticmethod
    def forward(ctx, dy, x, cg, w_shape):
        """the weight gradient of a bf16 convolution: fp32 (the parameter's dtype)"""
        ctx.cg = cg
        dy, x = _c16(dy), _c16(x)
        ctx.save_for_backward(dy, x)
        if not conv16_supported(2, cg):
            return ops._conv_gemm(2, cast_f32(dy), cast_f32(x), w_shape, cg)
        return _conv16(2, dy, x, w_shape, cg, out_f32=True)


@_twin(ops.ConvFwdBnAct)
class ConvFwdBnAct16(ops.ConvFwdBnAct):
    @staticmethod
    def forward(ctx, x, w, scale, shift, residual, relu, cg, fan=1):
        ctx.set_materialize_grads(False)
        ctx.w_key = ops._param_key(w)
        x = _c16(x)
        w, scale, shift = _reqd(w, "conv weight"), ops._req(scale), ops._req(shift)
        residual = _c16(residual) if residual is not None else None
        shape = (cg.E * cg.imgs, cg.OH, cg.OW, cg.Cout)
        if conv16_supported(0, cg):
            out = _conv16(0, x, _as_b16(w), shape, cg, False, scale, shift, residual, relu)
        else:
            y = ops._conv_gemm(0, cast_f32(x), cast_f32(w), shape, cg)
            out = cast_b16(ops._channel_affine(y, scale, shift, cast_f32(residual) if residual is not None else None, relu))
        ctx.cg, ctx.relu, ctx.has_res = cg, relu, residual is not None
        ctx.save_for_backward(x, w, scale, out if relu else None)
        return out if fan == 1 else (out, out.view_as(out))


# ---- attention in the mode: bf16 q / k / v / dO into the flash kernels' planes, head dim 64 on the single-term passes -------------------
SINGLE_TERM = 0 if os.environ.get("IX_B16_SINGLE_TERM", "1") == "0" else 1   # "0": the three-term passes on the same planes (A/B runs)


class _single_term:
    """the head-dim-64 passes of csrc/flash16.hip in their single-term build for the duration of one launch sequence (operands that are
    16-bit values fit the h plane exactly: one matrix instruction per k-slice instead of three)"""

    def __enter__(self):
        self.old = ops._L().ix_flash_set_single_term(SINGLE_TERM)
        return self

    def __exit__(self, *exc):
        ops._L().ix_flash_set_single_term(self.old)
        return False


@_twin(ops.FlashAttention)
class FlashAttention16(ops.FlashAttention):
    @staticmethod
    def forward(ctx, q, k, v, g, mask, p, seed):
        done = {}
        for t in (q, k, v):
            if id(t) not in done:
                done[id(t)] = _c16(t)
        q, k, v = done[id(q)], done[id(k)], done[id(v)]
        with _single_term():
            out = ops.FlashAttention._forward(ctx, q, k, v, g, mask, p, seed)   # (saves q, k, v, the fp32 output and lse)
        return cast_b16(out)


@_twin(ops.FlashAttentionBwd)
class FlashAttentionBwd16(ops.FlashAttentionBwd):
    @staticmethod
    def forward(ctx, q, k, v, out, lse, do, g, p, seed, pl, same_qk):
        do = _c16(do)
        with _single_term():
            gq, gk, gv = ops.FlashAttentionBwd._forward(ctx, q, k, v, ops._req(out), lse, do, g, p, seed, pl, same_qk)
        return tuple(cast_b16(t) if t is not None else None for t in (gq, gk, gv))

    @staticmethod
    @once_differentiable
    def backward(ctx, hq, hk, hv):
        """double backward (the MAML meta-gradient in the mode): bf16 cotangents into the same three passes"""
        q, k, v = ctx.saved_tensors[:3]
        prep = lambda h, ref: _c16(h.contiguous()) if h is not None else torch.zeros(ref.shape, dtype=B16, device=ref.device)
        hq = prep(hq, q)
        hk = hq if ctx.same_qk[0] else prep(hk, k)
        hv = hq if ctx.same_qk[1] else prep(hv, v)
        with _single_term():
            outs = ops.FlashAttentionBwd._backward_impl(ctx, hq, hk, hv)
        return tuple(cast_b16(t) if torch.is_tensor(t) else t for t in outs)


def gemm_rowsum(dc, a, sp, groups):
    """(dW, dbias) = (alpha dC^T a, colsum(dC)) in ONE launch of the bf16 GEMM (ix_gemm_rowsum_b16: four more matrix instructions per
    k-slice against a fragment of ones in the workgroups of the first N tile) -- the 16-bit twin of hipops.GemmRowsum for the
    unrecorded backward.  sp: the weight-gradient contraction (A = dC^T m-contiguous, B = a n-contiguous).  None when the operands
    do not qualify (recorded backward, unaligned rows): the caller then takes the two separate nodes."""
    if torch.is_grad_enabled() and (dc.requires_grad or a.requires_grad):
        return None
    if not (sp.A.trans and sp.A.offset == 0 and sp.A.ld == sp.M and sp.bi == 1 and not sp.B.trans and not sp.C.trans):
        return None
    L = ops._L()
    if L.ix_gemm_b16_supported((dc.data_ptr() if dc.dtype == B16 else 0), (a.data_ptr() if a.dtype == B16 else 0) + 2 * sp.B.offset,
                               4 * sp.C.offset, sp.M, sp.N, sp.K, 0, 0, sp.A.ld, sp.B.ld, sp.C.ld, sp.A.so, 0, sp.B.so, 0, sp.C.so, 0) != 1:
        return None
    dc16, a16 = _c16(dc), _c16(a)
    covered = sp.bo * sp.M * sp.N == ops._numel(sp.out_shape)
    out = (torch.empty if covered else torch.zeros)(sp.out_shape, device=dc.device, dtype=torch.float32)
    rs = torch.empty((groups, sp.M) if groups else (sp.M,), device=dc.device, dtype=torch.float32)
    key = ("b16", sp.M, sp.N, sp.K, sp.bo)
    nws = ops._ws_bytes.get(key)
    if nws is None:
        n = ctypes.c_size_t()
        ops._chk(L.ix_workspace_bytes_gemm_b16(sp.M, sp.N, sp.K, sp.bo, ctypes.byref(n)), "ix_workspace_bytes_gemm_b16")
        nws = ops._ws_bytes[key] = n.value
    ws = ops._workspace(nws, dc.device) if nws else None
    ops._chk(L.ix_gemm_rowsum_b16(dc16.data_ptr(), a16.data_ptr() + 2 * sp.B.offset, out.data_ptr() + 4 * sp.C.offset, rs.data_ptr(),
                                  sp.M, sp.N, sp.K, sp.A.ld, sp.B.ld, sp.C.ld, sp.bo, sp.A.so, sp.B.so, sp.C.so, sp.M, sp.alpha,
                                  ws.data_ptr() if nws else None, nws, ops._stream()), "ix_gemm_rowsum_b16")
    _stats["native_gemms"] += 1
    return out, rs
