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

from . import hipops as ops
from ._lib import HipLibraryError

B16 = torch.bfloat16
_seen = ops._b16_seen   # [False] until the first bf16 activation exists in this process (keeps the fp32 path's per-call cost at one list read)


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
    cap = ops._capture[0]
    if cap is not None:
        store, tag = cap, ("b16", ops._scratch_slot[0], id(w), ops._wp_epoch[0])
    else:
        store, tag = w.__dict__.setdefault("_ix_b16", {}), ops._wp_epoch[0]
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
