"""hipops.contraction -- strided batched contractions (GemmSpec), their routing (12-wave kernels, weight planes, bf16 GEMM) and the
Linear-layer Functions."""
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
from .core import (Function, _L, _capture, _chk, _is_unwanted, _numel, _param_key, _req, _scratch_slot, _stream, _workspace, _wp_epoch, mark_weight)
from . import elementwise
from .elementwise import (AddRowVec, ColSum, fanout)


# ---------------------------------------------------------------------------------------------------------
# GEMM
# ---------------------------------------------------------------------------------------------------------
# A strided matrix view into a flat tensor: elem(r, c) = base[offset + bo*so + bi*si + (c*ld + r if trans else r*ld + c)]
View = namedtuple("View", "offset ld trans so si")
# One batched contraction C = alpha * A(MxK) B(KxN): views for A, B and for C inside a fresh tensor `out_shape`
# (odt: dtype of C in the 16-bit mode -- None = bf16 when an operand is bf16; torch.float32: heads, parameter gradients)
GemmSpec = namedtuple("GemmSpec", "M N K bo bi A B C out_shape alpha odt", defaults=(None,))


def _flip(v):
    return View(v.offset, v.ld, not v.trans, v.so, v.si)


# Pre-split fp16x3 contraction route (csrc/gemm_x3.hip), opt-in: IX_GEMM_X3=1.  Measured on the step's shapes
# (tools/x3_bench.py, profiles/r2b_x3_bench.json): 1.2-1.4x the bf16x6 kernel where both operands are stored with the
# contracted index contiguous and N >= 1024 (the operand-split passes amortise over N), slower elsewhere -- so only those
# shapes are routed, and the default stays bf16x6 for every contraction.
GEMM_X3 = os.environ.get("IX_GEMM_X3", "0") == "1"
_ws_bytes = {}   # contraction signature -> scratch bytes (split-K planes; fp16x3 operand planes on the opt-in route)


def _gemm_workspace_bytes(pa, pb, sp, presplit=True):
    """presplit=False: the caller's entry point never takes the opt-in pre-split route (row-sum / BN-fused contractions): size
    the scratch for ITS plan (split-K planes), not for the fp16 operand planes of a route it will not use."""
    x3 = presplit and GEMM_X3 and not sp.A.trans and sp.B.trans and sp.N >= 1024 and sp.M >= 1024
    key = (sp.M, sp.N, sp.K, sp.bo, sp.bi, sp.A.trans, sp.B.trans, sp.A.ld, sp.B.ld, sp.A.so, sp.B.so, pa & 15, pb & 15, x3)
    n = _ws_bytes.get(key)
    if n is None:
        out = ctypes.c_size_t(0)
        L = _L()
        L.ix_gemm_presplit_enable(1 if x3 else 0)
        _chk(L.ix_workspace_bytes_gemm_f32(sp.M, sp.N, sp.K, 0 if sp.A.trans else 1, 1 if sp.B.trans else 0, sp.A.ld,
                                           sp.B.ld, sp.bo, sp.bi, sp.A.so, sp.B.so, pa, pb, 0, 0, ctypes.byref(out)),
             "ix_workspace_bytes_gemm_f32")
        L.ix_gemm_presplit_enable(0)
        n = _ws_bytes[key] = (out.value, x3)
    return n


# ---- activation x WEIGHT PLANES (csrc/gemm_wp.hip) ---------------------------------------------------------------------------
# A Linear layer's forward and input gradient multiply an activation by a WEIGHT.  The 12-wave kernel converts both fp32
# operands to fp16 planes in its producer waves, once per output tile; a weight is read by every row tile of every
# contraction that uses it, several times per step.  Route: the weight is converted ONCE per (tensor, version, orientation) into
# the kernel's own LDS image (ix_wp_split_f32) -- cached on the tensor object, re-made when the tensor changes -- and the
# contraction runs on gemm_wp_kernel (weight planes and raw fp32 activation tiles by LDS-DMA, activation split in the
# consumers' registers, four workgroups per CU).  Same arithmetic class as the fp16x3 form (tests/test_ops_gpu.py::
# test_weight_planes_contraction_*).  Which operands are weights: `mark_weight` (set by linear() / weight_view()).
GEMM_WP = os.environ.get("IX_GEMM_WP", "1") == "1"
WP_MIN_ROWS = int(os.environ.get("IX_GEMM_WP_MIN_ROWS", "8192"))
_wp_stats = {"routed": 0, "splits": 0}


def _wp_plan(a, b, bias, sp):
    """-> (k_contig, ld, batch_stride, shared) when this contraction takes the weight-planes route, else None"""
    if not (GEMM_WP and getattr(b, "_ix_weight", False)) or core.COMPUTE_DTYPE != "f32":
        return None
    A, B, C = sp.A, sp.B, sp.C
    if A.trans or C.trans or sp.K % 32 or sp.M < 128 or sp.N < 128 or sp.alpha != 1.0:
        return None
    if (A.ld | A.offset | A.so | A.si) & 3 or sp.M * A.ld >= (1 << 29) or a.data_ptr() & 15:
        return None
    # the weight view: contiguous [N, K] rows (k-contiguous) or contiguous [K, N] (n-contiguous), one per outer slice or shared
    if B.offset & 3 or B.si != 0 and sp.bi > 1:
        return None
    if B.ld != (sp.K if B.trans else sp.N) or B.so not in (0, sp.N * sp.K):
        return None
    if bias is not None and bias.dim() == 2 and bias.shape[0] != sp.bo:
        return None
    # measured (tools/wp_bench.py, profiles/r4i_wp_bench.txt): 1.25-1.35x the 12-wave kernel on long activations (M >= 12 500
    # rows per slice), 1.15-1.28x at 1805 rows x 16 episodes when N >= 512 and a tie or a loss at N = 256 with K >= 1024 --
    # and a weight is used about once per orientation and weight set per step, so at 1805 rows the split (10-12 % of such a
    # contraction) eats the gain (r4g / r4h: routed launches 40.6 -> 35.9 ms, splits + 6.5 ms).  The route is taken where the
    # split is noise: long activations.  IX_GEMM_WP_MIN_ROWS moves the threshold (tests: 128).
    if sp.M < WP_MIN_ROWS or (sp.N <= 256 and sp.K >= 1024 and sp.M < 8192):
        return None
    if -(-sp.M // 128) * -(-sp.N // 128) * sp.bo * sp.bi < (512 if WP_MIN_ROWS > 128 else 96):
        return None   # too few tiles for four workgroups on each of 256 CUs (the 12-wave kernel is persistent and splits K)
    if torch.cuda.is_current_stream_capturing() and _capture[0] is None:
        return None   # a capture this module was not told about: nowhere safe to keep planes that only exist at replay
    lib = _L()
    if lib.ix_gemm_set_x3(1) != 1:   # the bf16x6 form was asked for (IX_GEMM_KERNEL=x6, the tests' kernel_form): this route is
        lib.ix_gemm_set_x3(0)        # the fp16x3 arithmetic -- leave the contraction to the 12-wave kernel's bf16x6 form
        return None
    return (1 if B.trans else 0, B.ld, B.so, B.so == 0 or sp.bo == 1)


def _wp_planes(b, sp, plan):
    """(planes, unscale) of weight operand `b` for this orientation: cached on the tensor object while its version stands.
    Inside a HIP-graph capture the cache is the capture's own (the split is part of the graph: a replay must redo it, weights
    change between replays while their addresses stay)."""
    kc, ld, so, shared = plan
    nb = 1 if shared else sp.bo
    key = (kc, ld, so, sp.B.offset, sp.N, sp.K, nb, _wp_epoch[0])
    if _capture[0] is not None:
        # (+ the scratch slot = the stream a captured segment replays on: segment C -- slot 1, the side stream -- and segment B
        #  run concurrently at replay, so neither may read planes that the other one's graph writes)
        store, tag = _capture[0], ("wp", _scratch_slot[0], id(b)) + key
    else:
        store = b.__dict__.setdefault("_ix_wp", {})
        tag = key
        if len(store) > 4:   # (superseded epochs / versions of this tensor)
            store.clear()
    hit = store.get(tag)
    if hit is not None and hit[2] == b._version and hit[3] == b.data_ptr():
        return hit[0], hit[1]
    L = _L()
    pb, ub = ctypes.c_size_t(), ctypes.c_size_t()
    _chk(L.ix_wp_planes_bytes(sp.N, sp.K, nb, ctypes.byref(pb), ctypes.byref(ub)), "ix_wp_planes_bytes")
    planes = torch.empty(pb.value, dtype=torch.uint8, device=b.device)
    unscale = torch.empty(ub.value // 4, dtype=torch.float32, device=b.device)
    _chk(L.ix_wp_split_f32(b.data_ptr() + sp.B.offset * 4, ld, so if not shared else 0, sp.N, sp.K, kc, nb, planes.data_ptr(),
                           unscale.data_ptr(), _stream()), "ix_wp_split_f32")
    store[tag] = (planes, unscale, b._version, b.data_ptr(), b if _capture[0] is not None else None)
    _wp_stats["splits"] += 1
    return planes, unscale


def _run_gemm(a, b, bias, sp, fill=True):
    """fill=False: the caller guarantees nobody reads the elements of `out` the product does not write (the pad columns
    of attention tensors: every consumer stops at the row length) -- saves a memset of the whole tensor."""
    covered = not fill or sp.bo * sp.bi * sp.M * sp.N == _numel(sp.out_shape)
    out = (torch.empty if covered else torch.zeros)(sp.out_shape, device=a.device, dtype=torch.float32)
    assert not sp.C.trans
    esz = 4
    pa, pb = a.data_ptr() + sp.A.offset * esz, b.data_ptr() + sp.B.offset * esz
    plan = _wp_plan(a, b, bias, sp)
    if plan is not None:
        planes, unscale = _wp_planes(b, sp, plan)
        _wp_stats["routed"] += 1
        _chk(_L().ix_gemm_wp_f32(pa, sp.A.ld, sp.A.so, sp.A.si, planes.data_ptr(), unscale.data_ptr(), 1 if plan[3] else 0,
                                 out.data_ptr() + sp.C.offset * esz, sp.C.ld, sp.C.so, sp.C.si,
                                 bias.data_ptr() if bias is not None else None,
                                 sp.N if (bias is not None and bias.dim() == 2) else 0, sp.M, sp.N, sp.K, sp.bo, sp.bi, sp.alpha,
                                 _stream()), "ix_gemm_wp_f32")
        return out
    nws, x3 = _gemm_workspace_bytes(pa, pb, sp)
    ws = _workspace(nws, a.device) if nws else None
    L = _L()
    if x3:
        L.ix_gemm_presplit_enable(1)
    rc = L.ix_gemm_f32_ws(pa, pb,
                          out.data_ptr() + sp.C.offset * esz, bias.data_ptr() if bias is not None else None,
                          sp.M, sp.N, sp.K, 0 if sp.A.trans else 1, 1 if sp.B.trans else 0,
                          sp.A.ld, sp.B.ld, sp.C.ld, sp.bo, sp.bi, sp.A.so, sp.A.si, sp.B.so, sp.B.si, sp.C.so, sp.C.si,
                          sp.N if (bias is not None and bias.dim() == 2) else 0, sp.alpha, 0, 0,
                          ws.data_ptr() if nws else None, nws, _stream())
    if x3:
        L.ix_gemm_presplit_enable(0)
    _chk(rc, "ix_gemm_f32_ws")
    return out


class CatParams(Function):
    """torch.cat(params, 0) whose result keeps standing for its sources in skip_param_grads (fusion key / query / value
    projections evaluated as one contraction); the gradient goes back as row views."""

    @staticmethod
    def forward(ctx, *ws):
        ctx.sizes = [w.shape[0] for w in ws]
        out = torch.cat(ws, 0)
        keys = []
        for w in ws:
            k = _param_key(w)
            keys.extend(k if isinstance(k, tuple) else (k,))
        out._ix_of_param = tuple(keys)
        return out

    @staticmethod
    def backward(ctx, g):
        return tuple(g.split(ctx.sizes, 0))


class Gemm(Function):
    """out = alpha * A B (+ bias) for strided views A of `a` and B of `b`; see GemmSpec."""

    b16 = "native"

    @staticmethod
    def forward(ctx, a, b, bias, sp):
        ctx.a_key, ctx.b_key = _param_key(a), _param_key(b)
        ctx.bias_key = _param_key(bias) if bias is not None else None
        if a.dtype == torch.bfloat16 or b.dtype == torch.bfloat16:
            return Gemm._forward_b16(ctx, a, b, bias, sp)
        a, b = _req(a, "gemm A"), _req(b, "gemm B")
        ctx.sp = sp
        ctx.has_bias = bias is not None
        ctx.bias_groups = bias.shape[0] if (bias is not None and bias.dim() == 2) else 0
        if bias is not None:
            bias = _req(bias, "gemm bias")
            assert bias.numel() == (sp.bo if ctx.bias_groups else 1) * sp.N, (tuple(bias.shape), sp.bo, sp.N)
        ctx.a_shape, ctx.b_shape = tuple(a.shape), tuple(b.shape)
        ctx.save_for_backward(a, b)
        return _run_gemm(a, b, bias, sp)

    @staticmethod
    def _forward_b16(ctx, a, b, bias, sp):
        """the 16-bit mode: bf16 operands (an fp32 weight is read through its cached bf16 copy), C in sp.odt; the gradients come back
        in each operand's OWN dtype -- bf16 for activations, fp32 for parameters"""
        from .. import b16
        wa, wb = getattr(a, "_ix_weight", False), getattr(b, "_ix_weight", False)
        a, b = b16._reqd(a, "gemm A"), b16._reqd(b, "gemm B")
        if wa:
            mark_weight(a)
        if wb:
            mark_weight(b)
        ctx.sp = sp
        ctx.has_bias = bias is not None
        ctx.bias_groups = bias.shape[0] if (bias is not None and bias.dim() == 2) else 0
        if bias is not None:
            bias = _req(bias, "gemm bias")
            assert bias.numel() == (sp.bo if ctx.bias_groups else 1) * sp.N, (tuple(bias.shape), sp.bo, sp.N)
        ctx.a_shape, ctx.b_shape = tuple(a.shape), tuple(b.shape)
        ctx.save_for_backward(a, b)
        return b16.run_gemm(a, b, bias, sp)

    @staticmethod
    def backward(ctx, dc):
        a, b = ctx.saved_tensors
        sp = ctx.sp
        dc = dc.contiguous()
        skip = core._unwanted
        need_a = ctx.needs_input_grad[0] and not _is_unwanted(ctx.a_key, skip)
        need_b = ctx.needs_input_grad[1] and not _is_unwanted(ctx.b_key, skip)
        need_bias = ctx.has_bias and ctx.needs_input_grad[2] and not _is_unwanted(ctx.bias_key, skip)
        # the bias gradient colsum(dC) rides on the weight-gradient contraction dB^T = dC^T A (ix_gemm_rowsum_f32: the
        # A-producer waves of that launch sum the dC tiles they stream anyway) whenever dC is its plain m-contiguous A operand
        fuse = (GEMM_ROWSUM and need_bias and need_b and sp.B.trans and sp.bi == 1 and sp.C.offset == 0
                and (dc.dtype == a.dtype or b.dtype == torch.float32)   # (16-bit mode: bf16 dC and activation, fp32 weight)
                and sp.C.ld == sp.N and (ctx.bias_groups == sp.bo or (ctx.bias_groups == 0 and sp.bo == 1))
                and (sp.bo == 1 or sp.C.so == sp.M * sp.N))
        # (recorded backward: dC feeds up to three nodes -- one alias each, so that ITS gradient is one sum, Fanout)
        dcs = list(fanout(dc, int(need_a) + int(need_b) + int(need_bias and not fuse)))
        da, db = _gemm_backward(sp, a, b, ctx.a_shape, ctx.b_shape, dc, need_a, need_b and not fuse, dcs)
        dbias = None
        if fuse:
            s = GemmSpec(sp.N, sp.K, sp.M, sp.bo, sp.bi, _flip(sp.C), sp.A,
                         View(sp.B.offset, sp.B.ld, False, sp.B.so, sp.B.si), ctx.b_shape, sp.alpha)
            if dc.dtype == torch.bfloat16 or a.dtype == torch.bfloat16:
                from .. import b16
                fused = b16.gemm_rowsum(dcs[-1], a, s, ctx.bias_groups)
                if fused is None:   # (rows not 16-byte aligned: the two separate nodes)
                    fuse = False
                    db = _gemm_backward(sp, a, b, ctx.a_shape, ctx.b_shape, dc, False, True, [dcs.pop()])[1]
                    d = dc
                    dbias = ColSum.call(d.reshape(ctx.bias_groups, -1, sp.N) if ctx.bias_groups else d.reshape(-1, sp.N))
                else:
                    dcs.pop()
                    db, dbias = fused
                return da, db, dbias, None
            db, dbias = GemmRowsum.call(dcs.pop(), a, s, ctx.bias_groups)
        elif need_bias:
            d = dcs.pop()
            dbias = ColSum.call(d.reshape(ctx.bias_groups, -1, sp.N) if ctx.bias_groups else d.reshape(-1, sp.N))
        return da, db, dbias, None


def _gemm_backward(sp, a, b, a_shape, b_shape, dc, need_a, need_b, dcs=None):
    """Gradients of C = alpha A B w.r.t. the storage of A and of B (each again one strided contraction).  `dcs`: aliases of dC
    to consume, one per node (hipops.fanout), or None."""
    da = db = None
    # (16-bit mode: a gradient has its operand's own dtype -- bf16 activations, fp32 parameters; None outside the mode)
    mixed = dc.dtype == torch.bfloat16 or a.dtype == torch.bfloat16 or b.dtype == torch.bfloat16
    oa, ob = (a.dtype, b.dtype) if mixed else (None, None)
    if need_a:
        dc = dcs.pop() if dcs else dc
        if not sp.A.trans:   # dA (MxK) = alpha * dC (MxN) * B^T (NxK)
            s = GemmSpec(sp.M, sp.K, sp.N, sp.bo, sp.bi, sp.C, _flip(sp.B),
                         View(sp.A.offset, sp.A.ld, False, sp.A.so, sp.A.si), a_shape, sp.alpha, oa)
            da = Gemm.call(dc, b, None, s)
        else:                # storage holds A^T (KxM): dA^T = alpha * B (KxN) * dC^T (NxM)
            s = GemmSpec(sp.K, sp.M, sp.N, sp.bo, sp.bi, sp.B, _flip(sp.C),
                         View(sp.A.offset, sp.A.ld, False, sp.A.so, sp.A.si), a_shape, sp.alpha, oa)
            da = Gemm.call(b, dc, None, s)
    if need_b:
        dc = dcs.pop() if dcs else dc
        if not sp.B.trans:   # dB (KxN) = alpha * A^T (KxM) * dC (MxN)
            s = GemmSpec(sp.K, sp.N, sp.M, sp.bo, sp.bi, _flip(sp.A), sp.C,
                         View(sp.B.offset, sp.B.ld, False, sp.B.so, sp.B.si), b_shape, sp.alpha, ob)
            db = Gemm.call(a, dc, None, s)
        else:                # storage holds B^T (NxK): dB^T = alpha * dC^T (NxM) * A (MxK)
            s = GemmSpec(sp.N, sp.K, sp.M, sp.bo, sp.bi, _flip(sp.C), sp.A,
                         View(sp.B.offset, sp.B.ld, False, sp.B.so, sp.B.si), b_shape, sp.alpha, ob)
            db = Gemm.call(dc, a, None, s)
    return da, db


GEMM_ROWSUM = os.environ.get("IX_GEMM_ROWSUM", "1") == "1"   # "0": bias gradients by a separate ix_colsum_f32 launch (A/B runs)


class GemmRowsum(Function):
    """(alpha A B, rowsum(A)) for a contiguous m-fastest A view of `a` (spec A.trans, offset 0): one launch of
    ix_gemm_rowsum_f32.  rowsum: [M], or [groups, M] for per-episode operands."""

    @staticmethod
    def forward(ctx, a, b, sp, groups):
        a, b = _req(a, "gemm A"), _req(b, "gemm B")
        assert sp.A.trans and sp.A.offset == 0 and sp.A.ld == sp.M and sp.bi == 1 and not sp.C.trans
        ctx.sp, ctx.groups = sp, groups
        ctx.a_shape, ctx.b_shape = tuple(a.shape), tuple(b.shape)
        ctx.save_for_backward(a, b)
        covered = sp.bo * sp.M * sp.N == _numel(sp.out_shape)
        out = (torch.empty if covered else torch.zeros)(sp.out_shape, device=a.device, dtype=torch.float32)
        rs = torch.empty((groups, sp.M) if groups else (sp.M,), device=a.device, dtype=torch.float32)
        pa, pb = a.data_ptr(), b.data_ptr() + sp.B.offset * 4
        nws, _ = _gemm_workspace_bytes(pa, pb, sp, presplit=False)
        ws = _workspace(nws, a.device) if nws else None
        _chk(_L().ix_gemm_rowsum_f32(pa, pb, out.data_ptr() + sp.C.offset * 4,
                                     sp.M, sp.N, sp.K, 0, 1 if sp.B.trans else 0, sp.A.ld, sp.B.ld, sp.C.ld, sp.bo, sp.A.so,
                                     sp.B.so, sp.C.so, sp.alpha, rs.data_ptr(), sp.M, ws.data_ptr() if nws else None, nws,
                                     _stream()), "ix_gemm_rowsum_f32")
        return out, rs

    @staticmethod
    def backward(ctx, gc, gr):
        a, b = ctx.saved_tensors
        sp = ctx.sp
        da = db = None
        if gc is not None:
            da, db = _gemm_backward(sp, a, b, ctx.a_shape, ctx.b_shape, gc.contiguous(), ctx.needs_input_grad[0],
                                    ctx.needs_input_grad[1])
        if gr is not None and ctx.needs_input_grad[0]:   # rowsum[m] = sum_k A(m, k): every k line of A's storage gets gr
            if da is None:
                da = torch.zeros(ctx.a_shape, device=a.device, dtype=torch.float32)
            da = AddRowVec.call(da, gr.contiguous(), max(ctx.groups, 1))
        return da, db, None, None


def linear(x, weight, bias=None, out_dtype=None):
    """y[..., o] = sum_i x[..., i] * weight[o, i] + bias[o]   (nn.Linear semantics).

    Episode-batched form: weight [E, N, K] (+ bias [E, N]) holds one set of MAML fast weights per episode and the
    leading dim of x is E * (rows per episode); episode e's rows meet episode e's weights in ONE batched launch.
    out_dtype (16-bit mode only): torch.float32 for a result that leaves the 16-bit part of the graph (heads)."""
    mark_weight(weight)
    if x.dtype != torch.bfloat16:
        out_dtype = None
    if weight.dim() == 3:
        E, N, K = weight.shape
        assert x.shape[-1] == K and x.numel() % (E * K) == 0, (tuple(x.shape), tuple(weight.shape))
        R = x.numel() // (E * K)
        sp = GemmSpec(R, N, K, E, 1, View(0, K, False, R * K, 0), View(0, K, True, N * K, 0), View(0, N, False, R * N, 0),
                      tuple(x.shape[:-1]) + (N,), 1.0, out_dtype)
        return Gemm.call(x, weight, bias, sp)
    K = x.shape[-1]
    N = weight.shape[0]
    R = x.numel() // K
    out_shape = tuple(x.shape[:-1]) + (N,)
    sp = GemmSpec(R, N, K, 1, 1, View(0, K, False, 0, 0), View(0, K, True, 0, 0), View(0, N, False, 0, 0),
                  out_shape, 1.0, out_dtype)
    return Gemm.call(x, weight, bias, sp)


def matmul_nn(a, b):
    """[M,K] @ [K,N] for plain contiguous 2-D tensors."""
    M, K = a.shape
    N = b.shape[1]
    sp = GemmSpec(M, N, K, 1, 1, View(0, K, False, 0, 0), View(0, N, False, 0, 0), View(0, N, False, 0, 0), (M, N), 1.0)
    return Gemm.call(a, b, None, sp)


class SplitRows(Function):
    """Row blocks (views, no copies) of a packed parameter -- nn.MultiheadAttention's in_proj_weight / in_proj_bias -- whose
    gradient comes back as ONE concatenation instead of a zero-fill + slice copy per block + an accumulation of the
    full-size pieces (five small launches per block pair in autograd's slice backward)."""

    @staticmethod
    def forward(ctx, w, *sizes):
        ctx.sizes, ctx.tail = sizes, tuple(w.shape[1:])
        parts = tuple(w.split(list(sizes), 0))
        for t in parts:
            t._ix_of_param = id(w)   # (skip_param_grads: the blocks stand for the Parameter they were cut from)
        fl = w.__dict__.get("_ix_b16_flat")   # 16-bit mode: the blocks of the parameter's bf16 shadow go with them (b16.weight_b16)
        if fl is not None and fl[1] == core._wp_epoch[0] and fl[2] == w._version and fl[3] == w.data_ptr():
            for t, sh in zip(parts, fl[0].split(list(sizes), 0)):
                t.__dict__["_ix_b16_flat"] = (sh, fl[1], t._version, t.data_ptr())
        return parts

    @staticmethod
    def backward(ctx, *gs):
        parts = [g if g is not None else gs_zero(ctx, n, gs) for g, n in zip(gs, ctx.sizes)]
        return (torch.cat(parts, 0),) + (None,) * len(ctx.sizes)


def gs_zero(ctx, n, gs):
    ref = next(g for g in gs if g is not None)
    return torch.zeros((n,) + ctx.tail, device=ref.device, dtype=ref.dtype)
