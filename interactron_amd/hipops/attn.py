"""hipops.attn -- materialised and flash attention Functions, operand planes, key biases."""
import ctypes
import gc as _gc
import os
import os as _os
from collections import namedtuple

import torch
from torch.autograd import Function as _TorchFunction
from torch.autograd.function import once_differentiable

from .. import _lib
from . import contraction
from .contraction import (Gemm, GemmSpec, View, _flip, _run_gemm)
from . import elementwise
from .elementwise import (Axpby)
from . import core
from .core import (Function, _L, _NullCtx, _bias_cache, _capture, _chk, _req, _stream)


def attn_pitch(S):
    """Row pitch (floats) of the [L, S] attention tensors: rows start on 128-byte lines.  Measured with the C-tile store
    pattern of the score product (tools/tile_fill_probe.py): rows that straddle cache lines (pitch 2060) drain at 3.4
    TB/s chip-wide, aligned rows at 5.6 -- the K = 64 score product sits exactly on that limit."""
    return (S + 31) // 32 * 32


def attention_scores(q, k, nbatch, heads, L, S, hd, q_ld, k_ld, q_off, k_off, scale):
    """scores[b,h,l,s] = scale * sum_d q[b,l,h*hd+d] * k[b,s,h*hd+d]  -> [nbatch, heads, L, Sp], Sp = attn_pitch(S).

    q / k are [nbatch, L|S, ld] activations (possibly packed side by side: offsets q_off / k_off inside a row)."""
    Sp = attn_pitch(S)
    sp = GemmSpec(L, S, hd, nbatch, heads, View(q_off, q_ld, False, L * q_ld, hd), View(k_off, k_ld, True, S * k_ld, hd),
                  View(0, Sp, False, heads * L * Sp, L * Sp), (nbatch, heads, L, Sp), scale)
    return Gemm.call(q, k, None, sp)


def attention_apply(p, v, nbatch, heads, L, S, hd, v_ld, v_off):
    """out[b,l,h*hd+d] = sum_s p[b,h,l,s] * v[b,s,h*hd+d]  -> [nbatch, L, heads*hd]."""
    Sp = p.shape[-1]
    E = heads * hd
    sp = GemmSpec(L, hd, S, nbatch, heads, View(0, Sp, False, heads * L * Sp, L * Sp),
                  View(v_off, v_ld, False, S * v_ld, hd), View(0, E, False, L * E, hd), (nbatch, L, E), 1.0)
    return Gemm.call(p, v, None, sp)


# ---------------------------------------------------------------------------------------------------------
# attention core as ONE autograd node (and its backward as one node)
# ---------------------------------------------------------------------------------------------------------
def _spec_dA(sp, a_shape):
    """spec of dA for C = alpha A B with A stored untransposed: dA (MxK) = alpha dC (MxN) B^T; operands (dC, b)."""
    assert not sp.A.trans
    return GemmSpec(sp.M, sp.K, sp.N, sp.bo, sp.bi, sp.C, _flip(sp.B), View(sp.A.offset, sp.A.ld, False, sp.A.so, sp.A.si),
                    a_shape, sp.alpha)


def _spec_dB(sp, b_shape):
    """spec of dB: B stored untransposed -> dB (KxN) = alpha A^T dC, operands (a, dC); B stored transposed (NxK rows)
    -> dB^T = alpha dC^T A, operands (dC, a).  Returns (spec, dc_first)."""
    out = View(sp.B.offset, sp.B.ld, False, sp.B.so, sp.B.si)
    if not sp.B.trans:
        return GemmSpec(sp.K, sp.N, sp.M, sp.bo, sp.bi, _flip(sp.A), sp.C, out, b_shape, sp.alpha), False
    return GemmSpec(sp.N, sp.K, sp.M, sp.bo, sp.bi, _flip(sp.C), sp.A, out, b_shape, sp.alpha), True


def _attn_specs(g):
    Sp = attn_pitch(g.S)
    E = g.heads * g.hd
    tt = View(0, Sp, False, g.heads * g.L * Sp, g.L * Sp)
    scores = GemmSpec(g.L, g.S, g.hd, g.n, g.heads, View(g.q_off, g.q_ld, False, g.L * g.q_ld, g.hd),
                      View(g.k_off, g.k_ld, True, g.S * g.k_ld, g.hd), tt, (g.n, g.heads, g.L, Sp), g.scale)
    apply_ = GemmSpec(g.L, g.hd, g.S, g.n, g.heads, tt, View(g.v_off, g.v_ld, False, g.S * g.v_ld, g.hd),
                      View(0, E, False, g.L * E, g.hd), (g.n, g.L, E), 1.0)
    return scores, apply_, Sp


AttnGeom = namedtuple("AttnGeom", "n heads L S hd q_ld k_ld q_off k_off v_ld v_off scale")


def _sum2(a, b):
    if a is None:
        return b
    if b is None:
        return a
    return Axpby.call(a, b, 1.0, 1.0)


ATTN_LEAN_BYTES = 4 << 30   # [L, S] tensors at least this large: AttentionCore keeps two of them per layer instead of four


def _dropout_of(y, p, seed):
    """d = dropout(y) with the mask of (seed, flat index) -- what ix_attn_prob_fwd_f32 wrote as its second output."""
    if p <= 0.0:
        return y
    d = torch.empty_like(y)
    _chk(_L().ix_dropout_f32(y.data_ptr(), d.data_ptr(), y.numel(), p, seed, _stream()), "ix_dropout_f32")
    return d


class AttentionCore(Function):
    """out[b,l,h*hd+:] = dropout(softmax(scale q k^T [+ key mask])) v  per (batch, head), as one node.

    Same arithmetic as attention_scores -> Softmax -> dropout -> attention_apply, but the [L, S] tensors meet exactly
    one elementwise kernel per pass (softmax+dropout fused, mask regenerated from the seed), autograd never sums
    [L, S]-sized gradients, and the double backward (AttentionCoreBwd.backward) is written out by hand."""

    @staticmethod
    def forward(ctx, q, k, v, g, mask, p, seed):
        q, k, v = _req(q, "attention q"), _req(k, "attention k"), _req(v, "attention v")
        sp_s, sp_a, Sp = _attn_specs(g)
        y = _run_gemm(q, k, None, sp_s, fill=False)
        d = torch.empty_like(y) if p > 0.0 else None
        _chk(_L().ix_attn_prob_fwd_f32(y.data_ptr(), y.data_ptr(), d.data_ptr() if d is not None else None,
                                       g.n * g.heads * g.L, g.S, Sp, mask.data_ptr() if mask is not None else None,
                                       g.heads * g.L, mask.shape[-1] if mask is not None else 0, p, seed, _stream()),
             "ix_attn_prob_fwd_f32")
        if d is None:
            d = y
        out = _run_gemm(d, v, None, sp_a)
        # lean mode (800x800 frames: one [L, S] tensor is 5 GB per episode): keep y only, regenerate d = dropout(y)
        # and gs = softmax_bwd(y, dropout_bwd(gd)) where they are needed -- two saved [L, S] tensors per layer, not four
        ctx.lean = ATTN_LEAN_BYTES is not None and y.numel() * 4 >= ATTN_LEAN_BYTES
        ctx.g, ctx.p, ctx.seed = g, p, seed
        if ctx.lean and d is not y:
            ctx.save_for_backward(q, k, v, y)
        else:
            ctx.save_for_backward(q, k, v, y, d)
        return out

    @staticmethod
    def backward(ctx, do):
        if len(ctx.saved_tensors) == 4:
            q, k, v, y = ctx.saved_tensors
            d = None
        else:
            q, k, v, y, d = ctx.saved_tensors
        gq, gk, gv = AttentionCoreBwd.call(q, k, v, y, d, do, ctx.g, ctx.p, ctx.seed, ctx.lean)
        return gq, gk, gv, None, None, None, None


class AttentionCoreBwd(Function):
    @staticmethod
    def forward(ctx, q, k, v, y, d, do, g, p, seed, lean=False):
        do = _req(do.contiguous(), "attention dO")
        sp_s, sp_a, Sp = _attn_specs(g)
        rows = g.n * g.heads * g.L
        if d is None:
            d = _dropout_of(y, p, seed)
        gd = _run_gemm(do, v, None, _spec_dA(sp_a, tuple(y.shape)), fill=False)                 # dO v^T            [n,H,L,Sp]
        gs = torch.empty_like(y)
        _chk(_L().ix_attn_prob_bwd_f32(y.data_ptr(), gd.data_ptr(), gs.data_ptr(), rows, g.S, Sp, p, seed, _stream()),
             "ix_attn_prob_bwd_f32")
        gq = _run_gemm(gs, k, None, _spec_dA(sp_s, tuple(q.shape)))                 # scale gs k
        s_k, dc_first = _spec_dB(sp_s, tuple(k.shape))
        gk = _run_gemm(gs, q, None, s_k) if dc_first else _run_gemm(q, gs, None, s_k)   # scale gs^T q
        s_v, dc_first_v = _spec_dB(sp_a, tuple(v.shape))
        gv = _run_gemm(do, d, None, s_v) if dc_first_v else _run_gemm(d, do, None, s_v)  # d^T dO
        ctx.g, ctx.p, ctx.seed, ctx.lean = g, p, seed, lean
        if lean:
            ctx.save_for_backward(q, k, v, y, do, gd)
        else:
            ctx.save_for_backward(q, k, v, y, do, gd, d, gs)
        return gq, gk, gv

    @staticmethod
    @once_differentiable
    def backward(ctx, hq, hk, hv):
        g, p, seed = ctx.g, ctx.p, ctx.seed
        sp_s, sp_a, Sp = _attn_specs(g)
        rows = g.n * g.heads * g.L
        if ctx.lean:
            q, k, v, y, do, gd = ctx.saved_tensors
            d = gs = None
        else:
            q, k, v, y, do, gd, d, gs = ctx.saved_tensors
        hq = _req(hq.contiguous()) if hq is not None else None
        hk = _req(hk.contiguous()) if hk is not None else None
        hv = _req(hv.contiguous()) if hv is not None else None
        # G = dL/d gs = scale (hq k^T + q hk^T);  HD = dL/d d = dO hv^T
        G2 = None
        if hq is not None and hk is not None:
            # both terms as ONE product over a doubled head dim, [hq | q] [k | hk]^T: the two small operand copies
            # replace a whole [L, S] tensor written by one GEMM and read again by the kernel below
            heads = lambda t, off, ld, rows: t.view(g.n, rows, ld)[..., off:off + g.heads * g.hd].reshape(g.n, rows, g.heads, g.hd)
            a2 = torch.cat([heads(hq, g.q_off, g.q_ld, g.L), heads(q, g.q_off, g.q_ld, g.L)], -1)
            b2 = torch.cat([heads(k, g.k_off, g.k_ld, g.S), heads(hk, g.k_off, g.k_ld, g.S)], -1)
            E2 = 2 * g.heads * g.hd
            sp2 = GemmSpec(g.L, g.S, 2 * g.hd, g.n, g.heads, View(0, E2, False, g.L * E2, 2 * g.hd),
                           View(0, E2, True, g.S * E2, 2 * g.hd), sp_s.C, sp_s.out_shape, g.scale)
            G1 = _run_gemm(a2, b2, None, sp2, fill=False)
            del a2, b2
        else:
            G1 = _run_gemm(hq, k, None, sp_s, fill=False) if hq is not None else (_run_gemm(q, hk, None, sp_s, fill=False) if hk is not None else None)
        HD = _run_gemm(do, hv, None, _spec_dA(sp_a, tuple(y.shape)), fill=False) if hv is not None else None
        HgD, HS = torch.empty_like(y), torch.empty_like(y)
        nul = lambda t: t.data_ptr() if t is not None else None
        _chk(_L().ix_attn_prob_bwd_bwd_f32(nul(G1), nul(G2), y.data_ptr(), gd.data_ptr(), nul(HD), HgD.data_ptr(),
                                           HS.data_ptr(), rows, g.S, Sp, p, seed, _stream()), "ix_attn_prob_bwd_bwd_f32")
        del G1, G2, HD
        s_q = _spec_dA(sp_s, tuple(q.shape))
        s_k, kf = _spec_dB(sp_s, tuple(k.shape))
        s_v, vf = _spec_dB(sp_a, tuple(v.shape))
        rk = lambda dc, a: _run_gemm(dc, a, None, s_k) if kf else _run_gemm(a, dc, None, s_k)
        need = ctx.needs_input_grad
        grad_q = grad_k = grad_v = grad_do = None
        if gs is None and (need[0] or need[1]):
            gs = torch.empty_like(y)
            _chk(_L().ix_attn_prob_bwd_f32(y.data_ptr(), gd.data_ptr(), gs.data_ptr(), rows, g.S, Sp, p, seed, _stream()),
                 "ix_attn_prob_bwd_f32")
        if need[0]:   # scale (gs hk + HS k)
            grad_q = _sum2(_run_gemm(gs, hk, None, s_q) if hk is not None else None, _run_gemm(HS, k, None, s_q))
        if need[1]:   # scale (gs^T hq + HS^T q)
            grad_k = _sum2(rk(gs, hq) if hq is not None else None, rk(HS, q))
        if need[2]:   # HgD^T dO
            grad_v = _run_gemm(do, HgD, None, s_v) if vf else _run_gemm(HgD, do, None, s_v)
        del gs, HS
        if need[5]:   # d hv + HgD v
            if d is None and hv is not None:
                d = _dropout_of(y, p, seed)
            grad_do = _sum2(_run_gemm(d, hv, None, sp_a) if hv is not None else None, _run_gemm(HgD, v, None, sp_a))
        return grad_q, grad_k, grad_v, None, None, grad_do, None, None, None, None


# ---------------------------------------------------------------------------------------------------------
# flash-style attention core (csrc/flash.hip): no [L, S] tensor in HBM
# ---------------------------------------------------------------------------------------------------------
# "flash": csrc/flash.hip (no [L, S] tensor in HBM; head dims 32 / 64); "materialised": the AttentionCore node above (scores
# and probabilities as [n, H, L, S] tensors).  IX_ATTENTION in the environment overrides the default (A/B runs).
import os as _os
ATTENTION_IMPL = _os.environ.get("IX_ATTENTION", "flash")
# "fp32" (default, the parity path): fp32-grade arithmetic everywhere.  "fp8": the two products of the flash forward kernel on OCP
# e4m3 operands in calls that are not differentiated (predict; BASELINE.json configs[4], the 1600 / 200-query stress configuration);
# differentiated calls stay fp32-grade throughout (flash_forward says why).
ATTENTION_DTYPE = _os.environ.get("IX_ATTENTION_DTYPE", "fp32")


def _pad128(R):
    return (R + 127) // 128 * 128


class _PlanesC(ctypes.Structure):   # struct ix_attn_planes of include/interactron_hip.h
    _fields_ = [("row", ctypes.c_void_p), ("unscale", ctypes.c_void_p), ("tr", ctypes.c_void_p), ("tr_form", ctypes.c_int)]


# How the flash kernels run their token-contracting products (P v, dS k, ... and the second-order ones): "f16" = tr form 1, two
# fp16 planes and three matrix instructions per k-slice, the [L, S] intermediates scaled into fp16 range in registers (default
# since round 3); "bf16" = tr form 0, three bf16 planes and six instructions.  Both carry the parity record (tests/conftest.py
# kernel_form).  Read when an operand is split; the derivative passes follow the form their forward was split with.
FLASH_TR = _os.environ.get("IX_FLASH_TR", "f16")
FLASH_SPLIT_DOT = _os.environ.get("IX_FLASH_SPLIT_DOT", "1") == "1"   # "0": delta = dO . O by its own launch (A/B runs)
FLASH_NOBIAS = _os.environ.get("IX_FLASH_NOBIAS", "1") == "1"   # "0": always hand the kernels a key-bias tensor (A/B runs)
_TR_FORMS = {"bf16": 0, "f16": 1}


def flash_m16(on=None):
    """Head dim 64 in the fp16 form runs the 16x16x32 passes of csrc/flash16.hip (row planes only, no tr planes are written);
    ``flash_m16(False)`` switches back to the 32x32x16 passes of csrc/flash.hip (A/B runs, and the tests pin both).  Returns
    the previous setting.  Operands split under one setting must be consumed under the same one."""
    return bool(_L().ix_flash_set_m16(-1 if on is None else int(bool(on))))


def _rows_only(hd, form):
    return hd == 64 and form == 1 and flash_m16()


class AttnPlanes:
    """One attention operand as the flash kernels read it: fp16 row planes [2][n*H][Rp][hd] with their block unscale factors
    [n*H][Rp/32], and tr planes -- bf16 [3][n*H][hd][Rp] (form 0) or fp16 [2][n*H][hd][Rp] (form 1) (csrc/flash.hip);
    ``ref`` is the C view handed to the library."""

    def __init__(self, row, unscale, tr, tr_form=0):
        self.row, self.unscale, self.tr, self.tr_form = row, unscale, tr, tr_form
        self.c = _PlanesC(row.data_ptr() if row is not None else None, unscale.data_ptr() if unscale is not None else None,
                          tr.data_ptr() if tr is not None else None, tr_form)
        self.ref = ctypes.byref(self.c)


def attn_split(x, n, R, ld, off, H, hd, row=True, tr=True, tr_form=None, dot=None):
    """fp32 activations [n, R, ld] (head h at columns off + h*hd) -> AttnPlanes (Rp = R rounded up to 128).
    dot = (y, ldy, offy): also t[n*H, Rp] = sum_d x[.., h, d] * y[.., h, d] from the same read of x -> (AttnPlanes, t)."""
    in16 = x.dtype == torch.bfloat16   # (16-bit activation mode: the same planes from bf16 values, ix_attn_split_*_b16)
    x = _req(x, "attention operand") if not in16 else (x if x.is_contiguous() else x.contiguous())
    Rp = _pad128(R)
    dev = x.device
    form = _TR_FORMS[FLASH_TR] if tr_form is None else tr_form
    if _rows_only(hd, form):
        row, tr = row or tr, False
    rowp = torch.empty(2 * n * H * Rp * hd, dtype=torch.float16, device=dev) if row else None
    us = torch.empty(n * H * (Rp // 32), dtype=torch.float32, device=dev) if row or (tr and form == 1) else None
    trp = None
    if tr:
        trp = (torch.empty(2 * n * H * Rp * hd, dtype=torch.float16, device=dev) if form == 1 else
               torch.empty(3 * n * H * Rp * hd, dtype=torch.bfloat16, device=dev))
    if dot is not None:
        y, ldy, offy = dot
        t = torch.empty(n * H, Rp, dtype=torch.float32, device=dev)
        fn = _L().ix_attn_split_dot_b16 if in16 else _L().ix_attn_split_dot_f32
        _chk(fn(x.data_ptr(), rowp.data_ptr() if row else None, us.data_ptr() if us is not None else None,
                trp.data_ptr() if tr else None, form, n, R, Rp, ld, off, H, hd, _req(y).data_ptr(), ldy, offy,
                t.data_ptr(), _stream()), "ix_attn_split_dot")
        return AttnPlanes(rowp, us, trp, form), t
    if in16:
        return attn_split_multi([(x, R, ld, off, row, tr)], n, H, hd, tr_form=form)[0]
    _chk(_L().ix_attn_split_f32(x.data_ptr(), rowp.data_ptr() if row else None, us.data_ptr() if us is not None else None,
                                trp.data_ptr() if tr else None, form, n, R, Rp, ld, off, H, hd, _stream()), "ix_attn_split_f32")
    return AttnPlanes(rowp, us, trp, form)


def attn_split_multi(ops, n, H, hd, tr_form=None):
    """attn_split for up to three operands of one attention call in ONE launch.  ops: [(x, R, ld, off, row, tr), ...]."""
    form = _TR_FORMS[FLASH_TR] if tr_form is None else tr_form
    cnt = len(ops)
    xs, rows, uss, trs = [], [], [], []
    rows_only = _rows_only(hd, form)
    in16 = ops[0][0].dtype == torch.bfloat16
    assert all((o[0].dtype == torch.bfloat16) == in16 for o in ops), "the operands of one attention call share a storage dtype"
    for x, R, ld, off, row, tr in ops:
        if rows_only:
            row, tr = row or tr, False
        x = _req(x, "attention operand") if not in16 else (x if x.is_contiguous() else x.contiguous())
        Rp, dev = _pad128(R), x.device
        xs.append(x)
        rows.append(torch.empty(2 * n * H * Rp * hd, dtype=torch.float16, device=dev) if row else None)
        uss.append(torch.empty(n * H * (Rp // 32), dtype=torch.float32, device=dev) if row or (tr and form == 1) else None)
        trs.append(None if not tr else torch.empty(2 * n * H * Rp * hd, dtype=torch.float16, device=dev) if form == 1 else
                   torch.empty(3 * n * H * Rp * hd, dtype=torch.bfloat16, device=dev))
    ptr = lambda ts: (ctypes.c_void_p * cnt)(*[t.data_ptr() if t is not None else None for t in ts])
    ints = lambda vs: (ctypes.c_int * cnt)(*vs)
    fn = _L().ix_attn_split_multi_b16 if in16 else _L().ix_attn_split_multi_f32
    _chk(fn(cnt, ptr(xs), ptr(rows), ptr(uss), ptr(trs), form, n, ints([o[1] for o in ops]),
            ints([_pad128(o[1]) for o in ops]), (ctypes.c_int64 * cnt)(*[o[2] for o in ops]),
            ints([o[3] for o in ops]), H, hd, _stream()), "ix_attn_split_multi")
    return [AttnPlanes(r, u, t, form) for r, u, t in zip(rows, uss, trs)]


def attn_bias(mask, n, S, device):
    """additive key bias [n, Sp]: 0 valid / -inf masked or tail (Sp = S rounded up to 128); mask uint8 [n, S] or None.
    Without a mask the bias depends on (n, S) only and is kept (the GPT fusion asks for the same one in every layer of
    every step); with a mask it is kept for as long as the SAME mask tensor (address, version) is presented -- the six
    encoder and six decoder layers of one detector pass share one key_padding_mask."""
    Sb = _pad128(S)
    key = (device.index, n, S) if mask is None else (device.index, n, S, mask.data_ptr(), mask._version)
    if _capture[0] is not None:   # inside a graph capture: a private cache that dies with the capture (see capture_begin)
        hit = _capture[0].get(key)
        if hit is None:
            bias = torch.empty(n, Sb, dtype=torch.float32, device=device)
            _chk(_L().ix_attn_bias_f32(mask.data_ptr() if mask is not None else None, bias.data_ptr(), n, S, Sb,
                                       mask.shape[-1] if mask is not None else 0, _stream()), "ix_attn_bias_f32")
            hit = _capture[0][key] = (bias, mask)
        return hit[0]
    hit = _bias_cache.get(key)
    if hit is not None and (mask is None or hit[1] is mask):
        return hit[0]
    bias = torch.empty(n, Sb, dtype=torch.float32, device=device)
    _chk(_L().ix_attn_bias_f32(mask.data_ptr() if mask is not None else None, bias.data_ptr(), n, S, Sb,
                               mask.shape[-1] if mask is not None else 0, _stream()), "ix_attn_bias_f32")
    if mask is not None:   # one masked entry at a time (masks change with every batch)
        for k in [k for k in _bias_cache if len(k) == 5]:
            del _bias_cache[k]
    _bias_cache[key] = (bias, mask)
    return bias


def flash_dropmask(BH, L, S, p, seed, device="cuda"):
    """The flash kernels' dropout mask as a tensor [BH, L, S] (1/keep or 0) -- for tests."""
    m = torch.empty(BH, L, S, dtype=torch.float32, device=device)
    _chk(_L().ix_flash_dropmask_f32(m.data_ptr(), BH, L, S, p, seed, _stream()), "ix_flash_dropmask_f32")
    return m


def attn_split_fp8(x, n, R, ld, off, H, hd, row=True, tr=False):
    """fp32 activations -> one OCP e4m3 plane in row and / or tr layout + block unscale factors (csrc/flash.hip, fp8 forward)."""
    x = _req(x, "attention operand")
    Rp = _pad128(R)
    rowp = torch.empty(n * H * Rp * hd, dtype=torch.uint8, device=x.device) if row else None
    trp = torch.empty(n * H * Rp * hd, dtype=torch.uint8, device=x.device) if tr else None
    us = torch.empty(n * H * (Rp // 32), dtype=torch.float32, device=x.device)
    _chk(_L().ix_attn_split_fp8_f32(x.data_ptr(), rowp.data_ptr() if row else None, trp.data_ptr() if tr else None,
                                    us.data_ptr(), n, R, Rp, ld, off, H, hd, _stream()), "ix_attn_split_fp8_f32")
    return rowp, trp, us


def _bias_ptr(pl):
    return pl["bias"].data_ptr() if pl["bias"] is not None else None


def flash_forward(q, k, v, g, mask, p, seed, need_backward=True, dtype=None):
    """-> (out [n, L, H*hd], lse [n*H, Lp] (+inf beyond L), operand planes) for geometry g (AttnGeom).
    dtype "fp8" (ATTENTION_DTYPE): the two forward products on e4m3 operands -- for calls that are NOT differentiated
    (need_backward False: predict / no_grad).  A differentiated call runs the fp32-grade forward whatever the switch says: the
    derivative passes recompute the probabilities from the fp16 planes and take delta = dO . O from the saved output, and an
    output of other probabilities (fp8: 6 % per element) breaks the identity sum_j dS_ij = 0 that cancels the component all keys
    of a row share -- with DETR's biased key projections that component dominates, and the gradients of a training step came out
    with norms 30 x off and whole-gradient cosine 0.175 against the oracle (profiles/r6k_16_bit_step_survey.txt, round 6; the op
    test on zero-mean Gaussian keys had not shown it).  A consistent output costs the fp32-grade forward itself, so that is what
    runs."""
    dev = q.device
    dtype = dtype or ATTENTION_DTYPE
    fp8 = dtype == "fp8" and not need_backward
    # no key mask + the head-dim-64 fp16-form kernels (csrc/flash16.hip): no bias tensor at all (NULL: they blank the keys beyond
    # S of the last tile themselves and skip the bias loads / adds of every other tile)
    no_bias = FLASH_NOBIAS and mask is None and not fp8 and _rows_only(g.hd, _TR_FORMS[FLASH_TR])
    pl = {"bias": None if no_bias else attn_bias(mask, g.n, g.S, dev)}
    if not fp8:
        pl["q"], pl["k"], pl["v"] = attn_split_multi([(q, g.L, g.q_ld, g.q_off, True, need_backward),
                                                      (k, g.S, g.k_ld, g.k_off, True, need_backward),
                                                      (v, g.S, g.v_ld, g.v_off, need_backward, True)], g.n, g.heads, g.hd)
    Lp, Sp, E = _pad128(g.L), _pad128(g.S), g.heads * g.hd
    out = torch.empty(g.n, g.L, E, dtype=torch.float32, device=dev)
    lse = torch.empty(g.n * g.heads, Lp, dtype=torch.float32, device=dev)   # (rows L..Lp come back as +inf: P = 0 there)
    if fp8:
        q8, _, qus = attn_split_fp8(q, g.n, g.L, g.q_ld, g.q_off, g.heads, g.hd)
        k8, _, kus = attn_split_fp8(k, g.n, g.S, g.k_ld, g.k_off, g.heads, g.hd)
        _, v8, vus = attn_split_fp8(v, g.n, g.S, g.v_ld, g.v_off, g.heads, g.hd, row=False, tr=True)
        _chk(_L().ix_flash_fwd_fp8_f32(q8.data_ptr(), qus.data_ptr(), k8.data_ptr(), kus.data_ptr(), v8.data_ptr(), vus.data_ptr(),
                                       pl["bias"].data_ptr(), out.data_ptr(), lse.data_ptr(), g.n, g.heads, g.L, Lp, g.S, Sp,
                                       g.hd, E, 0, g.scale, p, seed, _stream()), "ix_flash_fwd_fp8_f32")
        return out, lse, pl
    _chk(_L().ix_flash_fwd_f32(pl["q"].ref, pl["k"].ref, pl["v"].ref, _bias_ptr(pl), out.data_ptr(), lse.data_ptr(),
                               g.n, g.heads, g.L, Lp, g.S, Sp, g.hd, E, 0, g.scale, p, seed, _stream()), "ix_flash_fwd_f32")
    return out, lse, pl


def flash_supported(g):
    return g.hd in (32, 64) and g.q_ld % 4 == 0 and g.k_ld % 4 == 0 and g.v_ld % 4 == 0 and g.q_off % 4 == 0 \
        and g.k_off % 4 == 0 and g.v_off % 4 == 0 and g.n * g.heads <= 65535


class FlashAttention(Function):
    """out[b,l,h*hd+:] = dropout(softmax(scale q k^T [+ key mask])) v per (batch, head) without [L, S] tensors in HBM
    (csrc/flash.hip).  Saves q, k, v, out, the row normalisers and the 16-bit operand planes."""

    @staticmethod
    def forward(ctx, q, k, v, g, mask, p, seed):
        q, k, v = _req(q, "attention q"), _req(k, "attention k"), _req(v, "attention v")
        return FlashAttention._forward(ctx, q, k, v, g, mask, p, seed)

    @staticmethod
    def _forward(ctx, q, k, v, g, mask, p, seed):
        """(shared with the 16-bit twin, b16.FlashAttention16: q / k / v fp32 or bf16; the kernels write the output in fp32)"""
        # (a derivative can follow only through a real context with an operand that asks for one)
        recorded = not isinstance(ctx, _NullCtx) and any(ctx.needs_input_grad[:3])
        out, lse, pl = flash_forward(q, k, v, g, mask, p, seed, need_backward=recorded)
        ctx.g, ctx.p, ctx.seed, ctx.pl = g, p, seed, pl
        # packed projection buffers: [q | k] (nn.MultiheadAttention self-attention) or [k | q | v] (fusion blocks) in one tensor.
        # ONE gradient buffer serves the operands of a shared tensor only when their column ranges are disjoint and cover the
        # rows (the two packed layouts); the same tensor passed with overlapping columns (attention(x, x, x) with equal
        # offsets) gets separate buffers, which autograd then sums.
        E_ = g.heads * g.hd
        alias_qk = q.data_ptr() == k.data_ptr() and q.shape == k.shape
        alias_qv = q.data_ptr() == v.data_ptr() and q.shape == v.shape
        packed3 = alias_qk and alias_qv and g.q_ld == 3 * E_ and g.k_ld == 3 * E_ and g.v_ld == 3 * E_ \
            and sorted((g.q_off, g.k_off, g.v_off)) == [0, E_, 2 * E_]
        packed2 = alias_qk and not alias_qv and g.q_ld == 2 * E_ and g.k_ld == 2 * E_ and sorted((g.q_off, g.k_off)) == [0, E_]
        ctx.same_qk = (packed2 or packed3, packed3)
        ctx.save_for_backward(q, k, v, out, lse)
        return out

    @staticmethod
    def backward(ctx, do):
        q, k, v, out, lse = ctx.saved_tensors
        gq, gk, gv = FlashAttentionBwd.call(q, k, v, out, lse, do, ctx.g, ctx.p, ctx.seed, ctx.pl, ctx.same_qk)
        return gq, gk, gv, None, None, None, None


def _grad_buffers(g, q, k, v, same):
    """Gradient buffers in the operands' own (packed) layouts; columns that belong to other tensors stay zero.
    same = (k shares q's tensor, v shares q's tensor): shared tensors get ONE buffer."""
    same_qk, same_qv = same
    E, dev = g.heads * g.hd, q.device
    full = lambda ld, off: ld == E and off == 0
    packed2 = same_qk and not same_qv and g.q_ld == 2 * E and sorted((g.q_off, g.k_off)) == [0, E]   # [q | k]: fully covered
    packed3 = same_qk and same_qv and g.q_ld == 3 * E and sorted((g.q_off, g.k_off, g.v_off)) == [0, E, 2 * E]
    gq = (torch.empty if packed2 or packed3 or (full(g.q_ld, g.q_off) and not (same_qk or same_qv)) else torch.zeros)(
        q.shape, dtype=torch.float32, device=dev)
    gk = gq if same_qk else (torch.empty if full(g.k_ld, g.k_off) else torch.zeros)(k.shape, dtype=torch.float32, device=dev)
    gv = gq if same_qv else (torch.empty if full(g.v_ld, g.v_off) else torch.zeros)(v.shape, dtype=torch.float32, device=dev)
    return gq, gk, gv


class FlashAttentionBwd(Function):
    """(gq, gk, gv) of FlashAttention; gq / gk come back in the layout of the packed q / k projection buffers (one shared
    buffer when q and k are the same tensor: autograd then has nothing to add)."""

    @staticmethod
    def forward(ctx, q, k, v, out, lse, do, g, p, seed, pl, same_qk):
        do = _req(do.contiguous(), "attention dO")
        return FlashAttentionBwd._forward(ctx, q, k, v, out, lse, do, g, p, seed, pl, same_qk)

    @staticmethod
    def _forward(ctx, q, k, v, out, lse, do, g, p, seed, pl, same_qk):
        """(shared with the 16-bit twin: q / k / v / dO fp32 or bf16, `out` and the gradients fp32)"""
        dev = q.device
        Lp, Sp, E = _pad128(g.L), _pad128(g.S), g.heads * g.hd
        if FLASH_SPLIT_DOT:   # the planes of dO and delta = dO . O (per query and head) from one read of dO
            dop, delta = attn_split(do, g.n, g.L, E, 0, g.heads, g.hd, tr_form=pl["q"].tr_form, dot=(out, E, 0))
        else:
            dop = attn_split(do, g.n, g.L, E, 0, g.heads, g.hd, tr_form=pl["q"].tr_form)
            delta = torch.empty(g.n * g.heads, Lp, dtype=torch.float32, device=dev)
            _chk(_L().ix_attn_rowdot_f32(do.data_ptr(), out.data_ptr(), delta.data_ptr(), g.n, g.heads, g.L, Lp, g.hd, E, 0, E, 0,
                                         _stream()), "ix_attn_rowdot_f32")
        gq, gk, gv = _grad_buffers(g, q, k, v, same_qk)
        _chk(_L().ix_flash_bwd_f32(pl["q"].ref, pl["k"].ref, pl["v"].ref, dop.ref, _bias_ptr(pl), lse.data_ptr(),
                                   delta.data_ptr(), gq.data_ptr(), gk.data_ptr(), gv.data_ptr(), g.n, g.heads, g.L, Lp, g.S, Sp,
                                   g.hd, g.q_ld, g.q_off, g.k_ld, g.k_off, g.v_ld, g.v_off, g.scale, p, seed, _stream()),
             "ix_flash_bwd_f32")
        ctx.g, ctx.p, ctx.seed, ctx.same_qk = g, p, seed, same_qk
        ctx.pl = dict(pl, do=dop, delta=delta)
        ctx.save_for_backward(q, k, v, out, lse, do)
        # a shared buffer carries the gradients of everything packed in it: hand it to q, nothing to the others
        return gq, (None if same_qk[0] else gk), (None if same_qk[1] else gv)

    @staticmethod
    @once_differentiable
    def backward(ctx, hq, hk, hv):
        """Double backward: (dq, dk, dv, ddO) for the cotangents of (gq, gk, gv); three passes of csrc/flash.hip."""
        q, k, v, out, lse, do = ctx.saved_tensors
        g, pl, dev = ctx.g, ctx.pl, q.device
        Lp, Sp, E = _pad128(g.L), _pad128(g.S), g.heads * g.hd
        zeros = lambda t: torch.zeros(t.shape, dtype=torch.float32, device=dev)
        hq = _req(hq.contiguous()) if hq is not None else zeros(q)
        if ctx.same_qk[0]:
            hk = hq          # one cotangent buffer for the packed gradient buffer
        else:
            hk = _req(hk.contiguous()) if hk is not None else zeros(k)
        if ctx.same_qk[1]:
            hv = hq
        else:
            hv = _req(hv.contiguous()) if hv is not None else zeros(v)
        return FlashAttentionBwd._backward_impl(ctx, hq, hk, hv)

    @staticmethod
    def _backward_impl(ctx, hq, hk, hv):
        """(shared with the 16-bit twin: cotangents fp32 or bf16, all of one dtype; the results are fp32)"""
        q, k, v, out, lse, do = ctx.saved_tensors
        g, pl, dev = ctx.g, ctx.pl, q.device
        Lp, Sp, E = _pad128(g.L), _pad128(g.S), g.heads * g.hd
        hqp, hkp, hvp = attn_split_multi([(hq, g.L, g.q_ld, g.q_off, True, True), (hk, g.S, g.k_ld, g.k_off, True, True),
                                          (hv, g.S, g.v_ld, g.v_off, True, True)], g.n, g.heads, g.hd, tr_form=pl["q"].tr_form)
        dq, dk, dv = _grad_buffers(g, q, k, v, ctx.same_qk)
        ddo = torch.empty(g.n, g.L, E, dtype=torch.float32, device=dev)
        need = ctypes.c_size_t()
        _chk(_L().ix_workspace_bytes_flash_bwd_bwd(g.n, g.heads, g.L, ctypes.byref(need)), "ix_workspace_bytes_flash_bwd_bwd")
        ws = torch.empty(need.value // 4, dtype=torch.float32, device=dev)
        _chk(_L().ix_flash_bwd_bwd_f32(pl["q"].ref, pl["k"].ref, pl["v"].ref, pl["do"].ref, hqp.ref, hkp.ref, hvp.ref,
                                       _bias_ptr(pl), lse.data_ptr(), pl["delta"].data_ptr(), dq.data_ptr(),
                                       dk.data_ptr(), dv.data_ptr(), ddo.data_ptr(), g.n, g.heads, g.L, Lp, g.S, Sp, g.hd,
                                       g.q_ld, g.q_off, g.k_ld, g.k_off, g.v_ld, g.v_off, E, 0, g.scale, ctx.p, ctx.seed,
                                       ws.data_ptr(), need.value, _stream()), "ix_flash_bwd_bwd_f32")
        need_in = ctx.needs_input_grad
        return (dq if need_in[0] else None, (None if ctx.same_qk[0] else dk) if need_in[1] else None,
                (None if ctx.same_qk[1] else dv) if need_in[2] else None,
                None, None, ddo if need_in[5] else None, None, None, None, None, None)


def attention(q, k, v, nbatch, heads, L, S, hd, q_ld, k_ld, q_off, k_off, v_ld, v_off, scale, mask, p, training):
    """Scaled-dot-product attention out of packed projection buffers (see attention_scores / attention_apply for the
    layouts); `mask`: optional uint8 key-padding mask [nbatch, S]."""
    p = float(p) if training else 0.0
    g = AttnGeom(nbatch, heads, L, S, hd, q_ld, k_ld, q_off, k_off, v_ld, v_off, float(scale))
    seed = core._next_seed() if p > 0.0 else 0
    if ATTENTION_IMPL == "flash" and flash_supported(g):
        return FlashAttention.call(q, k, v, g, mask, p, seed)
    return AttentionCore.call(q, k, v, g, mask, p, seed)
