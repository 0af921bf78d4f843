"""hipops.criterion_ops -- matcher cost, assignment, set criterion, policy cross-entropy, position embedding kernels."""
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
from .core import (Function, _L, _chk, _reduce_ws, _req, _stream, h2d_async)


# ---------------------------------------------------------------------------------------------------------
# criterion kernels
# ---------------------------------------------------------------------------------------------------------
def match_cost(logits, boxes, tgt_ids, tgt_boxes, w_class, w_bbox, w_giou):
    """Hungarian cost matrix [rows, T] (no grad)."""
    logits, boxes, tgt_boxes = _req(logits.detach()), _req(boxes.detach()), _req(tgt_boxes)
    rows, C = logits.shape
    T = tgt_ids.numel()
    cost = torch.empty(rows, T, device=logits.device, dtype=torch.float32)
    tgt_ids = tgt_ids.contiguous()
    assert tgt_ids.dtype == torch.int64 and tgt_ids.is_cuda
    _chk(_L().ix_match_cost_f32(logits.data_ptr(), boxes.data_ptr(), tgt_ids.data_ptr(), tgt_boxes.data_ptr(),
                                cost.data_ptr(), rows, C, T, w_class, w_bbox, w_giou, _stream()), "ix_match_cost_f32")
    return cost


def lsap(cost_cpu):
    """Host rectangular assignment on a CPU float32 [nr, nc] tensor -> (rows int64[k], cols int64[k])."""
    cost_cpu = cost_cpu.contiguous()
    assert not cost_cpu.is_cuda and cost_cpu.dtype == torch.float32
    nr, nc = cost_cpu.shape
    k = min(nr, nc)
    r = torch.empty(k, dtype=torch.int64)
    c = torch.empty(k, dtype=torch.int64)
    _chk(_L().ix_lsap_f32(cost_cpu.data_ptr(), nr, nc, r.data_ptr(), c.data_ptr()), "ix_lsap_f32")
    return r, c


# ---- device-resident matcher + set criterion (csrc/criterion.hip, second half) --------------------------------------------
class Targets:
    """Ground truth of I images as one CSR list on the device: ids int64 [T], boxes [T, 4], off int32 [I + 1]; ``sizes`` is
    the host copy of the per-image counts, ``ldn`` the column pitch of the cost matrices (max count, rounded up to 8)."""

    def __init__(self, ids, boxes, off, sizes):
        self.ids, self.boxes, self.off, self.sizes = ids, boxes, off, list(sizes)
        self.I = len(self.sizes)
        self.ldn = (max(self.sizes + [1]) + 7) // 8 * 8



def pack_targets(targets):
    """list of {"labels": int64 [n_i], "boxes": [n_i, 4]} (device tensors) -> Targets; one cat per field + one small upload"""
    sizes = [int(t["labels"].shape[0]) for t in targets]
    dev = targets[0]["boxes"].device if targets else torch.device("cuda")
    if sum(sizes) == 0:
        ids = torch.zeros(1, dtype=torch.int64, device=dev)
        boxes = torch.full((1, 4), 0.5, dtype=torch.float32, device=dev)
    else:
        ids = torch.cat([t["labels"] for t in targets]).contiguous()
        boxes = _req(torch.cat([t["boxes"] for t in targets]), "target boxes")
    off = [0]
    for n in sizes:
        off.append(off[-1] + n)
    tg = Targets(ids, boxes, h2d_async(torch.tensor(off, dtype=torch.int32)), sizes)
    tg.targets = targets   # (the per-image dicts: the host assignment route and the tests' pinning hook read them)
    return tg


LSAP_DEVICE_MAX = 256


def match_cost_csr(logits, boxes, tg, w_class, w_bbox, w_giou):
    """[I, Q, ldn] cost matrices of all images (columns beyond an image's target count are not written)"""
    I, Q, C = logits.shape
    logits, boxes = _req(logits.detach()), _req(boxes.detach())
    cost = torch.empty(I, Q, tg.ldn, device=logits.device, dtype=torch.float32)
    _chk(_L().ix_match_cost_csr_f32(logits.data_ptr(), boxes.data_ptr(), tg.ids.data_ptr(), tg.boxes.data_ptr(), tg.off.data_ptr(),
                                    cost.data_ptr(), I, Q, C, tg.ldn, w_class, w_bbox, w_giou, _stream()), "ix_match_cost_csr_f32")
    return cost


def lsap_device(cost, tg):
    """-> (tgt_of_q int32 [I, Q], q_of_tgt int32 [T]) -- scipy's assignment per image, computed on the GPU"""
    I, Q, ldn = cost.shape
    toq = torch.empty(I, Q, dtype=torch.int32, device=cost.device)
    qot = torch.empty(max(int(tg.ids.shape[0]), 1), dtype=torch.int32, device=cost.device)
    _chk(_L().ix_lsap_device_f32(cost.data_ptr(), tg.off.data_ptr(), I, Q, ldn, toq.data_ptr(), qot.data_ptr(), _stream()),
         "ix_lsap_device_f32")
    return toq, qot


class SetLoss(Function):
    """DETR set criterion of image groups: apply(logits [I, Q, C], boxes [I, Q, 4], tg, tgt_of_q, w_noobj, specs) with
    specs = ((stride, len), ...) -> one [G, 5] tensor per spec, G = I // stride, columns (loss_ce, class_error, loss_bbox,
    loss_giou, cardinality_error) of the group's images g * stride .. g * stride + len - 1, each with its own normalisers
    (reference detr.py:220-265 called once per group).  Only the FIRST spec is differentiable (the others are bookkeeping:
    the frame-0 reward of interactron.py:104-108)."""

    @staticmethod
    def forward(ctx, logits, boxes, tg, tgt_of_q, w_noobj, specs):
        logits, boxes = _req(logits, "criterion logits"), _req(boxes, "criterion boxes")
        I, Q, C = logits.shape
        dev = logits.device
        rowstat = torch.empty(I * Q, 4, dtype=torch.float32, device=dev)
        lse = torch.empty(I * Q, dtype=torch.float32, device=dev)
        flags = torch.empty(I * Q, dtype=torch.int32, device=dev)
        L = _L()
        _chk(L.ix_set_loss_rows_f32(logits.data_ptr(), boxes.data_ptr(), tg.ids.data_ptr(), tg.boxes.data_ptr(), tg.off.data_ptr(),
                                    tgt_of_q.data_ptr(), rowstat.data_ptr(), lse.data_ptr(), flags.data_ptr(), I, Q, C, w_noobj,
                                    _stream()), "ix_set_loss_rows_f32")
        outs, norm0 = [], None
        for k, (stride, ln) in enumerate(specs):
            assert I % stride == 0 and 1 <= ln <= stride, (I, stride, ln)
            G = I // stride
            out = torch.empty(G, 5, dtype=torch.float32, device=dev)
            norm = torch.empty(G, 2, dtype=torch.float32, device=dev)
            _chk(L.ix_set_loss_groups_f32(rowstat.data_ptr(), flags.data_ptr(), tg.off.data_ptr(), stride, ln, G, Q, out.data_ptr(),
                                          norm.data_ptr(), _stream()), "ix_set_loss_groups_f32")
            outs.append(out)
            if k == 0:
                norm0 = norm
        ctx.tg, ctx.w_noobj, ctx.spec0 = tg, w_noobj, specs[0]
        ctx.save_for_backward(logits, boxes, tgt_of_q, lse, norm0)
        for o in outs[1:]:
            ctx.mark_non_differentiable(o)
        return tuple(outs)

    @staticmethod
    @once_differentiable
    def backward(ctx, g0, *_):
        logits, boxes, tgt_of_q, lse, norm0 = ctx.saved_tensors
        tg = ctx.tg
        I, Q, C = logits.shape
        stride, ln = ctx.spec0
        g0 = _req(g0.contiguous())
        dl, db = torch.empty_like(logits), torch.empty_like(boxes)
        _chk(_L().ix_set_loss_bwd_f32(logits.data_ptr(), boxes.data_ptr(), tg.ids.data_ptr(), tg.boxes.data_ptr(), tg.off.data_ptr(),
                                      tgt_of_q.data_ptr(), lse.data_ptr(), g0.data_ptr(), norm0.data_ptr(), stride, ln, I, Q, C,
                                      ctx.w_noobj, dl.data_ptr(), db.data_ptr(), _stream()), "ix_set_loss_bwd_f32")
        return dl, db, None, None, None, None


class WeightedCE(Function):
    """F.cross_entropy(logits [R,C], target [R], weight [C]) with mean reduction; also returns per-row argmax."""

    @staticmethod
    def forward(ctx, logits, target, weight):
        logits, weight = _req(logits), _req(weight)
        R, C = logits.shape
        lse = torch.empty(R, device=logits.device, dtype=torch.float32)
        argmax = torch.empty(R, device=logits.device, dtype=torch.int64)
        sums = torch.empty(2, device=logits.device, dtype=torch.float32)
        target = target.contiguous()
        wp, wn = _reduce_ws("wce", R, 0, 0, logits.device)
        _chk(_L().ix_weighted_ce_fwd_f32(logits.data_ptr(), target.data_ptr(), weight.data_ptr(), lse.data_ptr(),
                                         argmax.data_ptr(), sums.data_ptr(), R, C, wp, wn, _stream()), "ix_weighted_ce_fwd_f32")
        ctx.save_for_backward(logits, target, weight, lse, sums)
        ctx.mark_non_differentiable(argmax)
        return sums[0] / sums[1], argmax

    @staticmethod
    @once_differentiable
    def backward(ctx, g, _):
        logits, target, weight, lse, sums = ctx.saved_tensors
        R, C = logits.shape
        d = torch.empty_like(logits)
        g = g.contiguous()
        _chk(_L().ix_weighted_ce_bwd_f32(logits.data_ptr(), target.data_ptr(), weight.data_ptr(), lse.data_ptr(),
                                         sums.data_ptr(), g.data_ptr(), d.data_ptr(), R, C, _stream()),
             "ix_weighted_ce_bwd_f32")
        return d, None, None


class BoxLoss(Function):
    """Sums of L1 and (1 - GIoU) over matched (prediction row, target box) pairs -> tensor [2]."""

    @staticmethod
    def forward(ctx, pred, src_idx, tgt):
        pred, tgt = _req(pred), _req(tgt)
        out = torch.empty(2, device=pred.device, dtype=torch.float32)
        src_idx = src_idx.contiguous()
        _chk(_L().ix_box_loss_fwd_f32(pred.data_ptr(), src_idx.data_ptr(), tgt.data_ptr(), out.data_ptr(),
                                      src_idx.numel(), _stream()), "ix_box_loss_fwd_f32")
        ctx.save_for_backward(pred, src_idx, tgt)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        pred, src_idx, tgt = ctx.saved_tensors
        d = torch.empty_like(pred)
        g = g.contiguous()
        _chk(_L().ix_box_loss_bwd_f32(pred.data_ptr(), src_idx.data_ptr(), tgt.data_ptr(), g.data_ptr(), d.data_ptr(),
                                      pred.shape[0], src_idx.numel(), _stream()), "ix_box_loss_bwd_f32")
        return d, None, None


def sine_position(mask_u8, num_pos_feats=128, temperature=10000.0, scale=6.283185307179586):
    """mask uint8 [n,h,w] (1 = padded) -> [n, h*w, 2*num_pos_feats] token-major position embedding."""
    n, h, w = mask_u8.shape
    pos = torch.empty(n, h * w, 2 * num_pos_feats, device=mask_u8.device, dtype=torch.float32)
    _chk(_L().ix_sine_pos_f32(mask_u8.data_ptr(), pos.data_ptr(), n, h, w, num_pos_feats, temperature, scale, _stream()),
         "ix_sine_pos_f32")
    return pos


def mask_nearest(mask_u8, h, w):
    n, H, W = mask_u8.shape
    out = torch.empty(n, h, w, device=mask_u8.device, dtype=torch.uint8)
    _chk(_L().ix_mask_nearest_u8(mask_u8.data_ptr(), out.data_ptr(), n, H, W, h, w, _stream()), "ix_mask_nearest_u8")
    return out
