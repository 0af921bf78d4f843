"""Hungarian matcher and DETR set criterion on the HIP kernels.

reference: models/detr_models/matcher.py:12-81 (HungarianMatcher), models/detr_models/detr.py:86-265 (SetCriterion),
models/detr_models/util/box_ops.py.  Same constructor arguments, call signature, return types (matcher indices are
CPU int64 tensors) and loss keys / order.

Everything between the predictions and the loss scalars stays on the device: the targets of all images of a call are one
CSR list (``hipops.Targets``), one launch builds every cost matrix, one launch solves every assignment (one wavefront
per image, the algorithm and tie-breaking of scipy / ``ix_lsap_f32``), and ``hipops.SetLoss`` evaluates the losses of
whole groups of images -- the reference's host LSAP in the middle of the step (matcher.py:73-76) is gone.  The host
assignment ``HungarianMatcher.assign`` remains as the fallback for sides beyond 256 and as the hook tests use to pin
assignments (a replaced ``assign`` routes the matching through the host).
"""
import torch
from torch import nn

from . import hipops as ops


class HungarianMatcher(nn.Module):
    def __init__(self, cost_class=1.0, cost_bbox=1.0, cost_giou=1.0):
        super().__init__()
        self.cost_class, self.cost_bbox, self.cost_giou = cost_class, cost_bbox, cost_giou
        assert cost_class != 0 or cost_bbox != 0 or cost_giou != 0, "all costs cant be 0"

    # ---- device route ---------------------------------------------------------------------------------------------
    @torch.no_grad()
    def match(self, outputs, tg):
        """tgt_of_q int32 [I, Q] on the device (image-local target index matched to each query, -1 = none) for the
        images of ``tg`` (hipops.Targets).  No host round trip unless ``assign`` has been replaced (tests pin assignments
        there) or a side exceeds what the one-wavefront kernel takes."""
        logits, boxes = outputs["pred_logits"], outputs["pred_boxes"]
        I, Q = logits.shape[:2]
        assert I == tg.I, (I, tg.I)
        host = type(self).assign is not HungarianMatcher._host_assign or Q > ops.LSAP_DEVICE_MAX or tg.ldn > ops.LSAP_DEVICE_MAX
        cost = ops.match_cost_csr(logits, boxes, tg, float(self.cost_class), float(self.cost_bbox), float(self.cost_giou))
        if not host:
            return ops.lsap_device(cost, tg)[0]
        c = cost.cpu()   # (synchronises: test / fallback route only)
        costs = [c[i, :, :n].contiguous() for i, n in enumerate(tg.sizes)]
        return indices_to_device(self.assign(costs, tg.as_list()), Q)

    # ---- host pieces (reference surface, fallback, test hook) ------------------------------------------------------
    @torch.no_grad()
    def begin(self, outputs, targets):
        """Enqueue the cost kernel (reference matcher.py:54-73: one launch builds the whole [bs*Q, sum(N_i)] matrix) and
        its D2H copy into pinned memory; returns a handle for ``finish`` (host assignment)."""
        bs, num_queries = outputs["pred_logits"].shape[:2]
        sizes = [len(v["boxes"]) for v in targets]
        if sum(sizes) == 0:
            return None, None, sizes, bs, num_queries, targets
        tgt_ids = torch.cat([v["labels"] for v in targets])
        tgt_bbox = torch.cat([v["boxes"] for v in targets])
        C = ops.match_cost(outputs["pred_logits"].flatten(0, 1), outputs["pred_boxes"].flatten(0, 1), tgt_ids, tgt_bbox,
                           float(self.cost_class), float(self.cost_bbox), float(self.cost_giou))
        host = torch.empty(C.shape, dtype=C.dtype, pin_memory=True)
        host.copy_(C, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        return host, ev, sizes, bs, num_queries, targets

    @torch.no_grad()
    def costs(self, handle):
        """Per-image cost matrices [Q, N_i] as CPU tensors (waits for the copy started by ``begin``)."""
        host, ev, sizes, bs, num_queries, _ = handle
        if host is None:
            return [torch.empty(num_queries, 0) for _ in sizes]
        ev.synchronize()
        C = host.view(bs, num_queries, -1)
        return [c[i].contiguous() for i, c in enumerate(C.split(sizes, -1))]

    @torch.no_grad()
    def assign(self, costs, targets):
        """Host assignment per image (reference matcher.py:76: scipy.optimize.linear_sum_assignment)."""
        empty = torch.empty(0, dtype=torch.int64)
        return [ops.lsap(c) if c.shape[1] else (empty.clone(), empty.clone()) for c in costs]

    _host_assign = assign

    def finish(self, handle):
        return self.assign(self.costs(handle), handle[5])

    @torch.no_grad()
    def cost_matrices(self, outputs, targets):
        return self.costs(self.begin(outputs, targets))

    @torch.no_grad()
    def forward(self, outputs, targets):
        """-> list of (query indices, target indices) CPU int64 tensors per image, sorted by query (reference matcher.py:77),
        computed by the device assignment."""
        tg = ops.pack_targets(targets)
        toq = self.match(outputs, tg).cpu()
        out = []
        for i in range(tg.I):
            src = torch.nonzero(toq[i] >= 0).reshape(-1)
            out.append((src.to(torch.int64), toq[i][src].to(torch.int64)))
        return out


def _as_list(self):
    t = getattr(self, "targets", None)
    assert t is not None, "this Targets pack does not carry its per-image dicts"
    return t


ops.Targets.as_list = _as_list


def indices_to_device(indices, Q):
    """[(query idx, target idx)] per image (CPU int64) -> tgt_of_q int32 [I, Q] on the device (one pinned upload)"""
    toq = torch.full((len(indices), Q), -1, dtype=torch.int32)
    for i, (src, tgt) in enumerate(indices):
        if len(src):
            toq[i, src] = tgt.to(torch.int32)
    return ops.h2d_async(toq)


def build_matcher(args):
    return HungarianMatcher(cost_class=args.SET_COST_CLASS, cost_bbox=args.SET_COST_BBOX, cost_giou=args.SET_COST_GIOU)


LOSS_COLUMNS = {"labels": (("loss_ce", 0), ("class_error", 1)), "boxes": (("loss_bbox", 2), ("loss_giou", 3)),
                "cardinality": (("cardinality_error", 4),)}


class SetCriterion(nn.Module):
    def __init__(self, num_classes, matcher, weight_dict, eos_coef, losses):
        super().__init__()
        self.num_classes, self.matcher, self.weight_dict, self.eos_coef, self.losses = \
            num_classes, matcher, weight_dict, eos_coef, losses
        for loss in losses:
            assert loss in LOSS_COLUMNS, "do you really want to compute %s loss?" % loss
        empty_weight = torch.ones(self.num_classes + 1)
        empty_weight[-1] = self.eos_coef
        self.register_buffer("empty_weight", empty_weight)

    def grouped(self, outputs, tg, tgt_of_q, specs, background_c=0.1):
        """Losses of image groups in one pass: ``specs`` = ((stride, len), ...) -> one [G, 5] tensor per spec (columns
        loss_ce, class_error, loss_bbox, loss_giou, cardinality_error); see hipops.SetLoss.  The class weights are the
        reference's (detr.py:124-126): 1 everywhere, ``background_c`` for the no-object class."""
        return ops.SetLoss.apply(outputs["pred_logits"], outputs["pred_boxes"], tg, tgt_of_q, float(background_c), tuple(specs))

    def as_dict(self, row):
        """one [5] row of a grouped result -> the reference's loss dict (keys / order of detr.py:253-255)"""
        return {name: row[col] for loss in self.losses for name, col in LOSS_COLUMNS[loss]}

    def forward(self, outputs, targets, detector_out=None, background_c=0.1, indices=None):
        """``indices``: optional precomputed matcher output for exactly these images."""
        logits = outputs["pred_logits"]
        I, Q = logits.shape[:2]
        tg = ops.pack_targets(targets)
        if indices is not None:
            toq = indices_to_device(indices, Q)
        else:
            match_on = detector_out if detector_out is not None else outputs
            toq = self.matcher.match({k: v for k, v in match_on.items() if k != "aux_outputs"}, tg)
        (out,) = self.grouped(outputs, tg, toq, ((I, I),), background_c)
        return self.as_dict(out[0])
