"""Hungarian matcher and DETR set criterion on the HIP kernels.

reference: models/detr_models/matcher.py:12-81 (HungarianMatcher), models/detr_models/detr.py:86-265 (SetCriterion),
models/detr_models/util/box_ops.py.  Same constructor arguments, call signature, return types (matcher indices are
CPU int64 tensors) and loss keys / order.
"""
import torch
from torch import nn

from . import hipops as ops


class HungarianMatcher(nn.Module):
    def __init__(self, cost_class=1.0, cost_bbox=1.0, cost_giou=1.0):
        super().__init__()
        self.cost_class, self.cost_bbox, self.cost_giou = cost_class, cost_bbox, cost_giou
        assert cost_class != 0 or cost_bbox != 0 or cost_giou != 0, "all costs cant be 0"

    @torch.no_grad()
    def begin(self, outputs, targets):
        """Enqueue the cost kernel (reference matcher.py:54-73: one launch builds the whole [bs*Q, sum(N_i)] matrix) and
        its D2H copy into pinned memory; returns a handle for ``finish``.  Splitting the call lets the caller queue more
        GPU work behind the copy and run the host-side assignment while that work executes."""
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

    def finish(self, handle):
        return self.assign(self.costs(handle), handle[5])

    @torch.no_grad()
    def cost_matrices(self, outputs, targets):
        return self.costs(self.begin(outputs, targets))

    @torch.no_grad()
    def forward(self, outputs, targets):
        return self.finish(self.begin(outputs, targets))


def build_matcher(args):
    return HungarianMatcher(cost_class=args.SET_COST_CLASS, cost_bbox=args.SET_COST_BBOX, cost_giou=args.SET_COST_GIOU)


class SetCriterion(nn.Module):
    def __init__(self, num_classes, matcher, weight_dict, eos_coef, losses):
        super().__init__()
        self.num_classes, self.matcher, self.weight_dict, self.eos_coef, self.losses = \
            num_classes, matcher, weight_dict, eos_coef, losses
        empty_weight = torch.ones(self.num_classes + 1)
        empty_weight[-1] = self.eos_coef
        self.register_buffer("empty_weight", empty_weight)

    @staticmethod
    def _src_idx(indices):
        batch_idx = torch.cat([torch.full_like(src, i) for i, (src, _) in enumerate(indices)])
        return batch_idx, torch.cat([src for (src, _) in indices])

    def forward(self, outputs, targets, detector_out=None, background_c=0.1, indices=None):
        """``indices``: optional precomputed matcher output for exactly these images (the episode-batched step runs the
        matcher once over all episodes of a chunk: one cost kernel, one D2H, instead of one host sync per episode)."""
        if indices is None:
            match_on = detector_out if detector_out is not None else outputs
            indices = self.matcher({k: v for k, v in match_on.items() if k != "aux_outputs"}, targets)
        logits, boxes = outputs["pred_logits"], outputs["pred_boxes"]
        dev = logits.device
        bs, Q, C = logits.shape
        num_boxes = max(float(sum(len(t["labels"]) for t in targets)), 1.0)
        batch_idx, src_idx = self._src_idx(indices)
        # matched (query, target) pairs of all images in ONE non-blocking upload: row 0 = flat query index, row 1 =
        # index into the concatenated targets (reference detr.py:104-108,150-152 gathers image by image)
        offs, n = [], 0
        for t in targets:
            offs.append(n)
            n += len(t["labels"])
        tgt_idx = torch.cat([J + o for (_, J), o in zip(indices, offs)])
        pairs = ops.h2d_async(torch.stack([batch_idx * Q + src_idx, tgt_idx]))
        flat_idx, tgt_idx = pairs[0], pairs[1]
        labels_o = torch.cat([t["labels"] for t in targets])[tgt_idx]
        losses = {}
        argmax = None
        for loss in self.losses:
            if loss == "labels":
                target = torch.full((bs * Q,), self.num_classes, dtype=torch.int64, device=dev)
                target[flat_idx] = labels_o
                weight = torch.ones_like(self.empty_weight)
                weight[-1] *= background_c
                ce, argmax = ops.WeightedCE.apply(logits.reshape(bs * Q, C), target, weight)
                losses["loss_ce"] = ce
                if labels_o.numel() == 0:
                    acc = torch.zeros([], device=dev)
                else:
                    acc = (argmax[flat_idx] == labels_o).float().sum() * (100.0 / labels_o.numel())
                losses["class_error"] = 100 - acc
            elif loss == "boxes":
                tgt_boxes = torch.cat([t["boxes"] for t in targets], dim=0)[tgt_idx]
                sums = ops.BoxLoss.apply(boxes.reshape(bs * Q, 4), flat_idx, tgt_boxes)
                losses["loss_bbox"] = sums[0] / num_boxes
                losses["loss_giou"] = sums[1] / num_boxes
            elif loss == "cardinality":
                with torch.no_grad():
                    if argmax is None:
                        argmax = logits.argmax(-1).reshape(-1)
                    lengths = ops.h2d_async(torch.tensor([float(len(v["labels"])) for v in targets]))
                    card = (argmax.view(bs, Q) != C - 1).sum(1).float()
                    losses["cardinality_error"] = (card - lengths).abs().mean()
            else:
                raise AssertionError("do you really want to compute %s loss?" % loss)
        return losses
