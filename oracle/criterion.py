"""Oracle: box ops, Hungarian matcher and the DETR set criterion.

TEST INFRASTRUCTURE (see oracle/__init__.py).
"""
import torch
import torch.nn.functional as F
from scipy.optimize import linear_sum_assignment


def cxcywh_to_xyxy(b):
    """reference box_ops.py:8-12."""
    cx, cy, w, h = b.unbind(-1)
    return torch.stack((cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h), dim=-1)


def pairwise_iou(a, b):
    """reference box_ops.py:23-36 -> (iou [N,M], union [N,M])."""
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    lt = torch.max(a[:, None, :2], b[:, :2])
    rb = torch.min(a[:, None, 2:], b[:, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    union = area_a[:, None] + area_b - inter
    return inter / union, union


def pairwise_giou(a, b):
    """reference box_ops.py:39-58 (xyxy boxes)."""
    iou, union = pairwise_iou(a, b)
    lt = torch.min(a[:, None, :2], b[:, :2])
    rb = torch.max(a[:, None, 2:], b[:, 2:])
    wh = (rb - lt).clamp(min=0)
    hull = wh[..., 0] * wh[..., 1]
    return iou - (hull - union) / hull


@torch.no_grad()
def matching_cost(pred_logits, pred_boxes, tgt_ids, tgt_boxes, w_class=1.0, w_bbox=5.0, w_giou=2.0):
    """Cost matrix of reference matcher.py:54-73: [bs*Q, sum(N_i)] before the per-image split."""
    prob = pred_logits.flatten(0, 1).softmax(-1)
    boxes = pred_boxes.flatten(0, 1)
    c_class = -prob[:, tgt_ids]
    c_bbox = torch.cdist(boxes, tgt_boxes, p=1)
    c_giou = -pairwise_giou(cxcywh_to_xyxy(boxes), cxcywh_to_xyxy(tgt_boxes))
    return w_bbox * c_bbox + w_class * c_class + w_giou * c_giou


@torch.no_grad()
def hungarian_match(pred_logits, pred_boxes, targets, w_class=1.0, w_bbox=5.0, w_giou=2.0):
    """reference matcher.py:32-77 -> list of (query_idx int64[k], target_idx int64[k])."""
    bs, q = pred_logits.shape[:2]
    tgt_ids = torch.cat([t["labels"] for t in targets])
    tgt_boxes = torch.cat([t["boxes"] for t in targets])
    C = matching_cost(pred_logits, pred_boxes, tgt_ids, tgt_boxes, w_class, w_bbox, w_giou).view(bs, q, -1).cpu()
    sizes = [len(t["boxes"]) for t in targets]
    out = []
    for i, c in enumerate(C.split(sizes, -1)):
        r, col = linear_sum_assignment(c[i])
        out.append((torch.as_tensor(r, dtype=torch.int64), torch.as_tensor(col, dtype=torch.int64)))
    return out


def set_criterion(pred_logits, pred_boxes, targets, num_classes=1235, background_c=0.1, indices=None,
                  w_class=1.0, w_bbox=5.0, w_giou=2.0):
    """SetCriterion.forward with losses (labels, boxes, cardinality) -- reference detr.py:220-265."""
    if indices is None:
        indices = hungarian_match(pred_logits, pred_boxes, targets, w_class, w_bbox, w_giou)
    num_boxes = max(float(sum(len(t["labels"]) for t in targets)), 1.0)
    batch_idx = torch.cat([torch.full_like(s, i) for i, (s, _) in enumerate(indices)])
    src_idx = torch.cat([s for s, _ in indices])
    matched_cls = torch.cat([t["labels"][j] for t, (_, j) in zip(targets, indices)])
    # classification: weighted CE, every unmatched query is "no object" (detr.py:111-132)
    target_classes = torch.full(pred_logits.shape[:2], num_classes, dtype=torch.int64)
    target_classes[batch_idx, src_idx] = matched_cls
    weight = torch.ones(num_classes + 1)
    weight[-1] = background_c
    losses = {"loss_ce": F.cross_entropy(pred_logits.transpose(1, 2), target_classes, weight)}
    matched_logits = pred_logits[batch_idx, src_idx]
    if matched_cls.numel() == 0:
        acc = torch.zeros([])
    else:
        acc = (matched_logits.argmax(-1) == matched_cls).float().sum() * (100.0 / matched_cls.numel())
    losses["class_error"] = 100 - acc
    # boxes: L1 and GIoU on matched pairs (detr.py:148-167)
    src_boxes = pred_boxes[batch_idx, src_idx]
    tgt_boxes = torch.cat([t["boxes"][j] for t, (_, j) in zip(targets, indices)], dim=0)
    losses["loss_bbox"] = F.l1_loss(src_boxes, tgt_boxes, reduction="none").sum() / num_boxes
    giou = torch.diag(pairwise_giou(cxcywh_to_xyxy(src_boxes), cxcywh_to_xyxy(tgt_boxes)))
    losses["loss_giou"] = (1 - giou).sum() / num_boxes
    # cardinality (logging only, detr.py:134-146)
    with torch.no_grad():
        lengths = torch.as_tensor([len(t["labels"]) for t in targets]).float()
        card = (pred_logits.argmax(-1) != pred_logits.shape[-1] - 1).sum(1).float()
        losses["cardinality_error"] = F.l1_loss(card, lengths)
    return losses
