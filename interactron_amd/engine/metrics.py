"""Detection bookkeeping and the AP metric of the reference evaluators, on the host.

reference: engine/random_policy_evaluator.py:60-157 (per-image post-processing), :213-329 (compute_cat_ap / compute_ap /
compute_pr), utils/detection_utils.py:401-421 (match_predictions_to_detections), torchvision.ops.{nms, box_iou}.
O(100) episodes of 50 boxes: host code by design ("OUT OF SCOPE for kernels"); it must be faithful to
the reference's quirks because it DEFINES the acceptance metric:
  * the confidence sweep pops detections in place, so a threshold removes them for all later thresholds too;
  * the area filter uses strict inequalities on both sides;
  * recall list gets r[0] + 1e-6 prepended and precision 0.0, then a 101-point interpolation walks recall downwards.
"""
import numpy as np
import torch

from ..constants import NUM_CLASSES, THOR_CLASS_IDS


def box_cxcywh_to_xyxy(x):
    cx, cy, w, h = x.unbind(-1)
    return torch.stack([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], dim=-1)


def box_iou(a, b):
    """Pairwise IoU of xyxy boxes [N,4] x [M,4] -> [N,M] (torchvision.ops.box_iou)."""
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    lt = torch.max(a[:, None, :2], b[None, :, :2])
    rb = torch.min(a[:, None, 2:], b[None, :, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    return inter / (area_a[:, None] + area_b[None, :] - inter)


def nms(boxes, scores, iou_threshold):
    """Greedy non-maximum suppression, indices of kept boxes by decreasing score (torchvision.ops.nms semantics:
    a box is dropped when its IoU with an already kept, higher-scoring box is > iou_threshold)."""
    if boxes.numel() == 0:
        return torch.empty(0, dtype=torch.long, device=boxes.device)
    order = torch.argsort(scores, descending=True)
    iou = box_iou(boxes[order], boxes[order])
    n = order.numel()
    suppressed = [False] * n
    over = (iou > iou_threshold).tolist()
    keep = []
    for i in range(n):
        if suppressed[i]:
            continue
        keep.append(i)
        row = over[i]
        for j in range(i + 1, n):
            if row[j]:
                suppressed[j] = True
    return order[torch.tensor(keep, dtype=torch.long, device=boxes.device)]


def match_predictions_to_detections(ious):
    """Which prediction (row) answers for which ground-truth box (column): a proposal game on the IoU table
    -> (IoU of the holder per GT, holder index per GT or -1).  Pinned by fixture G15 against the reference's
    utils/detection_utils.py:401-421, whose observable behaviour (it defines the AP metric) is:

    * every prediction keeps a cursor into its own ranking of the GT boxes (best IoU first) and, each round, points at the
      box under the cursor -- also while it is holding a box;
    * boxes are visited in index order; a box hands itself to the pointer with the largest IoU (lowest index on ties).  A
      box nobody points at with a positive IoU falls to prediction 0 -- the arg-max of an all-zero column -- and that
      still counts as "prediction 0 is taken";
    * a prediction displaced from a box becomes loose again, loose predictions move their cursor on after the round, and
      the game stops after one round per box or once min(#predictions, #boxes) predictions are taken;
    * holders with IoU 0 are reported as "no match" (-1).
    The bookkeeping runs on host lists (the reference loops over tensors element by element)."""
    n_p, n_g = ious.shape
    table = ious.tolist()
    ranking = torch.argsort(ious, dim=1, descending=True).tolist()   # (tie order inside a row: torch's, as in the reference)
    cursor, taken, holder = [0] * n_p, [False] * n_p, [-1] * n_g
    for _ in range(n_g):
        pointed = [ranking[p][cursor[p]] for p in range(n_p)]
        for j in range(n_g):
            winner, top = 0, 0.0
            for p in range(n_p):
                if pointed[p] == j and table[p][j] > top:
                    winner, top = p, table[p][j]
            if holder[j] not in (-1, winner):
                taken[holder[j]] = False
            holder[j] = winner
            taken[winner] = True
        for p in range(n_p):
            if not taken[p]:
                cursor[p] += 1
        if sum(taken) >= min(n_p, n_g):
            break
    held = [table[h][j] if h != -1 else 0.0 for j, h in enumerate(holder)]
    best_iou = torch.tensor(held, dtype=torch.float32, device=ious.device).reshape(n_g)
    best_idx = torch.tensor([h if v != 0.0 else -1 for h, v in zip(holder, held)], dtype=torch.long, device=ious.device).reshape(n_g)
    return best_iou, best_idx


def _area(box):
    return ((box[2] - box[0]) * (box[3] - box[1])).item()


def _entry(kind, iou, matched, cat, score, box, img):
    return {"iou": iou, "category_match": matched, "type": kind, "pred_cat": cat, "pred_score": score,
            "box": [c.item() for c in box], "area": _area(box), "img": img}


def frame_detections(pred_logits, pred_boxes, gt_boxes, gt_cats, img):
    """TP / FP / FN records of ONE image (reference random_policy_evaluator.py:62-157).

    pred_logits [Q, C+1], pred_boxes [Q,4] cxcywh, gt_boxes [N,4] cxcywh, gt_cats int64 [N] (any device; the
    bookkeeping runs on the host)."""
    pred_logits, pred_boxes = pred_logits.detach().float().cpu(), pred_boxes.detach().float().cpu()
    gt_boxes, gt_cats = box_cxcywh_to_xyxy(gt_boxes.detach().float().cpu()), gt_cats.detach().cpu()
    boxes = box_cxcywh_to_xyxy(pred_boxes)
    scores, cats = pred_logits.softmax(dim=-1).max(dim=-1)
    fg = cats != NUM_CLASSES
    boxes, cats, scores = boxes[fg], cats[fg], scores[fg]
    keep = nms(boxes, scores, 0.5)
    boxes, cats, scores = boxes[keep], cats[keep], scores[keep]
    pred_set, gt_set = set(int(c) for c in cats), set(int(c) for c in gt_cats)
    pred_only = set(THOR_CLASS_IDS).intersection(pred_set - gt_set)
    out = []
    for cat in gt_set:
        cat_gt = gt_boxes[gt_cats == cat]
        if torch.any(cats == cat):
            cb, cs = boxes[cats == cat], scores[cats == cat]
            ious = box_iou(cb, cat_gt)
            best_iou, best_idx = match_predictions_to_detections(ious)
            for i in range(ious.shape[0]):
                kind = "tp" if torch.any(best_idx == i) else "fp"
                out.append(_entry(kind, ious[i].max().item(), True, cat, cs[i].item(), cb[i], img))
            for j in range(ious.shape[1]):
                if best_iou[j] == 0.0:
                    out.append(_entry("fn", 0.0, False, cat, 0.0, cat_gt[j], img))
        else:
            for j in range(cat_gt.shape[0]):
                out.append(_entry("fn", 0.0, False, cat, 0.0, cat_gt[j], img))
    for cat in pred_only:
        cb, cs = boxes[cats == cat], scores[cats == cat]
        for i in range(cs.shape[0]):
            out.append(_entry("fp", 0.0, False, cat, cs[i].item(), cb[i], img))
    return out


def _pr_sweep(detections, nsamples, iou_thresh):
    tps = [d for d in detections if d["type"] == "tp"]
    fps = [d for d in detections if d["type"] == "fp"]
    fns = [d for d in detections if d["type"] == "fn"]
    low = [d for d in tps if d["iou"] < iou_thresh]      # matched but below the IoU bar: counted as false positives
    tps = [d for d in tps if d["iou"] >= iou_thresh]
    fps = fps + low
    p, r = [], []
    for conf in np.arange(0.0, 1.0, 1.0 / nsamples):
        tps = [d for d in tps if not d["pred_score"] < conf]
        fps = [d for d in fps if not d["pred_score"] < conf]
        p.append(0 if len(tps) == 0 else len(tps) / (len(tps) + len(fps)))
        r.append(0 if len(tps) == 0 else len(tps) / (len(tps) + len(fns)))
    return p, r


def _interpolated_ap_samples(p, r):
    p = [0.0] + p
    r = [r[0] + 0.000001] + r
    samples, r_idx = [], 0
    for cutoff in np.arange(1.0, -0.0001, -0.01):
        while r_idx < len(r) - 1 and r[r_idx] > cutoff:
            r_idx += 1
        samples.append(max(p[:r_idx + 1]))
    return samples


def compute_pr(detections, nsamples=100, iou_thresh=0.5, min_area=0.0, max_area=1.0):
    return _pr_sweep([d for d in detections if min_area < d["area"] < max_area], nsamples, iou_thresh)


def compute_ap(detections, nsamples=100, iou_thresholds=(0.5,), min_area=0.0, max_area=1.0):
    """101-point interpolated AP averaged over IoU thresholds (reference random_policy_evaluator.py:277-329)."""
    detections = [d for d in detections if min_area < d["area"] < max_area]
    aps = []
    for t in iou_thresholds:
        p, r = _pr_sweep(detections, nsamples, t)
        aps.append(np.mean(_interpolated_ap_samples(p, r)))
    return np.mean(aps)


def compute_cat_ap(detections, nsamples=100, iou_thresholds=(0.5,), min_area=0.0, max_area=1.0, verbose=False):
    """Per-category AP averaged over the categories with at least five ground-truth instances (reference :213-275;
    the reference appends the RUNNING mean of the interpolation samples once per recall cutoff -- kept)."""
    aps = []
    for cat in list(set(d["pred_cat"] for d in detections)):
        dets = [d for d in detections if d["pred_cat"] == cat and min_area < d["area"] < max_area]
        if len([d for d in dets if d["type"] in ("tp", "fn")]) < 5:
            continue
        cat_aps = []
        for t in iou_thresholds:
            p, r = _pr_sweep(dets, nsamples, t)
            samples = _interpolated_ap_samples(p, r)
            cat_aps.extend(np.mean(samples[:k + 1]) for k in range(len(samples)))
        aps.append(np.mean(cat_aps))
        if verbose:
            print("{}: {:06f}".format(cat, np.mean(cat_aps)))
    return np.mean(aps)


def summarize(detections):
    """The six numbers the reference prints after an evaluation (random_policy_evaluator.py:183-194)."""
    ious = list(np.arange(0.5, 1.0, 0.05))
    s, m, l = 32 ** 2 / 300 ** 2, 96 ** 2 / 300 ** 2, 1.0
    return {"AP_50": compute_ap(detections, 100, [0.5]), "AP_75": compute_ap(detections, 100, [0.75]),
            "AP": compute_ap(detections, 100, ious), "AP_small": compute_ap(detections, 100, ious, 0.0, s),
            "AP_medium": compute_ap(detections, 100, ious, s, m), "AP_large": compute_ap(detections, 100, ious, m, l)}
