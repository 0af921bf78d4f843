"""Oracle: the four per-episode models (configs 1-4) as pure functions.

TEST INFRASTRUCTURE (see oracle/__init__.py).  ``det`` / ``fus`` are state
dicts of CPU tensors (reference key names without the ``detector.`` /
``fusion.`` prefix).  The functions return what the reference's ``forward`` /
``predict`` / ``get_next_action`` return plus, for the training paths, a dict
``grads`` with what the reference leaves in ``.grad`` (None = never touched).
"""
import random

import torch
import torch.nn.functional as F

from .criterion import set_criterion
from .detector import detr_forward, theta_names, trainable_names
from .fusion import fusion_decoder_forward, fusion_gpt_forward


class PathTrie:
    """Best-reward action per path prefix (reference utils/storage_utils.py:4-50)."""

    def __init__(self):
        self.root = {"cost": float("inf"), "action": None, "next": {}}

    def add_path(self, path, reward):
        node = self.root
        for a in path:
            a = int(a)
            if reward < node["cost"]:
                node["cost"], node["action"] = reward, a
            node = node["next"].setdefault(a, {"cost": float("inf"), "action": None, "next": {}})

    def get_label(self, path):
        out, node = [], self.root
        for a in path:
            out.append(node["action"])
            node = node["next"][int(a)]
        return out


def clipped_sgd(params, grads, lr, clip=0.01):
    """reference meta_utils.py:135-142."""
    return [p if g is None else p - torch.clip(lr * g, min=-clip, max=clip) for p, g in zip(params, grads)]


def _labels(data, b):
    return [{"labels": data["category_ids"][b][j], "boxes": data["boxes"][b][j]}
            for j in range(len(data["category_ids"][b]))]


def _unsq(out):
    o = dict(out)
    for k in ("embedded_memory_features", "box_features", "pred_logits", "pred_boxes"):
        o[k] = o[k].unsqueeze(0)
    return o


def _leafify(sd, names):
    """Fresh leaves requiring grad for ``names``, everything else shared/no-grad."""
    out = {k: v.detach() for k, v in sd.items()}
    for k in names:
        out[k] = sd[k].detach().clone().requires_grad_(True)
    return out


def _fusion_fn(style):
    return fusion_gpt_forward if style == "gpt" else fusion_decoder_forward


def _weighted(l):
    # interactron weighs GIoU by 5 and L1 by 2 (reference interactron.py:108,121,133)
    return l["loss_ce"] + 5 * l["loss_giou"] + 2 * l["loss_bbox"]


def _inner_steps(cfg):
    """MODEL.INNER_STEPS: how often the learned-loss step of reference interactron.py:94-102 is repeated (the reference does it
    once; SURVEY section 0 row 2 / BASELINE.json's "5-step adapt loop" make it a parameter, default 1)."""
    k = int(cfg.get("INNER_STEPS", 1)) if isinstance(cfg, dict) else int(getattr(cfg, "INNER_STEPS", 1))
    assert k >= 1, "MODEL.INNER_STEPS must be >= 1"
    return k


def interactron_predict(det, fus, data, cfg, style="gpt"):
    """reference interactron.py:31-59 (and interactron_random.py:27-56); the adapt step repeated INNER_STEPS times."""
    b, s, c, h, w = data["frames"].shape
    img, mask = data["frames"].view(s, c, h, w), data["masks"].view(s, h, w)
    names = theta_names()
    fast = _leafify(det, names)
    for _ in range(_inner_steps(cfg)):
        pre = _unsq(detr_forward(fast, img, mask))
        learned = torch.norm(_fusion_fn(style)(fus, pre, cfg)["loss"])
        g = torch.autograd.grad(learned, [fast[k] for k in names], allow_unused=True)
        nxt = clipped_sgd([fast[k] for k in names], g, cfg["ADAPTIVE_LR"])
        fast = dict(fast)
        fast.update(zip(names, [t.detach().requires_grad_(True) for t in nxt]))
    with torch.no_grad():
        post = detr_forward(fast, img[0:1], mask[0:1])
    return {k: v.unsqueeze(0) for k, v in post.items()}


def interactron_next_action(det, fus, data, cfg):
    """reference interactron.py:174-197."""
    b, s, c, h, w = data["frames"].shape
    with torch.no_grad():
        pre = _unsq(detr_forward(det, data["frames"].view(b * s, c, h, w), data["masks"].view(b * s, h, w)))
        actions = fusion_gpt_forward(fus, pre, cfg)["actions"]
    return int(actions[s - 1].argmax(dim=-1).item()), actions


def interactron_forward(det, fus, data, cfg, path_storage, style="gpt", training=False, ridx_fn=None):
    """One meta-train step over a batch of episodes (reference interactron.py:61-151).

    Returns (predictions, losses, grads) where grads = {"detector": {name: tensor|None},
    "fusion": {name: tensor|None}} accumulated (summed) over the episodes of the batch.
    ``style`` "decoder" gives interactron_random.py:58-130 (no policy loss, no path storage).
    """
    b, s, c, h, w = data["frames"].shape
    img, mask = data["frames"].view(b, s, c, h, w), data["masks"].view(b, s, h, w)
    names = theta_names()
    det_train = trainable_names()
    outer = {k: det[k].detach().clone().requires_grad_(True) for k in det_train}   # the real nn.Parameters
    fus_leaf = {k: (v.detach().clone().requires_grad_(True) if v.is_floating_point() and k.split(".")[-1] != "mask"
                    and k != "pos_embed" else v) for k, v in fus.items()}
    det_losses, sup_losses, logits_out, boxes_out = [], [], [], []
    ridx_fn = ridx_fn or (lambda: random.randint(0, 4))
    for task in range(b):
        labels = _labels(data, task)
        theta_task = {k: outer[k].clone() for k in names}                         # clone_parameters (keeps graph to outer)
        d = {k: v.detach() for k, v in det.items()}
        d.update({k: outer[k] for k in det_train if k not in names})             # in_proj_* stay real parameters
        dtheta = {k: theta_task[k].clone().detach().requires_grad_(True) for k in names}
        d.update(dtheta)
        # the learned-loss step (reference :94-102), INNER_STEPS times with the second-order graph through all of them; the
        # policy logits are those of the first (un-adapted) pass, as in the reference's single step
        fast, cur, fout, g_steps = d, [dtheta[k] for k in names], None, []
        for step in range(_inner_steps(cfg)):
            pre = _unsq(detr_forward(fast, img[task], mask[task], training))
            fo = _fusion_fn(style)(fus_leaf, pre, cfg, training)
            fout = fout or fo
            learned = torch.norm(fo["loss"])
            g = torch.autograd.grad(learned, cur, create_graph=True, retain_graph=True, allow_unused=True)
            g_steps.append(g)
            cur = clipped_sgd(cur, g, cfg["ADAPTIVE_LR"])
            fast = dict(d)
            fast.update(zip(names, cur))
        post = detr_forward(fast, img[task], mask[task], training)
        sup = set_criterion(post["pred_logits"], post["pred_boxes"], labels, cfg["NUM_CLASSES"], 0.1)
        if style == "gpt":
            gt = set_criterion(post["pred_logits"][[0]], post["pred_boxes"][[0]], [labels[0]], cfg["NUM_CLASSES"], 0.1)
            gt_loss = _weighted(gt)
            key = data["initial_image_path"][task]
            store = path_storage.setdefault(key, PathTrie())
            store.add_path(data["actions"][task][:4], torch.mean(gt_loss).item())
            best = torch.tensor(store.get_label(data["actions"][task][:4]), dtype=torch.long)
            sup["loss_path"] = F.cross_entropy(fout["actions"].view(4, 4), best)
            sup["policy_reward"] = gt_loss
        sup_losses.append({k: v.detach() for k, v in sup.items()})
        total = _weighted(sup) + (sup["loss_path"] if style == "gpt" else 0)
        total.backward()
        # first-order detector update (reference interactron.py:126-134)
        cur1 = [theta_task[k] for k in names]
        for g in g_steps:   # the same steps with the gradients as constants, from the attached copy of theta
            cur1 = clipped_sgd(cur1, [None if x is None else x.detach().clone() for x in g], cfg["ADAPTIVE_LR"])
        fast1 = dict(d)
        fast1.update(zip(names, cur1))
        ridx = ridx_fn()
        post1 = detr_forward(fast1, img[task][ridx:ridx + 1], mask[task][ridx:ridx + 1], training)
        dl = set_criterion(post1["pred_logits"], post1["pred_boxes"], labels[ridx:ridx + 1], cfg["NUM_CLASSES"], 0.1)
        det_losses.append({k: v.detach() for k, v in dl.items()})
        _weighted(dl).backward()
        logits_out.append(post1["pred_logits"])
        boxes_out.append(post1["pred_boxes"])
    preds = {"pred_logits": torch.stack(logits_out, 0).detach(), "pred_boxes": torch.stack(boxes_out, 0).detach()}
    losses = {k.replace("loss", "loss_detector"): torch.mean(torch.stack([x[k] for x in det_losses]))
              for k in det_losses[0]}
    losses.update({k.replace("loss", "loss_supervisor"): torch.mean(torch.stack([x[k] for x in sup_losses]))
                   for k in sup_losses[0]})
    grads = {"detector": {k: outer[k].grad for k in det_train},
             "fusion": {k: (v.grad if v.requires_grad else None) for k, v in fus_leaf.items()}}
    return preds, losses, grads


def multiframe_predict(det, fus, data, cfg):
    """reference detr_multiframe.py:24-53."""
    b, s, c, h, w = data["frames"].shape
    with torch.no_grad():
        out = fusion_gpt_forward(fus, _unsq(detr_forward(det, data["frames"].view(b * s, c, h, w),
                                                         data["masks"].view(b * s, h, w))), cfg)
    return {"pred_boxes": out["pred_boxes"].view(b, s, *out["pred_boxes"].shape[1:]),
            "pred_logits": out["pred_logits"].view(b, s, *out["pred_logits"].shape[1:])}


def multiframe_forward(det, fus, data, cfg, training=False):
    """reference detr_multiframe.py:55-109: criterion on the *fusion* boxes/logits, plain backward per episode."""
    b, s, c, h, w = data["frames"].shape
    det_train = trainable_names()
    d = {k: v.detach() for k, v in det.items()}
    d.update({k: det[k].detach().clone().requires_grad_(True) for k in det_train})
    f = {k: (v.detach().clone().requires_grad_(True) if v.is_floating_point() and k.split(".")[-1] != "mask" else v)
         for k, v in fus.items()}
    losses, lo, bo = [], [], []
    for task in range(b):
        # reference keeps the detector in eval() except its decoder (detr_multiframe.py:114-119); oracle runs are eval
        out = fusion_gpt_forward(f, _unsq(detr_forward(d, data["frames"][task], data["masks"][task], training)), cfg,
                                 training)
        loss = set_criterion(out["pred_logits"], out["pred_boxes"], _labels(data, task), cfg["NUM_CLASSES"], 0.1)
        _weighted(loss).backward()
        losses.append({k: v.detach() for k, v in loss.items()})
        lo.append(out["pred_logits"][0:1].detach())
        bo.append(out["pred_boxes"][0:1].detach())
    res = {k.replace("loss", "loss_detector"): torch.mean(torch.stack([x[k] for x in losses])) for k in losses[0]}
    grads = {"detector": {k: d[k].grad for k in det_train},
             "fusion": {k: (v.grad if v.requires_grad else None) for k, v in f.items()}}
    return {"pred_logits": torch.stack(lo, 0), "pred_boxes": torch.stack(bo, 0)}, res, grads


def detr_predict(det, data):
    """reference models/detr.py:20-41."""
    b, s, c, h, w = data["frames"].shape
    with torch.no_grad():
        out = detr_forward(det, data["frames"].view(b * s, c, h, w), data["masks"].view(b * s, h, w))
    return {k: v.view(b, s, *v.shape[1:]) for k, v in out.items()}


def detr_train_forward(det, data, num_classes=1235, training=False):
    """reference models/detr.py:43-64: loss = CE + 5 L1 + 2 GIoU over all b*s frames, one backward."""
    b, s, c, h, w = data["frames"].shape
    det_train = trainable_names()
    d = {k: v.detach() for k, v in det.items()}
    d.update({k: det[k].detach().clone().requires_grad_(True) for k in det_train})
    labels = [l for i in range(b) for l in _labels(data, i)]
    out = detr_forward(d, data["frames"].view(b * s, c, h, w), data["masks"].view(b * s, h, w), training)
    losses = set_criterion(out["pred_logits"], out["pred_boxes"], labels, num_classes, 0.1)
    (losses["loss_ce"] + 5 * losses["loss_bbox"] + 2 * losses["loss_giou"]).backward()
    preds = {k: v.detach().view(b, s, *v.shape[1:]) for k, v in out.items()}
    return preds, {k.replace("loss", "loss_detector"): v.detach() for k, v in losses.items()}, \
        {"detector": {k: d[k].grad for k in det_train}}
