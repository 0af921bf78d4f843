"""Fixture G18: the reference's two evaluators END TO END on the committed tiny dataset (tests/golden/data).

    python tests/golden/make_golden_evalrun.py [--ref /root/reference]     # build container only; writes golden_evalrun.pt

What runs is the imported reference itself: ``InteractiveEvaluator.evaluate`` (engine/interactive_evaluator.py:35-262 --
``InteractiveDaatset.reset`` -> 4 x ``interactron.get_next_action`` + ``step`` -> ``interactron.predict`` -> softmax / max,
background drop, NMS 0.5, per-category matching, TP / FP / FN records, ``compute_ap``) and ``RandomPolicyEvaluator.evaluate``
(engine/random_policy_evaluator.py:37-211, the fixed test rollout) on an ``interactron`` model with the RNG-free procedural
weights.  It is the only available stand-in for ``north_star``'s "AP within +-0.002": data and trained weights are S3
tarballs.

Procedural weights alone make the 50 queries indistinguishable: one arbitrary class, one box, every AP zero, and a policy
that always picks the same move.  ``evalrun_weight_edit`` (interactron_amd/synthetic.py -- applied to BOTH sides' state dicts,
it is part of the fixture's weight recipe) sharpens the decoder's cross-attention in closed form; ``calibrate`` below then
re-centres the class rows of the dataset's categories, the last box-head layer and the policy head's bias on the mean of a
calibration pass (the detector on the test rollouts, the policy on its first observation of every scene) and stores those few
tensors in the fixture as ``overrides`` -- inputs of the fixture, like the images.

Stored: the overrides, the four chosen actions per scene (and the smallest arg-max margin), every detection record in order, #tp / #fp / #fn, and the six AP numbers the
reference prints (computed with its own ``compute_ap`` from the very list ``evaluate`` built).  Only data, never reference
source.
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)

import make_golden as mg            # noqa: E402
import make_golden_data as mgd      # noqa: E402
from interactron_amd.synthetic import evalrun_weight_edit   # noqa: E402


ACTIONS = ["MoveAhead", "MoveBack", "RotateLeft", "RotateRight"]
POOL = [3, 11, 18, 22, 57, 67]          # labels (THOR class ids, utils/constants.py); annotation category_id = label - 1


def write_eval_annotations(root):
    """Six 300 x 300 scenes over the committed JPEGs of FloorPlan200 / FloorPlan201 (the reference model cannot take the
    400 x 300 scene: models/interactron.py:178 views the mask with the resized frame's shape): every state as a root once,
    a different move table per scene so that the policy's choices matter, two to five large central objects per state out
    of six categories (several queries then share a category: NMS and the prediction <-> ground-truth game are exercised)."""
    scenes = []
    for s in range(6):
        plan = "FloorPlan%d" % (200 + s % 2)
        states = ["%s|%d" % (plan, k) for k in range(3)]
        table = {}
        for k, st in enumerate(states):
            dets = {}
            for j in range(2 + (s + 2 * k) % 4):
                w, h = 90 + 23 * ((s + j + k) % 5), 80 + 19 * ((2 * s + j + k) % 6)
                x, y = 20 + 31 * ((s + 2 * j + k) % 4), 15 + 27 * ((j + 3 * k + s) % 5)
                dets["obj|%d|%d|%d" % (s, k, j)] = {"category_id": POOL[(s + j + 2 * k) % len(POOL)] - 1,
                                                     "bbox": [x, y, min(w, 298 - x), min(h, 298 - y)]}
            table[st] = {"detections": dets,
                         "actions": {a: states[(k + 1 + (i * (s + 1)) % 3) % 3] for i, a in enumerate(ACTIONS)}}
        scenes.append({"scene_name": plan, "root": states[s // 2], "state_table": table})
    ann = os.path.join(root, "annotations_eval.json")
    with open(ann, "w") as f:
        json.dump({"data": scenes, "metadata": {"actions": ACTIONS}}, f, indent=1)
    return ann


def fit_annotations(ann_path, ds, records):
    """Round 5: ground truth that the (untrained, calibrated) detector can actually find.  The procedural boxes of
    write_eval_annotations have nothing to do with what the detector predicts on the noise JPEGs: 12 tp of 138 records, AP_50 =
    0.006 -- a +-0.002 check on such numbers passes for almost any detector.  The annotations are part of the fixture, so the
    root state of every scene gets its objects FROM the frame-0 predictions of a first evaluation pass (`records`: the tp / fp
    records of that pass, grouped by root image): most predictions become an object of their category -- the predicted box
    moved along one axis by w (1 - t) / (1 + t), which makes the pair's IoU exactly t, for t = 0.925 / 0.775 / 0.625 / 0.425:
    hits at every threshold of the AP sweep, every IoU 0.025 away from the nearest threshold (a record ON a threshold would
    turn the +-0.002 AP check into a coin toss between two float32 implementations: one hit more or less moves AP_50 by
    0.009) --, every fifth prediction gets none (a false positive), and every other scene gets one object nobody predicts (a
    false negative).  Nothing here looks at the second pass: predict() never sees labels, so the predictions do not move."""
    ann = json.load(open(ann_path))
    by_img = {}
    for r in records:
        if r["type"] in ("tp", "fp"):
            by_img.setdefault(r["img"], []).append(r)
    targets = [0.925, 0.775, 0.625, 0.425]
    for s, scene in enumerate(ann["data"]):
        img = ds[s]["initial_image_path"]
        preds = sorted(by_img.get(img, []), key=lambda r: -r["pred_score"])
        dets = {}
        for j, r in enumerate(preds):
            if j % 5 == 4:
                continue
            x0, y0, x1, y1 = [300.0 * c for c in r["box"]]
            w, h, t = x1 - x0, y1 - y0, targets[(j + s) % 4]
            if (j + s) % 2 == 0:
                d = w * (1 - t) / (1 + t)
                x0 = x0 + d if x1 + d < 299.0 else x0 - d
            else:
                d = h * (1 - t) / (1 + t)
                y0 = y0 + d if y1 + d < 299.0 else y0 - d
            if x0 < 0.5 or y0 < 0.5 or x0 + w > 299.5 or y0 + h > 299.5:
                continue   # (would be clipped by the dataset: its IoU would not be the designed one)
            dets["fit|%d|%d" % (s, j)] = {"category_id": int(r["pred_cat"]) - 1, "bbox": [round(x0, 3), round(y0, 3), round(w, 3), round(h, 3)]}
        if s % 2 == 0:
            unused = [c for c in POOL if c not in {int(r["pred_cat"]) for r in preds}] or [POOL[s % len(POOL)]]
            dets["miss|%d" % s] = {"category_id": unused[0] - 1, "bbox": [40 + 13 * s, 30 + 11 * s, 70, 60]}
        scene["state_table"][scene["root"]]["detections"] = dets
    with open(ann_path, "w") as f:
        json.dump(ann, f, indent=1)
    return ann_path


def calibrate(model, ds, collate_fn, env, rounds=3, class_gain=2.5, class_lift=5.0, box_gain=25.0):
    """-> overrides for evalrun_weight_edit (see there).  `model` already carries the closed-form part of the edit.

    Calibrated on what the evaluators score: the POST-adaptation frame-0 outputs of ``predict`` on the test rollouts (the
    clipped inner step moves every detector weight by up to 0.01, a common shift of the query features that the gains
    would otherwise turn into one saturated class and boxes on the image border).  The adaptation in turn sees the heads
    through the fusion's inputs, hence a few rounds."""
    model.eval()
    det, labels = model.detector, torch.tensor(POOL)
    overrides = {}
    target = torch.logit(torch.tensor([0.45, 0.44, 0.45, 0.42], dtype=torch.float64))
    for r in range(rounds):
        feats, pre, acts = [], [], []
        for i in range(len(ds)):
            data = collate_fn([ds[i]])
            out = model.predict(data)
            feats.append(out["box_features"].detach().reshape(-1, 256))
            pre.append(torch.logit(out["pred_boxes"].detach().double().reshape(-1, 4).clamp(1e-6, 1 - 1e-6)))
        H, P = torch.cat(feats).double(), torch.cat(pre)
        mean = H.mean(0)
        sd = model.state_dict()
        if r == 0:
            _, _, vt = torch.linalg.svd(H - mean, full_matrices=False)
            rows = torch.stack([vt[i] * (class_gain / float(((H - mean) @ vt[i]).std())) for i in range(len(POOL))])
            overrides["detector.class_embed.weight"] = {"rows": labels, "values": rows.float()}
            overrides["detector.bbox_embed.layers.2.weight"] = (sd["detector.bbox_embed.layers.2.weight"].double() * box_gain).float()
            overrides["detector.bbox_embed.layers.2.bias"] = (box_gain * (sd["detector.bbox_embed.layers.2.bias"].double()
                                                                          - P.mean(0)) + target).float()
        else:
            overrides["detector.bbox_embed.layers.2.bias"] = (sd["detector.bbox_embed.layers.2.bias"].double()
                                                              - (P.mean(0) - target)).float()
        rows = overrides["detector.class_embed.weight"]["values"].double()
        bias = sd["detector.class_embed.bias"].double().clone()
        bias[labels] = class_lift - rows @ mean
        overrides["detector.class_embed.bias"] = bias.float()
        evalrun_weight_edit(sd, overrides, sharpen=1.0, query_gain=1.0, loss_gain=1.0)
        model.load_state_dict(sd)
        print("calibration round", r, "post-adaptation box pre-sigmoid mean", P.mean(0).tolist(), "std", P.std(0).tolist())
    # the policy head: its four logits minus their mean over the decisions of a greedy evaluation (which the bias moves:
    # two rounds), then the candidate with the largest smallest arg-max margin among a few fixed small offsets -- a
    # decision that hangs on 1e-4 would make the fixture a coin toss for any other fp32 implementation
    key = "fusion.action_decoder.layers.2.bias"
    for r in range(2):
        trace = policy_trace(model, env)
        sd = model.state_dict()
        overrides[key] = (sd[key].double() - torch.stack(trace).double().mean(0)).float()
        evalrun_weight_edit(sd, {key: overrides[key]}, sharpen=1.0, query_gain=1.0, loss_gain=1.0)
        model.load_state_dict(sd)
    base, best = overrides[key].clone(), None
    for c, off in enumerate([(0, 0, 0, 0), (2, -1, 1, -2), (-2, 1, 2, -1), (1, 2, -2, -1), (-1, -2, -1, 2)]):
        cand = base + 1e-3 * torch.tensor(off, dtype=torch.float32)
        sd = model.state_dict()
        evalrun_weight_edit(sd, {key: cand}, sharpen=1.0, query_gain=1.0, loss_gain=1.0)
        model.load_state_dict(sd)
        trace = policy_trace(model, env)
        margins = [float(t.sort(descending=True).values[0] - t.sort(descending=True).values[1]) for t in trace]
        moves = [int(t.argmax()) for t in trace]
        print("policy candidate", c, "min margin %.2e" % min(margins), "moves", moves)
        if len(set(moves)) >= 3 and (best is None or min(margins) > best[0]):
            best = (min(margins), cand)
    assert best is not None and best[0] > 1e-3, "no candidate with a usable policy margin"
    overrides[key] = best[1]
    sd = model.state_dict()
    evalrun_weight_edit(sd, {key: best[1]}, sharpen=1.0, query_gain=1.0, loss_gain=1.0)
    model.load_state_dict(sd)
    return overrides


def policy_trace(model, env):
    """the policy's logits at every decision of a greedy interactive evaluation (6 scenes x 4 moves)"""
    seen, trace = {}, []
    orig = model.fusion.forward

    def fusion(x):
        out = orig(x)
        seen["logits"] = out["actions"].detach().clone()
        return out
    model.fusion.forward = fusion
    try:
        env.idx = -1
        for _ in range(len(env)):
            data = env.reset()
            for _ in range(4):
                a = model.get_next_action(data)
                trace.append(seen["logits"][data["frames"].shape[1] - 1])
                data = env.step(a)
    finally:
        del model.fusion.forward
    env.idx = -1
    return trace


def six(cls, dets):
    ious = list(np.arange(0.5, 1.0, 0.05))
    s, m = 32 ** 2 / 300 ** 2, 96 ** 2 / 300 ** 2
    return {"AP_50": float(cls.compute_ap(list(dets), 100, [0.5])), "AP_75": float(cls.compute_ap(list(dets), 100, [0.75])),
            "AP": float(cls.compute_ap(list(dets), 100, ious)), "AP_small": float(cls.compute_ap(list(dets), 100, ious, 0.0, s)),
            "AP_medium": float(cls.compute_ap(list(dets), 100, ious, s, m)),
            "AP_large": float(cls.compute_ap(list(dets), 100, ious, m, 1.0))}


def run(evaluator_cls, model, cfg):
    """evaluate(save_results=False) with two taps: the detection list as handed to the first compute_ap, and the actions"""
    seen, actions, margins = {}, [], []
    orig_ap = evaluator_cls.compute_ap

    def tap(detections, *a, **k):
        seen.setdefault("detections", [dict(d) for d in detections])
        return orig_ap(detections, *a, **k)

    evaluator_cls.compute_ap = staticmethod(tap)
    if hasattr(model, "get_next_action"):
        orig_act = model.get_next_action

        orig_fusion = model.fusion.forward

        def fusion(x):
            out = orig_fusion(x)
            seen["policy_logits"] = out["actions"].detach().clone()
            return out

        def act(data):
            a = orig_act(data)
            actions.append(int(a))
            top = seen["policy_logits"][data["frames"].shape[1] - 1].sort(descending=True).values
            margins.append(float(top[0] - top[1]))
            return a
        model.get_next_action, model.fusion.forward = act, fusion
    try:
        ev = evaluator_cls(model, cfg)
        ap50, ap, tp, fp, fn = ev.evaluate(save_results=False)
    finally:
        evaluator_cls.compute_ap = staticmethod(orig_ap)
        if hasattr(model, "get_next_action"):
            del model.get_next_action, model.fusion.forward
    dets = seen["detections"]
    root = os.path.join(HERE, "data")
    for d in dets:
        d["img"] = os.path.relpath(d["img"], root)
    return {"returned": (float(ap50), float(ap), int(tp), int(fp), int(fn)), "detections": dets, "actions": actions,
            "min_policy_margin": min(margins) if margins else None, "six": six(evaluator_cls, dets)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    args = ap.parse_args()
    mgd.install_functional()
    mg.install_reference(args.ref)
    torch.set_num_threads(8)
    from utils.config_utils import Config
    from engine.interactive_evaluator import InteractiveEvaluator
    from engine.random_policy_evaluator import RandomPolicyEvaluator
    root = os.path.join(HERE, "data")
    model, _ = mg.build("interactron", Config)
    ann = write_eval_annotations(root)
    from datasets.sequence_dataset import SequenceDataset
    from models.detr_models.util.misc import NestedTensor
    from utils.storage_utils import collate_fn
    from utils.transform_utis import transform
    sd = model.state_dict()
    evalrun_weight_edit(sd)
    model.load_state_dict(sd)
    ds = SequenceDataset(os.path.join(root, "imgs") + "/", ann, "test", transform=transform)
    global NestedTensor_
    NestedTensor_ = NestedTensor
    from datasets.interactive_dataset import InteractiveDaatset
    env = InteractiveDaatset(os.path.join(root, "imgs") + "/", ann, "test", transform=transform)
    overrides = calibrate(model, ds, collate_fn, env)
    cfg = Config(**{"EVALUATOR": {"BATCH_SIZE": 1, "NUM_WORKERS": 0, "OUTPUT_DIRECTORY": "/tmp/g18", "CHECKPOINT": ""},
                    "DATASET": {"TEST": {"IMAGE_ROOT": os.path.join(root, "imgs") + "/", "ANNOTATION_ROOT": ann, "MODE": "test"}}})
    first = run(InteractiveEvaluator, model, cfg)
    print("first pass (procedural boxes):", first["returned"], first["six"])
    for d in first["detections"]:
        d["img"] = os.path.join(root, d["img"])
    fit_annotations(ann, ds, first["detections"])
    thresholds = np.arange(0.5, 1.0, 0.05)

    def near(d):
        return d["type"] == "tp" and min(abs(d["iou"] - t) for t in thresholds) < 5e-3

    # config 3 (configs/interactron_random.yaml, the README's "Interactron-Rand" row): the decoder-fusion model with the same
    # detector recipe through the fixed-rollout evaluator
    model_r, _ = mg.build("interactron_random", Config)
    sd_r = model_r.state_dict()
    evalrun_weight_edit(sd_r, overrides)
    model_r.load_state_dict(sd_r)
    RUNS = ("interactive", "random_policy", "random_policy_interactron_random")
    for attempt in range(6):
        G = {"overrides": overrides, "annotations": os.path.relpath(ann, root), "interactive": run(InteractiveEvaluator, model, cfg),
             "random_policy": run(RandomPolicyEvaluator, model, cfg),
             "random_policy_interactron_random": run(RandomPolicyEvaluator, model_r, cfg)}
        bad = [d for k in RUNS for d in G[k]["detections"] if near(d)]
        margin = min(abs(d["iou"] - t) for k in RUNS for d in G[k]["detections"] if d["type"] == "tp" for t in thresholds)
        print("attempt", attempt, "smallest distance of a hit's IoU from a sweep threshold: %.2e" % margin, "(%d hits within 5e-3)" % len(bad))
        if not bad:
            break
        # The designed IoUs hold for the pass the objects were fitted to; the other evaluator adapts on other rollouts and its
        # boxes land elsewhere.  Objects whose hit sits within 5e-3 of a threshold in EITHER pass are taken out of the ground
        # truth (found by the record's own IoU against the objects of its image and category) and both passes are run again.
        a = json.load(open(ann))
        for d in bad:
            scene = next(sc for s_, sc in enumerate(a["data"]) if os.path.relpath(ds[s_]["initial_image_path"], root) == d["img"])
            objs = scene["state_table"][scene["root"]]["detections"]
            px0, py0, px1, py1 = [300.0 * c for c in d["box"]]
            best, best_err = None, 1e9
            for name, o in objs.items():
                if o["category_id"] + 1 != int(d["pred_cat"]):
                    continue
                gx0, gy0, gw, gh = o["bbox"]
                iw, ih = max(0.0, min(px1, gx0 + gw) - max(px0, gx0)), max(0.0, min(py1, gy0 + gh) - max(py0, gy0))
                iou = iw * ih / ((px1 - px0) * (py1 - py0) + gw * gh - iw * ih)
                if abs(iou - d["iou"]) < best_err:
                    best, best_err = name, abs(iou - d["iou"])
            if best is not None and best_err < 2e-2:
                del objs[best]
        with open(ann, "w") as f:
            json.dump(a, f, indent=1)
    assert not bad, "hits still sit on thresholds of the AP sweep"
    G["iou_threshold_margin"] = float(margin)
    for k in RUNS:
        v = G[k]
        kinds = [d["type"] for d in v["detections"]]
        print(k, "returned", v["returned"], "actions", v["actions"], "margin", v["min_policy_margin"], "records", len(kinds),
              {t: kinds.count(t) for t in ("tp", "fp", "fn")}, v["six"])
    torch.save(G, os.path.join(HERE, "golden_evalrun.pt"))
    print("wrote golden_evalrun.pt")


if __name__ == "__main__":
    main()
