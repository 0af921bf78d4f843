"""G15: fixtures for the AP metric and the prediction<->ground-truth matching, captured from the reference's own
functions (engine/random_policy_evaluator.py compute_ap / compute_cat_ap / compute_pr static methods and
utils/detection_utils.py:match_predictions_to_detections) on synthetic detection lists / IoU matrices.

    python tests/golden/make_golden_eval.py      # needs /root/reference; writes tests/golden/golden_eval.pt
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, HERE)
import _torchvision_stub  # noqa: E402
_torchvision_stub.install()   # stand-in torchvision modules (torchvision is not installed)
sys.path.insert(0, REF)
np.float = float   # the reference targets numpy < 1.24

from utils.detection_utils import match_predictions_to_detections  # noqa: E402
from engine.random_policy_evaluator import RandomPolicyEvaluator as Ref  # noqa: E402


def synthetic_detections(n, seed):
    rng = np.random.RandomState(seed)
    dets = []
    for i in range(n):
        kind = ["tp", "fp", "fn"][int(rng.choice(3, p=[0.5, 0.3, 0.2]))]
        w, h = rng.uniform(0.02, 0.6, 2)
        x, y = rng.uniform(0, 1 - w), rng.uniform(0, 1 - h)
        dets.append({"iou": float(rng.uniform(0.3, 1.0)) if kind == "tp" else (float(rng.uniform(0, 0.4)) if kind == "fp" else 0.0),
                     "category_match": kind == "tp", "type": kind, "pred_cat": int(rng.choice([3, 11, 18, 22, 57])),
                     "pred_score": float(rng.uniform(0, 1)) if kind != "fn" else 0.0,
                     "box": [float(x), float(y), float(x + w), float(y + h)], "area": float(w * h), "img": "img%d" % (i % 7)})
    return dets


def main():
    out = {"ap": [], "match": []}
    ious = list(np.arange(0.5, 1.0, 0.05))
    for seed, n in [(0, 60), (1, 200), (2, 17), (3, 400)]:
        dets = synthetic_detections(n, seed)
        rec = {"detections": dets,
               "ap50": float(Ref.compute_ap(list(dets), 100, [0.5])),
               "ap75": float(Ref.compute_ap(list(dets), 100, [0.75])),
               "ap": float(Ref.compute_ap(list(dets), 100, ious)),
               "ap_small": float(Ref.compute_ap(list(dets), 100, ious, 0.0, 32 ** 2 / 300 ** 2)),
               "ap_medium": float(Ref.compute_ap(list(dets), 100, ious, 32 ** 2 / 300 ** 2, 96 ** 2 / 300 ** 2)),
               "cat_ap": float(Ref.compute_cat_ap(list(dets), 100, ious)),
               "pr": [list(map(float, v)) for v in Ref.compute_pr(list(dets), 100, 0.5)]}
        out["ap"].append(rec)
    g = torch.Generator().manual_seed(5)
    for (p, t) in [(1, 1), (3, 2), (2, 5), (6, 6), (8, 3), (4, 7)]:
        m = torch.rand(p, t, generator=g)
        m[m < 0.25] = 0.0                      # disjoint boxes
        if p > 2 and t > 1:
            m[1] = m[0]                        # duplicated prediction -> ties
        bi, bx = match_predictions_to_detections(m.clone())
        out["match"].append({"ious": m, "best_ious": bi, "best_idx": bx})
    torch.save(out, os.path.join(HERE, "golden_eval.pt"))
    print("wrote golden_eval.pt")


if __name__ == "__main__":
    main()
