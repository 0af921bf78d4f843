"""Fixture G17 (N3, data path): the reference's own datasets and transforms on a tiny on-disk dataset.

Usage:  python tests/golden/make_golden_data.py [--ref /root/reference]

Writes tests/golden/data/ (three scenes x three states of small JPEGs -- 300x300 like the shipped data, one scene 400x300 so
that the resize is not the identity -- plus an annotation JSON in the reference schema,
data_collection/collect_ithor_tree_data.py:77-137) and tests/golden/golden_data.pt with what the imported reference
returns for it: ``SequenceDataset`` in test mode (datasets/sequence_dataset.py:30-95, deterministic action script),
``collate_fn`` (utils/storage_utils.py:53-64) and ``InteractiveDaatset.reset / step`` (datasets/interactive_dataset.py).

torchvision is not installed; the reference's transforms (models/detr_models/util/transforms.py) only need
``torchvision.transforms.functional`` on PIL images, which is restated here from its documented behaviour:
resize = PIL bilinear resize to (w, h); to_tensor = HWC uint8 -> CHW float / 255; normalize = (x - mean) / std;
hflip / crop / pad = the PIL operations.  Only inputs' recipes and outputs are stored -- never reference source.
"""
import argparse
import json
import os
import sys

import numpy as np
import torch
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)

ACTIONS = ["MoveAhead", "MoveBack", "RotateLeft", "RotateRight"]
SIZES = [(300, 300), (300, 300), (400, 300)]   # (width, height) per scene


def write_dataset(root):
    """Deterministic images (seeded noise over a colour gradient) and annotations; returns the annotation path."""
    scenes = []
    for s, (w, h) in enumerate(SIZES):
        name = "FloorPlan%d" % (200 + s)
        os.makedirs(os.path.join(root, "imgs", name), exist_ok=True)
        states = ["%s|%d" % (name, k) for k in range(3)]
        table = {}
        for k, st in enumerate(states):
            rs = np.random.RandomState(100 * s + k)
            yy, xx = np.mgrid[0:h, 0:w]
            img = np.stack([(xx * 255 // w), (yy * 255 // h), ((xx + yy) * 255 // (w + h))], -1).astype(np.int32)
            img = np.clip(img + rs.randint(-20, 20, (h, w, 3)), 0, 255).astype(np.uint8)
            Image.fromarray(img).save(os.path.join(root, "imgs", name, st + ".jpg"), quality=90)
            dets = {}
            for j in range((k + s) % 3 + (1 if s != 1 or k != 2 else 0)):   # one state without any detection
                bw, bh = 30 + 17 * j + 5 * k, 40 + 11 * j + 3 * s
                dets["obj|%d|%d" % (s, j)] = {"category_id": (37 * s + 11 * j + 5 * k) % 1234,
                                              "bbox": [10 + 23 * j + k, 20 + 19 * j + 2 * k, bw, bh]}
            table[st] = {"detections": dets, "actions": {a: states[(k + 1 + i) % 3] for i, a in enumerate(ACTIONS)}}
        scenes.append({"scene_name": name, "root": states[0], "state_table": table})
    ann = os.path.join(root, "annotations.json")
    with open(ann, "w") as f:
        json.dump({"data": scenes, "metadata": {"actions": ACTIONS}}, f, indent=1)
    return ann


def install_functional():
    import _torchvision_stub
    _torchvision_stub.install()
    F = sys.modules["torchvision.transforms.functional"]
    T = sys.modules["torchvision.transforms"]
    F.resize = lambda img, size, *a, **k: img.resize((size[1], size[0]), Image.BILINEAR)
    F.to_tensor = lambda pic: torch.from_numpy(np.asarray(pic.convert("RGB"), dtype=np.uint8).copy()).permute(2, 0, 1).float().div(255)
    F.normalize = lambda t, mean, std: (t - torch.tensor(mean).view(-1, 1, 1)) / torch.tensor(std).view(-1, 1, 1)
    F.hflip = lambda img: img.transpose(Image.FLIP_LEFT_RIGHT)
    F.crop = lambda img, top, left, h, w: img.crop((left, top, left + w, top + h))

    class Compose:
        def __init__(self, ts):
            self.transforms = ts

        def __call__(self, x):
            for t in self.transforms:
                x = t(x)
            return x
    T.Compose = Compose


def summarize(t):
    t = t.detach()
    flat = t.reshape(-1)
    idx = torch.linspace(0, flat.numel() - 1, min(512, flat.numel())).long()
    return {"shape": tuple(t.shape), "dtype": str(t.dtype), "norm": float(flat.double().norm()), "idx": idx,
            "sample": flat[idx].clone()}


def record_sample(s, root):
    return {"frames": [summarize(f) for f in s["frames"]], "masks": [tuple(m.shape) for m in s["masks"]],
            "actions": list(s["actions"]), "n_objects": [len(o) for o in s["object_ids"]],
            "category_ids": [c.clone() for c in s["category_ids"]], "boxes": [b.clone() for b in s["boxes"]],
            "episode_ids": s["episode_ids"], "initial_image_path": os.path.relpath(s["initial_image_path"], root)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    args = ap.parse_args()
    root = os.path.join(HERE, "data")
    ann = write_dataset(root)
    install_functional()
    sys.path.insert(0, args.ref)
    from datasets.interactive_dataset import InteractiveDaatset
    from datasets.sequence_dataset import SequenceDataset
    from utils.storage_utils import collate_fn
    from utils.transform_utis import transform
    imgs = os.path.join(root, "imgs")
    ds = SequenceDataset(imgs + "/", ann, "test", transform=transform)
    G = {"len": len(ds), "samples": [record_sample(ds[i], root) for i in range(len(ds))]}
    # an explicit action script through __getitem__(idx, actions) (what the interactive evaluator replays)
    G["scripted"] = record_sample(ds.__getitem__(1, actions=["MoveBack", "MoveBack", "RotateRight", "MoveAhead", "RotateLeft"]), root)
    batch = collate_fn([ds[0], ds[1]])
    G["collate"] = {"frames": summarize(batch["frames"]), "masks": tuple(batch["masks"].shape), "actions": batch["actions"].clone(),
                    "episode_ids": batch["episode_ids"].clone(), "category_ids": [[c.clone() for c in ep] for ep in batch["category_ids"]],
                    "initial_image_path": [os.path.relpath(p, root) for p in batch["initial_image_path"]]}
    env = InteractiveDaatset(imgs, ann, "test", transform=transform)
    trace = []
    for script in ([2, 0, 3, 1], [1, 1, 0, 2], [3, 2, 2, 0], [0, 0, 0, 0]):   # four resets: wraps around three scenes
        d = env.reset()
        steps = [{"frames": summarize(d["frames"]), "actions": d["actions"].clone(), "episode_ids": int(d["episode_ids"]),
                  "boxes": [b.clone() for b in d["boxes"][0]], "category_ids": [c.clone() for c in d["category_ids"][0]],
                  "initial_image_path": [os.path.relpath(p, root) for p in d["initial_image_path"]]}]
        for a in script:
            d = env.step(a)
            steps.append({"frames": summarize(d["frames"]), "actions": d["actions"].clone(), "episode_ids": int(d["episode_ids"]),
                          "boxes": [b.clone() for b in d["boxes"][0]], "category_ids": [c.clone() for c in d["category_ids"][0]],
                          "initial_image_path": [os.path.relpath(p, root) for p in d["initial_image_path"]]})
        trace.append({"script": script, "steps": steps})
    G["interactive"] = trace
    torch.save(G, os.path.join(HERE, "golden_data.pt"))
    print("wrote golden_data.pt:", G["len"], "episodes;", sum(len(t["steps"]) for t in trace), "interactive observations")


if __name__ == "__main__":
    main()
