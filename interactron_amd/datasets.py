"""Data path in front of the hot path ("next" row N3 of SURVEY.md 8f): the reference's rollout datasets and image /
box transforms, restated on PIL + numpy (torchvision is not a dependency).

reference: datasets/sequence_dataset.py:10-97, datasets/interactive_dataset.py:10-225 (annotation walk: ``data[i] =
{scene_name, root, state_table[state] = {detections{id: {category_id, bbox xywh}}, actions{name: next_state}}}``,
``metadata.actions``), utils/transform_utis.py:5-22, models/detr_models/util/transforms.py (resize / flip / crop /
normalise to cxcywh in [0,1]), utils/storage_utils.py:53-64 (collate_fn, in interactron_amd.storage).
"""
import json
import random

import numpy as np
import torch
from PIL import Image
from torch.utils.data import Dataset

from .constants import ACTIONS

MEAN = np.array([0.485, 0.456, 0.406], dtype=np.float32)
STD = np.array([0.229, 0.224, 0.225], dtype=np.float32)
TEST_ACTIONS = ["RotateLeft", "MoveAhead", "RotateLeft", "MoveBack", "RotateRight"]


# ------------------------------------------------------------------------------------------------------------
# transforms: (PIL image, target dict | None) -> (tensor [3,H,W], target | None)
# ------------------------------------------------------------------------------------------------------------
def _resize(img, target, size, max_size=None):
    w, h = img.size
    if max_size is not None:
        lo, hi = float(min(w, h)), float(max(w, h))
        if hi / lo * size > max_size:
            size = int(round(max_size * lo / hi))
    if (w <= h and w == size) or (h <= w and h == size):
        oh, ow = h, w
    elif w < h:
        ow, oh = size, int(size * h / w)
    else:
        oh, ow = size, int(size * w / h)
    out = img.resize((ow, oh), Image.BILINEAR)
    if target is None:
        return out, None
    rw, rh = float(ow) / float(w), float(oh) / float(h)
    target = dict(target)
    target["boxes"] = target["boxes"] * torch.as_tensor([rw, rh, rw, rh])
    if "areas" in target:
        target["areas"] = target["areas"] * (rw * rh)
    return out, target


def _hflip(img, target):
    w, _ = img.size
    out = img.transpose(Image.FLIP_LEFT_RIGHT)
    if target is not None:
        target = dict(target)
        b = target["boxes"]
        target["boxes"] = b[:, [2, 1, 0, 3]] * torch.as_tensor([-1, 1, -1, 1]) + torch.as_tensor([w, 0, w, 0])
    return out, target


def _crop(img, target, top, left, h, w):
    out = img.crop((left, top, left + w, top + h))
    if target is not None:
        target = dict(target)
        b = target["boxes"] - torch.as_tensor([left, top, left, top], dtype=torch.float32)
        b = torch.min(b.reshape(-1, 2, 2), torch.as_tensor([w, h], dtype=torch.float32)).clamp(min=0)
        keep = torch.all(b[:, 1, :] > b[:, 0, :], dim=1)
        target["boxes"] = b.reshape(-1, 4)[keep]
        for k in ("labels", "areas", "iscrowd"):
            if k in target:
                target[k] = target[k][keep]
    return out, target


def _to_normalised_tensor(img, target):
    arr = np.asarray(img.convert("RGB"), dtype=np.float32) / 255.0
    x = torch.from_numpy(((arr - MEAN) / STD).transpose(2, 0, 1).copy())
    if target is not None:
        target = dict(target)
        h, w = x.shape[-2:]
        b = target["boxes"]
        cxcywh = torch.stack([(b[:, 0] + b[:, 2]) / 2, (b[:, 1] + b[:, 3]) / 2, b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]], -1)
        target["boxes"] = cxcywh / torch.tensor([w, h, w, h], dtype=torch.float32)
    return x, target


def transform(img, target):
    """Test-time transform (utils/transform_utis.py:5-11): resize to 300 + ImageNet normalisation."""
    img, target = _resize(img, target, 300, max_size=300)
    return _to_normalised_tensor(img, target)


def train_transform(img, target):
    """Training augmentation (utils/transform_utis.py:13-22): flip, random resize, 300x300 random crop, resize."""
    if random.random() < 0.5:
        img, target = _hflip(img, target)
    img, target = _resize(img, target, random.choice([400, 500, 600]))
    w = random.randint(300, min(img.width, 300))
    h = random.randint(300, min(img.height, 300))
    top, left = random.randint(0, img.height - h), random.randint(0, img.width - w)
    img, target = _crop(img, target, top, left, h, w)
    img, target = _resize(img, target, 300, max_size=300)
    return _to_normalised_tensor(img, target)


# ------------------------------------------------------------------------------------------------------------
# datasets
# ------------------------------------------------------------------------------------------------------------
class _RolloutBase(Dataset):
    label_offset = 1   # reference adds 1 to category_id (sequence_dataset.py:62); InteractiveDaatset.__getitem__ does not

    def __init__(self, img_root, annotations_path, mode="train", transform=None):
        assert mode in ["train", "test"], "Only train and test modes supported"
        self.mode = mode
        with open(annotations_path) as f:
            self.annotations = json.load(f)
        self.img_dir = img_root if img_root[-1] != "/" else img_root[:-1]
        self.transform = transform

    def __len__(self):
        return len(self.annotations["data"])

    def _path(self, scene, state_name):
        return "{}/{}/{}.jpg".format(self.img_dir, scene["scene_name"], state_name)

    def _frame(self, scene, state_name, label_offset):
        state = scene["state_table"][state_name]
        frame = Image.open(self._path(scene, state_name))
        imgw, imgh = frame.size
        mask = torch.zeros((imgw, imgh), dtype=torch.long)
        ids, cls, boxes = [], [], []
        for k, v in state["detections"].items():
            ids.append(hash(k.encode()))
            cls.append(v["category_id"] + label_offset)
            x, y, w, h = v["bbox"]
            boxes.append([x, y, x + w, y + h])
        target = None
        if boxes:
            b = torch.tensor(boxes, dtype=torch.float)
            target = {"boxes": b, "labels": torch.tensor(cls, dtype=torch.long),
                      "areas": (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1]), "iscrowd": torch.zeros(len(ids)).bool()}
        if self.transform:
            frame, target = self.transform(frame, target)
        return (frame, mask, ids, target["labels"] if target is not None else torch.zeros(0).long(),
                target["boxes"] if target is not None else torch.zeros(0, 4), state)

    def _rollout(self, scene, actions, steps, label_offset, next_state):
        state_name = scene["root"]
        out = {"frames": [], "masks": [], "object_ids": [], "category_ids": [], "boxes": []}
        for i in range(steps):
            frame, mask, ids, labels, boxes, state = self._frame(scene, state_name, label_offset)
            out["frames"].append(frame)
            out["masks"].append(mask)
            out["object_ids"].append(ids)
            out["category_ids"].append(labels)
            out["boxes"].append(boxes)
            if i < steps - 1:
                state_name = next_state(i, state)
        return out


class SequenceDataset(_RolloutBase):
    """Five-frame random (train) / fixed (test) rollouts (reference datasets/sequence_dataset.py)."""

    def __getitem__(self, idx, actions=None):
        if torch.is_tensor(idx):
            idx = idx.tolist()
        scene = self.annotations["data"][idx]
        if self.mode == "test" and actions is None:
            actions = list(TEST_ACTIONS)
        if actions is None:
            actions = [random.choice(self.annotations["metadata"]["actions"]) for _ in range(5)]
        out = self._rollout(scene, actions, 5, 1, lambda i, state: state["actions"][actions[i]])
        out.update({"actions": [ACTIONS.index(a) for a in actions], "episode_ids": idx,
                    "initial_image_path": self._path(scene, scene["root"])})
        return out


class InteractiveDataset(_RolloutBase):
    """Agent-driven rollouts: ``reset()`` opens the next scene with one frame, ``step(action)`` replays the scene from
    its root along all actions so far (reference datasets/interactive_dataset.py:30-154; batch dim of 1)."""

    def __init__(self, img_root, annotations_path, mode="train", transform=None):
        super().__init__(img_root, annotations_path, mode, transform)
        self.idx = -1
        self.actions = []

    def _episode(self):
        scene = self.annotations["data"][self.idx]
        acts = self.actions
        out = self._rollout(scene, acts, len(acts) + 1, 1, lambda i, state: state["actions"][acts[i]])
        return {"frames": torch.stack(out["frames"], dim=0).unsqueeze(0),
                "masks": torch.stack(out["masks"], dim=0).unsqueeze(0),
                "actions": torch.tensor([ACTIONS.index(a) for a in acts], dtype=torch.long).unsqueeze(0),
                "object_ids": out["object_ids"], "category_ids": [out["category_ids"]], "boxes": [out["boxes"]],
                "episode_ids": self.idx, "initial_image_path": [self._path(scene, scene["root"])]}

    def reset(self):
        self.idx += 1
        if self.idx >= len(self.annotations["data"]):
            self.idx = 0
        self.actions = []
        return self._episode()

    def step(self, action):
        self.actions.append(ACTIONS[action])
        return self._episode()

    def __getitem__(self, idx):
        if torch.is_tensor(idx):
            idx = idx.tolist()
        scene = self.annotations["data"][idx]
        actions = [random.choice(self.annotations["metadata"]["actions"]) for _ in range(5)]
        if self.mode == "test":
            nxt = lambda i, state: state["actions"][actions[i]]
        else:
            nxt = lambda i, state: random.choice(list(scene["state_table"]))
        out = self._rollout(scene, actions, 5, 0, nxt)   # (no +1 label offset here, reference :190)
        out.update({"actions": [ACTIONS.index(a) for a in actions], "episode_ids": idx,
                    "initial_image_path": self._path(scene, scene["root"])})
        return out


InteractiveDaatset = InteractiveDataset   # the reference's spelling (datasets/interactive_dataset.py:10)


# ------------------------------------------------------------------------------------------------------------
# data-parallel batches: every rank agrees on the global batch, decodes only its own episodes
# ------------------------------------------------------------------------------------------------------------
class _RankItems(Dataset):
    """dataset[(idx, actions)] -> one decoded rollout along `actions` (what a DataLoader worker of this rank produces)"""

    def __init__(self, base):
        self.base = base

    def __len__(self):
        return len(self.base)

    def __getitem__(self, item):
        idx, actions = item
        return self.base.__getitem__(idx, actions=list(actions) if actions is not None else None)


class EpisodeBatchLoader:
    """The batches ``DataLoader(SequenceDataset, batch_size=B, shuffle=...)`` of the reference trainer
    (engine/interactron_trainer.py:78-90) under one-process-per-GPU data parallelism.

    Every rank derives the SAME global batches from one seeded generator -- the permutation of an epoch from
    ``seed + epoch``, each training rollout's five random actions (sequence_dataset.py:42-43) from ``(seed, epoch, idx)`` --
    but its DataLoader workers open and transform only the episodes ``rank::world`` of each batch (1 / world of the JPEG
    decodes).  What the others contribute to this rank's PathStorage replay (models/interactron.py:109-115) is their root
    image path and their actions, both read off the annotation table without touching an image.  Yields
    ``(local batch with dp_roots / dp_actions / dp_index / dp_world, global batch size)``; a rank whose shard of a short last
    batch is empty gets a batch with zero episodes."""

    def __init__(self, dataset, batch_size, shuffle, rank=0, world=1, seed=42, num_workers=0, pin_memory=True, collate=None):
        from torch.utils.data.dataloader import DataLoader
        self.dataset, self.batch_size, self.shuffle, self.rank, self.world, self.seed = dataset, batch_size, shuffle, rank, world, seed
        self.num_workers, self.pin_memory, self.collate, self._DataLoader = num_workers, pin_memory, collate, DataLoader
        self.epoch = 0

    def __len__(self):
        return (len(self.dataset) + self.batch_size - 1) // self.batch_size

    def _plan(self, epoch):
        """global batches of one epoch: [[(idx, actions | None), ...], ...] -- identical on every rank"""
        n = len(self.dataset)
        if self.shuffle:
            order = torch.randperm(n, generator=torch.Generator().manual_seed(self.seed + epoch)).tolist()
        else:
            order = list(range(n))
        names = self.dataset.annotations["metadata"]["actions"]
        plan = []
        for b0 in range(0, n, self.batch_size):
            batch = []
            for idx in order[b0:b0 + self.batch_size]:
                if self.dataset.mode == "test":
                    acts = list(TEST_ACTIONS)
                else:
                    rng = random.Random((self.seed * 1000003 + epoch) * 1000003 + idx)
                    acts = [rng.choice(names) for _ in range(5)]
                batch.append((idx, acts))
            plan.append(batch)
        return plan

    def __iter__(self):
        plan = self._plan(self.epoch)
        self.epoch += 1
        mine = [[item for item in batch[self.rank::self.world]] for batch in plan]
        loader = self._DataLoader(_RankItems(self.dataset), batch_sampler=[b for b in mine if b], num_workers=self.num_workers,
                                  pin_memory=self.pin_memory, collate_fn=self.collate)
        it = iter(loader)
        for batch, local in zip(plan, mine):
            data = next(it) if local else _empty_batch()
            data["dp_roots"] = [self.dataset._path(self.dataset.annotations["data"][idx], self.dataset.annotations["data"][idx]["root"])
                                for idx, _ in batch]
            data["dp_actions"] = [[ACTIONS.index(a) for a in acts[:4]] for _, acts in batch]
            data["dp_index"] = list(range(self.rank, len(batch), self.world))
            data["dp_world"] = self.world
            yield data, len(batch)


def _empty_batch():
    return {"frames": torch.zeros(0, 5, 3, 1, 1), "masks": torch.zeros(0, 5, 1, 1, dtype=torch.long),
            "actions": torch.zeros(0, 5, dtype=torch.long), "object_ids": [], "category_ids": [], "boxes": [],
            "episode_ids": torch.zeros(0, dtype=torch.long), "initial_image_path": []}
