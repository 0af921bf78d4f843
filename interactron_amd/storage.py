"""Host-side policy label store and batch collation (reference utils/storage_utils.py).

``PathStorage`` is a trie over action prefixes that remembers, at every prefix, the action of the lowest-reward
(= lowest detection loss) path seen so far; ``get_label`` reads those best actions back along a path.
"""
import torch


class _Node:
    __slots__ = ("cost", "action", "children")

    def __init__(self):
        self.cost, self.action, self.children = float("inf"), None, {}


def _as_int(a):
    return int(a.item()) if hasattr(a, "item") else int(a)


class PathStorage:
    def __init__(self):
        self.root = _Node()

    def add_path(self, path, ifga):
        node = self.root
        for a in path:
            a = _as_int(a)
            if ifga < node.cost:
                node.cost, node.action = ifga, a
            if a not in node.children:
                node.children[a] = _Node()
            node = node.children[a]

    def get_label(self, path):
        actions, node = [], self.root
        for a in path:
            actions.append(node.action)
            node = node.children[_as_int(a)]
        return actions


def collate_fn(batch):
    return {
        "frames": torch.stack([torch.stack(b["frames"]) for b in batch]),
        "masks": torch.stack([torch.stack(b["masks"]) for b in batch]),
        "actions": torch.stack([torch.tensor(b["actions"], dtype=torch.long) for b in batch]),
        "object_ids": [[torch.tensor(inst, dtype=torch.long) for inst in b["object_ids"]] for b in batch],
        "category_ids": [list(b["category_ids"]) for b in batch],
        "boxes": [list(b["boxes"]) for b in batch],
        "episode_ids": torch.stack([torch.tensor(b["episode_ids"], dtype=torch.long) for b in batch]),
        "initial_image_path": [b["initial_image_path"] for b in batch],
    }
