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


def best_path_labels(path_storage, roots, actions, rewards, mine=None):
    """The PathStorage bookkeeping of reference models/interactron.py:109-115 for a run of episodes, in order: add each
    episode's (actions, reward) to the trie of its root image, then read that episode's label back.  ``mine`` (optional
    set of positions) selects the episodes whose labels are returned -- under data parallelism every rank replays the
    WHOLE global batch in global order (rewards of the other ranks' episodes arrive by ``exchange_rewards``) so that its
    tries, and hence its labels, are exactly those of a single process; it needs labels only for its own episodes."""
    labels = []
    for i, (root, acts, rew) in enumerate(zip(roots, actions, rewards)):
        store = path_storage.setdefault(root, PathStorage())
        store.add_path(acts, rew)
        if mine is None or i in mine:
            labels.append(store.get_label(acts))
    return labels


def exchange_rewards(local_rewards, slots, total):
    """Data-parallel hand-over of the per-episode rewards: this rank's ``local_rewards`` belong to positions ``slots`` of a
    run of ``total`` episodes; returns all ``total`` rewards (list of float).  One tiny all-reduce (SUM of disjoint
    one-hot placements) -- every rank of the group must call it the same number of times."""
    import torch.distributed as dist
    on_gpu = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1 and dist.get_backend() == "nccl"
    buf = torch.zeros(total, dtype=torch.float64, pin_memory=on_gpu)
    for s, r in zip(slots, local_rewards):
        buf[s] = r
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        if on_gpu:   # RCCL reduces device memory: pinned staging both ways, no pageable (host-blocking, stream-draining) copy
            dev = buf.to("cuda", non_blocking=True)
            dist.all_reduce(dev, op=dist.ReduceOp.SUM)
            buf.copy_(dev, non_blocking=True)
            torch.cuda.current_stream().synchronize()
        else:
            dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    return buf.tolist()


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
