"""Evaluators with the reference's interface (engine/random_policy_evaluator.py, engine/interactive_evaluator.py):
``Evaluator(model, config, load_checkpoint=False).evaluate(save_results=False) -> (AP50, AP, #tp, #fp, #fn)``; with
``save_results=True`` the six AP numbers are printed and ``results.json`` is written under
``EVALUATOR.OUTPUT_DIRECTORY/<timestamp>/``.  Image dumps (the reference draws boxes with PIL fonts) are left out.
"""
import json
import os
from datetime import datetime

import torch
from torch.utils.data.dataloader import DataLoader

from ..datasets import InteractiveDataset, SequenceDataset, transform
from ..storage import collate_fn
from . import metrics


def _to_device(data, device):
    data["frames"] = data["frames"].to(device)
    data["masks"] = data["masks"].to(device)
    data["category_ids"] = [[j.to(device) for j in i] for i in data["category_ids"]]
    data["boxes"] = [[j.to(device) for j in i] for i in data["boxes"]]
    return data


class _EvaluatorBase:
    dataset_cls = SequenceDataset

    def __init__(self, model, config, load_checkpoint=False, dataset=None):
        self.model = model
        if load_checkpoint:
            self.model.load_state_dict(torch.load(config.EVALUATOR.CHECKPOINT, map_location=torch.device("cpu"))["model"],
                                       strict=False)
        self.test_dataset = dataset if dataset is not None else self.dataset_cls(
            config.DATASET.TEST.IMAGE_ROOT, config.DATASET.TEST.ANNOTATION_ROOT, config.DATASET.TEST.MODE, transform=transform)
        self.config = config
        # one process per GPU: bind this rank's device (LOCAL_RANK) before anything is moved or launched -- train.py
        # builds the evaluator BEFORE the trainer, and an evaluator that caches the then-current device 0 uploads every
        # rank's episodes to GPU 0 while the trainer later moves the model to cuda:LOCAL_RANK
        if torch.cuda.is_available():
            from ..trainer import init_distributed
            init_distributed()
            self.model.to(torch.cuda.current_device())
        self.out_dir = config.EVALUATOR.OUTPUT_DIRECTORY + "/" + datetime.now().strftime("%m-%d-%Y-%H:%M:%S") + "/"

    @property
    def device(self):
        """Where the model lives NOW (never a cached index: the trainer may have moved the model since __init__)."""
        return next(self.model.parameters()).device

    def _episodes(self, mine):
        """Yields (test-set indices, data, predictions) per evaluated batch, for the test episodes `mine` (ascending)."""
        raise NotImplementedError

    @staticmethod
    def _ranks():
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            return dist.get_rank(), dist.get_world_size()
        return 0, 1

    def evaluate(self, save_results=False):
        """Episodes are independent (SURVEY 8e "Eval"): rank r of W evaluates test episodes r::W, the per-episode
        detection lists are gathered (`all_gather_object`, a few KB) and merged back into test-set order, so the record
        list -- and therefore every AP number -- is exactly a single process's; every rank returns the same result, rank 0
        alone writes results.json."""
        rank, world = self._ranks()
        per_episode = []
        for indices, data, predictions in self._episodes(list(range(rank, len(self.test_dataset), world))):
            with torch.no_grad():
                for b in range(predictions["pred_boxes"].shape[0]):   # frame 0 of every episode is scored
                    per_episode.append((int(indices[b]), metrics.frame_detections(
                        predictions["pred_logits"][b][0], predictions["pred_boxes"][b][0], data["boxes"][b][0],
                        data["category_ids"][b][0], data["initial_image_path"][b])))
        if world > 1:
            import torch.distributed as dist
            shards = [None] * world
            dist.all_gather_object(shards, per_episode)
            per_episode = sorted((item for shard in shards for item in shard), key=lambda item: item[0])
        detections = [d for _, dets in per_episode for d in dets]
        n = {k: sum(1 for d in detections if d["type"] == k) for k in ("tp", "fp", "fn")}
        if not save_results:
            ap_50 = metrics.compute_ap(detections, nsamples=100, iou_thresholds=[0.5])
            ap = metrics.compute_ap(detections, nsamples=100, iou_thresholds=list(metrics.np.arange(0.5, 1.0, 0.05)))
            return ap_50, ap, n["tp"], n["fp"], n["fn"]
        s = metrics.summarize(detections)
        if rank == 0:
            print("AP_50:", s["AP_50"], "AP_75", s["AP_75"], "AP", s["AP"], "AP_small", s["AP_small"], "AP_medium",
                  s["AP_medium"], "AP_large", s["AP_large"])
            os.makedirs(self.out_dir, exist_ok=True)
            with open(self.out_dir + "results.json", "w") as f:
                json.dump({"AP_50": s["AP_50"], "detections": detections}, f)
        return s


class RandomPolicyEvaluator(_EvaluatorBase):
    """Fixed test rollout (SequenceDataset test actions), ``model.predict`` on the 5 frames."""

    dataset_cls = SequenceDataset

    def _episodes(self, mine):
        cfg = self.config.EVALUATOR
        self.model.eval()
        batches = [mine[i:i + cfg.BATCH_SIZE] for i in range(0, len(mine), cfg.BATCH_SIZE)]
        loader = DataLoader(self.test_dataset, batch_sampler=batches, pin_memory=torch.cuda.is_available(),
                            num_workers=cfg.NUM_WORKERS, collate_fn=collate_fn)
        for indices, data in zip(batches, loader):
            data = _to_device(data, self.device)
            yield indices, data, self.model.predict(data)


class InteractiveEvaluator(_EvaluatorBase):
    """The learned policy picks the four moves (``model.get_next_action``), then ``model.predict`` on the rollout."""

    dataset_cls = InteractiveDataset

    def _episodes(self, mine):
        env = self.test_dataset
        for idx in mine:
            self.model.eval()
            env.idx = idx - 1           # reset() opens scene idx + 1 (reference interactive_dataset.py:30-154 walks them in order)
            data = _to_device(env.reset(), self.device)
            for _ in range(4):
                data = _to_device(env.step(self.model.get_next_action(data)), self.device)
            yield [idx], data, self.model.predict(data)
