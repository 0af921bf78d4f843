"""Training loops with the reference's interface (engine/interactron_trainer.py, interactron_random_trainer.py,
direct_supervision_trainer.py): ``Trainer(model, config, evaluator=None).train()``.

Same schedule as the reference: an evaluation first, then ``MAX_EPOCHS - 1`` epochs of [train epoch, test-loss epoch +
evaluator, log], a uniform running average of the last ``SAVE_WINDOW`` epochs' ``state_dict`` saved as
``{"model": ...}`` (``detector.pt``).  What differs is the MI355X-native update: instead of ``nn.DataParallel`` +
``clip_grad_norm_`` + two ``torch.optim.Adam`` it is one process per GPU, every rank running the episodes
``rank::world`` of a batch, ONE RCCL all-reduce of the flat gradient buffer and the fused clip + Adam kernels
(``interactron_amd.trainer.FlatOuterStep``) -- numerically the same update (fixture G16).
"""
import math
import os
from datetime import datetime

import numpy as np
import torch

from ..datasets import EpisodeBatchLoader, SequenceDataset, train_transform, transform
from ..storage import collate_fn
from ..trainer import FlatOuterStep, init_distributed
from .logging import TBLogger


def _to_device(data, device):
    data["frames"] = data["frames"].to(device)
    data["masks"] = data["masks"].to(device)
    data["category_ids"] = [[j.to(device) for j in i] for i in data["category_ids"]]
    data["boxes"] = [[j.to(device) for j in i] for i in data["boxes"]]
    return data


class _TrainerBase:
    fixed_lrs = None          # InteractronRandomTrainer hard-codes 1e-5 / 1e-4 (interactron_random_trainer.py:70-71)
    shard_by_root = True      # multi-GPU: rank r takes episodes r::world of every batch; the batch's roots + actions travel
                              # with it so that every rank replays the whole batch's PathStorage updates (EpisodeBatchLoader)
    replica_check_every = 100 # outer steps between two replica-equality checks (flat parameter checksum over the ranks)
    pass_train_flag = False   # ... and calls model(data, train=is_train) (:91)

    def __init__(self, model, config, evaluator=None, train_dataset=None, test_dataset=None):
        self.model, self.config, self.evaluator = model, config, evaluator
        self.rank, self.local_rank, self.world = init_distributed()
        self.out_dir = os.path.join(config.TRAINER.OUTPUT_DIRECTORY, datetime.now().strftime("%m-%d-%Y:%H:%M:%S"))
        os.makedirs(self.out_dir, exist_ok=True)
        self.logger = TBLogger(os.path.join(self.out_dir, "logs"))
        self.model.set_logger(self.logger)
        self.checkpoint_path = os.path.join(self.out_dir, "detector.pt")
        self.saved_checkpoints = None
        d = config.DATASET
        self.train_dataset = train_dataset if train_dataset is not None else SequenceDataset(
            d.TRAIN.IMAGE_ROOT, d.TRAIN.ANNOTATION_ROOT, d.TRAIN.MODE, transform=train_transform)
        self.test_dataset = test_dataset if test_dataset is not None else SequenceDataset(
            d.TEST.IMAGE_ROOT, d.TEST.ANNOTATION_ROOT, d.TEST.MODE, transform=transform)
        assert torch.cuda.is_available(), "interactron_amd trains on the HIP kernels only (no CPU path)"
        torch.cuda.set_device(self.local_rank % torch.cuda.device_count())
        self.device = torch.cuda.current_device()
        self.model.to(self.device)

    # ---- checkpoint averaging (reference :48-65) ----------------------------------------------------------
    def record_checkpoint(self, w=1.0):
        sd = self.model.state_dict()
        if self.saved_checkpoints is None:
            self.saved_checkpoints = {k: w * v for k, v in sd.items()}
        else:
            for k, v in sd.items():
                self.saved_checkpoints[k] += w * v

    def save_checkpoint(self):
        sd = self.saved_checkpoints if self.saved_checkpoints is not None else self.model.state_dict()
        if self.rank == 0:
            torch.save({"model": sd}, self.checkpoint_path)

    # ---- loss bookkeeping ------------------------------------------------------------------------------------
    def _total_loss(self, losses):
        det = losses["loss_detector_ce"] + 5 * losses["loss_detector_giou"] + 2 * losses["loss_detector_bbox"]
        sup = losses["loss_supervisor_ce"] + 5 * losses["loss_supervisor_giou"] + 2 * losses["loss_supervisor_bbox"]
        return det + sup

    def _make_outer(self, cfg):
        det_lr, fus_lr = self.fixed_lrs if self.fixed_lrs else (cfg.DETECTOR_LR, cfg.SUPERVISOR_LR)
        return FlatOuterStep(self.model, detector_lr=det_lr, fusion_lr=fus_lr, max_norm=cfg.GRAD_NORM_CLIP)

    def _base_lr(self, cfg):
        return self.fixed_lrs[1] if self.fixed_lrs else cfg.SUPERVISOR_LR

    def _tokens_of(self, global_episodes, data):
        """LR-schedule tokens of one GLOBAL batch (reference interactron_trainer.py:116: episodes x frames)"""
        return global_episodes * 5

    # ---- replica hygiene under data parallelism ------------------------------------------------------------------
    def _sync_replicas(self, outer):
        import torch.distributed as dist
        if self.world > 1 and dist.is_initialized():
            dist.broadcast(outer.flat.params, src=0)
            with torch.no_grad():
                for b in self.model.buffers():
                    dist.broadcast(b, src=0)      # the buffer itself (bumps its version: cached FrozenBN folds see the change)
            if hasattr(self.model, "invalidate_graphs"):
                self.model.invalidate_graphs()    # captured graphs / folds made before the sync read the old values
            from .. import hipops
            hipops.weights_changed()              # (the flat buffer was rewritten under the parameters' views)
            if getattr(outer, "shadow_b16", False):
                outer.flat.sync_b16()             # (16-bit mode: the bf16 shadow of the flat buffer follows)
            self._check_replicas(outer)

    def _check_replicas(self, outer):
        """Every rank must hold the same parameters: (sum, sum of squares) of the flat buffer, MIN and MAX over the ranks."""
        import torch.distributed as dist
        p = outer.flat.params.double()
        stat = torch.stack([p.sum(), (p * p).sum()])
        finite = torch.isfinite(stat).all().to(stat.dtype).reshape(1)
        stat = torch.cat([torch.nan_to_num(stat, nan=0.0, posinf=0.0, neginf=0.0), finite])
        lo, hi = stat.clone(), stat.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        if float(lo[2]) == 0.0:    # (every rank sees it and raises: NaN != NaN would otherwise read as "drifted apart")
            raise RuntimeError("non-finite parameters on %s rank(s): the step diverged (this rank: %s)"
                               % ("every" if float(hi[2]) == 0.0 else "some", "non-finite" if float(finite) == 0.0 else "finite"))
        if not torch.equal(lo, hi):
            raise RuntimeError("data-parallel replicas have drifted apart: parameter checksums differ between ranks "
                               "(min %s, max %s)" % (lo[:2].tolist(), hi[:2].tolist()))

    def _decay_lr(self, cfg, outer, global_tokens):
        """Cosine schedule of the reference (interactron_trainer.py:113-127), counted in tokens of the GLOBAL batch so that
        every rank -- including one whose shard of a short last batch is empty -- holds the same counter and therefore
        the same learning rate (the replicas apply Adam independently; different rates would let them drift apart)."""
        lr = self._base_lr(cfg)
        if cfg.LR_DECAY:
            self.tokens += global_tokens
            if self.tokens < cfg.WARMUP_TOKENS:
                mult = float(self.tokens) / float(max(1, cfg.WARMUP_TOKENS))
            else:
                prog = float(self.tokens - cfg.WARMUP_TOKENS) / float(max(1, cfg.FINAL_TOKENS - cfg.WARMUP_TOKENS))
                mult = max(0.1, 0.5 * (1.0 + math.cos(math.pi * prog)))
            lr = lr * mult
            outer.lrs[-1] = lr          # the reference decays only the supervisor (fusion) optimiser
        return lr

    def train(self):
        model, cfg = self.model, self.config.TRAINER
        outer = self._make_outer(cfg)
        model.train()
        state = {"epoch": 0}

        # replicas start from rank 0's weights (every rank built the same model, but a checkpoint path or an RNG-dependent
        # initialiser may differ between hosts) and are checked against each other every `replica_check_every` steps
        self._sync_replicas(outer)
        loaders = {}

        def run_epoch(split):
            is_train = split == "train"
            if split not in loaders:   # (kept across epochs: the loader's epoch counter drives the shared permutation)
                loaders[split] = EpisodeBatchLoader(self.train_dataset if is_train else self.test_dataset, cfg.BATCH_SIZE,
                                                    shuffle=is_train, rank=self.rank, world=self.world, seed=42,
                                                    num_workers=cfg.NUM_WORKERS, pin_memory=True, collate=collate_fn)
            tag = "Train" if is_train else "Test"
            loss_list = []
            for it, (data, global_episodes) in enumerate(loaders[split]):
                global_tokens = self._tokens_of(global_episodes, data)   # of the GLOBAL batch: identical on every rank
                if self.world == 1 or not self.shard_by_root:
                    for k in ("dp_roots", "dp_actions", "dp_index", "dp_world"):
                        data.pop(k, None)
                data = _to_device(data, self.device)
                if data["frames"].shape[0] == 0:    # a short last batch can leave a rank without episodes: it still
                    if hasattr(model, "dp_idle_step"):   # has to take part in the reward exchange, the gradient
                        model.dp_idle_step(data)         # all-reduce and the LR schedule
                    if is_train:
                        outer.step()
                        self._decay_lr(cfg, outer, global_tokens)
                    continue
                _, losses = model(data, train=is_train) if self.pass_train_flag else model(data)
                for name, comp in losses.items():
                    self.logger.add_value("{}/{}".format(tag, name), comp.mean())
                total = self._total_loss(losses)
                self.logger.add_value("{}/Total Loss".format(tag), total.mean())
                loss_list.append(total.item())
                if is_train:
                    outer.step()   # all-reduce(SUM) of the flat grads + clip_grad_norm_ + Adam x2, grads zeroed
                    lr = self._decay_lr(cfg, outer, global_tokens)
                    self.logger.add_value("{}/LR".format(tag), lr)
                    self.steps += 1
                    if self.world > 1 and self.steps % self.replica_check_every == 0:
                        self._check_replicas(outer)
                    if self.rank == 0 and it % 10 == 0:
                        print("epoch %d iter %d: train loss %.5f. lr %e" % (state["epoch"], it, float(np.mean(loss_list)), lr))
            if not is_train:
                return float(np.mean(loss_list)) if loss_list else float("nan")

        def run_evaluation():
            run_epoch("test")   # (the reference's test epoch also back-propagates inside model(); those grads are dropped)
            outer.zero_grads()
            if self.evaluator is None:
                return None
            m50, m, tps, fps, fns = self.evaluator.evaluate(save_results=False)
            for k, v in (("TP", tps), ("FP", fps), ("FN", fns), ("mAP_50", m50), ("mAP", m)):
                self.logger.add_value("Test/" + k, float(v))
            return m

        self.tokens, self.steps = 0, 0
        run_evaluation()
        self.logger.log_values()
        for epoch in range(1, cfg.MAX_EPOCHS):
            state["epoch"] = epoch
            run_epoch("train")
            if self.test_dataset is not None and self.evaluator is not None:
                run_evaluation()
            self.logger.log_values()
            if self.test_dataset is not None and cfg.MAX_EPOCHS - epoch <= cfg.SAVE_WINDOW:
                self.record_checkpoint(w=1 / cfg.SAVE_WINDOW)
        self.save_checkpoint()


class InteractronTrainer(_TrainerBase):
    """reference engine/interactron_trainer.py:22-163."""


class InteractronRandomTrainer(_TrainerBase):
    """reference engine/interactron_random_trainer.py (fixed learning rates, ``model(data, train=...)``)."""

    fixed_lrs = (1e-5, 1e-4)
    pass_train_flag = True


class DirectSupervisionTrainer(_TrainerBase):
    """reference engine/direct_supervision_trainer.py: one Adam over ``LEARNING_RATE`` (configs 1 and 2), loss
    1*CE + 5*L1 + 2*GIoU, LR schedule counted in batches."""

    def _total_loss(self, losses):
        return losses["loss_detector_ce"] + 5 * losses["loss_detector_bbox"] + 2 * losses["loss_detector_giou"]

    def _make_outer(self, cfg):
        # one optimiser over every trainable tensor (``detr`` holds its network as ``.model``, ``detr_multiframe`` as
        # ``.detector`` + ``.fusion``)
        return FlatOuterStep(self.model, max_norm=cfg.GRAD_NORM_CLIP, groups=[list(self.model.parameters())],
                             lrs=[cfg.LEARNING_RATE])

    def _base_lr(self, cfg):
        return cfg.LEARNING_RATE

    def _tokens_of(self, global_episodes, data):
        return global_episodes
