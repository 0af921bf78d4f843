#!/usr/bin/env python3
"""``python train.py --config_file configs/interactron.yaml`` -- the reference's train.py:13-24 on the MI355X path.
Multi-GPU: ``python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 train.py --config_file ...``."""
import random

import numpy
import torch

from interactron_amd import build_evaluator, build_model, build_trainer, get_args, get_config, manual_seed


def train():
    random.seed(42)
    torch.manual_seed(42)
    torch.cuda.manual_seed(42)
    numpy.random.seed(42)
    manual_seed(42)   # dropout stream of the HIP kernels
    args = get_args()
    cfg = get_config(args.config_file)
    from interactron_amd.trainer import init_distributed
    rank, _, _ = init_distributed()   # torchrun: binds cuda:LOCAL_RANK and creates the RCCL group before anything touches the GPU
    # every rank draws its OWN dropout masks and first-order frames (a single process draws distinct ones per episode; with one
    # seed the replicas' masks would coincide).  What the ranks must agree on -- the batches -- comes from EpisodeBatchLoader's
    # own generator.
    manual_seed(42 + rank)
    random.seed(42 + rank)
    model = build_model(cfg.MODEL)
    evaluator = build_evaluator(model, cfg)
    trainer = build_trainer(model, cfg, evaluator=evaluator)
    trainer.train()


if __name__ == "__main__":
    train()
