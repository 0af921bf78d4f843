#!/usr/bin/env python3
"""``python evaluate.py --config_file configs/interactron.yaml`` -- the reference's evaluate.py:9-14 on the MI355X path."""
from interactron_amd import build_evaluator, build_model, get_args, get_config


def evaluate():
    args = get_args()
    cfg = get_config(args.config_file)
    model = build_model(cfg.MODEL)
    evaluator = build_evaluator(model, cfg, load_checkpoint=True)
    evaluator.evaluate(save_results=True)


if __name__ == "__main__":
    evaluate()
