"""Trainer / evaluator orchestration ("next" rows N1/N2) end to end on a tiny synthetic dataset on disk: the reference's
entry-point sequence build_model -> build_evaluator -> build_trainer -> train() -> evaluate(), on the HIP path."""
import json
import os

import numpy as np
import pytest
import torch
from PIL import Image

from interactron_amd.constants import ACTIONS

pytestmark = pytest.mark.gpu

MODEL = dict(TYPE="interactron", WEIGHTS="procedural", NUM_CLASSES=1235, SET_COST_CLASS=1.0, SET_COST_BBOX=5.0,
             SET_COST_GIOU=2.0, NUM_LAYERS=4, NUM_HEADS=8, EMBEDDING_DIM=512, BLOCK_SIZE=2060, IMG_FEATURE_SIZE=256,
             OUTPUT_SIZE=512, BOX_EMB_SIZE=256, EMBEDDING_PDROP=0.1, RESIDUAL_PDROP=0.1, ATTENTION_PDROP=0.1, ADAPTIVE_LR=1e-3)


def make_dataset(tmp):
    root = os.path.join(tmp, "imgs")
    scenes = []
    for s in range(3):
        name = "FloorPlan%d" % s
        os.makedirs(os.path.join(root, name))
        states = ["s%d_%d" % (s, k) for k in range(3)]
        table = {}
        for k, st in enumerate(states):
            Image.fromarray(np.random.RandomState(10 * s + k).randint(0, 255, (300, 300, 3), dtype=np.uint8)).save(
                os.path.join(root, name, st + ".jpg"))
            table[st] = {"detections": {"o%d" % j: {"category_id": 10 + j, "bbox": [40 + 30 * j, 30 + 10 * j, 100, 80]} for j in range(k + 1)},
                         "actions": {a: states[(k + 1 + i) % 3] for i, a in enumerate(ACTIONS)}}
        scenes.append({"scene_name": name, "root": states[0], "state_table": table})
    ann = os.path.join(tmp, "ann.json")
    json.dump({"data": scenes, "metadata": {"actions": ACTIONS}}, open(ann, "w"))
    return root, ann


def test_train_then_evaluate_drop_in(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from interactron_amd import Config, build_evaluator, build_model, build_trainer, manual_seed
    root, ann = make_dataset(str(tmp_path))
    split = {"TYPE": "sequence", "MODE": "test", "IMAGE_ROOT": root, "ANNOTATION_ROOT": ann}
    cfg = Config(**{
        "MODEL": MODEL,
        "DATASET": {"TRAIN": dict(split, MODE="train"), "TEST": split},
        "TRAINER": {"TYPE": "interactron", "BATCH_SIZE": 2, "NUM_WORKERS": 0, "MAX_EPOCHS": 2, "SAVE_WINDOW": 1,
                    "DETECTOR_LR": 1e-5, "SUPERVISOR_LR": 1e-4, "GRAD_NORM_CLIP": 1.0, "LR_DECAY": 1, "WARMUP_TOKENS": 10,
                    "FINAL_TOKENS": 100, "OUTPUT_DIRECTORY": str(tmp_path / "train")},
        "EVALUATOR": {"TYPE": "random_policy_evaluator", "BATCH_SIZE": 1, "NUM_WORKERS": 0,
                      "OUTPUT_DIRECTORY": str(tmp_path / "eval"), "CHECKPOINT": ""},
    })
    manual_seed(1)
    model = build_model(cfg.MODEL)
    before = {k: v.detach().clone() for k, v in model.state_dict().items()}
    evaluator = build_evaluator(model, cfg)
    trainer = build_trainer(model, cfg, evaluator=evaluator)
    trainer.train()
    # checkpoint in the reference's format, same keys, weights moved
    ckpt = torch.load(trainer.checkpoint_path, map_location="cpu")
    assert set(ckpt["model"]) == set(before)
    moved = [k for k in before if before[k].is_floating_point() and not torch.equal(before[k].cpu(), ckpt["model"][k].cpu())]
    assert any(k.startswith("fusion.") for k in moved) and any(k.startswith("detector.") for k in moved)
    frozen = "detector.backbone.0.body.layer1.0.conv1.weight"
    assert torch.equal(before[frozen].cpu(), ckpt["model"][frozen].cpu())
    logs = [json.loads(l) for l in open(os.path.join(trainer.out_dir, "logs", "scalars.jsonl"))]
    assert len(logs) == 2 and "Train/loss_supervisor_ce" in logs[1] and "Test/mAP_50" in logs[0]
    assert all(np.isfinite(v) for v in logs[1].values())
    # evaluate.py path: load the checkpoint, interactive policy rollout, results.json
    cfg.EVALUATOR.TYPE, cfg.EVALUATOR.CHECKPOINT = "interactive_evaluator", trainer.checkpoint_path
    ev = build_evaluator(build_model(cfg.MODEL), cfg, load_checkpoint=True)
    summary = ev.evaluate(save_results=True)
    assert set(summary) == {"AP_50", "AP_75", "AP", "AP_small", "AP_medium", "AP_large"}
    res = json.load(open(ev.out_dir + "results.json"))
    assert {d["type"] for d in res["detections"]} <= {"tp", "fp", "fn"} and len(res["detections"]) >= 3
    # non-interactive evaluation with several episodes per predict() call: every ground-truth box is still scored once
    cfg.EVALUATOR.TYPE = "random_policy_evaluator"
    counts = []
    for bs in (1, 3):
        cfg.EVALUATOR.BATCH_SIZE = bs
        _, _, tp, fp, fn = build_evaluator(build_model(cfg.MODEL), cfg, load_checkpoint=True).evaluate()
        counts.append(tp + fn)
    assert counts[0] == counts[1] and counts[0] > 0
