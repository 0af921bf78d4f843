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


# ---- fixture G18: the reference's evaluators end to end (tests/golden/make_golden_evalrun.py) ---------------------------
def _g18_model(G, model_type="interactron"):
    from interactron_amd import Config, build_model
    from interactron_amd.synthetic import evalrun_weight_edit, load_procedural
    model = build_model(Config(**dict(MODEL, TYPE=model_type)))
    load_procedural(model.fusion, "fusion.")      # (build_model's "procedural" covers the detector; as in the parity tests)
    sd = model.state_dict()
    sd = {k: v.clone() for k, v in sd.items()}
    # the conv weights of this package live as [out, kh, kw, in]; state_dict() hands out the reference layout, so the
    # recipe -- closed-form edit + the stored overrides -- applies name for name
    evalrun_weight_edit(sd, G["overrides"])
    model.load_state_dict(sd)
    return model.cuda()


def _g18_cfg(G, tmp_path, kind):
    from interactron_amd import Config
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "data")
    split = {"TYPE": "sequence", "MODE": "test", "IMAGE_ROOT": os.path.join(root, "imgs") + "/",
             "ANNOTATION_ROOT": os.path.join(root, G["annotations"])}
    return root, Config(**{"MODEL": MODEL, "DATASET": {"TEST": split},
                           "EVALUATOR": {"TYPE": kind, "BATCH_SIZE": 1, "NUM_WORKERS": 0, "OUTPUT_DIRECTORY": str(tmp_path),
                                         "CHECKPOINT": ""}})


@pytest.mark.parametrize("kind,model_type", [("interactive_evaluator", "interactron"), ("random_policy_evaluator", "interactron"),
                                             ("random_policy_evaluator", "interactron_random")])
def test_g18_evaluators_end_to_end_against_the_reference(kind, model_type, golden, tmp_path):
    """The only available stand-in for north_star's "AP within +-0.002": the imported reference's evaluators
    (engine/interactive_evaluator.py:35-262, engine/random_policy_evaluator.py:37-211) were run on tests/golden/data with the
    fixture's weight recipe -- `interactron` through both, `interactron_random` (config 3) through the fixed rollout; the HIP
    evaluators must choose the same moves, produce the same records (kind, category, image exactly; IoU and score to 1e-3, box
    corners to 2e-3) in the same order, the same counts and the six AP numbers to 0.002.  Since round 5 the ground truth of the
    root states is fitted to what the calibrated detector finds (make_golden_evalrun.py:fit_annotations): AP_50 = 0.39 / 0.39 /
    0.33, AP = 0.24 / 0.24 / 0.13, AP_75 / AP_medium / AP_large non-zero, 54 % / 54 % / 38 % of the records hits, every hit's IoU
    >= 5e-3 away from the sweep's thresholds -- numbers on which +-0.002 discriminates (round 4: AP_50 = 0.006)."""
    from interactron_amd import build_evaluator
    G = golden("golden_evalrun.pt")
    want = G[kind.replace("_evaluator", "") + ("_interactron_random" if model_type == "interactron_random" else "")]
    assert want["six"]["AP_50"] >= 0.3 and want["six"]["AP"] >= 0.1 and min(want["six"][k] for k in ("AP_75", "AP_medium", "AP_large")) > 0
    root, cfg = _g18_cfg(G, tmp_path, kind)
    model = _g18_model(G, model_type)
    moves = []
    if kind == "interactive_evaluator":
        orig = model.get_next_action
        model.get_next_action = lambda data: (moves.append(int(orig(data))), moves[-1])[1]
    ev = build_evaluator(model, cfg)
    summary = ev.evaluate(save_results=True)
    got = json.load(open(ev.out_dir + "results.json"))["detections"]
    assert moves == want["actions"], (moves, want["actions"])
    assert len(got) == len(want["detections"]), (len(got), len(want["detections"]))
    for i, (g, w) in enumerate(zip(got, want["detections"])):
        assert (g["type"], g["pred_cat"], g["category_match"]) == (w["type"], w["pred_cat"], w["category_match"]), (i, g, w)
        assert os.path.relpath(g["img"], root) == w["img"], (i, g["img"], w["img"])
        assert abs(g["iou"] - w["iou"]) <= 1e-3 and abs(g["pred_score"] - w["pred_score"]) <= 1e-3, (i, g, w)
        assert max(abs(a - b) for a, b in zip(g["box"], w["box"])) <= 2e-3 and abs(g["area"] - w["area"]) <= 2e-3, (i, g, w)
    for k, v in want["six"].items():
        assert abs(float(summary[k]) - v) <= 2e-3, (k, float(summary[k]), v)
    ap50, ap, tp, fp, fn = ev.evaluate(save_results=False)
    assert (tp, fp, fn) == tuple(want["returned"][2:]) and abs(ap50 - want["returned"][0]) <= 2e-3 and abs(ap - want["returned"][1]) <= 2e-3
