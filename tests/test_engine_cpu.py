"""Host-side "next" rows: AP metric and matching against fixtures captured from the reference's own
functions (G15, tests/golden/make_golden_eval.py), the dataset walk / transforms on a tiny synthetic dataset on disk,
the scalar logger and the factories.  No GPU needed."""
import json
import os
import random

import numpy as np
import pytest
import torch
from PIL import Image

from interactron_amd import Config
from interactron_amd.constants import ACTIONS
from interactron_amd.datasets import InteractiveDataset, SequenceDataset, train_transform, transform
from interactron_amd.engine import metrics
from interactron_amd.storage import collate_fn

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_g15_ap_metric_matches_reference(golden):
    G = golden("golden_eval.pt")
    ious = list(np.arange(0.5, 1.0, 0.05))
    for rec in G["ap"]:
        d = rec["detections"]
        assert metrics.compute_ap(list(d), 100, [0.5]) == pytest.approx(rec["ap50"], abs=1e-12)
        assert metrics.compute_ap(list(d), 100, [0.75]) == pytest.approx(rec["ap75"], abs=1e-12)
        assert metrics.compute_ap(list(d), 100, ious) == pytest.approx(rec["ap"], abs=1e-12)
        assert metrics.compute_ap(list(d), 100, ious, 0.0, 32 ** 2 / 300 ** 2) == pytest.approx(rec["ap_small"], abs=1e-12)
        assert metrics.compute_ap(list(d), 100, ious, 32 ** 2 / 300 ** 2, 96 ** 2 / 300 ** 2) == \
            pytest.approx(rec["ap_medium"], abs=1e-12)
        assert metrics.compute_cat_ap(list(d), 100, ious) == pytest.approx(rec["cat_ap"], abs=1e-12)
        p, r = metrics.compute_pr(list(d), 100, 0.5)
        assert p == pytest.approx(rec["pr"][0]) and r == pytest.approx(rec["pr"][1])
        assert len(d) == len(rec["detections"])   # the caller's list is not consumed


def test_g15_prediction_matching_matches_reference(golden):
    for rec in golden("golden_eval.pt")["match"]:
        bi, bx = metrics.match_predictions_to_detections(rec["ious"].clone())
        assert torch.equal(bx, rec["best_idx"])
        torch.testing.assert_close(bi, rec["best_ious"])


def test_nms_and_box_iou():
    boxes = torch.tensor([[0, 0, 10, 10], [1, 1, 11, 11], [20, 20, 30, 30], [0, 0, 10, 10.5], [50, 50, 60, 60]], dtype=torch.float)
    scores = torch.tensor([0.9, 0.8, 0.7, 0.95, 0.1])
    keep = metrics.nms(boxes, scores, 0.5)
    assert keep.tolist() == [3, 2, 4]          # 0 and 1 overlap box 3 (IoU > 0.5)
    iou = metrics.box_iou(boxes[:2], boxes[:3])
    assert iou[0, 0] == pytest.approx(1.0) and iou[0, 2] == 0.0
    assert iou[0, 1] == pytest.approx(81.0 / 119.0)
    assert metrics.nms(boxes[:0], scores[:0], 0.5).numel() == 0


def test_frame_detections_bookkeeping():
    """One image: a matched prediction (tp), a duplicate of it (fp after NMS keeps both at IoU<=0.5? no: suppressed), a
    missed ground-truth box (fn), a THOR-class prediction with no ground truth (fp) and a non-THOR one (ignored)."""
    Q, C = 6, 1236
    logits = torch.full((Q, C), -10.0)
    logits[:, 1235] = 0.0                                   # default: background
    boxes = torch.tensor([[0.3, 0.3, 0.2, 0.2]] * Q)
    logits[0, 11] = 8.0                                     # tp for gt class 11
    logits[1, 11] = 7.0; boxes[1] = torch.tensor([0.31, 0.3, 0.2, 0.2])   # near-duplicate: removed by NMS
    logits[2, 18] = 6.0; boxes[2] = torch.tensor([0.7, 0.7, 0.1, 0.1])    # THOR class without ground truth -> fp
    logits[3, 5] = 6.0; boxes[3] = torch.tensor([0.7, 0.2, 0.1, 0.1])     # non-THOR class without ground truth -> dropped
    gt_boxes = torch.tensor([[0.3, 0.3, 0.2, 0.2], [0.8, 0.2, 0.1, 0.1]])
    gt_cats = torch.tensor([11, 22])
    dets = metrics.frame_detections(logits, boxes, gt_boxes, gt_cats, "img")
    kinds = sorted((d["type"], d["pred_cat"]) for d in dets)
    assert kinds == [("fn", 22), ("fp", 18), ("tp", 11)]
    tp = [d for d in dets if d["type"] == "tp"][0]
    assert tp["iou"] == pytest.approx(1.0, abs=1e-5) and tp["area"] == pytest.approx(0.04, rel=1e-4)


@pytest.fixture()
def tiny_dataset(tmp_path):
    """Two scenes, 3 states each, 40x30 JPEGs, in the reference annotation schema (masks keep the ORIGINAL image size,
    reference sequence_dataset.py:55-56 -- real data is 300x300, so only the GPU smoke test needs full-size images)."""
    random.seed(0)
    root = tmp_path / "imgs"
    scenes = []
    for s in range(2):
        name = "FloorPlan%d" % s
        os.makedirs(root / name)
        table = {}
        states = ["s%d_%d" % (s, k) for k in range(3)]
        for k, st in enumerate(states):
            Image.fromarray(np.random.RandomState(10 * s + k).randint(0, 255, (30, 40, 3), dtype=np.uint8)).save(root / name / (st + ".jpg"))
            table[st] = {"detections": {"obj%d" % j: {"category_id": 10 + j + k, "bbox": [4 + j, 3 + j, 10, 8]} for j in range(k + 1)},
                         "actions": {a: states[(k + 1 + i) % 3] for i, a in enumerate(ACTIONS)}}
        scenes.append({"scene_name": name, "root": states[0], "state_table": table})
    ann = tmp_path / "ann.json"
    ann.write_text(json.dumps({"data": scenes, "metadata": {"actions": ACTIONS}}))
    return str(root), str(ann)


def test_sequence_dataset_and_collate(tiny_dataset):
    root, ann = tiny_dataset
    ds = SequenceDataset(root + "/", ann, "test", transform=transform)
    assert len(ds) == 2
    s = ds[0]
    assert [ACTIONS[a] for a in s["actions"]] == ["RotateLeft", "MoveAhead", "RotateLeft", "MoveBack", "RotateRight"]
    assert len(s["frames"]) == 5 and s["frames"][0].shape[0] == 3 and max(s["frames"][0].shape[1:]) == 300
    assert s["initial_image_path"] == root + "/FloorPlan0/s0_0.jpg"
    # labels carry the +1 offset, boxes are normalised cxcywh
    assert s["category_ids"][0].tolist() == [11]
    torch.testing.assert_close(s["boxes"][0], torch.tensor([[(4 + 5) / 40, (3 + 4) / 30, 10 / 40, 8 / 30]]), atol=1e-5, rtol=1e-5)
    batch = collate_fn([ds[0], ds[1]])
    assert batch["frames"].shape[:3] == (2, 5, 3) and batch["actions"].shape == (2, 5)
    assert batch["masks"].dtype == torch.long and len(batch["boxes"][1]) == 5
    tr = SequenceDataset(root, ann, "train", transform=train_transform)[1]
    assert all(f.shape == (3, 300, 300) for f in tr["frames"])
    assert all(bool(((b >= 0) & (b <= 1)).all()) for b in tr["boxes"])


def test_interactive_dataset_reset_step(tiny_dataset):
    root, ann = tiny_dataset
    env = InteractiveDataset(root, ann, "test", transform=transform)
    d = env.reset()
    assert d["frames"].shape[:2] == (1, 1) and d["actions"].shape == (1, 0)
    d = env.step(2)
    d = env.step(0)
    assert d["frames"].shape[:2] == (1, 3) and d["actions"].tolist() == [[2, 0]]
    assert len(d["boxes"][0]) == 3 and d["initial_image_path"] == [root + "/FloorPlan0/s0_0.jpg"]
    assert env.reset()["episode_ids"] == 1 and env.reset()["episode_ids"] == 0     # wraps around


def test_logger_and_factories(tmp_path):
    from interactron_amd.engine.logging import TBLogger
    log = TBLogger(str(tmp_path / "logs"))
    log.add_value("Train/x", 1.0)
    log.add_value("Train/x", torch.tensor(3.0))
    log.log_values()
    rec = json.loads(open(tmp_path / "logs" / "scalars.jsonl").read().strip())
    assert rec == {"iter": 0, "Train/x": 2.0}
    from interactron_amd import build_evaluator, build_trainer
    cfg = Config(**{"TRAINER": {"TYPE": "nope"}, "EVALUATOR": {"TYPE": "nope"}})
    with pytest.raises(AssertionError):
        build_trainer(None, cfg)
    with pytest.raises(AssertionError):
        build_evaluator(None, cfg)


# ---- N3: the data path against fixture G17 (the reference's own datasets / transforms on tests/golden/data) ---------------
def _check_summary(rec, t, what):
    assert tuple(t.shape) == tuple(rec["shape"]) and str(t.dtype) == rec["dtype"], (what, tuple(t.shape), rec["shape"])
    flat = t.reshape(-1)
    torch.testing.assert_close(flat[rec["idx"]], rec["sample"], atol=1e-6, rtol=1e-6, msg=lambda m: what + ": " + m)
    assert abs(float(flat.double().norm()) - rec["norm"]) <= 1e-6 * max(rec["norm"], 1.0), what


def _check_sample(rec, s, root, what):
    for i, (f, r) in enumerate(zip(s["frames"], rec["frames"])):
        _check_summary(r, f, "%s/frame%d" % (what, i))
    assert [tuple(m.shape) for m in s["masks"]] == rec["masks"] and all(m.dtype == torch.long for m in s["masks"])
    assert list(s["actions"]) == rec["actions"] and [len(o) for o in s["object_ids"]] == rec["n_objects"]
    for a, b in zip(s["category_ids"], rec["category_ids"]):
        assert a.dtype == b.dtype and torch.equal(a, b), what
    for a, b in zip(s["boxes"], rec["boxes"]):
        torch.testing.assert_close(a, b, atol=1e-7, rtol=1e-6, msg=lambda m: what + " boxes: " + m)
    assert s["episode_ids"] == rec["episode_ids"] and os.path.relpath(s["initial_image_path"], root) == rec["initial_image_path"]


def test_g17_sequence_dataset_collate_and_interactive_match_reference(golden):
    """SequenceDataset (test mode + an explicit action script), collate_fn and InteractiveDataset reset / step on the
    committed tiny dataset == what the imported reference returned (tests/golden/make_golden_data.py): decoded and
    normalised pixels, label offset, cxcywh boxes (incl. a 400x300 scene that is really resized and a state without
    detections), mask shapes, action indices, episode ids and paths."""
    G = golden("golden_data.pt")
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "data")
    imgs, ann = os.path.join(root, "imgs"), os.path.join(root, "annotations.json")
    ds = SequenceDataset(imgs + "/", ann, "test", transform=transform)
    assert len(ds) == G["len"]
    for i, rec in enumerate(G["samples"]):
        _check_sample(rec, ds[i], root, "episode%d" % i)
    _check_sample(G["scripted"], ds.__getitem__(1, actions=["MoveBack", "MoveBack", "RotateRight", "MoveAhead", "RotateLeft"]),
                  root, "scripted")
    batch = collate_fn([ds[0], ds[1]])
    c = G["collate"]
    _check_summary(c["frames"], batch["frames"], "collate/frames")
    assert tuple(batch["masks"].shape) == c["masks"] and torch.equal(batch["actions"], c["actions"])
    assert torch.equal(batch["episode_ids"], c["episode_ids"])
    assert [os.path.relpath(p, root) for p in batch["initial_image_path"]] == c["initial_image_path"]
    for ep, rep in zip(batch["category_ids"], c["category_ids"]):
        assert all(torch.equal(a, b) for a, b in zip(ep, rep))
    env = InteractiveDataset(imgs, ann, "test", transform=transform)
    for trace in G["interactive"]:
        obs = [env.reset()] + [env.step(a) for a in trace["script"]]
        for k, (d, rec) in enumerate(zip(obs, trace["steps"])):
            what = "interactive/%s/%d" % (trace["script"], k)
            _check_summary(rec["frames"], d["frames"], what)
            assert torch.equal(d["actions"], rec["actions"]) and int(d["episode_ids"]) == rec["episode_ids"], what
            assert [os.path.relpath(p, root) for p in d["initial_image_path"]] == rec["initial_image_path"]
            for a, b in zip(d["boxes"][0], rec["boxes"]):
                torch.testing.assert_close(a, b, atol=1e-7, rtol=1e-6)
            assert all(torch.equal(a, b) for a, b in zip(d["category_ids"][0], rec["category_ids"]))


def test_episode_batch_loader_shards_the_decode_not_the_batch():
    """EpisodeBatchLoader under world size 2 (plain objects, no process group needed): both ranks derive the same global
    batches (permutation and per-episode action scripts from one seed), each decodes only its episodes r::2 -- together they
    are exactly the single-process batch, frame for frame -- and both carry the whole batch's roots / actions for the
    PathStorage replay.  A second epoch draws another permutation; a rank whose shard of the short last batch is empty gets
    an empty batch and the same global size."""
    from interactron_amd.datasets import EpisodeBatchLoader, SequenceDataset, transform
    from interactron_amd.storage import collate_fn
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "data")
    imgs, ann = os.path.join(root, "imgs"), os.path.join(root, "annotations.json")

    def loaders(world, mode, shuffle):   # (the deterministic eval transform: the shards must be comparable pixel for pixel)
        ds = SequenceDataset(imgs + "/", ann, mode, transform=transform)
        return [EpisodeBatchLoader(ds, 2, shuffle, rank=r, world=world, seed=7, num_workers=0, pin_memory=False, collate=collate_fn)
                for r in range(world)]

    for mode, shuffle in (("train", True), ("test", False)):
        (one,), (r0, r1) = loaders(1, mode, shuffle), loaders(2, mode, shuffle)
        for epoch in range(2):
            full, parts = list(one), [list(r0), list(r1)]
            assert len(full) == len(parts[0]) == len(parts[1]) == 2   # 3 scenes, batches of 2
            for b, (whole, n) in enumerate(full):
                for r in range(2):
                    part, n_r = parts[r][b]
                    assert n_r == n and part["dp_world"] == 2 and part["dp_index"] == list(range(r, n, 2))
                    assert part["dp_roots"] == whole["initial_image_path"] == whole["dp_roots"]
                    assert part["dp_actions"] == whole["actions"][:, :4].tolist() == whole["dp_actions"]
                    idx = list(range(r, n, 2))
                    assert part["frames"].shape[0] == len(idx)
                    if idx:
                        assert torch.equal(part["frames"], whole["frames"][idx]) and torch.equal(part["actions"], whole["actions"][idx])
                        assert part["initial_image_path"] == [whole["initial_image_path"][i] for i in idx]
                        for i, ep in zip(idx, part["boxes"]):
                            assert all(torch.equal(a, b) for a, b in zip(ep, whole["boxes"][i]))
            if epoch == 0:
                first = [w["episode_ids"].tolist() for w, _ in full]
            elif shuffle:
                assert sorted(sum(first, [])) == sorted(sum([w["episode_ids"].tolist() for w, _ in full], []))
        assert full[-1][1] == 1 and parts[1][-1][0]["frames"].shape[0] == 0   # short last batch: rank 1 idles


# ---- evaluation sharded over ranks (SURVEY 8e "Eval"): world_size 2 over gloo == one process ---------------------------------
class _StubModel(torch.nn.Module):
    """predict / get_next_action as deterministic functions of the frames (the evaluators' host logic is what is under test:
    which rank evaluates which episode, the order of the gathered records, the AP numbers -- no GPU, no kernels)"""

    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(torch.zeros(1))

    def get_next_action(self, data):
        return int(data["frames"].double().abs().sum().item() * 1000) % 4

    def predict(self, data):
        b = data["frames"].shape[0]
        logits, boxes = torch.zeros(b, 1, 12, 1236), torch.zeros(b, 1, 12, 4)
        for i in range(b):     # episode by episode: the answer must not depend on who shares the batch
            g = torch.Generator().manual_seed(int(data["frames"][i].double().abs().sum().item() * 10) % (2 ** 31))
            boxes[i] = torch.rand(1, 12, 4, generator=g) * 0.4 + 0.2
            cats = data["category_ids"][i][0]
            for q in range(12):
                c = int(cats[q % len(cats)]) if len(cats) and q % 3 else 11
                logits[i, 0, q, c] = 2.0 + 3.0 * float(torch.rand(1, generator=g))
            if len(cats):
                boxes[i, 0, 1] = data["boxes"][i][0][0] * torch.tensor([1.0, 1.0, 0.9, 1.1])
        return {"pred_logits": logits, "pred_boxes": boxes}


def _eval_cfg(tmp, kind):
    root = os.path.join(GOLDEN, "data")
    return Config(**{"EVALUATOR": {"TYPE": kind, "BATCH_SIZE": 2, "NUM_WORKERS": 0, "OUTPUT_DIRECTORY": tmp, "CHECKPOINT": ""},
                     "DATASET": {"TEST": {"TYPE": "sequence", "MODE": "test", "IMAGE_ROOT": os.path.join(root, "imgs") + "/",
                                          "ANNOTATION_ROOT": os.path.join(root, "annotations_eval.json")}}})


def _eval_worker(rank, world, port, kind, tmp, out):
    import torch.distributed as dist
    from interactron_amd import build_evaluator
    if world > 1:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
    ev = build_evaluator(_StubModel(), _eval_cfg(tmp + "/w%dr%d" % (world, rank), kind))
    seen = []
    orig = ev._episodes
    ev._episodes = lambda mine: (seen.extend(mine), orig(mine))[1]
    five = ev.evaluate(save_results=False)
    six = ev.evaluate(save_results=True)
    wrote = os.path.exists(ev.out_dir + "results.json")
    dets = json.load(open(ev.out_dir + "results.json"))["detections"] if wrote else None
    out[(world, rank)] = (tuple(float(x) for x in five), {k: float(v) for k, v in six.items()}, sorted(set(seen)), wrote, dets)
    if world > 1:
        dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["random_policy_evaluator", "interactive_evaluator"])
def test_sharded_evaluation_world_size_2_gloo_equals_single_process(kind, tmp_path):
    """Rank r evaluates test episodes r::2, the detection lists are gathered and merged back into test-set order: same
    records in the same order, same six AP numbers on both ranks as one process (reference loop:
    engine/interactive_evaluator.py:35-63, random_policy_evaluator.py:37-60); only rank 0 writes results.json."""
    import torch.multiprocessing as mp
    from tests.test_host_cpu import _free_port
    out = mp.Manager().dict()
    _eval_worker(0, 1, 0, kind, str(tmp_path), out)
    mp.spawn(_eval_worker, args=(2, _free_port(), kind, str(tmp_path), out), nprocs=2, join=True)
    single, r0, r1 = out[(1, 0)], out[(2, 0)], out[(2, 1)]
    assert single[2] == [0, 1, 2, 3, 4, 5] and r0[2] == [0, 2, 4] and r1[2] == [1, 3, 5]
    assert r0[0] == r1[0] == single[0] and r0[1] == r1[1] == single[1]
    assert single[3] and r0[3] and not r1[3]
    assert r0[4] == single[4] and len(single[4]) > 20 and {d["type"] for d in single[4]} == {"tp", "fp", "fn"}
