"""Generate the golden fixtures by importing the reference (build container only).

Usage:  python tests/golden/make_golden.py [--ref /root/reference] [--out tests/golden]

The reference (allenai/interactron) is pure Python; it is imported from
``--ref`` behind ``_torchvision_stub`` (torchvision is not installed), its
models are loaded with the RNG-free procedural weights of
``interactron_amd.synthetic`` and run on the synthetic episodes of SURVEY.md 8d.
Only inputs' *recipes* (tags, shapes) and outputs are stored -- never reference
source.  The fixtures are what pins both the CPU oracle (tests/test_oracle_golden.py)
and, on the GPU box, the HIP path (tests/test_parity_gpu.py).

Large tensors are stored as (shape, L2 norm, strided sample); small ones in full.
"""
import argparse
import os
import random
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)

from interactron_amd.synthetic import hash_normal, hash_randint, hash_uniform, procedural_tensor, synthetic_episodes  # noqa: E402

MODEL_CFG = dict(
    WEIGHTS="__procedural__", NUM_CLASSES=1235, BACKBONE="resnet50", SET_COST_CLASS=1.0, SET_COST_BBOX=5.0,
    SET_COST_GIOU=2.0, TEST_RESOLUTION=300, PREDICT_ACTIONS=True, NUM_LAYERS=4, NUM_HEADS=8, EMBEDDING_DIM=512,
    BLOCK_SIZE=2060, IMG_FEATURE_SIZE=256, OUTPUT_SIZE=512, BOX_EMB_SIZE=256, EMBEDDING_PDROP=0.1,
    RESIDUAL_PDROP=0.1, ATTENTION_PDROP=0.1, ADAPTIVE_LR=1e-3)


def summarize(t, full_limit=70000, samples=256):
    """Compact, comparison-friendly record of a tensor."""
    if t is None:
        return None
    t = t.detach().cpu()
    rec = {"shape": tuple(t.shape), "dtype": str(t.dtype)}
    if t.numel() <= full_limit:
        rec["full"] = t.clone()
        return rec
    flat = t.reshape(-1).double()
    rec["norm"] = float(flat.norm())
    rec["sum"] = float(flat.sum())
    idx = torch.linspace(0, flat.numel() - 1, samples).long()
    rec["idx"] = idx
    rec["sample"] = t.reshape(-1)[idx].clone()
    return rec


def grad_record(named, samples=256):
    """Per gradient tensor: None flag, shape, L2 norm, the first 8 values and `samples` values on an even stride over
    the whole tensor (a slice routed to the wrong place beyond element 8 changes the strided sample even when it
    preserves the norm)."""
    out = {}
    for k, g in named.items():
        if g is None:
            out[k] = None
            continue
        flat = g.detach().reshape(-1)
        idx = torch.linspace(0, flat.numel() - 1, min(samples, flat.numel())).long()
        out[k] = {"shape": tuple(g.shape), "norm": float(g.double().norm()), "head": flat[:8].clone(),
                  "idx": idx, "sample": flat[idx].clone()}
    return out


def install_reference(ref):
    import _torchvision_stub
    _torchvision_stub.install()
    np.float = float  # reference gpt.py:245 / new_transformer.py:118 use the removed alias
    sys.path.insert(0, ref)
    orig_load = torch.load

    def fake_load(path, *a, **k):
        if path == "__procedural__":
            return {"model": None}
        return orig_load(path, *a, **k)

    torch.load = fake_load
    # detector.load_state_dict(None) would fail: patch after construction instead
    from torch import nn
    orig_lsd = nn.Module.load_state_dict

    def lsd(self, sd, *a, **k):
        if sd is None:
            return None
        return orig_lsd(self, sd, *a, **k)

    nn.Module.load_state_dict = lsd


def load_procedural(module, prefix):
    sd = module.state_dict()
    new = {}
    for k, v in sd.items():
        t = procedural_tensor(prefix + k, tuple(v.shape))
        new[k] = v if t is None else t.to(v.dtype)
    module.load_state_dict(new)


def build(model_type, cfg_cls):
    from utils.config_utils import build_model
    cfg = cfg_cls(**dict(MODEL_CFG, TYPE=model_type))
    model = build_model(cfg)
    if model_type == "detr":
        load_procedural(model.model, "detector.")
    else:
        load_procedural(model.detector, "detector.")
        load_procedural(model.fusion, "fusion.")
    return model, cfg


def to_sd(module, strip=""):
    return {k[len(strip):] if k.startswith(strip) else k: v.detach().clone() for k, v in module.state_dict().items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=HERE)
    ap.add_argument("--skip-train", action="store_true")
    args = ap.parse_args()
    install_reference(args.ref)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    from utils.config_utils import Config
    from models.detr_models.util import box_ops
    from models.detr_models.util.misc import NestedTensor
    from models.detr_models.matcher import HungarianMatcher
    from models.detr_models.detr import SetCriterion
    from models.detr_models.position_encoding import PositionEmbeddingSine
    from utils.meta_utils import get_parameters, sgd_step
    from utils.storage_utils import PathStorage

    G = {}

    # ---------------- G1: box ops --------------------------------------------------------
    pb = torch.from_numpy(np.concatenate([hash_uniform("g1/c", 100, 0.1, 0.9).reshape(50, 2),
                                          hash_uniform("g1/wh", 100, 0.02, 0.6).reshape(50, 2)], 1).astype(np.float32))
    tb = torch.from_numpy(np.concatenate([hash_uniform("g1/tc", 14, 0.3, 0.7).reshape(7, 2),
                                          hash_uniform("g1/twh", 14, 0.05, 0.35).reshape(7, 2)], 1).astype(np.float32))
    G["g1"] = {"pred": pb, "tgt": tb, "xyxy": box_ops.box_cxcywh_to_xyxy(pb),
               "giou": box_ops.generalized_box_iou(box_ops.box_cxcywh_to_xyxy(pb), box_ops.box_cxcywh_to_xyxy(tb))}

    # ---------------- G2 / G3: matcher and criterion ------------------------------------
    logits = torch.from_numpy((hash_normal("g2/logits", 5 * 50 * 1236) * 2.0).astype(np.float32)).reshape(5, 50, 1236)
    boxes = torch.from_numpy(np.concatenate([hash_uniform("g2/c", 500, 0.2, 0.8).reshape(250, 2),
                                             hash_uniform("g2/wh", 500, 0.05, 0.5).reshape(250, 2)], 1)
                             .astype(np.float32)).reshape(5, 50, 4)
    sizes = [0, 3, 4, 7, 5]
    targets = []
    for i, n in enumerate(sizes):
        lab = torch.from_numpy(hash_randint("g2/lab%d" % i, n, 1, 1234))
        bx = torch.from_numpy(np.concatenate([hash_uniform("g2/tc%d" % i, 2 * n, 0.3, 0.7).reshape(n, 2),
                                              hash_uniform("g2/twh%d" % i, 2 * n, 0.05, 0.35).reshape(n, 2)], 1)
                              .astype(np.float32)).reshape(n, 4)
        if n >= 4:  # repeated GT -> exact ties in the cost matrix
            bx[1] = bx[0]
            lab[1] = lab[0]
        targets.append({"labels": lab, "boxes": bx})
    matcher = HungarianMatcher(1.0, 5.0, 2.0)
    indices = matcher({"pred_logits": logits, "pred_boxes": boxes}, targets)
    crit = SetCriterion(1235, matcher, {"loss_ce": 1, "loss_bbox": 5, "loss_giou": 2}, 0.1,
                        ["labels", "boxes", "cardinality"])
    lg, bxg = logits.clone().requires_grad_(True), boxes.clone().requires_grad_(True)
    losses = crit({"pred_logits": lg, "pred_boxes": bxg}, targets, background_c=0.1)
    (losses["loss_ce"] + 5 * losses["loss_giou"] + 2 * losses["loss_bbox"]).backward()
    G["g2"] = {"sizes": sizes, "indices": indices,
               "targets": targets}
    G["g3"] = {"losses": {k: v.detach() for k, v in losses.items()}, "grad_logits": summarize(lg.grad),
               "grad_boxes": bxg.grad.clone()}

    # ---------------- G4: sine position embedding ---------------------------------------
    pe = PositionEmbeddingSine(128, normalize=True)
    g4 = {}
    for hw in (19, 50):
        m = torch.zeros(1, hw, hw, dtype=torch.bool)
        g4["zero_%d" % hw] = summarize(pe(NestedTensor(torch.zeros(1, 1, hw, hw), m)))
    m = torch.zeros(2, 19, 19, dtype=torch.bool)
    m[0, 15:, :] = True
    m[1, :, 12:] = True
    g4["padded_19"] = summarize(pe(NestedTensor(torch.zeros(2, 1, 19, 19), m)))
    G["g4"] = g4

    # ---------------- G10: clipped SGD ---------------------------------------------------
    p = [torch.from_numpy(hash_normal("g10/p%d" % i, n).astype(np.float32)) for i, n in enumerate((1000, 37, 4096))]
    g = [torch.from_numpy((hash_normal("g10/g%d" % i, n) * 12.0).astype(np.float32)) for i, n in enumerate((1000, 37, 4096))]
    g[1] = None
    G["g10"] = {"out": [t.clone() for t in sgd_step(p, g, 1e-3)]}

    # ---------------- G14: PathStorage ----------------------------------------------------
    ps = PathStorage()
    script = [([0, 1, 2, 3], 3.0), ([0, 1, 3, 3], 2.0), ([1, 1, 2, 3], 2.5), ([0, 2, 2, 0], 1.0), ([0, 1, 2, 3], 0.5)]
    labels = []
    for path, rew in script:
        pt = torch.tensor(path)
        ps.add_path(pt, rew)
        labels.append(ps.get_label(pt))
    G["g14"] = {"script": script, "labels": labels}

    torch.save(G, os.path.join(args.out, "golden_small.pt"))
    print("wrote golden_small.pt", {k: None for k in G})

    # ---------------- model-level goldens (G5-G9, G11-G13, G16) -------------------------
    M = {}
    data1 = synthetic_episodes(1, tag="golden")
    model, cfg = build("interactron", Config)
    model.eval()
    det = model.detector
    M["theta_names"] = None
    named = dict(det.named_parameters())
    inv = {id(v): k for k, v in named.items()}
    M["theta_names"] = [inv[id(t)] for t in get_parameters(det)]
    M["detector_state_keys"] = list(det.state_dict().keys())
    M["fusion_state_keys"] = {k: tuple(v.shape) for k, v in model.fusion.state_dict().items()}
    M["detector_trainable"] = [k for k, v in named.items() if v.requires_grad]

    frames = data1["frames"][0]
    masks = data1["masks"][0]
    with torch.no_grad():
        # G5: one strided + one dilated bottleneck
        body = det.backbone[0].body
        x = torch.from_numpy(hash_normal("g5/x", 2 * 256 * 20 * 20).astype(np.float32)).reshape(2, 256, 20, 20).abs()
        M["g5_layer2_0"] = summarize(body.layer2[0](x))
        x4 = torch.from_numpy(hash_normal("g5/x4", 2 * 1024 * 10 * 10).astype(np.float32)).reshape(2, 1024, 10, 10).abs()
        M["g5_layer4_0"] = summarize(body.layer4[0](x4))
        M["g5_layer4_1"] = summarize(body.layer4[1](body.layer4[0](x4)))
        # G6: one encoder and one decoder layer
        src = torch.from_numpy(hash_normal("g6/src", 30 * 2 * 256).astype(np.float32)).reshape(30, 2, 256)
        pos = torch.from_numpy(hash_normal("g6/pos", 30 * 2 * 256).astype(np.float32)).reshape(30, 2, 256)
        kpm = torch.zeros(2, 30, dtype=torch.bool)
        kpm[1, 25:] = True
        tr = det.transformer
        enc = tr.encoder.layers[0](src, src_key_padding_mask=kpm, pos=pos)
        M["g6_enc"] = summarize(enc)
        tgt = torch.from_numpy(hash_normal("g6/tgt", 7 * 2 * 256).astype(np.float32)).reshape(7, 2, 256)
        qp = torch.from_numpy(hash_normal("g6/qp", 7 * 2 * 256).astype(np.float32)).reshape(7, 2, 256)
        M["g6_dec"] = summarize(tr.decoder.layers[0](tgt, enc, memory_key_padding_mask=kpm, pos=pos, query_pos=qp))
        # G7: full detector forward
        out = det(NestedTensor(frames, masks))
        M["g7"] = {k: summarize(v) for k, v in out.items()}
        # G8: fusion forward on the detector outputs
        pre = {k: (v.unsqueeze(0) if k != "image_features" else v) for k, v in out.items()}
        fo = model.fusion(pre)
        M["g8"] = {k: summarize(v) for k, v in fo.items()}
    # G9: grad of the learned loss w.r.t. theta
    theta = get_parameters(det)
    out = det(NestedTensor(frames, masks))
    pre = {k: (v.unsqueeze(0) if k != "image_features" else v) for k, v in out.items()}
    learned = torch.norm(model.fusion(pre)["loss"])
    g = torch.autograd.grad(learned, theta, allow_unused=True)
    M["g9"] = {"learned_loss": float(learned), "grads": grad_record(dict(zip(M["theta_names"], g)))}
    # G11 / G12: predict and get_next_action
    pred = model.predict(data1)
    M["g11"] = {k: summarize(v) for k, v in pred.items()}
    g12, g12_logits = [], []
    hook = model.fusion.register_forward_hook(lambda mod, inp, out: g12_logits.append(out["actions"].detach().clone()))
    for s in range(1, 5):
        d = {"frames": data1["frames"][:, :s], "masks": data1["masks"][:, :s],
             "category_ids": [data1["category_ids"][0][:s]], "boxes": [data1["boxes"][0][:s]]}
        g12.append(model.get_next_action(d))
    hook.remove()
    M["g12"] = g12
    M["g12_logits"] = g12_logits   # the [4, 4] policy logits behind each int (row s-1 is the one argmax'ed)
    torch.save(M, os.path.join(args.out, "golden_model.pt"))
    print("wrote golden_model.pt")
    if args.skip_train:
        return

    # ---------------- G13 / G16: meta-train step (dropout off), then clip + Adam x2 -----
    T = {}
    data2 = synthetic_episodes(2, tag="golden")
    # both episodes share the root image so the second one sees the first one's PathStorage entry
    data2["initial_image_path"] = ["golden/ep0", "golden/ep0"]
    model.eval()
    model.zero_grad()
    random.seed(7)
    preds, losses = model(data2)
    T["g13"] = {
        "preds": {k: summarize(v) for k, v in preds.items()},
        "losses": {k: v.detach().clone() for k, v in losses.items()},
        "detector_grads": grad_record({k: v.grad for k, v in model.detector.named_parameters()}),
        "fusion_grads": grad_record({k: v.grad for k, v in model.fusion.named_parameters()}),
        "ridx_seed": 7,
        "path_labels": {k: v.get_label(data2["actions"][0][:4]) for k, v in model.path_storage.items()},
    }
    before = {k: v.detach().clone() for k, v in model.named_parameters()}
    opt_d = torch.optim.Adam(model.detector.parameters(), lr=1e-5)
    opt_f = torch.optim.Adam(model.fusion.parameters(), lr=1e-4)
    total_norm = torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
    opt_d.step()
    opt_f.step()
    T["g16"] = {"total_norm": float(total_norm),
                "delta": {k: {"norm": float((v.detach() - before[k]).double().norm()),
                              "head": (v.detach() - before[k]).reshape(-1)[:8].clone()}
                          for k, v in model.named_parameters()}}
    torch.save(T, os.path.join(args.out, "golden_train.pt"))
    print("wrote golden_train.pt")

    # ---------------- configs 1-3 ----------------------------------------------------------
    O = {}
    m1, _ = build("detr", Config)
    m1.eval()
    with torch.no_grad():
        O["detr_predict"] = {k: summarize(v) for k, v in m1.predict(data1).items()}
    m1.zero_grad()
    p1, l1 = m1(data1)
    O["detr_forward"] = {"losses": {k: v.detach().clone() for k, v in l1.items()},
                         "grads": grad_record({k: v.grad for k, v in m1.model.named_parameters()})}
    m2, _ = build("detr_multiframe", Config)
    m2.eval()
    with torch.no_grad():
        O["multiframe_predict"] = {k: summarize(v) for k, v in m2.predict(data1).items()}
    m2.zero_grad()
    p2, l2 = m2(data1)
    O["multiframe_forward"] = {"preds": {k: summarize(v) for k, v in p2.items()},
                               "losses": {k: v.detach().clone() for k, v in l2.items()},
                               "detector_grads": grad_record({k: v.grad for k, v in m2.detector.named_parameters()}),
                               "fusion_grads": grad_record({k: v.grad for k, v in m2.fusion.named_parameters()})}
    m3, _ = build("interactron_random", Config)
    m3.eval()
    O["random_fusion_state_keys"] = {k: tuple(v.shape) for k, v in m3.fusion.state_dict().items()}
    O["random_predict"] = {k: summarize(v) for k, v in m3.predict(data1).items()}
    m3.zero_grad()
    random.seed(7)
    p3, l3 = m3(data1)
    O["random_forward"] = {"preds": {k: summarize(v) for k, v in p3.items()},
                           "losses": {k: v.detach().clone() for k, v in l3.items()},
                           "detector_grads": grad_record({k: v.grad for k, v in m3.detector.named_parameters()}),
                           "fusion_grads": grad_record({k: v.grad for k, v in m3.fusion.named_parameters()})}
    torch.save(O, os.path.join(args.out, "golden_configs.pt"))
    print("wrote golden_configs.pt")


if __name__ == "__main__":
    main()
