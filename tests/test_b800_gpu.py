"""The north-star shape (BASELINE.json: 5 x 3x800x800 episodes -> 50x50 = 2500 tokens per frame, fusion T = 12 755): the HIP
path against the CPU oracle at sizes the oracle finishes in seconds (one 800x800 frame through the detector; the fusion on
two frames' worth of tokens, T = 5105), and size-independent properties of one FULL 5-frame 800x800 meta-train step.
``pytest -m gpu``."""
import random

import pytest
import torch

from interactron_amd.synthetic import load_procedural, procedural_state_dict, synthetic_episodes
from tests.test_parity_gpu import MODEL_CFG, to_gpu

pytestmark = pytest.mark.gpu

CFG800 = dict(MODEL_CFG, BLOCK_SIZE=5 * (2500 + 50) + 5)


def _close(a, b, what):
    tol = 1e-3 * float(b.abs().max()) + 1e-4    # the forward tolerance of tests/test_parity_gpu.py
    torch.testing.assert_close(a.detach().cpu(), b, atol=tol, rtol=1e-3, msg=lambda m: what + ": " + m)


def test_detector_800x800_frame_against_oracle():
    from interactron_amd import Config, NestedTensor, build_model
    from oracle import detector as od
    m = build_model(Config(**dict(MODEL_CFG, TYPE="detr"))).cuda().eval()
    data = synthetic_episodes(1, frames=1, height=800, width=800, tag="b800-det")
    frames, masks = data["frames"][0], data["masks"][0]
    with torch.no_grad():
        out = m.model(NestedTensor(frames.cuda(), masks.cuda()))
    assert out["embedded_memory_features"].shape == (1, 256, 50, 50) and out["pred_logits"].shape == (1, 50, 1236)
    det = {k[len("detector."):]: v for k, v in
           procedural_state_dict({"detector." + k: v for k, v in od.detr_state_shapes().items()}).items()}
    with torch.no_grad():
        ref = od.detr_forward(det, frames, masks)
    for k, v in ref.items():
        _close(out[k], v, "detector@800/" + k)


def test_fusion_two_800x800_frames_against_oracle():
    """GPT fusion with the 12 755-entry position table on the tokens of two 800x800 frames (T = 2 * 2550 + 5 = 5105)."""
    from interactron_amd import Config, build_model
    from interactron_amd.synthetic import hash_normal, hash_uniform
    from oracle import fusion as of
    import numpy as np
    m = build_model(Config(**dict(CFG800, TYPE="interactron")))
    assert m.fusion.model.seq_pos_embed.shape == (1, 12755, 512)
    load_procedural(m.fusion, "fusion.")
    fusion = m.fusion.cuda().eval()
    t = lambda name, shape, scale=1.0: torch.from_numpy((hash_normal("b800/" + name, int(np.prod(shape))) * scale)
                                                         .astype(np.float32)).reshape(shape)
    x = {"embedded_memory_features": t("mem", (1, 2, 256, 50, 50)), "box_features": t("box", (1, 2, 50, 256)),
         "pred_logits": t("logits", (1, 2, 50, 1236), 2.0),
         "pred_boxes": torch.from_numpy(hash_uniform("b800/pb", 400, 0.1, 0.9).astype(np.float32)).reshape(1, 2, 50, 4)}
    with torch.no_grad():
        out = fusion({k: v.cuda() for k, v in x.items()})
    sd = {k[len("fusion."):]: v for k, v in
          procedural_state_dict({"fusion." + k: v for k, v in of.fusion_state_shapes(CFG800, "gpt").items()}).items()}
    with torch.no_grad():
        ref = of.fusion_gpt_forward(sd, x, CFG800)
    for k in ("loss", "actions", "pred_boxes", "pred_logits"):
        _close(out[k].reshape(ref[k].shape), ref[k], "fusion@T5105/" + k)


def test_full_800x800_meta_train_step_properties(golden):
    """One complete 5-frame 800x800 meta-train step (T = 12 755, encoder rows of 2500 tokens): everything finite, shapes
    and None-pattern as at 300x300, and gradients ACCUMULATE -- a second identical pass doubles every .grad (the
    reference sums over the tasks of a batch, models/interactron.py:123,134) -- the size-independent check that no part
    of a gradient is dropped or written twice at this size."""
    from interactron_amd import Config, build_model
    m = build_model(Config(**dict(CFG800, TYPE="interactron", EPISODE_CHUNK=1)))
    load_procedural(m.fusion, "fusion.")
    m = m.cuda().eval()
    assert m.fusion.model.block_size == 12755
    data = to_gpu(synthetic_episodes(1, height=800, width=800, tag="b800-step"))
    m.zero_grad()
    random.seed(5)
    preds, losses = m(data)
    assert preds["pred_logits"].shape == (1, 1, 50, 1236) and preds["pred_boxes"].shape == (1, 1, 50, 4)
    assert len(losses) >= 10 and all(bool(torch.isfinite(v).all()) for v in losses.values())
    first = {k: (None if p.grad is None else p.grad.detach().clone()) for k, p in m.named_parameters()}
    trainable = [k for k, p in m.named_parameters() if p.requires_grad]
    # None-pattern as at 300x300 (fixture G13): frozen tensors and the never-used fusion.model.pos_emb stay None
    T = golden("golden_train.pt")["g13"]
    ref_none = {"detector." + k: v is None for k, v in T["detector_grads"].items()}
    ref_none.update({"fusion." + k: v is None for k, v in T["fusion_grads"].items()})
    for k in first:
        if k != "fusion.model.seq_pos_embed":   # (same tensor, longer table)
            assert (first[k] is None) == ref_none[k], k
    have = [k for k in trainable if first[k] is not None]
    assert all(bool(torch.isfinite(first[k]).all()) for k in have)
    assert sum(float(first[k].double().norm()) > 0 for k in have) > 0.9 * len(have)
    random.seed(5)
    m(data)
    tot1 = tot2 = worst = 0.0
    for k in have:
        n1 = float(first[k].double().norm())
        n2 = float(m.get_parameter(k).grad.double().norm())
        tot1, tot2 = tot1 + n1 * n1, tot2 + n2 * n2
        if n1 < 1e-6:
            continue
        # the second pass computes bit-identical contributions (ordered reductions), so .grad doubles up to the rounding of
        # the accumulation itself; a part dropped or written twice is off by 50 % or more
        worst = max(worst, abs(n2 - 2 * n1) / (2 * n1))
        assert abs(n2 - 2 * n1) <= ACC_TENSOR * 2 * n1, (k, n1, n2)
    print("800x800 accumulate twice: worst tensor %.2e, whole %.2e" % (worst, abs(tot2 ** 0.5 - 2 * tot1 ** 0.5) / (2 * tot1 ** 0.5)))
    assert abs(tot2 ** 0.5 - 2 * tot1 ** 0.5) <= ACC_WHOLE * 2 * tot1 ** 0.5, (tot1 ** 0.5, tot2 ** 0.5)


ACC_TENSOR, ACC_WHOLE = 1e-6, 1e-7   # measured (r3, deterministic reductions): 4.8e-9, 2.2e-11 (round 2, atomics: 0.1, 1e-2)
FP8_LOGIT_MAX, FP8_LOGIT_L2, FP8_BOX_ABS = 0.08, 0.05, 0.03   # measured: 0.038, 0.030, 0.012


def test_stress_config_200_queries_fp32_and_fp8_attention():
    """BASELINE.json configs[4] at reduced frames: NUM_QUERIES = 200 (reference hard-codes 50, detr.py:331) on 2 frames of
    320x256 through detector + GPT fusion, HIP vs CPU oracle -- once on the parity path (fp32-grade attention, the usual
    forward tolerance) and once with the opt-in fp8 forward attention (IX_ATTENTION_DTYPE=fp8 / hipops.ATTENTION_DTYPE).
    Stated fp8 tolerance on this model (FP8_* below): logits within 8 % of max|logit| element-wise and 5 % relative L2,
    boxes (sigmoid outputs in [0, 1]) within 0.03 absolute."""
    from interactron_amd import Config, NestedTensor, build_model, hipops
    from oracle import detector as od, fusion as of
    Q, s, h, w = 200, 2, 320, 256
    tokens = (h // 16) * (w // 16)
    cfg = dict(MODEL_CFG, TYPE="interactron", NUM_QUERIES=Q, BLOCK_SIZE=5 * (tokens + Q) + 5)
    m = build_model(Config(**cfg))
    load_procedural(m.fusion, "fusion.")
    m = m.cuda().eval()
    data = synthetic_episodes(1, frames=s, height=h, width=w, tag="stress-q200")
    frames, masks = data["frames"][0], data["masks"][0]
    det = {k[len("detector."):]: v for k, v in
           procedural_state_dict({"detector." + k: v for k, v in od.detr_state_shapes(num_queries=Q).items()}).items()}
    fus = {k[len("fusion."):]: v for k, v in
           procedural_state_dict({"fusion." + k: v for k, v in of.fusion_state_shapes(cfg, "gpt").items()}).items()}
    with torch.no_grad():
        rd = od.detr_forward(det, frames, masks)
        rf = of.fusion_gpt_forward(fus, {k: v.unsqueeze(0) for k, v in rd.items() if k != "image_features"}, cfg)

    def run():
        with torch.no_grad():
            d = m.detector(NestedTensor(frames.cuda(), masks.cuda()))
            f = m.fusion({k: v.unsqueeze(0) for k, v in d.items() if k != "image_features"})
        return d, f

    assert hipops.ATTENTION_DTYPE == "fp32"
    d, f = run()
    assert d["pred_logits"].shape == (s, Q, 1236) and f["pred_logits"].shape[-2:] == (Q, 1236)
    for k in ("pred_logits", "pred_boxes", "box_features"):
        _close(d[k], rd[k], "q200/detector/" + k)
    for k in ("pred_logits", "pred_boxes", "loss", "actions"):
        _close(f[k].reshape(rf[k].shape), rf[k], "q200/fusion/" + k)
    hipops.ATTENTION_DTYPE = "fp8"
    try:
        d8, f8 = run()
    finally:
        hipops.ATTENTION_DTYPE = "fp32"
    for name, got, ref in (("detector", d8, rd), ("fusion", f8, rf)):
        lg, rl = got["pred_logits"].cpu().reshape(ref["pred_logits"].shape), ref["pred_logits"]
        bx, rb = got["pred_boxes"].cpu().reshape(ref["pred_boxes"].shape), ref["pred_boxes"]
        worst, l2, box = float((lg - rl).abs().max() / rl.abs().max()), float((lg - rl).norm() / rl.norm()), float((bx - rb).abs().max())
        print("fp8 attention, %s: logits max err / max|logit| = %.4f, relative L2 = %.4f; boxes max abs err = %.4f" % (name, worst, l2, box))
        assert worst <= FP8_LOGIT_MAX and l2 <= FP8_LOGIT_L2 and box <= FP8_BOX_ABS, (name, worst, l2, box)
    assert float((d8["pred_logits"] - d["pred_logits"]).abs().max()) > 0   # (the fp8 path really ran)


def test_stress_config_training_step_under_the_fp8_switch_against_the_oracle():
    """BASELINE.json configs[4] as a TRAINING step at a size the oracle finishes in seconds: one 5-frame episode of 256 x 256
    frames, NUM_QUERIES = 200 (fusion T = 5 (256 + 200) + 5 = 2285), the whole meta-train step of models/interactron.py:61-151
    against the float32 CPU oracle -- every loss, every gradient tensor's norm and direction (cosine >= 0.999) -- and then once more
    with ``hipops.ATTENTION_DTYPE = "fp8"``: a training step differentiates every attention call, differentiated calls keep the
    fp32-grade forward (hipops/attn.py flash_forward), so the step under the switch is the SAME step, loss for loss and gradient norm
    for gradient norm.  History (round 6, profiles/r6k_16_bit_step_survey.txt): with the e4m3 forward products left on in
    differentiated calls this test measured whole-gradient cosine 0.175 against the oracle and backbone gradient norms 30-40 x too
    large, at losses within 3.5 % -- the forward was fine and the derivative passes were fine, their combination was not
    (tests/test_ops_gpu.py::test_fp8_attention_is_for_calls_that_are_not_differentiated has the arithmetic).  fp8 is a predict-time
    option (test_stress_config_200_queries_fp32_and_fp8_attention, test_stress_config_full_size_predict_properties)."""
    import __graft_entry__ as entry
    from interactron_amd import hipops
    extra = dict(NUM_QUERIES=200, BLOCK_SIZE=5 * (16 * 16 + 200) + 5)
    # (200 near-identical queries on RNG-free weights: the Hungarian optimum is full of ties -- the runs take the oracle's assignments)
    # (1e-2 on the norms without the float64 slack run -- the first trainable convolution sits at 7e-3 here, as in the 128 x 128 smoke
    #  step before its slack is counted; losses at 5 %: with 200 near-identical queries an image presented to several criterion calls
    #  may take another call's recorded assignment -- a tie for the matcher, not for the loss of THAT call)
    ref = entry.smoke_check(256, cfg_extra=extra, f64_slack=False, norm_tol=1e-2, loss_tol=5e-2, cos_min=0.999, pin_matching="ties")
    assert ref["whole_gradient_cosine"] >= 0.9999, ref["whole_gradient_cosine"]
    assert hipops.ATTENTION_DTYPE == "fp32"
    hipops.ATTENTION_DTYPE = "fp8"
    try:
        got = entry.smoke_check(256, cfg_extra=extra, f64_slack=False, norm_tol=1e-2, loss_tol=5e-2, cos_min=0.999, pin_matching="ties")
    finally:
        hipops.ATTENTION_DTYPE = "fp32"
    assert got["checked"] == ref["checked"] >= 300
    for part in ("losses", "norms"):   # (1e-5: the step is the same sequence of launches)
        for k, v in ref[part].items():
            assert abs(got[part][k] - v) <= 1e-5 * max(abs(v), 1e-6), (k, got[part][k], v, "a training step changed under ATTENTION_DTYPE = fp8")


def test_stress_config_full_size_predict_properties():
    """BASELINE.json configs[4] at FULL size: one 5-frame episode of 3x1600x1600 frames (100 x 100 = 10 000 tokens per frame),
    200 queries, fusion T = 5 (10 000 + 200) + 5 = 51 005 (reference shapes: models/gpt.py:39-57,191, detr.py:331), forward
    attention products on e4m3 MFMA -- ``predict()`` (adapt on the 5 frames through the learned loss, detect frame 0 through
    theta').  The oracle cannot run this size; checked are the size-independent properties: shapes, finiteness, box range,
    that the adaptation moved the prediction (theta' != theta) and that a second call reproduces the first bit for bit."""
    from interactron_amd import Config, build_model, hipops
    Q, size = 200, 1600
    cfg = dict(MODEL_CFG, TYPE="interactron", NUM_QUERIES=Q, BLOCK_SIZE=5 * (100 * 100 + Q) + 5, PREDICT_GRAPH=False)
    m = build_model(Config(**cfg))
    load_procedural(m.fusion, "fusion.")
    m = m.cuda().eval()
    assert m.fusion.model.block_size == 51005
    data = synthetic_episodes(1, height=size, width=size, tag="stress-full")
    ep = {"frames": data["frames"].cuda(), "masks": data["masks"].cuda()}
    torch.cuda.reset_peak_memory_stats()
    hipops.ATTENTION_DTYPE = "fp8"
    try:
        out = m.predict(ep)
        again = m.predict(ep)
        with torch.no_grad():
            from interactron_amd import NestedTensor
            plain = m.detector(NestedTensor(ep["frames"][0, :1], ep["masks"][0, :1]))
    finally:
        hipops.ATTENTION_DTYPE = "fp32"
    torch.cuda.synchronize()
    print("1600x1600 / Q = 200 / T = 51005 predict: peak memory %.1f GB" % (torch.cuda.max_memory_allocated() / 1e9))
    assert out["pred_logits"].shape == (1, 1, Q, 1236) and out["pred_boxes"].shape == (1, 1, Q, 4)
    assert out["embedded_memory_features"].shape == (1, 1, 256, 100, 100)
    for k, v in out.items():
        assert bool(torch.isfinite(v).all()), k
        assert torch.equal(v, again[k]), (k, "predict is not reproducible")
    assert float(out["pred_boxes"].min()) >= 0.0 and float(out["pred_boxes"].max()) <= 1.0
    moved = float((out["pred_logits"][0] - plain["pred_logits"]).abs().max())
    assert 0.0 < moved < 1e3, moved   # one clipped inner SGD step (|delta theta| <= 0.01 per element) changes the prediction
