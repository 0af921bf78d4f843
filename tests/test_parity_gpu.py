"""Model-level parity of the HIP path (called through the reference's Python surface, computing through the C-ABI)
against (a) the golden fixtures captured from the imported reference and (b) the CPU oracle on fresh inputs.

Tolerances (fp32 kernels, f32-in MFMA = exact fmaf chains, different summation order than the CPU):
  forward tensors   atol 1e-3 * max|ref| (+1e-4)            gradients: per-tensor L2 norm within 5e-3 relative
  (G13 second-order gradients: + 3x the reference's own float32 rounding error on that tensor, measured against the
  float64 oracle by tests/golden/make_f64_noise.py -- e.g. 0.56 % on the scalar fusion.loss_decoder.layers.2.bias)
``pytest -m gpu`` on a real MI355X.
"""
import random

import pytest
import torch

from interactron_amd.synthetic import load_procedural, synthetic_episodes
from tests.helpers import ReferenceMatching, check_grad, check_record, image_key

pytestmark = pytest.mark.gpu

MODEL_CFG = dict(WEIGHTS="procedural", NUM_CLASSES=1235, SET_COST_CLASS=1.0, SET_COST_BBOX=5.0, SET_COST_GIOU=2.0,
                 NUM_LAYERS=4, NUM_HEADS=8, EMBEDDING_DIM=512, BLOCK_SIZE=2060, IMG_FEATURE_SIZE=256, OUTPUT_SIZE=512,
                 BOX_EMB_SIZE=256, EMBEDDING_PDROP=0.1, RESIDUAL_PDROP=0.1, ATTENTION_PDROP=0.1, ADAPTIVE_LR=1e-3)


def to_gpu(data):
    d = dict(data)
    d["frames"], d["masks"], d["actions"] = data["frames"].cuda(), data["masks"].cuda(), data["actions"].cuda()
    d["category_ids"] = [[t.cuda() for t in ep] for ep in data["category_ids"]]
    d["boxes"] = [[t.cuda() for t in ep] for ep in data["boxes"]]
    return d


def make(model_type, **extra):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from interactron_amd import Config, build_model
    model = build_model(Config(**dict(MODEL_CFG, TYPE=model_type, **extra)))
    if hasattr(model, "fusion"):
        load_procedural(model.fusion, "fusion.")
    return model.cuda().eval()


@pytest.fixture(scope="module")
def interactron_model():
    return make("interactron")


@pytest.fixture(scope="module")
def episode1():
    return to_gpu(synthetic_episodes(1, tag="golden"))


def rec_tol(rec):
    ref = rec.get("full", rec.get("sample"))
    return 1e-3 * float(ref.abs().max()) + 1e-4


def _g2_inputs():
    import numpy as np
    from interactron_amd.synthetic import hash_normal, hash_uniform
    logits = torch.from_numpy((hash_normal("g2/logits", 5 * 50 * 1236) * 2.0).astype(np.float32)).reshape(5, 50, 1236)
    boxes = torch.from_numpy(np.concatenate([hash_uniform("g2/c", 500, 0.2, 0.8).reshape(250, 2),
                                             hash_uniform("g2/wh", 500, 0.05, 0.5).reshape(250, 2)], 1)
                             .astype(np.float32)).reshape(5, 50, 4)
    return logits, boxes


def test_g1_pairwise_giou_through_the_cost_kernel(golden):
    """G1 (reference box_ops.generalized_box_iou, [50, 7]) read back out of the HIP cost kernel: with the class and L1
    weights at zero and the GIoU weight at -1 the cost matrix IS the pairwise GIoU (matcher.py:63-70)."""
    from interactron_amd import hipops as ops
    g = golden("golden_small.pt")["g1"]
    pred, tgt = g["pred"].cuda(), g["tgt"].cuda()
    logits = torch.zeros(50, 1236, device="cuda")
    ids = torch.zeros(7, dtype=torch.int64, device="cuda")
    cost = ops.match_cost(logits, pred, ids, tgt, 0.0, 0.0, -1.0)
    torch.testing.assert_close(cost.cpu(), g["giou"], atol=2e-6, rtol=1e-5)


def test_g2_hungarian_indices_bit_exact_through_the_hip_matcher(golden):
    """The product matcher (HIP cost kernel -> one-wavefront-per-image assignment on the GPU, ix_lsap_device_f32) on the
    reference's G2 inputs: the int64 index pairs must EQUAL the reference's (matcher.py:54-77 + scipy) for all five images
    -- an empty image, and repeated ground-truth boxes whose cost columns tie exactly."""
    from interactron_amd import HungarianMatcher
    g = golden("golden_small.pt")["g2"]
    logits, boxes = _g2_inputs()
    targets = [{"labels": t["labels"].cuda(), "boxes": t["boxes"].cuda()} for t in g["targets"]]
    got = HungarianMatcher(1.0, 5.0, 2.0)({"pred_logits": logits.cuda(), "pred_boxes": boxes.cuda()}, targets)
    assert len(got) == 5
    for i, ((a, b), (ra, rb)) in enumerate(zip(got, g["indices"])):
        assert a.dtype == torch.int64 and b.dtype == torch.int64 and not a.is_cuda
        assert torch.equal(a, ra) and torch.equal(b, rb), ("image %d" % i, a, ra, b, rb)
    # one image at a time (the per-episode call shape of the sequential schedule) gives the same pairs
    for i in range(5):
        (a, b), = HungarianMatcher(1.0, 5.0, 2.0)({"pred_logits": logits[i:i + 1].cuda(), "pred_boxes": boxes[i:i + 1].cuda()},
                                                  targets[i:i + 1])
        assert torch.equal(a, g["indices"][i][0]) and torch.equal(b, g["indices"][i][1])


def test_g2_through_the_raw_c_abi_as_in_integration_md(golden):
    """The ctypes stub of INTEGRATION.md section 2 (ix_match_cost_csr_f32 + ix_lsap_device_f32 called with plain pointers, no
    package code in between) on the reference's G2 inputs: scipy's pairs for all five images."""
    import ctypes
    from interactron_amd._lib import LIB_PATH
    lib = ctypes.CDLL(LIB_PATH)
    P, I, F = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
    lib.ix_match_cost_csr_f32.argtypes = [P] * 6 + [I] * 4 + [F] * 3 + [P]
    lib.ix_lsap_device_f32.argtypes = [P, P, I, I, I, P, P, P]
    g = golden("golden_small.pt")["g2"]
    logits, boxes = _g2_inputs()
    pred_logits, pred_boxes = logits.cuda(), boxes.cuda()
    targets = [{"labels": t["labels"].cuda(), "boxes": t["boxes"].cuda()} for t in g["targets"]]
    bs, Q, C = pred_logits.shape
    sizes = [len(t["labels"]) for t in targets]
    ids, tb = torch.cat([t["labels"] for t in targets]), torch.cat([t["boxes"] for t in targets])
    off = torch.tensor([0] + torch.tensor(sizes).cumsum(0).tolist(), dtype=torch.int32).cuda()
    ldn = (max(sizes + [1]) + 7) // 8 * 8
    cost = torch.empty(bs, Q, ldn, device="cuda")
    toq = torch.empty(bs, Q, dtype=torch.int32, device="cuda")
    qot = torch.empty(max(ids.numel(), 1), dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    assert lib.ix_match_cost_csr_f32(pred_logits.data_ptr(), pred_boxes.data_ptr(), ids.data_ptr(), tb.data_ptr(), off.data_ptr(),
                                     cost.data_ptr(), bs, Q, C, ldn, 1.0, 5.0, 2.0, st) == 0
    assert lib.ix_lsap_device_f32(cost.data_ptr(), off.data_ptr(), bs, Q, ldn, toq.data_ptr(), qot.data_ptr(), st) == 0
    toq = toq.cpu()
    for i, (ra, rb) in enumerate(g["indices"]):
        src = torch.nonzero(toq[i] >= 0).reshape(-1)
        assert torch.equal(src, ra) and torch.equal(toq[i][src].long(), rb), ("image %d" % i)


def test_device_assignment_equals_host_assignment_bit_for_bit():
    """ix_lsap_device_f32 (one wavefront per image) against ix_lsap_f32 (the host restatement of scipy's algorithm, pinned to
    the reference by G2): identical assignments on random cost matrices -- fewer, as many and more targets than queries,
    empty images, costs quantised to a handful of values (exact ties everywhere), duplicated columns and duplicated rows,
    50 and 200 queries (the stress configuration)."""
    from interactron_amd import hipops as ops
    gen = torch.Generator().manual_seed(3)
    for Q in (50, 200):
        sizes = [0, 1, 3, 7, Q - 1, Q, Q + 1, 96, 0, 5, 12, 2]
        for kind in ("smooth", "ties", "dup"):
            mats = []
            for n in sizes:
                c = torch.rand(Q, n, generator=gen) * 4 - 2
                if kind == "ties":
                    c = (c * 2).round() / 2
                elif kind == "dup" and n >= 2:
                    c[:, 1::2] = c[:, 0:(n // 2) * 2:2]          # duplicated ground truth: columns tie exactly
                    c[Q // 2:Q // 2 + 3] = c[0:3]                # ... and three queries predicting the same thing
                mats.append(c)
            ldn = (max(sizes) + 7) // 8 * 8
            cost = torch.full((len(sizes), Q, ldn), float("nan"))
            for i, c in enumerate(mats):
                cost[i, :, :c.shape[1]] = c
            off = [0]
            for n in sizes:
                off.append(off[-1] + n)
            tg = ops.Targets(torch.zeros(max(off[-1], 1), dtype=torch.int64, device="cuda"),
                             torch.zeros(max(off[-1], 1), 4, device="cuda"), torch.tensor(off, dtype=torch.int32).cuda(), sizes)
            tg.ldn = ldn
            toq, qot = ops.lsap_device(cost.cuda(), tg)
            toq, qot = toq.cpu(), qot.cpu()
            for i, c in enumerate(mats):
                want = torch.full((Q,), -1, dtype=torch.int32)
                if c.shape[1]:
                    r, col = ops.lsap(c.contiguous())
                    want[r] = col.to(torch.int32)
                assert torch.equal(toq[i], want), (Q, kind, sizes[i], toq[i], want)
                inv = torch.full((c.shape[1],), -1, dtype=torch.int32)
                m = want >= 0
                inv[want[m].long()] = torch.nonzero(m).reshape(-1).to(torch.int32)
                assert torch.equal(qot[off[i]:off[i + 1]], inv), (Q, kind, sizes[i])


def test_device_assignment_leaves_non_finite_images_unmatched():
    """A cost matrix without a finite path (a row of NaN / +inf: a diverged step) is INFEASIBLE for scipy and the host route
    (they raise); the device kernel must not augment along "infinite" paths and hand back an arbitrary assignment: the image
    comes back unmatched (-1 everywhere it could not assign), the finite image beside it is untouched."""
    from interactron_amd import hipops as ops
    Q, n = 50, 6
    gen = torch.Generator().manual_seed(11)
    good = torch.rand(Q, n, generator=gen)
    for bad_value in (float("nan"), float("inf")):
        bad = torch.rand(Q, n, generator=gen)
        bad[:, 2] = bad_value                      # one target nobody can take at a finite cost
        cost = torch.zeros(2, Q, 8)
        cost[0, :, :n], cost[1, :, :n] = bad, good
        tg = ops.Targets(torch.zeros(2 * n, dtype=torch.int64, device="cuda"), torch.zeros(2 * n, 4, device="cuda"),
                         torch.tensor([0, n, 2 * n], dtype=torch.int32).cuda(), [n, n])
        tg.ldn = 8
        toq, qot = ops.lsap_device(cost.cuda(), tg)
        toq, qot = toq.cpu(), qot.cpu()
        r, col = ops.lsap(good.contiguous())
        want = torch.full((Q,), -1, dtype=torch.int32)
        want[r] = col.to(torch.int32)
        assert torch.equal(toq[1], want)
        assert int(qot[2]) == -1 and not bool((toq[0] == 2).any()), (bad_value, qot[:n], toq[0])
        assigned = toq[0][toq[0] >= 0]
        assert assigned.numel() == assigned.unique().numel() and assigned.numel() <= n - 1


def test_grouped_set_loss_against_float64():
    """hipops.SetLoss (the losses of several image groups from one pass, each with its own normalisers) against a float64
    restatement of reference detr.py:111-167,238-242 evaluated group by group: forward values of both group layouts and
    the gradients of the first one w.r.t. logits and boxes."""
    from interactron_amd import HungarianMatcher, SetCriterion, hipops as ops
    import torch.nn.functional as F
    gen = torch.Generator().manual_seed(5)
    I, Q, C, s = 6, 50, 1236, 3
    sizes = [3, 0, 7, 5, 1, 4]
    logits = (torch.randn(I, Q, C, generator=gen) * 2).cuda().requires_grad_(True)
    boxes = torch.cat([torch.rand(I, Q, 2, generator=gen) * 0.4 + 0.3, torch.rand(I, Q, 2, generator=gen) * 0.3 + 0.05], -1).cuda().requires_grad_(True)
    targets = [{"labels": torch.randint(0, C - 1, (n,), generator=gen).cuda(),
                "boxes": torch.cat([torch.rand(n, 2, generator=gen) * 0.4 + 0.3, torch.rand(n, 2, generator=gen) * 0.3 + 0.05], -1).cuda()}
               for n in sizes]
    crit = SetCriterion(C - 1, HungarianMatcher(1.0, 5.0, 2.0), {}, 0.1, ["labels", "boxes", "cardinality"]).cuda()
    tg = ops.pack_targets(targets)
    toq = crit.matcher.match({"pred_logits": logits, "pred_boxes": boxes}, tg)
    rows, single = crit.grouped({"pred_logits": logits, "pred_boxes": boxes}, tg, toq, ((s, s), (s, 1)), background_c=0.1)
    gw = torch.rand(I // s, 5, generator=gen).cuda()
    (rows * gw).sum().backward()
    toq_h = toq.cpu()

    def ref_group(lg, bx, imgs):
        w = torch.ones(C, dtype=torch.float64); w[-1] = 0.1
        tcls = torch.full((len(imgs), Q), C - 1, dtype=torch.int64)
        l1 = gi = torch.zeros((), dtype=torch.float64)
        nmatch = ncorrect = 0
        card = 0.0
        for a, i in enumerate(imgs):
            for q in range(Q):
                t = int(toq_h[i, q])
                if t >= 0:
                    tcls[a, q] = int(targets[i]["labels"][t])
                    p, tb = bx[i, q], targets[i]["boxes"][t].double().cpu()
                    l1 = l1 + (p - tb).abs().sum()
                    def xyxy(b): return torch.stack([b[0] - b[2] / 2, b[1] - b[3] / 2, b[0] + b[2] / 2, b[1] + b[3] / 2])
                    A, B = xyxy(p), xyxy(tb)
                    inter = (torch.min(A[2], B[2]) - torch.max(A[0], B[0])).clamp(min=0) * (torch.min(A[3], B[3]) - torch.max(A[1], B[1])).clamp(min=0)
                    ua = (A[2] - A[0]) * (A[3] - A[1]) + (B[2] - B[0]) * (B[3] - B[1]) - inter
                    hull = (torch.max(A[2], B[2]) - torch.min(A[0], B[0])) * (torch.max(A[3], B[3]) - torch.min(A[1], B[1]))
                    gi = gi + 1 - (inter / ua - (hull - ua) / hull)
                    nmatch += 1
                    ncorrect += int(lg[i, q].argmax()) == int(tcls[a, q])
            card += abs(int((lg[i].argmax(-1) != C - 1).sum()) - sizes[i])
        nb = max(sum(sizes[i] for i in imgs), 1)
        ce = F.cross_entropy(lg[list(imgs)].reshape(-1, C), tcls.reshape(-1), w)
        return torch.stack([ce, torch.tensor(100.0 - (100.0 * ncorrect / nmatch if nmatch else 0.0), dtype=torch.float64), l1 / nb, gi / nb,
                            torch.tensor(card / len(imgs), dtype=torch.float64)])

    lg64 = logits.detach().double().cpu().requires_grad_(True)
    bx64 = boxes.detach().double().cpu().requires_grad_(True)
    ref_rows = torch.stack([ref_group(lg64, bx64, range(g * s, g * s + s)) for g in range(I // s)])
    ref_single = torch.stack([ref_group(lg64, bx64, [g * s]) for g in range(I // s)])
    (ref_rows * gw.double().cpu()).sum().backward()
    torch.testing.assert_close(rows.detach().double().cpu(), ref_rows.detach(), atol=2e-5, rtol=2e-5)
    torch.testing.assert_close(single.detach().double().cpu(), ref_single.detach(), atol=2e-5, rtol=2e-5)
    torch.testing.assert_close(logits.grad.double().cpu(), lg64.grad, atol=1e-8, rtol=2e-4)
    torch.testing.assert_close(boxes.grad.double().cpu(), bx64.grad, atol=1e-7, rtol=2e-4)


def test_g3_set_criterion_losses_and_gradients(golden):
    """SetCriterion on the G2 inputs with the HIP matcher's OWN assignment (no pinning): five loss scalars and the
    gradients w.r.t. logits and boxes against the reference's (detr.py:111-265, background_c = 0.1)."""
    from interactron_amd import HungarianMatcher, SetCriterion
    G = golden("golden_small.pt")
    logits, boxes = _g2_inputs()
    targets = [{"labels": t["labels"].cuda(), "boxes": t["boxes"].cuda()} for t in G["g2"]["targets"]]
    crit = SetCriterion(1235, HungarianMatcher(1.0, 5.0, 2.0), {"loss_ce": 1, "loss_bbox": 5, "loss_giou": 2}, 0.1,
                        ["labels", "boxes", "cardinality"]).cuda()
    lg, bx = logits.cuda().requires_grad_(True), boxes.cuda().requires_grad_(True)
    losses = crit({"pred_logits": lg, "pred_boxes": bx}, targets, background_c=0.1)
    (losses["loss_ce"] + 5 * losses["loss_giou"] + 2 * losses["loss_bbox"]).backward()
    assert list(losses) == list(G["g3"]["losses"])
    for k, v in G["g3"]["losses"].items():
        torch.testing.assert_close(losses[k].detach().cpu(), v, atol=1e-5, rtol=1e-5, msg=lambda m: k + ": " + m)
    check_record(G["g3"]["grad_logits"], lg.grad, atol=1e-8, rtol=1e-4, what="g3/grad_logits")
    torch.testing.assert_close(bx.grad.cpu(), G["g3"]["grad_boxes"], atol=1e-7, rtol=1e-4)


def test_g10_clipped_sgd_bit_exact(golden):
    """The fused multi-tensor inner step on the reference's G10 tensors (meta_utils.py:135-142: clipped elements, a None
    gradient): p - clamp(lr g, +-0.01) is one multiply, one clamp and one subtract per element, so the float32 results
    must be bit-identical."""
    import numpy as np
    from interactron_amd.meta import sgd_step
    from interactron_amd.synthetic import hash_normal
    g = golden("golden_small.pt")["g10"]
    p = [torch.from_numpy(hash_normal("g10/p%d" % i, n).astype(np.float32)).cuda() for i, n in enumerate((1000, 37, 4096))]
    gr = [torch.from_numpy((hash_normal("g10/g%d" % i, n) * 12.0).astype(np.float32)).cuda() for i, n in enumerate((1000, 37, 4096))]
    gr[1] = None
    out = sgd_step(p, gr, 1e-3)
    for a, b in zip(out, g["out"]):
        assert torch.equal(a.cpu(), b)


@pytest.mark.usefixtures("kernel_form")
def test_g7_detector_forward(golden, interactron_model, episode1):
    from interactron_amd import NestedTensor
    M = golden("golden_model.pt")
    with torch.no_grad():
        out = interactron_model.detector(NestedTensor(episode1["frames"][0], episode1["masks"][0]))
    for k, rec in M["g7"].items():
        check_record(rec, out[k], atol=rec_tol(rec), rtol=1e-3, what="g7/" + k)


def _hash_tensor(tag, *shape):
    import numpy as np
    from interactron_amd.synthetic import hash_normal
    n = 1
    for d in shape:
        n *= d
    return torch.from_numpy(hash_normal(tag, n).astype(np.float32)).reshape(*shape)


@pytest.mark.usefixtures("kernel_form")
def test_g5_bottlenecks_on_the_hip_modules(golden, interactron_model):
    """G5 (reference backbone.py:88-90 = torchvision v1.5 bottlenecks + FrozenBatchNorm2d backbone.py:19-54, recorded by running
    the reference's own modules): the strided block layer2.0 (stride on the 3x3, strided 1x1 downsample), layer4.0 (first block
    of the dilated layer: keeps dilation 1, 1x1 downsample without stride) and layer4.1 (dilation 2) on the HIP modules alone --
    implicit-GEMM convolutions with the frozen-BN affine / residual / ReLU riding on the contraction -- not through the whole
    detector (G7)."""
    M = golden("golden_model.pt")
    body = interactron_model.detector.backbone[0].body
    x = _hash_tensor("g5/x", 2, 256, 20, 20).abs().cuda()
    x4 = _hash_tensor("g5/x4", 2, 1024, 10, 10).abs().cuda()
    with torch.no_grad():
        y2 = body.layer2[0](x.permute(0, 2, 3, 1).contiguous())
        y40 = body.layer4[0](x4.permute(0, 2, 3, 1).contiguous())
        y41 = body.layer4[1](y40)
    for name, y in (("g5_layer2_0", y2), ("g5_layer4_0", y40), ("g5_layer4_1", y41)):
        rec = M[name]
        check_record(rec, y.permute(0, 3, 1, 2), atol=rec_tol(rec), rtol=1e-3, what=name)


@pytest.mark.usefixtures("kernel_form")
def test_g6_encoder_and_decoder_layer_on_the_hip_modules(golden, interactron_model):
    """G6 (reference transformer.py:148-161 forward_post of the encoder layer, :211-232 of the decoder layer, eval mode, recorded
    from the reference's own layers): 30 tokens x 2 sequences with a key-padding mask on the second one, 7 queries with a
    per-sequence query position table -- the HIP layers alone (packed in_proj contraction, flash attention at head dim 32 with a
    key bias, add + LayerNorm, FFN)."""
    M = golden("golden_model.pt")
    tr = interactron_model.detector.transformer
    src, pos = _hash_tensor("g6/src", 30, 2, 256).cuda(), _hash_tensor("g6/pos", 30, 2, 256).cuda()
    tgt, qp = _hash_tensor("g6/tgt", 7, 2, 256).cuda(), _hash_tensor("g6/qp", 7, 2, 256).cuda()
    kpm = torch.zeros(2, 30, dtype=torch.uint8, device="cuda")
    kpm[1, 25:] = 1
    bf = lambda t: t.permute(1, 0, 2).contiguous()   # the reference is sequence-first, the HIP modules batch-first
    with torch.no_grad():
        enc = tr.encoder.layers[0](bf(src), kpm, bf(pos))
        check_record(M["g6_enc"], enc.permute(1, 0, 2), atol=rec_tol(M["g6_enc"]), rtol=1e-3, what="g6_enc")
        # decoder layer: the memory is the REFERENCE's encoder output only up to fp32 rounding -- use ours (as the reference did its own)
        from interactron_amd import hipops
        memory_key = hipops.add(enc, bf(pos))
        dec = tr.decoder.layers[0](bf(tgt), enc, memory_key, kpm, bf(qp).reshape(2, 7 * 256))
        check_record(M["g6_dec"], dec.permute(1, 0, 2), atol=rec_tol(M["g6_dec"]), rtol=1e-3, what="g6_dec")


@pytest.mark.usefixtures("kernel_form")
def test_g8_g9_fusion_and_learned_loss_gradient(golden, interactron_model, episode1):
    from interactron_amd import NestedTensor, hipops
    from interactron_amd.meta import get_parameters, set_parameters
    M = golden("golden_model.pt")
    m = interactron_model
    theta = get_parameters(m.detector)
    dtheta = [p.detach().requires_grad_(True) for p in theta]
    set_parameters(m.detector, dtheta)
    try:
        out = m.detector(NestedTensor(episode1["frames"][0], episode1["masks"][0]))
        pre = {k: (v.unsqueeze(0) if k != "image_features" else v) for k, v in out.items()}
        fo = m.fusion(pre)
        for k, rec in M["g8"].items():
            check_record(rec, fo[k], atol=rec_tol(rec), rtol=1e-3, what="g8/" + k)
        learned = hipops.l2_norm(fo["loss"])
        assert abs(float(learned) - M["g9"]["learned_loss"]) < 1e-3 * abs(M["g9"]["learned_loss"])
        g = torch.autograd.grad(learned, dtheta, allow_unused=True)
    finally:
        set_parameters(m.detector, theta)
    for name, gi in zip(M["theta_names"], g):
        # FIRST-order gradients meet SURVEY 8d's proposal (1e-3 per tensor; measured worst 5.6e-5: tools/parity_survey.sh, r4)
        check_grad(M["g9"]["grads"][name], gi, rel=1e-3, what="g9/" + name)


@pytest.mark.usefixtures("kernel_form")
def test_g11_g12_predict_and_next_action(golden, interactron_model, episode1):
    M = golden("golden_model.pt")
    pred = interactron_model.predict(episode1)
    for k, rec in M["g11"].items():
        check_record(rec, pred[k], atol=rec_tol(rec), rtol=1e-3, what="g11/" + k)
    for s in range(1, 5):
        d = {"frames": episode1["frames"][:, :s], "masks": episode1["masks"][:, :s]}
        assert interactron_model.get_next_action(d) == M["g12"][s - 1]
        # ... and the [4, 4] policy logits behind the int (captured from the reference's fusion by a forward hook)
        ref = M["g12_logits"][s - 1].reshape(4, 4)
        got = interactron_model._policy_logits(d["frames"][0], d["masks"][0]).reshape(4, 4)
        torch.testing.assert_close(got.cpu(), ref, atol=1e-3 * float(ref.abs().max()) + 1e-4, rtol=1e-3)


# measured (r3): the MEDIAN tensor is closer to float64 than the reference's own float32 is, with either form of the contraction
# kernel (norm 1.8e-5 (x3) / 3.1e-5 (x6) vs 1.0e-4; strided sample 1.7e-4 / 2.6e-4 vs 6.6e-4 of the tensor).  Worst excess over
# twice the reference's error: x6: norm 1.05e-3 (layer3.3.conv1), sample 3.3e-5;  x3: norm 2.4e-3 (layer3.3.conv1), sample
# 1.4e-2 on ONE tensor (layer3.0.conv2, elements on ReLU / clip kinks: the reference itself is 0.7 % off the truth on that
# sample, HIP 2.7 %; it was 1.7 % before the round-3 retune of the tile / split plans -- another summation order), every
# other tensor below 3e-3.  The worst three are printed.
F64_SLACK = {"x6": (2e-3, 1e-3), "x3": (6e-3, 3e-2)}
# ... and that looser fp16x3 bound may only be USED by kink tensors: every tensor but at most F64_KINK_TENSORS of them must
# meet F64_TIGHT (norm, strided sample) in either form.  Round 4 looked for the cause (profiles/README.md): the outliers are
# backbone weights of the second-order pass (layer4.0.conv3 2.6 %, layer3.0.conv2 1.9 % of the sample's scale, the
# reference's own float32 0.7 % on both) and they do NOT move when the convolutions' weight-gradient contractions run on
# the bf16x6 form (24-bit operands) -- the difference is upstream: single elements whose ReLU mask / clipped inner step
# flips under another rounding, not the operand precision of one contraction.  Third-worst tensor: below 1e-5.
F64_TIGHT = (3e-3, 5e-3)
F64_KINK_TENSORS = 2


# Second-order gradients (G13, config 3): SURVEY 8d's 1e-3 per tensor + 3 x the reference's OWN float32 error on that tensor
# (|fp32 - fp64| from tests/golden/golden_train_f64.pt, passed to check_grad as norm64 / sample64) -- round 5 used a blanket
# 5e-3.  The tensors below need more than that in at least one contraction form (measured with IX_TEST_RECORD_ALL=1, the
# worst ratio over the three forms rounded up; BASELINE.md section 4 lists them with the measurements): elements of the
# clipped inner step and ReLUs upstream of them sit on kinks and flip under any other float32 summation order.
SECOND_ORDER_REL = {   # measured r6a (gpurun_out/r6a_survey.txt): worst ratio to the 1e-3 bound over the three forms, rounded up
    "g13/detector.backbone.0.body.layer3.3.conv1.weight": 2.5e-3,    # norm 1.83 x
    "g13/detector.backbone.0.body.layer3.5.conv2.weight": 1.5e-3,    # norm 1.04 x
    "g13/detector.backbone.0.body.layer2.1.conv3.weight": 2.5e-3,    # 10-12 of 256 strided samples beyond 20 x 1e-3 x RMS (5 allowed), worst 1.7 x
    "rand/detector.backbone.0.body.layer2.0.conv1.weight": 2.5e-3,   # strided sample L2 1.26 x of (4e-3 x sample norm + the float64 slack)
}
# ... and the whole-tensor direction (G13b): 0.9999 everywhere but on the first trainable convolution, the tensor furthest upstream
# of every kink (measured 0.99990 in the bf16x6 form, 0.99991 / 0.99992 in the others)
FULL_COS_MIN = {"detector.backbone.0.body.layer2.0.conv1.weight": 0.9998}


def second_order_rel(what):
    return SECOND_ORDER_REL.get(what, 1e-3)


def _decompress(rec):
    return rec["half"].float() * 2.0 ** rec["exp"]


def check_full_gradients(F, model, cos_min=0.9999, norm_rel=2e-3):
    """G13b (tests/golden/golden_train_full.pt, make_golden_full.py): WHOLE gradient tensors of the reference's step for a dozen
    representative parameters at the real 300 x 300 / T = 2060 shape: cosine >= 0.9999 over every element, norm within 2e-3."""
    worst = (1.0, "")
    for grp, mod in (("detector", model.detector), ("fusion", model.fusion)):
        named = dict(mod.named_parameters())
        for k, rec in F[grp].items():
            g = named[k].grad.detach().cpu()
            if g.dim() == 4 and tuple(g.shape) != tuple(rec["shape"]):
                g = g.permute(0, 3, 1, 2).contiguous()   # conv weights are stored [out, kh, kw, in]
            assert tuple(g.shape) == tuple(rec["shape"]), (k, tuple(g.shape), rec["shape"])
            ref = _decompress(rec).double()
            cos = float((g.double() * ref).sum() / (g.double().norm() * ref.norm()))
            worst = min(worst, (cos, grp + "." + k))
            assert cos >= FULL_COS_MIN.get(grp + "." + k, cos_min), (grp, k, "direction over the whole tensor", cos)
            n = float(g.double().norm())
            assert abs(n - rec["norm"]) <= max(norm_rel, 2 * second_order_rel("g13/%s.%s" % (grp, k))) * rec["norm"] + 1e-9, (grp, k, n, rec["norm"])
    print("G13b: %d whole gradient tensors, smallest cosine %.7f on %s" % (len(F["detector"]) + len(F["fusion"]), worst[0], worst[1]))


def test_g13_g16_meta_train_step_and_outer_update(golden, kernel_form):
    T = golden("golden_train.pt")
    F64 = golden("golden_train_f64.pt")   # exact (float64 oracle) norms: bounds the reference's own float32 noise
    m = make("interactron")
    data = to_gpu(synthetic_episodes(2, tag="golden"))
    data["initial_image_path"] = ["golden/ep0", "golden/ep0"]
    m.zero_grad()
    random.seed(T["g13"]["ridx_seed"])
    with ReferenceMatching(golden("golden_indices.pt")["g13"]) as rm:
        preds, losses = m(data)
    assert rm.calls >= 2 * (5 + 1)   # 5 supervised frames + the 1-frame detector loss per episode (frame-0 reward reuses)
    for k, rec in T["g13"]["preds"].items():
        check_record(rec, preds[k], atol=rec_tol(rec), rtol=1e-3, what="g13/" + k)
    assert list(losses) == list(T["g13"]["losses"])
    for k, v in T["g13"]["losses"].items():
        assert abs(float(losses[k]) - float(v)) <= 2e-3 * max(abs(float(v)), 1.0), (k, float(losses[k]), float(v))
    for k, p in m.detector.named_parameters():
        check_grad(T["g13"]["detector_grads"][k], p.grad, rel=second_order_rel("g13/detector." + k), what="g13/detector." + k,
                   norm64=F64["detector_grads"].get(k), sample64=F64["detector_grads_sample64"].get(k))
    for k, p in m.fusion.named_parameters():
        check_grad(T["g13"]["fusion_grads"][k], p.grad, rel=second_order_rel("g13/fusion." + k), what="g13/fusion." + k,
                   norm64=F64["fusion_grads"].get(k), sample64=F64["fusion_grads_sample64"].get(k))
    check_full_gradients(golden("golden_train_full.pt"), m)
    # "Is HIP as close to the truth as the reference is?"  Against the float64 oracle, per tensor, on the norm and on the 256
    # strided positions: |HIP - f64| <= 2 |reference fp32 - f64| + F64_SLACK |f64|  (the excess over twice the reference's
    # own float32 error, relative to the tensor; the worst tensors are printed).
    ex_norm, ex_samp = [], []
    for grp, mod in (("detector", m.detector), ("fusion", m.fusion)):
        for k, p in mod.named_parameters():
            rec, n64, s64 = T["g13"][grp + "_grads"][k], F64[grp + "_grads"].get(k), F64[grp + "_grads_sample64"].get(k)
            if rec is None or n64 is None or n64 < 1e-6:
                continue
            g = p.grad.detach().cpu()
            if g.dim() == 4 and tuple(g.shape) != tuple(rec["shape"]):
                g = g.permute(0, 3, 1, 2).contiguous()
            n = float(g.double().norm())
            ex_norm.append(((abs(n - n64) - 2 * abs(rec["norm"] - n64)) / n64, abs(n - n64) / n64, abs(rec["norm"] - n64) / n64, grp + "." + k))
            got, ref32 = g.reshape(-1)[rec["idx"]].double(), rec["sample"].double()
            e_hip, e_ref = float((got - s64).norm()), float((ref32 - s64).norm())
            # (the sample's scale: the tensor's RMS times sqrt(#samples) -- a sample of near-zero entries says nothing)
            scale = max(float(s64.norm()), n64 / max(g.numel(), 1) ** 0.5 * len(s64) ** 0.5)
            ex_samp.append(((e_hip - 2 * e_ref) / scale, e_hip / scale, e_ref / scale, grp + "." + k))
    ex_norm.sort(reverse=True)
    ex_samp.sort(reverse=True)
    print("g13 vs float64, worst three strided samples (excess, HIP, reference, tensor):", [(round(a, 5), round(b, 5), round(c, 5), n) for a, b, c, n in ex_samp[:3]])
    for name, ex in (("norm", ex_norm), ("strided sample", ex_samp)):
        print("g13 vs float64, %s: worst excess %.2e (HIP %.2e, reference fp32 %.2e) on %s; median HIP error %.2e, median "
              "reference error %.2e" % ((name,) + ex[0] + (sorted(e[1] for e in ex)[len(ex) // 2], sorted(e[2] for e in ex)[len(ex) // 2])))
    assert ex_norm[0][0] <= F64_SLACK[kernel_form][0], ex_norm[:3]
    assert ex_samp[0][0] <= F64_SLACK[kernel_form][1], ex_samp[:3]
    loose = {e[3] for e in ex_norm if e[0] > F64_TIGHT[0]} | {e[3] for e in ex_samp if e[0] > F64_TIGHT[1]}
    assert len(loose) <= F64_KINK_TENSORS, ("more than %d tensors need the loose bound" % F64_KINK_TENSORS, sorted(loose))
    assert all("backbone" in k for k in loose), sorted(loose)
    labels = {k: v.get_label(data["actions"][0][:4].tolist()) for k, v in m.path_storage.items()}
    assert labels == T["g13"]["path_labels"]
    # G16: clip_grad_norm_(all, 1.0) + Adam(detector, 1e-5) + Adam(fusion, 1e-4) as one fused flat-buffer step
    from interactron_amd.trainer import FlatOuterStep
    before = {k: v.detach().clone() for k, v in m.named_parameters()}
    step = FlatOuterStep(m, detector_lr=1e-5, fusion_lr=1e-4, max_norm=1.0)
    total = step.step()
    assert abs(float(total) - T["g16"]["total_norm"]) <= 5e-3 * T["g16"]["total_norm"]
    ref_grads = {"detector." + k: v for k, v in T["g13"]["detector_grads"].items()}
    ref_grads.update({"fusion." + k: v for k, v in T["g13"]["fusion_grads"].items()})
    for k, v in m.named_parameters():
        rec = T["g16"]["delta"][k]
        if ref_grads[k] is not None and ref_grads[k]["norm"] < 1e-6:
            continue   # rounding-noise gradient (attention key bias): Adam normalises noise to +-lr, nothing to compare
        dn = float((v.detach() - before[k]).double().norm())
        assert abs(dn - rec["norm"]) <= 2e-2 * max(rec["norm"], 1e-9) + 1e-9, (k, dn, rec["norm"])


def test_step_is_bit_reproducible():
    """Two runs of one meta-train step (train mode: dropout on, same seeds; episode-batched; second-order backward; clip +
    Adam) from the same state give BIT-identical losses, gradients and updated parameters.  Every sum that spans several
    workgroups is ordered (split-K partial planes, ticketed column / scalar sums), the flash kernels own their outputs, and
    autograd accumulates in graph order -- no fp32 atomics are left on the path.  (Round 2: up to 0.7 % per tensor.)"""
    from interactron_amd import Config, build_model, hipops
    from interactron_amd.trainer import FlatOuterStep
    data = to_gpu(synthetic_episodes(2, height=128, width=160, tag="repro"))
    runs = []
    for rep in range(2):
        m = build_model(Config(**dict(MODEL_CFG, TYPE="interactron", EPISODE_CHUNK=2)))
        load_procedural(m.fusion, "fusion.")
        m = m.cuda().train()
        outer = FlatOuterStep(m)
        random.seed(7)
        hipops.manual_seed(1234)
        _, losses = m(data)
        grads = {k: (None if p.grad is None else p.grad.clone()) for k, p in m.named_parameters()}
        outer.step()
        torch.cuda.synchronize()
        runs.append(({k: v.clone() for k, v in losses.items()}, grads, {k: p.detach().clone() for k, p in m.named_parameters()}))
    (l0, g0, p0), (l1, g1, p1) = runs
    for k in l0:
        assert torch.equal(l0[k], l1[k]), ("loss differs between two runs", k, float(l0[k]), float(l1[k]))
    bad = [k for k in g0 if (g0[k] is None) != (g1[k] is None) or (g0[k] is not None and not torch.equal(g0[k], g1[k]))]
    assert not bad, ("gradients differ between two runs of the same step", len(bad), bad[:5])
    bad = [k for k in p0 if not torch.equal(p0[k], p1[k])]
    assert not bad, ("updated parameters differ between two runs", len(bad), bad[:5])


@pytest.mark.parametrize("model_type,dtype", [("detr_multiframe", "f32"), ("detr_multiframe", "bf16"), ("detr", "f32")])
def test_outer_step_without_persistent_grad_views_equals_the_accumulating_one(model_type, dtype):
    """FlatOuterStep on the models whose step is one plain backward pass: the parameters enter the backward without a .grad (autograd keeps
    the incoming gradient tensors: no at::add per parameter), step() copies them into the flat buffer in one multi-tensor launch set --
    against the same three steps with persistent flat .grad views (steal_grads=False): losses and updated parameters BIT-identical
    (0 + g == g), no gradient left on a parameter after the step, and what run_evaluation drops is really dropped (zero_grads)."""
    from interactron_amd import Config, build_model
    from interactron_amd.trainer import FlatOuterStep
    hist = []
    for steal in (False, True):
        m = build_model(Config(**dict(MODEL_CFG, TYPE=model_type, COMPUTE_DTYPE=dtype)))
        if hasattr(m, "fusion"):
            load_procedural(m.fusion, "fusion.")
        m = m.cuda().eval()
        outer = FlatOuterStep(m, max_norm=1.0, groups=[list(m.parameters())], lrs=[1e-5], steal_grads=steal)
        assert outer.steal_grads == steal
        steps = []
        for k in range(3):
            data = to_gpu(synthetic_episodes(2, height=128, width=160, tag="steal-%d" % k))
            random.seed(4 + k)
            if k == 1:   # a discarded backward pass in between (the trainers' test epoch)
                m(data)
                outer.zero_grads()
            _, losses = m(data)
            if steal:
                assert sum(p.grad is not None and p.grad.data_ptr() != v.data_ptr() for p, v in outer.flat.grad_views) > 50
            outer.step()
            if steal:
                assert all(p.grad is None for p, _ in outer.flat.grad_views)
            steps.append(({k2: float(v) for k2, v in losses.items()}, outer.flat.params.clone()))
        hist.append(steps)
    for (l0, p0), (l1, p1) in zip(*hist):
        assert l0 == l1, (l0, l1)
        assert torch.equal(p0, p1)


def _graph_vs_eager_models(model_type, **extra):
    from interactron_amd import Config, build_model
    from interactron_amd.trainer import FlatOuterStep
    out = []
    for graph in ("false", "true"):
        m = build_model(Config(**dict(MODEL_CFG, TYPE=model_type, EPISODE_CHUNK=2, STEP_GRAPH=graph, PREDICT_GRAPH=(graph == "true"), **extra)))
        load_procedural(m.fusion, "fusion.")
        out.append((m.cuda(), None))
    return [(m, FlatOuterStep(m)) for m, _ in out]


@pytest.mark.parametrize("model_type", ["interactron", "interactron_random"])
def test_chunk_graph_replay_equals_eager_launches(model_type):
    """STEP_GRAPH: the three sync-free segments of a chunk replayed from captured HIP graphs (graphs.ChunkGraphs: static
    input buffers, device matcher, one host round trip for PathStorage) against the same segments issued launch by launch:
    four steps on alternating batches, each followed by clip + Adam -- losses, predictions, every gradient and every
    updated parameter must be BIT-identical (eval mode: no dropout; step 1 is the eager warm-up of the graph model, step 2
    captures, steps 3-4 replay with other inputs)."""
    _check_graph_replay(model_type)


@pytest.mark.parametrize("dtype", ["bf16", "bf16_fusion"])
def test_chunk_graph_replay_equals_eager_launches_in_the_16_bit_modes(dtype):
    """The same four steps with MODEL.COMPUTE_DTYPE bf16 / bf16_fusion: under a replay the parameters reach the matrix cores through the
    flat bf16 shadow that FlatOuterStep converts after every optimiser step (trainer.FlatBuffers.sync_b16) -- static addresses the
    captured segments read -- and the adapted fast weights are converted inside the graph; everything must stay BIT-identical to the
    launch-by-launch run, updated parameters included."""
    _check_graph_replay("interactron", COMPUTE_DTYPE=dtype)


def _check_graph_replay(model_type, **extra):
    (me, oe), (mg, og) = _graph_vs_eager_models(model_type, **extra)
    hw = (128, 160) if model_type == "interactron" else (300, 300)   # (the decoder fusion's position table is 19 x 19)
    batches = [to_gpu(synthetic_episodes(2, height=hw[0], width=hw[1], tag="graph-%s" % t)) for t in ("a", "b")]
    for m in (me, mg):
        m.eval()
    hist = []
    for m, o in ((me, oe), (mg, og)):
        random.seed(21)
        steps = []
        for k in range(4):
            preds, losses = m(batches[k % 2])
            grads = {n: (None if p.grad is None else p.grad.clone()) for n, p in m.named_parameters()}
            o.step()
            steps.append((preds, losses, grads, {n: p.detach().clone() for n, p in m.named_parameters()}))
        hist.append(steps)
    kinds = [type(v).__name__ for v in mg.__dict__.get("_chunk_graphs", {}).values()]
    assert kinds == ["ChunkGraphs"], ("the graph model did not capture", mg.__dict__.get("_chunk_graphs"))
    assert not me.__dict__.get("_chunk_graphs")
    for k, ((p0, l0, g0, w0), (p1, l1, g1, w1)) in enumerate(zip(*hist)):
        for n in p0:
            assert torch.equal(p0[n], p1[n]), (k, "prediction", n)
        for n in l0:
            assert torch.equal(l0[n], l1[n]), (k, "loss", n, float(l0[n]), float(l1[n]))
        bad = [n for n in g0 if (g0[n] is None) != (g1[n] is None) or (g0[n] is not None and not torch.equal(g0[n], g1[n]))]
        assert not bad, (k, "gradients", len(bad), bad[:4])
        bad = [n for n in w0 if not torch.equal(w0[n], w1[n])]
        assert not bad, (k, "parameters after clip + Adam", len(bad), bad[:4])
    if model_type == "interactron":
        assert {k: v.root.action for k, v in me.path_storage.items()} == {k: v.root.action for k, v in mg.path_storage.items()}


def test_chunk_graph_replays_draw_fresh_dropout_masks():
    """Train mode under graph replay: the dropout seeds are launch arguments frozen into the graph, the device salt
    (ix_set_dropout_salt) is bumped before every replay -- two replays on the same batch from the same weights must differ
    (and stay finite); the same salt reproduces the same step bit for bit."""
    from interactron_amd import Config, build_model
    m = build_model(Config(**dict(MODEL_CFG, TYPE="interactron", EPISODE_CHUNK=2, STEP_GRAPH="true")))
    load_procedural(m.fusion, "fusion.")
    m = m.cuda().train()
    for p in m.parameters():
        if p.requires_grad:
            p.grad = torch.zeros_like(p)
    data = to_gpu(synthetic_episodes(2, height=128, width=160, tag="graph-drop"))
    vals = []
    for k in range(5):
        random.seed(3)
        m.path_storage.clear()
        m.zero_grad(set_to_none=False)
        if k == 4:   # rewind the salt by one step: replay k = 3 again
            g = next(iter(m._chunk_graphs.values()))
            from interactron_amd.graphs import GOLDEN
            g.salt.sub_(GOLDEN).sub_(GOLDEN)
        _, losses = m(data)
        vals.append((float(losses["loss_supervisor_ce"]), m.fusion.loss_decoder.layers[0].weight.grad.clone()))
    assert all(v == v and abs(v) < 1e4 for v, _ in vals)
    assert vals[2][0] != vals[3][0] and not torch.equal(vals[2][1], vals[3][1]), "two replays drew the same dropout masks"
    assert vals[4][0] == vals[2][0] and torch.equal(vals[4][1], vals[2][1]), "same salt, different step"


def test_predict_graph_replay_equals_eager():
    """predict() of one episode from a captured graph (graphs.PredictGraph) == eager launches, bit for bit, on three calls with
    two different episodes (call 1 warms, call 2 captures, call 3 replays other inputs)."""
    (me, _), (mg, _) = _graph_vs_eager_models("interactron")
    eps = [to_gpu(synthetic_episodes(1, height=128, width=160, tag="pg-%d" % i)) for i in range(2)]
    for m in (me, mg):
        m.eval()
    for k in (0, 1, 0, 1):
        a, b = me.predict(eps[k]), mg.predict(eps[k])
        assert list(a) == list(b)
        for n in a:
            assert a[n].shape == b[n].shape and torch.equal(a[n], b[n]), (k, n)
    assert [type(v).__name__ for v in mg._predict_graphs.values()] == ["PredictGraph"]


def test_skipped_gradients_change_nothing_but_the_launch_count():
    """hipops.skip_param_grads: the weight-gradient contractions nobody asked for (nn.Parameters in the MAML inner
    gradient, the per-episode copies in the supervisor backward) are not launched.  Same model, same episodes, with and
    without the skip: identical losses and None-pattern, gradients equal up to split-K summation order, fewer launches."""
    import ctypes
    from interactron_amd import Config, build_model, _lib, hipops
    lib = _lib.load()
    data = to_gpu(synthetic_episodes(2, height=128, width=160, tag="skip"))
    res = []
    for skip in (True, False):
        hipops.SKIP_UNUSED_GRADS = skip
        try:
            m = build_model(Config(**dict(MODEL_CFG, TYPE="interactron", EPISODE_CHUNK=2)))
            load_procedural(m.fusion, "fusion.")
            m = m.cuda().eval()
            m.zero_grad()
            random.seed(5)
            lib.ix_gemm_stats(None, None, 1)
            _, losses = m(data)
            torch.cuda.synchronize()
            fl, n = ctypes.c_double(), ctypes.c_int64()
            lib.ix_gemm_stats(ctypes.byref(fl), ctypes.byref(n), 1)
            res.append((losses, {k: (None if p.grad is None else p.grad.clone()) for k, p in m.named_parameters()}, fl.value, n.value))
        finally:
            hipops.SKIP_UNUSED_GRADS = True
    (l1, g1, f1, n1), (l0, g0, f0, n0) = res
    assert n1 < n0 - 50 and f1 < 0.97 * f0, (n1, n0, f1, f0)
    # The gradients that ARE computed come from the same launches in the same order either way, and every reduction is
    # ordered: they must be bit-identical (round 2, with split-K atomics: whole gradient within 1 %, single tensors 30 %).
    for k in l0:
        assert torch.equal(l0[k], l1[k]), (k, float(l0[k]), float(l1[k]))
    for k in g0:
        assert (g0[k] is None) == (g1[k] is None), k
        if g0[k] is not None:
            assert torch.equal(g0[k], g1[k]), (k, float((g0[k] - g1[k]).abs().max()), float(g0[k].abs().max()))


def test_episode_batched_equals_sequential_schedule():
    """EPISODE_CHUNK > 1 (episodes of a batch run together with per-episode fast weights [E, ...]) must reproduce the
    reference's task-by-task schedule (EPISODE_CHUNK = 0): same losses, same accumulated gradients."""
    from interactron_amd import Config, build_model
    data = to_gpu(synthetic_episodes(3, height=128, width=160, tag="chunk"))
    from interactron_amd import criterion as cr
    res = []
    recorded, orig = {}, cr.HungarianMatcher.assign
    for chunk in (0, 2):
        m = build_model(Config(**dict(MODEL_CFG, TYPE="interactron", EPISODE_CHUNK=chunk)))
        load_procedural(m.fusion, "fusion.")
        m = m.cuda().eval()
        m.zero_grad()
        random.seed(11)
        if chunk == 0:   # record the sequential schedule's assignments, pin the batched run to them (ties: see helpers)
            def spy(matcher, costs, targets):
                out = orig(matcher, costs, targets)
                for t, rc in zip(targets, out):
                    recorded.setdefault(image_key(t), []).append(rc)
                return out
            cr.HungarianMatcher.assign = spy
            try:
                preds, losses = m(data)
            finally:
                cr.HungarianMatcher.assign = orig
        else:
            with ReferenceMatching(recorded, ordered=True):
                preds, losses = m(data)
        res.append((preds, losses, {k: (None if p.grad is None else p.grad.clone()) for k, p in m.named_parameters()},
                    {k: v.get_label(data["actions"][i][:4].tolist()) for i, (k, v) in enumerate(m.path_storage.items())}))
    (p0, l0, g0, s0), (p1, l1, g1, s1) = res
    assert s0 == s1
    for k in p0:
        torch.testing.assert_close(p1[k], p0[k], atol=1e-3 * float(p0[k].abs().max()) + 1e-4, rtol=1e-3)
    for k in l0:
        assert abs(float(l0[k]) - float(l1[k])) <= 1e-3 * max(abs(float(l0[k])), 1.0), k
    rel = {}
    for k in g0:
        assert (g0[k] is None) == (g1[k] is None), k
        if g0[k] is None:
            continue
        n0, n1 = float(g0[k].double().norm()), float(g1[k].double().norm())
        if max(n0, n1) < 1e-6:
            continue
        rel[k] = (abs(n0 - n1) / n0, float((g0[k] - g1[k]).double().norm()) / n0)
    have = [k for k in g0 if g0[k] is not None]
    num = sum(float((g0[k] - g1[k]).double().norm()) ** 2 for k in have)
    den = sum(float(g0[k].double().norm()) ** 2 for k in have)
    worst_n = max(rel.items(), key=lambda kv: kv[1][0])
    worst_d = max(rel.items(), key=lambda kv: kv[1][1])
    vals = sorted(v[1] for v in rel.values())
    print("batched vs sequential: whole gradient %.2e, median tensor %.2e, worst norm %.2e (%s), worst difference %.2e (%s)"
          % ((num / den) ** 0.5, vals[len(vals) // 2], worst_n[1][0], worst_n[0], worst_d[1][1], worst_d[0]))
    # Two different summation orders (batched vs per-episode shapes pick other tiles / splits) through ~50 ReLU layers and the
    # clipped inner step: deterministic now, so the bounds sit 3x above what this comparison measures (printed above) instead
    # of above the run-to-run scatter of the atomics (round 2: 5 % / 15 % per tensor).  A mis-routed episode weight is O(1).
    assert (num / den) ** 0.5 <= BATCHED_WHOLE, (num, den)
    assert vals[len(vals) // 2] <= BATCHED_MEDIAN, vals[len(vals) // 2]
    assert worst_n[1][0] <= BATCHED_WORST_NORM, worst_n
    assert worst_d[1][1] <= BATCHED_WORST_DIFF, worst_d


# measured (r3, deterministic): whole 8.4e-4, median 3.8e-4, worst norm 3.1e-3, worst difference 4.3e-2 (layer4.1.conv1)
# (the numbers move with the contraction plans -- other tiles, other summation orders -- hence the room)
BATCHED_WHOLE, BATCHED_MEDIAN, BATCHED_WORST_NORM, BATCHED_WORST_DIFF = 5e-3, 3e-3, 2e-2, 2e-1


def test_batched_predict_equals_per_episode_predict():
    """predict() on b > 1 episodes (per-episode fast weights, one batched pass) == the reference-shaped b = 1 calls."""
    from interactron_amd import Config, build_model
    data = to_gpu(synthetic_episodes(3, height=128, width=160, tag="predict-batch"))
    m = build_model(Config(**dict(MODEL_CFG, TYPE="interactron", EPISODE_CHUNK=2)))
    load_procedural(m.fusion, "fusion.")
    m = m.cuda().eval()
    together = m.predict(data)
    for i in range(3):
        one = m.predict({"frames": data["frames"][i:i + 1], "masks": data["masks"][i:i + 1]})
        for k in ("pred_logits", "pred_boxes"):
            assert together[k].shape[1:] == one[k].shape[1:]
            torch.testing.assert_close(together[k][i:i + 1], one[k], atol=1e-3 * float(one[k].abs().max()) + 1e-4, rtol=1e-3)


def test_policy_step_graph_replay_equals_eager():
    """get_next_action replayed from a captured HIP graph (eval mode) picks the same action as eager launches."""
    from interactron_amd import Config, build_model
    data = to_gpu(synthetic_episodes(2, height=128, width=160, tag="policy-graph"))
    m = build_model(Config(**dict(MODEL_CFG, TYPE="interactron")))
    load_procedural(m.fusion, "fusion.")
    m = m.cuda().eval()
    picks = {}
    for use in (True, False, True):
        m.config.POLICY_GRAPH = use
        picks.setdefault(use, []).append([m.get_next_action({"frames": data["frames"][i:i + 1, :s], "masks": data["masks"][i:i + 1, :s]})
                                           for i in range(2) for s in (1, 2, 3, 4)])
    assert m._graphs is not None and len(m._graphs) == 4, "the policy step was not captured"
    assert picks[True][0] == picks[False][0] == picks[True][1]


@pytest.mark.usefixtures("kernel_form")
def test_config1_detr(golden, episode1):
    O = golden("golden_configs.pt")
    m = make("detr")
    pred = m.predict(episode1)
    for k, rec in O["detr_predict"].items():
        check_record(rec, pred[k], atol=rec_tol(rec), rtol=1e-3, what="detr/" + k)
    m.zero_grad()
    with ReferenceMatching(golden("golden_indices.pt")["detr_forward"]):
        _, losses = m(episode1)
    for k, v in O["detr_forward"]["losses"].items():
        assert abs(float(losses[k]) - float(v)) <= 2e-3 * max(abs(float(v)), 1.0), k
    F64 = golden("golden_train_f64.pt")["configs"]["detr_forward"]["grads"]
    for k, p in m.model.named_parameters():
        check_grad(O["detr_forward"]["grads"][k], p.grad, rel=1e-3, what="detr/" + k, norm64=F64.get(k))   # first order: 1e-3


@pytest.mark.usefixtures("kernel_form")
def test_config2_multiframe(golden, episode1):
    O = golden("golden_configs.pt")
    m = make("detr_multiframe")
    pred = m.predict(episode1)
    for k, rec in O["multiframe_predict"].items():
        check_record(rec, pred[k], atol=rec_tol(rec), rtol=1e-3, what="mf/" + k)
    m.zero_grad()
    with ReferenceMatching(golden("golden_indices.pt")["multiframe_forward"]):
        preds, losses = m(episode1)
    for k, rec in O["multiframe_forward"]["preds"].items():
        check_record(rec, preds[k], atol=rec_tol(rec), rtol=1e-3, what="mf/" + k)
    for k, v in O["multiframe_forward"]["losses"].items():
        assert abs(float(losses[k]) - float(v)) <= 2e-3 * max(abs(float(v)), 1.0), k
    F64 = golden("golden_train_f64.pt")["configs"]["multiframe_forward"]
    for k, p in m.detector.named_parameters():
        check_grad(O["multiframe_forward"]["detector_grads"][k], p.grad, rel=1e-3, what="mf/detector." + k,   # first order: 1e-3
                   norm64=F64["detector_grads"].get(k))
    for k, p in m.fusion.named_parameters():
        check_grad(O["multiframe_forward"]["fusion_grads"][k], p.grad, rel=1e-3, what="mf/fusion." + k,
                   norm64=F64["fusion_grads"].get(k))


def _config2_16_bit(golden, episode1, dtype, grad_tol, logit_tol=3e-2, tie_tol=1e-4, box_tol=5e-3):
    """multi_frame_baseline in a 16-bit mode against the reference's fp32 recording at SURVEY 8d's bf16 tolerances: logits within 3e-2
    (`logit_tol`), boxes within 5e-3 (absolute), losses 2 %, the Hungarian assignments the reference's up to proven ties
    (ReferenceMatching: equal optimum to 1e-4 under this path's own cost matrix), gradient norms of the trained networks within `grad_tol`."""
    from interactron_amd import hipops
    O = golden("golden_configs.pt")
    m = make("detr_multiframe", COMPUTE_DTYPE=dtype)
    assert m.compute_dtype == hipops.normalize_compute_dtype(dtype) and hipops.COMPUTE_DTYPE == "f32"   # (the mode is the model's, in force inside its calls only)
    pred = m.predict(episode1)
    assert hipops.COMPUTE_DTYPE == "f32"
    errs = {}
    for k, tol in (("pred_logits", logit_tol), ("pred_boxes", box_tol)):
        rec = O["multiframe_predict"][k]
        got = pred[k].detach().float().cpu()
        if "full" in rec:
            err = float((got - rec["full"]).abs().max())
        else:
            err = float((got.reshape(-1)[rec["idx"]] - rec["sample"]).abs().max())
        print("%s, predict %s: max abs error %.2e (bound %.1e)" % (dtype, k, err, tol))
        errs[k] = (err, tol)
    for k, (err, tol) in errs.items():
        assert err <= tol, (k, err)
    m.zero_grad()
    # (every difference from the reference's assignment is still PROVEN a tie -- equal optimum to 1e-4 under this path's own
    #  cost matrix; 16-bit noise just decides more of the RNG-free weights' ties the other way: no cap on their number)
    with ReferenceMatching(golden("golden_indices.pt")["multiframe_forward"], max_flip_share=1.0, tie_tol=tie_tol) as rm:
        preds, losses = m(episode1)
    print("%s: %d of %d images matched differently from the reference's recording (proven ties)" % (dtype, rm.flips, rm.calls))
    for k, v in O["multiframe_forward"]["losses"].items():
        assert abs(float(losses[k]) - float(v)) <= 2e-2 * max(abs(float(v)), 1.0), (k, float(losses[k]), float(v))
    worst = (0.0, "")
    for grp, mod in (("detector", m.detector), ("fusion", m.fusion)):
        for k, p in mod.named_parameters():
            rec = O["multiframe_forward"][grp + "_grads"][k]
            if rec is None:
                assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
                continue
            assert p.grad is not None and p.grad.dtype == torch.float32, k   # parameter gradients stay fp32 in every mode
            n = float(p.grad.double().norm())
            if rec["norm"] < 1e-6:
                # a mathematically zero gradient (attention key bias: softmax is shift invariant): rounding noise in every arithmetic --
                # 1e-8 in fp32, up to 1e-3 of the neighbouring gradients with 16-bit activations; it must stay noise
                assert n <= 1e-2, (k, n)
                continue
            rel = abs(n - rec["norm"]) / rec["norm"]
            worst = max(worst, (rel, grp + "." + k))
    print("%s, worst gradient-norm deviation %.2e on %s" % ((dtype,) + worst))
    assert worst[0] <= grad_tol, worst
    return m


def test_config2_multiframe_single_pass_16_bit(golden, episode1):
    """MODEL.COMPUTE_DTYPE single_pass (round 4's 16-bit mode): fp32 storage, contractions round each operand once to 16 bits (the h
    plane of the fp16x3 form).  Reference arithmetic: models/gpt.py:39-57, models/detr_multiframe.py:55-109."""
    _config2_16_bit(golden, episode1, "single_pass", 5e-2)


def test_config2_multiframe_bf16_activations(golden, episode1):
    """BASELINE.json configs[1] (multi_frame_baseline ... bf16): MODEL.COMPUTE_DTYPE bf16 -- activations live in HBM as bf16, the
    contractions run on csrc/gemm16.hip, parameters / gradients / statistics stay fp32 (b16.py).  Reference arithmetic: models/gpt.py:
    39-78, models/detr_models/transformer.py:148-232, backbone.py:88-90, models/detr_multiframe.py:55-109.  And the mode is the
    model's: an fp32 model built AFTER the bf16 one still computes fp32-grade (its predict meets the fp32 tolerances), with the bf16
    model alive and used in between."""
    from interactron_amd import b16
    before = dict(b16._stats)
    # logits: SURVEY 8d proposed 3e-2 "to be finalised after first measurements".  Measured: 3.35e-2 on the worst sampled logit (|logit| up
    # to ~10) -- 8 mantissa bits in the matrix operands alone cost 8 x the 11-bit single-pass mode's 3.0e-3; finalised at 4e-2.
    # assignments: the bf16 cost matrix carries ~1e-3 of noise, so "equal optimum" is judged at 2e-3 (measured: 5.5e-4 on one image)
    # boxes: measured 4.5e-3 ... 5.0e-3 depending on which ops run natively (another rounding order): finalised at 6e-3.
    m16 = _config2_16_bit(golden, episode1, "bf16", 1.5e-1, logit_tol=4e-2, tie_tol=2e-3, box_tol=6e-3)
    assert b16._stats["native_gemms"] > before["native_gemms"] + 100, "the bf16 GEMM was not the one that ran"
    O = golden("golden_configs.pt")
    m32 = make("detr_multiframe")
    m16.predict(episode1)
    pred = m32.predict(episode1)
    for k, rec in O["multiframe_predict"].items():
        check_record(rec, pred[k], atol=rec_tol(rec), rtol=1e-3, what="f32 after bf16/" + k)


def test_interactron_step_in_the_16_bit_mode_against_the_oracle():
    """configs/interactron.yaml's meta-train step with MODEL.COMPUTE_DTYPE bf16 (one episode at 128 x 128, the smoke step): learned-loss
    gradient with create_graph, clipped SGD on fp32 fast weights, second-order backward -- through bf16 activations, the bf16 GEMM /
    convolution gathers, the single-term attention passes and, where an op has no 16-bit second-order kernel, its fp32 kernel between
    conversion passes (b16.py).  Against the fp32 CPU oracle (reference models/interactron.py:61-151), assignments pinned to the
    oracle's.  Two runs, because they answer different questions (measured r6j / r6k, profiles/r6k_16_bit_step_survey.txt):

    * ADAPTIVE_LR = 0 (the detector is not adapted; every kernel of the step still runs, second-order pass included): the arithmetic of
      the mode.  Losses within 1 % (measured 0.26 %), the WHOLE gradient (all tensors as one vector) within cosine >= 0.99 of the
      oracle's (measured 0.9953), every tensor >= 0.9 (0.941 on detector.query_embed.weight), norms within 15 % (6.1 %).
    * ADAPTIVE_LR = 1e-3 (the configuration's value): the clipped step moves each adapted weight by up to 0.01 -- half the scale of the
      procedural weights -- along the inner gradient, so the few-% noise of a gradient taken through 16-bit activations becomes a
      %-level perturbation of all 41 M adapted weights before the second pass starts.  Losses within 8 % (measured 5.3 %), whole-gradient
      cosine >= 0.9 (measured 0.935, and 0.9356 with every op of the mode computed by its fp32 kernel between bf16 stores, IX_B16_TWINS=0:
      it is the storage format, not a kernel); single tensors are NOT bounded (measured worst 0.34-0.68 on a layer3 convolution).  The
      fp32 mode is the one that carries this step's parity claim (test_config3_*, the smoke step); this run states what the 16-bit
      mode does to it."""
    import __graft_entry__ as entry
    from interactron_amd import b16
    before = b16._stats["native_gemms"]
    res = entry.smoke_check(128, cfg_extra={"COMPUTE_DTYPE": "bf16", "ADAPTIVE_LR": 0.0}, f64_slack=False, norm_tol=1.5e-1, loss_tol=1e-2,
                            cos_min=0.9, pin_matching="always", zero_grad_noise=1e-2)
    print("interactron step, 16-bit mode, no adaptation: whole-gradient cosine %.5f, worst tensor %.4f on %s, loss deviations %s"
          % ((res["whole_gradient_cosine"],) + res["worst_cosine"] + ({k: round(v, 4) for k, v in res["loss_deviations"].items()},)))
    assert res["whole_gradient_cosine"] >= 0.99, res["whole_gradient_cosine"]
    assert res["checked"] >= 300
    assert b16._stats["native_gemms"] > before + 500, "the 16-bit kernels did not run"
    res = entry.smoke_check(128, cfg_extra={"COMPUTE_DTYPE": "bf16"}, f64_slack=False, norm_tol=10.0, loss_tol=8e-2, cos_min=-1.0,
                            pin_matching="always", zero_grad_noise=1e-2)
    print("interactron step, 16-bit mode, ADAPTIVE_LR 1e-3: whole-gradient cosine %.5f, worst tensor %.4f on %s, loss deviations %s"
          % ((res["whole_gradient_cosine"],) + res["worst_cosine"] + ({k: round(v, 4) for k, v in res["loss_deviations"].items()},)))
    assert res["whole_gradient_cosine"] >= 0.9, res["whole_gradient_cosine"]


def test_interactron_step_with_the_fusion_transformer_in_the_16_bit_mode():
    """MODEL.COMPUTE_DTYPE bf16_fusion: the detector (backbone, encoder / decoder, heads: the weights the inner step adapts) computes
    fp32-grade, the GPT fusion transformer -- where the 800 x 800 step spends its time: T = 12 755 attention -- runs on bf16 activations.
    The same smoke step as above against the fp32 oracle, ADAPTIVE_LR = 1e-3, assignments pinned: losses within 1 % (measured 0.35 %),
    whole-gradient cosine >= 0.97 (measured 0.979; everything in bf16: 0.935), norms within 30 % (measured 23 %; the one-element bias of the loss
    head 48 %), every tensor >= 0.85 (measured 0.908, on the layer3
    convolution that falls to 0.34-0.68 with everything in bf16)."""
    import __graft_entry__ as entry
    from interactron_amd import b16
    before = b16._stats["native_gemms"]
    res = entry.smoke_check(128, cfg_extra={"COMPUTE_DTYPE": "bf16_fusion"}, f64_slack=False, norm_tol=3e-1, loss_tol=1e-2, cos_min=0.85,
                            pin_matching="always", zero_grad_noise=1e-2, scalar_tol=7e-1)   # (the one-element bias of the loss head: 48 % measured)
    print("interactron step, bf16 fusion transformer: whole-gradient cosine %.5f, worst tensor %.4f on %s, loss deviations %s"
          % ((res["whole_gradient_cosine"],) + res["worst_cosine"] + ({k: round(v, 4) for k, v in res["loss_deviations"].items()},)))
    assert res["whole_gradient_cosine"] >= 0.97, res["whole_gradient_cosine"]
    assert res["checked"] >= 300
    assert b16._stats["native_gemms"] > before + 100, "the 16-bit kernels did not run"
    # ... and without the adaptation (every kernel of the step still runs): the arithmetic of the mode by itself -- whole-gradient cosine
    # 0.99992, worst tensor 0.9959, norms within 5 % (measured 2.8 %; r6ab / r6ac; the same with every op on its fp32 kernel between bf16 stores, and with three-term attention:
    # what the 1e-3 step loses is the amplification of storage rounding by the clipped update, not a kernel)
    res = entry.smoke_check(128, cfg_extra={"COMPUTE_DTYPE": "bf16_fusion", "ADAPTIVE_LR": 0.0}, f64_slack=False, norm_tol=5e-2, loss_tol=5e-3,
                            cos_min=0.99, pin_matching="always", zero_grad_noise=1e-2, scalar_tol=5e-1)
    print("interactron step, bf16 fusion transformer, no adaptation: whole-gradient cosine %.5f, worst tensor %.4f on %s"
          % ((res["whole_gradient_cosine"],) + res["worst_cosine"]))
    assert res["whole_gradient_cosine"] >= 0.9995, res["whole_gradient_cosine"]


@pytest.mark.usefixtures("kernel_form")
def test_config3_interactron_random(golden, episode1):
    O = golden("golden_configs.pt")
    m = make("interactron_random")
    pred = m.predict(episode1)
    for k, rec in O["random_predict"].items():
        check_record(rec, pred[k], atol=rec_tol(rec), rtol=1e-3, what="rand/" + k)
    m.zero_grad()
    random.seed(7)
    with ReferenceMatching(golden("golden_indices.pt")["random_forward"]):
        preds, losses = m(episode1)
    for k, v in O["random_forward"]["losses"].items():
        assert abs(float(losses[k]) - float(v)) <= 2e-3 * max(abs(float(v)), 1.0), k
    F64 = golden("golden_train_f64.pt")["configs"]["random_forward"]
    for k, p in m.detector.named_parameters():
        check_grad(O["random_forward"]["detector_grads"][k], p.grad, rel=second_order_rel("rand/detector." + k), what="rand/detector." + k,
                   norm64=F64["detector_grads"].get(k))
    for k, p in m.fusion.named_parameters():
        check_grad(O["random_forward"]["fusion_grads"][k], p.grad, rel=second_order_rel("rand/fusion." + k), what="rand/fusion." + k,
                   norm64=F64["fusion_grads"].get(k))


@pytest.mark.usefixtures("kernel_form")
def test_oracle_parity_fresh_inputs_small_resolution():
    """HIP vs CPU oracle on inputs no fixture covers (different seed, 160x128 frames, padded masks)."""
    from interactron_amd import NestedTensor
    from interactron_amd.synthetic import procedural_state_dict
    from oracle import detector as od
    m = make("detr")
    data = synthetic_episodes(1, frames=3, height=160, width=128, tag="fresh")
    masks = torch.zeros(3, 160, 128, dtype=torch.long)
    masks[1, 120:, :] = 1
    masks[2, :, 100:] = 1
    with torch.no_grad():
        out = m.model(NestedTensor(data["frames"][0].cuda(), masks.cuda()))
    det = {k[len("detector."):]: v for k, v in
           procedural_state_dict({"detector." + k: v for k, v in od.detr_state_shapes().items()}).items()}
    with torch.no_grad():
        ref = od.detr_forward(det, data["frames"][0], masks)
    for k, v in ref.items():
        tol = 1e-3 * float(v.abs().max()) + 1e-4
        torch.testing.assert_close(out[k].cpu(), v, atol=tol, rtol=1e-3, msg=lambda s: k + ": " + s)


def test_inner_steps_2_against_the_oracle():
    """MODEL.INNER_STEPS = 2 (BASELINE.json's multi-step adapt loop): two learned-loss SGD steps with the
    second-order graph through both (reference step: models/interactron.py:94-102, utils/meta_utils.py:135-142), one episode
    at 128 x 128.  The episode-batched schedule against the CPU oracle: every loss (2e-3) and every gradient tensor of both
    networks -- None-pattern, direction (cosine >= 0.999), norm within 1e-3 + 3 x the float32 oracle's own distance from its
    float64 run on that tensor (two clipped steps double the elements that sit on kinks: the float32 oracle itself is up to
    7.5e-3 off its float64 run at two steps, 2e-3 at one; one tensor needs 2e-3: __graft_entry__.SMOKE_REL).  The reference's sequential schedule (EPISODE_CHUNK 0) against the
    batched one.  And the second step must matter: the supervisor losses differ from the single-step run's."""
    import __graft_entry__ as entry
    two = entry.smoke_check(128, inner_steps=2, episodes=1, chunk=16, f64_slack=True)
    assert two["checked"] >= 300
    seq = entry.smoke_check(128, inner_steps=2, episodes=1, chunk=0, oracle=False)
    assert seq["norms"].keys() == two["norms"].keys()
    for k, v in two["losses"].items():
        assert abs(seq["losses"][k] - v) <= 2e-3 * max(abs(v), 1.0), (k, seq["losses"][k], v)
    for k, v in two["norms"].items():
        if max(v, seq["norms"][k]) < 1e-6:
            continue   # mathematically zero gradient (attention key bias): rounding noise on both sides
        assert abs(seq["norms"][k] - v) <= (5e-2 if k.endswith("loss_decoder.layers.2.bias") else 5e-3) * max(v, 1e-6), (k, seq["norms"][k], v)
    one = entry.smoke_check(128, inner_steps=1, episodes=1, chunk=16, oracle=False)
    diff = max(abs(two["losses"][k] - one["losses"][k]) / max(abs(one["losses"][k]), 1e-6) for k in one["losses"] if "supervisor" in k)
    assert diff > 1e-5, "INNER_STEPS = 2 reproduced the single-step losses: the second step did nothing"


# ---- train mode (dropout ON) against the oracle, with the HIP path's own dropout masks handed to the oracle --------------------
def _hip_elementwise_keep(seed, n, p):
    """The counter-hash mask of csrc/elementwise.hip (dropout_kernel / relu_dropout_kernel / add_dropout_kernel: mix32 of
    seed ^ index * 0xD6E8FEB86659FD93, keep <=> draw >= p * 2^32), restated in numpy uint64 arithmetic: keep / (1 - p)."""
    import numpy as np
    M = np.uint64(0xFFFFFFFFFFFFFFFF)
    with np.errstate(over="ignore"):
        k = np.arange(n, dtype=np.uint64)
        z = (np.uint64(seed) ^ (k * np.uint64(0xD6E8FEB86659FD93))) & M
        z = (z + np.uint64(0x9E3779B97F4A7C15)) & M
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & M
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & M
        r = (z ^ (z >> np.uint64(31))) >> np.uint64(32)
    thresh = np.uint64(int(float(np.float32(p)) * 4294967296.0))
    scale = np.float32(1.0) / (np.float32(1.0) - np.float32(p))
    return torch.from_numpy((r >= thresh).astype(np.float32) * scale)


class _RecordDropoutSites:
    """Records, in call order, every dropout-carrying op the HIP path issues: (kind, shape of the masked tensor, p, seed)."""

    def __enter__(self):
        from interactron_amd import hipops as ops
        self.ops, self.sites, self._orig = ops, [], {}
        seeds = []
        self._orig["_next_seed"] = ops._next_seed
        ops._next_seed = lambda: (seeds.append(self._orig["_next_seed"]()), seeds[-1])[1]

        def wrap(name, shape_of, p_of):
            orig = getattr(ops, name)
            self._orig[name] = orig

            def f(*a):
                n0 = len(seeds)
                out = orig(*a)
                if len(seeds) > n0:
                    assert len(seeds) == n0 + 1
                    self.sites.append((name, tuple(shape_of(a)), float(p_of(a)), seeds[-1]))
                return out
            setattr(ops, name, f)
        wrap("dropout", lambda a: a[0].shape, lambda a: a[1])
        wrap("add_dropout", lambda a: a[1].shape, lambda a: a[2])
        wrap("relu_dropout", lambda a: a[0].shape, lambda a: a[1])
        wrap("attention", lambda a: (a[3] * a[4], a[5], a[6]), lambda a: a[16])   # (nbatch * heads, L, S), p
        return self

    def __exit__(self, *exc):
        for k, v in self._orig.items():
            setattr(self.ops, k, v)

    def masks(self):
        out = []
        for kind, shape, p, seed in self.sites:
            if kind == "attention":
                out.append(self.ops.flash_dropmask(shape[0], shape[1], shape[2], p, seed).cpu().float())
            else:
                n = 1
                for d in shape:
                    n *= d
                out.append(_hip_elementwise_keep(seed, n, p).reshape(shape))
        return out


def test_train_mode_step_with_the_kernels_own_dropout_masks_against_the_oracle():
    """The step bench.py times runs in TRAIN mode (dropout 0.1 in every transformer); model-level parity was eval-mode only
    because two implementations draw different masks.  Here the HIP path's masks -- pure functions of (seed, element index),
    recorded site by site in call order (78 + 13 sites: three detector passes, the fusion) -- are materialised and handed to
    the CPU oracle's F.dropout calls in the same order: one full meta-train step (reference models/interactron.py:61-151 with
    detector.train() / fusion.train(), :153-161) on one episode at 128 x 160, losses within 2e-3, every gradient tensor's norm
    within 5e-3 (at most two kink-bound backbone tensors within 1.5e-2) and direction (cosine) within 1e-3 of the oracle's."""
    import torch.nn.functional as F
    from interactron_amd import Config, build_model, hipops
    from interactron_amd.synthetic import procedural_state_dict
    from oracle import detector as od, episode as oe, fusion as of
    cfg = dict(MODEL_CFG, TYPE="interactron", BLOCK_SIZE=5 * (8 * 10 + 50) + 5, EPISODE_CHUNK=1, STEP_GRAPH="off")
    m = build_model(Config(**cfg))
    load_procedural(m.fusion, "fusion.")
    m = m.cuda().train()
    data = synthetic_episodes(1, height=128, width=160, tag="trainmode")
    random.seed(11)
    hipops.manual_seed(4321)
    m.zero_grad()
    with _RecordDropoutSites() as rec:
        _, losses = m(to_gpu(data))
    torch.cuda.synchronize()
    masks = rec.masks()
    kinds = [s[0] for s in rec.sites]
    assert kinds.count("attention") == 3 * 18 + 4 and len(kinds) == 3 * (18 + 42) + 13, (len(kinds), kinds.count("attention"))

    det = {k[len("detector."):]: v for k, v in
           procedural_state_dict({"detector." + k: v for k, v in od.detr_state_shapes().items()}).items()}
    ocfg = {k: v for k, v in cfg.items() if k not in ("TYPE", "EPISODE_CHUNK", "STEP_GRAPH", "WEIGHTS")}
    fus = {k[len("fusion."):]: v for k, v in
           procedural_state_dict({"fusion." + k: v for k, v in of.fusion_state_shapes(ocfg, "gpt").items()}).items()}
    queue = list(masks)
    real_dropout = F.dropout

    def fed_dropout(x, p=0.5, training=True, inplace=False):
        if not training or p <= 0.0:
            return x
        mk = queue.pop(0)
        if tuple(mk.shape) != tuple(x.shape):
            if mk.dim() == 3 and x.dim() == 3 and (mk.shape[1], mk.shape[0], mk.shape[2]) == tuple(x.shape):
                mk = mk.permute(1, 0, 2)            # [frames, tokens, d] here, [tokens, frames, d] in the reference's DETR
            else:
                assert mk.numel() == x.numel(), ("dropout site order differs", tuple(mk.shape), tuple(x.shape))
                mk = mk.reshape(x.shape)            # [batch * heads, L, S] -> [batch, heads, L, S]
        return x * mk.to(x.dtype)
    F.dropout = fed_dropout
    try:
        random.seed(11)
        _, ref_losses, ref_grads = oe.interactron_forward(det, fus, data, ocfg, {}, "gpt", training=True)
    finally:
        F.dropout = real_dropout
    assert not queue, "%d dropout sites of the HIP path were never reached by the oracle" % len(queue)
    for k, v in ref_losses.items():
        assert abs(float(losses[k]) - float(v)) <= 2e-3 * max(abs(float(v)), 1.0), (k, float(losses[k]), float(v))
    devs = []
    for group, module in (("fusion", m.fusion), ("detector", m.detector)):
        for name, p_ in module.named_parameters():
            ref = ref_grads[group].get(name)
            if ref is None:
                assert p_.grad is None or float(p_.grad.abs().max()) == 0.0, name
                continue
            g = p_.grad.detach().cpu()
            if g.dim() == 4 and tuple(g.shape) != tuple(ref.shape):
                g = g.permute(0, 3, 1, 2)
            rn, gn = float(ref.norm()), float(g.norm())
            if max(rn, gn) < 1e-6:
                continue
            cos = float((g.double().reshape(-1) @ ref.double().reshape(-1)) / (gn * rn)) if ref.numel() > 1 else 1.0
            devs.append((abs(gn - rn) / rn, 1.0 - cos, ref.numel(), group + "." + name))
    devs.sort(reverse=True)
    print("train-mode step with shared dropout masks: %d sites, %d gradient tensors; worst norm deviations %s; worst 1 - cosine %.2e"
          % (len(kinds), len(devs), [(round(d[0], 5), d[3]) for d in devs[:4]], max(d[1] for d in devs)))
    # every tensor within 5e-3 of the oracle's norm (one-element tensors 5e-2) -- except that a few second-order backbone weights
    # sit on ReLU / clip kinks of the inner step (the eval-mode G13 test shows the same tensors, and the reference's own float32
    # is 0.7 % off float64 there): at most two of them, within 1.5e-2 (measured: one, 8.7e-3); directions (1 - cosine) within
    # 1e-3 everywhere (measured 3.4e-4)
    loose = [d for d in devs if d[0] > (5e-2 if d[2] == 1 else 5e-3)]
    assert len(loose) <= 2 and all("backbone" in d[3] and d[0] <= 1.5e-2 for d in loose), loose
    assert max(d[1] for d in devs) <= 1e-3, sorted(devs, key=lambda d: -d[1])[:3]


def test_missing_library_fails_loudly(monkeypatch):
    from interactron_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libinteractron_hip.so")
    with pytest.raises(_lib.HipLibraryError):
        _lib.load()


def test_meta_train_step_with_the_stride2_data_gradients_in_class_form():
    """The smoke step (one episode at 128 x 128 against the CPU oracle: losses 2e-3, every gradient tensor's direction and norm)
    with csrc/gemm.hip conv_bwd_data_s2 FORCED (ix_conv_set_s2_split(2)): at this size the library would keep the one-launch
    gather, at the bench sizes it takes the per-parity-class form for layer2.0 / layer3.0 conv2 and the layer3.0 downsample --
    first- and second-order backward."""
    import __graft_entry__ as entry
    from interactron_amd import hipops
    lib = hipops._L()
    try:
        lib.ix_conv_set_s2_split(2)
        hipops._conv_ws.clear()
        res = entry.smoke_check(128)
    finally:
        lib.ix_conv_set_s2_split(1)
        hipops._conv_ws.clear()
    assert res["checked"] >= 300
