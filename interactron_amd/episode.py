"""The four per-episode models behind the reference's Python surface, computing on the HIP kernels.

    interactron         reference models/interactron.py:14-197        (config interactron.yaml)
    interactron_random  reference models/interactron_random.py:11-153 (config interactron_random.yaml)
    detr_multiframe     reference models/detr_multiframe.py:9-131     (config multi_frame_baseline.yaml)
    detr                reference models/detr.py:8-84                 (config single_frame_baseline.yaml)

Contract: ``forward(data) -> (predictions, losses)`` with gradients already accumulated into
``.grad`` of ``detector`` / ``fusion`` parameters; ``predict(data) -> dict``; ``get_next_action(data) -> int``;
``train()/eval()``; ``set_logger``; attributes ``detector``, ``fusion``, ``criterion``, ``path_storage``, ``config``.

Differences that do not change results: the adapted copy of theta shares storage with the parameters instead of
being cloned twice, and the meta-step's second-order backward is asked only for the tensors whose ``.grad`` the
reference keeps (fusion parameters and the never-adapted ``in_proj_*``), so the Hessian-vector product w.r.t. theta
that the reference computes and throws away is never formed.
"""
import os
import random
import time

import torch
from torch import nn

from . import hipops as ops
from .criterion import HungarianMatcher, SetCriterion
from .detector import NestedTensor, build_detector
from .fusion import DecoderTransformer, Transformer
from .meta import get_parameters, set_parameters, sgd_step
from .storage import PathStorage, best_path_labels, exchange_rewards
from .synthetic import load_procedural


def build(config):
    """reference detr.py:314-341 -> (detector, criterion, postprocessors)."""
    detector = build_detector(config.NUM_CLASSES, num_queries=int(getattr(config, "NUM_QUERIES", 50)))
    matcher = HungarianMatcher(config.SET_COST_CLASS, config.SET_COST_BBOX, config.SET_COST_GIOU)
    criterion = SetCriterion(config.NUM_CLASSES, matcher=matcher,
                             weight_dict={"loss_ce": 1, "loss_bbox": 5, "loss_giou": 2}, eos_coef=0.1,
                             losses=["labels", "boxes", "cardinality"])
    return detector, criterion, {}


def _load_detector_weights(detector, config):
    w = getattr(config, "WEIGHTS", None)
    if w in (None, "", "procedural", "__procedural__"):
        import warnings
        warnings.warn("MODEL.WEIGHTS is %r: the detector starts from RNG-free SYNTHETIC weights (bench / tests only) -- "
                      "metrics and checkpoints of this run say nothing about the reference's models" % (w,), stacklevel=2)
        load_procedural(detector, "detector.")
        return
    if not os.path.exists(str(w)):
        raise FileNotFoundError("MODEL.WEIGHTS %r not found (use WEIGHTS: \"procedural\" for RNG-free synthetic weights)" % w)
    detector.load_state_dict(torch.load(w, map_location="cpu")["model"])


def _labels(data, b):
    return [{"labels": data["category_ids"][b][j], "boxes": data["boxes"][b][j]}
            for j in range(len(data["category_ids"][b]))]


def _lift(out):
    o = dict(out)
    for k in ("embedded_memory_features", "box_features", "pred_logits", "pred_boxes"):
        o[k] = o[k].unsqueeze(0)
    return o


def _weighted(l):
    return l["loss_ce"] + 5 * l["loss_giou"] + 2 * l["loss_bbox"]


_const = {}


def _loss_weights(E, device):
    """[E, 5] constant that turns grouped criterion rows (loss_ce, class_error, loss_bbox, loss_giou, cardinality_error) into
    sum_e loss_ce + 5 loss_giou + 2 loss_bbox (reference interactron.py:119: the weights of `_weighted`)"""
    key = ("w", E, device.index)
    if key not in _const:
        _const[key] = torch.tensor([1.0, 0.0, 2.0, 5.0, 0.0], device=device).repeat(E, 1).contiguous()
    return _const[key]


def _ones(n, device):
    key = ("1", n, device.index)
    if key not in _const:
        _const[key] = torch.ones(n, device=device)
    return _const[key]


def _weighted_rows(rows):
    return rows[:, 0] + 5 * rows[:, 3] + 2 * rows[:, 2]


def _ldn(targets):
    """column pitch of the cost matrices of these images: the largest target count, rounded up to 8"""
    return (max([int(t["labels"].shape[0]) for t in targets] + [1]) + 7) // 8 * 8


def _named_means(criterion, row, tag):
    return {k.replace("loss", tag): v for k, v in criterion.as_dict(row).items()}


def _mean_losses(per_task, tag):
    return {k.replace("loss", tag): torch.mean(torch.stack([x[k] for x in per_task])) for k in per_task[0]}


def _in_compute_mode(fn):
    import functools

    @functools.wraps(fn)
    def run(self, *args, **kwargs):
        with ops.compute_mode(self.compute_dtype):
            return fn(self, *args, **kwargs)
    run._ix_mode_wrapped = True
    return run


class _EpisodeModel(nn.Module):
    """``compute_dtype`` (MODEL.COMPUTE_DTYPE, set by config.build_model): the arithmetic mode of THIS model -- "f32" (fp32-grade, the
    parity path), "bf16" (16-bit activations, b16.py) or "single_pass".  Every entry point of a subclass runs inside
    ``hipops.compute_mode(self.compute_dtype)``: two live models with different modes do not disturb each other."""
    compute_dtype = "f32"
    _ENTRY_POINTS = ("forward", "predict", "get_next_action", "_policy_logits", "dp_idle_step")

    def __init_subclass__(cls, **kwargs):
        super().__init_subclass__(**kwargs)
        for name in _EpisodeModel._ENTRY_POINTS:
            fn = cls.__dict__.get(name)
            if fn is not None and not getattr(fn, "_ix_mode_wrapped", False):
                setattr(cls, name, _in_compute_mode(fn))

    def __init__(self):
        super().__init__()
        self.logger = None
        self.mode = "train"

    def eval(self):
        return self.train(False)

    def set_logger(self, logger):
        assert self.logger is None, "This model already has a logger!"
        self.logger = logger


class _PhaseTimer:
    """Optional per-phase clock of one meta-train step, a diagnostic that is never on in timed runs.
    ``model.phase_times = {}``: synchronises at every phase boundary and accumulates wall-clock per phase.
    ``model.phase_times = []``: no synchronisation -- appends (name, host time, HIP event) per boundary, so the lag of
    the GPU behind the host (is the GPU ever starved?) can be read off afterwards (tools/phase_lag.py)."""

    def __init__(self, sink):
        self.sink = sink
        if isinstance(sink, dict):
            torch.cuda.synchronize()
            self.t = time.perf_counter()
        elif sink is not None:
            self.mark("start")

    def mark(self, name):
        if self.sink is None:
            return
        if isinstance(self.sink, list):
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self.sink.append((name, time.perf_counter(), ev))
            return
        torch.cuda.synchronize()
        now = time.perf_counter()
        self.sink[name] = self.sink.get(name, 0.0) + (now - self.t) * 1e3
        self.t = now


class _Adaptive(_EpisodeModel):
    """Shared body of interactron / interactron_random: learned-loss inner step + meta-gradient."""

    phase_times = None
    _graphs = {}   # (replaced by a per-instance dict on first use; None = graph capture disabled after a failure)

    use_policy = False

    def _graph_stamp(self):
        """Addresses of everything a captured graph reads by pointer: all parameters (and their .grad), buffers of detector
        and fusion plus the folded frozen-BN affines.  A re-homed parameter (FlatBuffers, .to()), a dropped .grad or a
        rebuilt fold changes it."""
        ptrs = [t.data_ptr() for t in self.detector.parameters()] + [t.data_ptr() for t in self.detector.buffers()]
        ptrs += [t.data_ptr() for t in self.fusion.parameters()] + [t.data_ptr() for t in self.fusion.buffers()]
        ptrs += [0 if t.grad is None else t.grad.data_ptr() for t in self.parameters()]
        folds = [m._fold for m in self.detector.modules() if getattr(m, "_fold", None) is not None]
        ptrs += [t.data_ptr() for f in folds for t in f[1]]
        # ... and the kernel forms a capture bakes in (the tests switch them at run time: a graph must not replay the other form)
        forms = (ops.FLASH_TR, ops.ATTENTION_IMPL, ops.ATTENTION_DTYPE, ops.contraction_form())
        return hash((tuple(ptrs), forms)), folds

    def invalidate_graphs(self):
        """Drop every captured graph (weights were reloaded / moved); the next call captures afresh."""
        if self.__dict__.get("_graphs"):
            self._graphs = {}
        self.__dict__.pop("_chunk_graphs", None)
        self.__dict__.pop("_predict_graphs", None)
        ops.weights_changed()

    def load_state_dict(self, *args, **kwargs):
        # load_state_dict copies IN PLACE (addresses unchanged) but FrozenBatchNorm2d drops its folded scale/shift, which
        # a captured graph reads by address: replaying it afterwards would read freed memory
        self.invalidate_graphs()
        return super().load_state_dict(*args, **kwargs)

    def _apply(self, fn, *args, **kwargs):   # .to() / .cuda() / .float(): storage moves
        self.invalidate_graphs()
        return super()._apply(fn, *args, **kwargs)

    def _second_order_targets(self):
        """Leaves that keep a .grad from the supervisor backward: fusion parameters and the detector parameters that
        are real nn.Parameters during the episode (MultiheadAttention.in_proj_*)."""
        theta_ids = {id(p) for p in self._theta}
        det = [p for p in self.detector.parameters() if p.requires_grad and id(p) not in theta_ids]
        return [p for p in self.fusion.parameters() if p.requires_grad] + det

    def _real_parameters(self):
        """get_parameters(detector) at the start of an episode pass, while every module still holds its own nn.Parameters
        (set_parameters swaps the adapted ones for plain tensors afterwards); also records the identity of ALL of the
        model's Parameters for _inner_grad."""
        self.__dict__["_param_ids"] = frozenset(id(p) for p in self.parameters())
        return get_parameters(self.detector)

    def _inner_grad(self, learned, dtheta, create_graph):
        """d learned / d dtheta -- the MAML inner gradient.  Only the per-episode copies `dtheta` are asked for; the
        weight-gradient contractions of every nn.Parameter on the way (fusion Linears, in_proj blocks) are skipped
        (hipops.skip_param_grads; their first-order gradients come from the supervisor / detector backward passes)."""
        ids = self.__dict__["_param_ids"]
        assert not any(id(t) in ids for t in dtheta)
        with ops.skip_param_grads(ids):
            return torch.autograd.grad(learned, dtheta, create_graph=create_graph, retain_graph=create_graph,
                                       allow_unused=True)

    def _inner_steps(self):
        """MODEL.INNER_STEPS (default 1 = the reference, models/interactron.py:94-102): how often the learned-loss step
        theta <- theta - clip(lr d||fusion(detector(frames | theta)).loss|| / d theta) is repeated before the adapted detector
        is used -- BASELINE.json's "5-step adapt loop", SURVEY section 0 row 2.  The second-order graph runs through all of
        them; the fast weights are device tensors from the first step to the last (one fused clipped-SGD launch per step,
        csrc/meta.hip)."""
        k = int(getattr(self.config, "INNER_STEPS", 1))
        assert k >= 1, "MODEL.INNER_STEPS must be >= 1"
        return k

    def _adapt(self, img, mask, create_graph):
        """-> (dtheta, [gradients of every inner step], fusion output of the first pass, adapted weights)"""
        dtheta = cur = [p.detach().requires_grad_(True) for p in self._theta]
        first, steps = None, []
        nt = NestedTensor(img, mask)
        for _ in range(self._inner_steps()):
            set_parameters(self.detector, cur)
            pre = _lift(self.detector(nt))
            fusion_out = self.fusion(pre)
            first = first or fusion_out
            learned_loss = ops.l2_norm(fusion_out["loss"])
            grads = self._inner_grad(learned_loss, cur, create_graph)
            steps.append(grads)
            cur = sgd_step(cur, grads, self.config.ADAPTIVE_LR)
            if not create_graph:   # (eval: nothing differentiates through the step)
                cur = [t.detach().requires_grad_(True) for t in cur]
        return dtheta, steps, first, cur

    def predict(self, data):
        b, s, c, w, h = data["frames"].shape
        if b > 1:
            return self._predict_batched(data)
        from . import graphs
        img, mask = data["frames"].view(s, c, w, h), data["masks"].view(s, w, h)
        return graphs.predict_runner(self, img, mask)(img, mask)

    def _predict_one(self, img, mask):
        """reference interactron.py:31-59 for one episode: adapt on the s frames, detect frame 0 through theta'"""
        self._theta = self._real_parameters()
        try:
            with torch.enable_grad():
                _, _, _, fast = self._adapt(img, mask, create_graph=False)
                # stays in grad mode: set_parameters only swaps tensors that require grad (reference meta_utils.py:76)
                set_parameters(self.detector, fast)
            with torch.no_grad():
                post = self.detector(NestedTensor(img[0:1], mask[0:1]))
        finally:
            set_parameters(self.detector, self._theta)
        return {k: v.unsqueeze(0) for k, v in post.items()}

    def _predict_batched(self, data):
        """predict() for b > 1 episodes at once (the reference's own predict only works for b = 1, its evaluators call
        it episode by episode): the episodes are adapted together with per-episode fast weights, exactly as in
        ``forward``; returns the same keys with shape [b, 1, ...]."""
        b, s, c, w, h = data["frames"].shape
        chunk = max(1, int(getattr(self.config, "EPISODE_CHUNK", 16)))
        self._theta = theta = self._real_parameters()
        outs = []
        try:
            for e0 in range(0, b, chunk):
                E = min(chunk, b - e0)
                frames = data["frames"][e0:e0 + E].reshape(E * s, c, w, h)
                masks = data["masks"][e0:e0 + E].reshape(E * s, w, h)
                with torch.enable_grad():
                    cur = [t.requires_grad_(True) for t in ops.expand_episodes(E, [p.detach() for p in theta])]
                    nt = NestedTensor(frames, masks)
                    nt.stem = self.detector.backbone[0].body.frozen_stem(frames)
                    for _ in range(self._inner_steps()):
                        set_parameters(self.detector, cur)
                        pre = self.detector(nt)
                        pre = {k: v.reshape((E, s) + tuple(v.shape[1:])) for k, v in pre.items()}
                        loss_map = self.fusion(pre)["loss"].reshape(E, -1)
                        learned = ops.rownorm_sum(loss_map)   # = sum_e torch.norm(loss_e): one launch
                        grads = self._inner_grad(learned, cur, False)
                        cur = [t.detach().requires_grad_(True) for t in sgd_step(cur, grads, self.config.ADAPTIVE_LR)]
                    set_parameters(self.detector, cur)
                with torch.no_grad():
                    first = NestedTensor(frames[0::s], masks[0::s])
                    first.stem = nt.stem[0::s]
                    outs.append(self.detector(first))
                del cur, grads, pre, loss_map, learned
        finally:
            set_parameters(self.detector, theta)
        return {k: torch.cat([o[k] for o in outs], 0).unsqueeze(1) for k in outs[0]}

    # Episodes of a batch are independent given theta (reference interactron.py:84 loops over them one by one).  On
    # MI355X a single 5-frame episode leaves most of the 256 CUs idle (M = 250 decoder rows, 1805 encoder rows), so the
    # episodes of a chunk are run TOGETHER: every adapted tensor gets a leading episode dim ([E, ...] fast weights),
    # every Linear / conv / LayerNorm over them becomes ONE batched launch in which episode e's rows meet episode e's
    # weights, and theta's gradient is the sum over that dim -- the same numbers as the sequential loop, E x fewer
    # launches.  EPISODE_CHUNK (config key, default 16 = the reference BATCH_SIZE; 0 = the sequential reference loop) bounds E.
    #
    # A chunk is three sync-free SEGMENTS of launches with one host round trip (PathStorage, learned policy only):
    #     A  expand theta, detector(5 frames), fusion, learned-loss gradient (create_graph), clipped SGD, detector again,
    #        device matcher + criterion of the 5 frames and of frame 0 (the reward), reward -> pinned host memory
    #     C  the first-order branch (reference :126-134): detector on one random frame through theta', criterion, backward
    #     --  host: PathStorage bookkeeping on the rewards, policy labels up
    #     B  policy cross-entropy, supervisor total, second-order backward
    #     D  .grad += C's gradients (C and B share no state, so a replay runs them concurrently on two streams)
    # Issued eagerly (default for large chunks: the step is GPU-bound) or replayed from three captured HIP graphs
    # (graphs.ChunkGraphs; small chunks are bound by the host issuing ~6 000 launches, STEP_GRAPH: auto / true / false).
    def _seg_a(self, st):
        E, s, theta, lr = st.E, st.s, self._theta, self.config.ADAPTIVE_LR
        # theta_task = clone(theta); dtheta = detach(theta_task)   (reference :86-90), one copy per episode
        st.dtheta = cur = [t.requires_grad_(True) for t in ops.expand_episodes(E, [p.detach() for p in theta])]
        # the frozen stem (conv1..layer1) sees the same frames in all forwards: computed once per chunk
        st.nt = nt = NestedTensor(st.frames, st.masks)
        nt.stem = self.detector.backbone[0].body.frozen_stem(st.frames)
        st.grads, first_out = [], None
        for _ in range(self._inner_steps()):   # (MODEL.INNER_STEPS, default 1: see _inner_steps)
            set_parameters(self.detector, cur)
            pre = self.detector(nt)
            st.mark("1 detector fwd (theta)")
            pre = {k: v.reshape((E, s) + tuple(v.shape[1:])) for k, v in pre.items()}
            fusion_out = self.fusion(pre)
            first_out = first_out or fusion_out
            st.mark("2 fusion fwd")
            loss_map = fusion_out["loss"].reshape(E, -1)
            learned = ops.rownorm_sum(loss_map)   # = sum_e torch.norm(loss_e): one launch
            grads = self._inner_grad(learned, cur, True)
            st.grads.append(grads)
            st.mark("3 learned-loss grad (create_graph)")
            cur = sgd_step(cur, grads, lr)
        fusion_out = first_out
        set_parameters(self.detector, cur)
        post = self.detector(nt)
        st.mark("4 inner SGD + detector fwd (theta')")
        st.actions_out = fusion_out["actions"].reshape(E * 4, 4)
        # Matcher + criterion of the whole chunk on the device (criterion.py): one cost launch and one assignment launch
        # for all E * s images (matching is per image, so the assignments are exactly those of per-episode calls), then
        # the losses of every episode's 5 frames AND of its frame 0 alone (the policy reward, reference :104-108) from one
        # pass over the rows.
        post_lb = {k: post[k] for k in ("pred_logits", "pred_boxes")}
        toq = self.criterion.matcher.match(post_lb, st.tg)
        specs = ((s, s), (s, 1)) if self.use_policy else ((s, s),)
        rows = self.criterion.grouped(post_lb, st.tg, toq, specs, background_c=0.1)
        st.sup_rows = rows[0]
        if self.use_policy:   # frame-0 loss = the reward PathStorage ranks action sequences by: its D2H copy starts now
            st.gts = _weighted_rows(rows[1])
            st.gts_host.copy_(st.gts, non_blocking=True)
        st.mark("5 matcher + criterion")

    def _seg_c(self, st):
        # The first-order branch (reference interactron.py:126-134) depends only on the learned-loss gradient, not on the
        # criterion.  The expansion of theta is differentiable; its backward sums the per-episode gradients into theta.grad.
        E, theta = st.E, self._theta
        fast1 = ops.expand_episodes(E, theta)
        for grads in st.grads:   # the inner steps again with their gradients as constants, from the attached copy of theta
            fast1 = sgd_step(fast1, [None if g is None else g.detach() for g in grads], self.config.ADAPTIVE_LR)
        set_parameters(self.detector, fast1)
        nt1 = NestedTensor(st.frames[st.sel], st.masks[st.sel])
        nt1.stem = st.nt.stem[st.sel]
        post1 = self.detector(nt1)
        post1_lb = {k: post1[k] for k in ("pred_logits", "pred_boxes")}
        toq1 = self.criterion.matcher.match(post1_lb, st.tg1)
        (det_rows,) = self.criterion.grouped(post1_lb, st.tg1, toq1, ((1, 1),), background_c=0.1)
        st.det_rows = det_rows.detach()
        st.logits1, st.boxes1 = post1_lb["pred_logits"].detach(), post1_lb["pred_boxes"].detach()
        st.mark("6 first-order SGD + 1-frame fwd + criterion")
        # the gradients come back as tensors and are added into .grad by segment D: this segment then touches no state the
        # supervisor backward (segment B) touches, and a graph replay runs the two on different streams at the same time
        st.c_params = [p for p in list(theta) + self._in_proj if p.requires_grad]
        st.c_grads = torch.autograd.grad(ops.Dot.apply(det_rows, _loss_weights(E, st.frames.device)), st.c_params, allow_unused=True)
        st.mark("7 first-order backward")

    @staticmethod
    def _accumulate(params, grads):
        """.grad += grads in one multi-tensor launch (a parameter without a .grad gets the tensor)"""
        dst, src = [], []
        for p, g in zip(params, grads):
            if g is None:
                continue
            if p.grad is None:
                p.grad = g
            else:
                dst.append(p.grad)
                src.append(g)
        ops.accumulate_multi(dst, src)

    def _seg_d(self, st):
        """.grad += the first-order branch's gradients"""
        self._accumulate(st.c_params, st.c_grads)

    def _seg_b(self, st):
        E = st.E
        total = ops.Dot.apply(st.sup_rows, _loss_weights(E, st.frames.device))
        if self.use_policy:
            # sum over episodes of F.cross_entropy(actions[4, 4], best_path[4]) (reference :116-118) = E x the mean over all
            # E * 4 rows (equal rows per episode, unit class weights)
            path_ce, _ = ops.WeightedCE.apply(st.actions_out, st.best_all, _ones(4, st.frames.device))
            total = total + path_ce * float(E)
            st.path_ce = path_ce.detach()
        # (the supervisor backward ends in the fusion parameters and the in_proj blocks; nothing keeps a gradient of the
        #  per-episode copies dtheta, so the weight-gradient contractions with respect to them are skipped)
        # gradients as tensors + ONE multi-tensor accumulation into .grad (torch.autograd.backward would run one AccumulateGrad
        # add_ per parameter: ~110 launches per chunk in a step whose small-batch form is bound by its launch count)
        # (only the LEAF copies: with INNER_STEPS > 1 the later fast weights are functions of the earlier steps' gradients and
        #  the supervisor gradient reaches the fusion parameters through them)
        with ops.skip_param_grads(frozenset(id(t) for t in st.dtheta)):
            grads = torch.autograd.grad(total, self._targets2, allow_unused=True)
        self._accumulate(self._targets2, grads)
        st.sup_rows = st.sup_rows.detach()
        st.mark("8 second-order backward")

    def _chunk_runner(self, E, s, shape, ldn, ldn1):
        """-> callable(inputs) -> results for chunks of this signature: eager segments, or graphs.ChunkGraphs replay"""
        from . import graphs
        return graphs.chunk_runner(self, E, s, shape, ldn, ldn1)

    def forward(self, data, train=True):
        chunk = int(getattr(self.config, "EPISODE_CHUNK", 16))
        if chunk <= 0:
            return self._forward_sequential(data)
        b, s, c, w, h = data["frames"].shape
        img, mask = data["frames"].view(b, s, c, w, h), data["masks"].view(b, s, w, h)
        det_out, sup_out, path_out, reward_out, logits_out, boxes_out = [], [], [], [], [], []
        self._theta = theta = self._real_parameters()
        self._targets2 = self._second_order_targets()
        theta_ids = {id(p) for p in theta}
        self._in_proj = [p for p in self.detector.parameters() if p.requires_grad and id(p) not in theta_ids]
        actions_host = data["actions"].tolist() if self.use_policy else None   # one D2H up front
        try:
            for e0 in range(0, b, chunk):
                E = min(chunk, b - e0)
                ep = range(e0, e0 + E)
                labels = [_labels(data, t) for t in ep]
                # the first-order branch's random frame per episode (reference :126) and both target lists: drawn, packed
                # and uploaded before anything is queued
                ridx = [random.randint(0, 4) for _ in ep]
                inputs = {"frames": img[e0:e0 + E].reshape(E * s, c, w, h), "masks": mask[e0:e0 + E].reshape(E * s, w, h),
                          "targets": [lab for ep_labels in labels for lab in ep_labels],
                          "targets1": [labels[i][ridx[i]] for i in range(E)],
                          "sel": [i * s + r for i, r in enumerate(ridx)]}

                def policy_labels(rewards, e0=e0, ep=ep):   # PathStorage bookkeeping on the host, in episode order
                    if "dp_index" in data:   # data parallel: replay the global batch's chunk (see _dp_chunk_labels)
                        return self._dp_chunk_labels(data, e0 // chunk, chunk, list(ep), rewards)
                    return best_path_labels(self.path_storage, [data["initial_image_path"][t] for t in ep],
                                            [actions_host[t][:4] for t in ep], rewards)

                run = self._chunk_runner(E, s, (c, w, h), _ldn(inputs["targets"]), _ldn(inputs["targets1"]))
                res = run(inputs, policy_labels if self.use_policy else None)
                sup_out.append(res["sup_rows"])
                det_out.append(res["det_rows"])
                logits_out.append(res["logits1"].unsqueeze(1))
                boxes_out.append(res["boxes1"].unsqueeze(1))
                if self.use_policy:
                    path_out.append(res["path_ce"].reshape(1).expand(E))
                    reward_out.append(res["gts"])
            if self.use_policy and "dp_index" in data:   # chunks this rank has no episodes in: still part of the exchange
                for c in range((b + chunk - 1) // chunk, self._dp_chunks(data, chunk)):
                    self._dp_chunk_labels(data, c, chunk, [], [])
        finally:
            set_parameters(self.detector, theta)
        predictions = {"pred_logits": torch.cat(logits_out, dim=0), "pred_boxes": torch.cat(boxes_out, dim=0)}
        # the reference's loss dict: per-task criterion dicts averaged over the tasks of the batch (:139-150)
        losses = _named_means(self.criterion, torch.cat(det_out).mean(0), "loss_detector")
        sup = _named_means(self.criterion, torch.cat(sup_out).mean(0), "loss_supervisor")
        if self.use_policy:
            sup["loss_supervisor_path"] = torch.cat(path_out).mean()
            sup["policy_reward"] = torch.cat(reward_out).mean()
        losses.update(sup)
        return predictions, losses

    # ---- PathStorage under data parallelism ----------------------------------------------------------------------
    # The reference keeps ONE trie per root image and fills it episode after episode (models/interactron.py:109-115);
    # with episodes r::W on rank r a root image's paths would be spread over the ranks' tries and the policy labels
    # would depend on W.  Every rank therefore replays the WHOLE global batch in global order: roots and actions of all
    # episodes come with the batch (trainer.shard_batch, by_root), the rewards of the other ranks' episodes by one
    # all-reduce of a few floats per chunk.  Local chunk c on every rank covers the contiguous global positions
    # [c*chunk*W, (c+1)*chunk*W), so replaying chunk after chunk IS the global order.
    @staticmethod
    def _dp_chunks(data, chunk):
        W, B = data["dp_world"], len(data["dp_roots"])
        return ((B + W - 1) // W + chunk - 1) // chunk

    def _dp_chunk_labels(self, data, c, chunk, local, rewards):
        W, B = data["dp_world"], len(data["dp_roots"])
        lo, hi = min(B, c * chunk * W), min(B, (c + 1) * chunk * W)
        mine = [data["dp_index"][t] - lo for t in local]
        allr = exchange_rewards(rewards, mine, hi - lo)
        return best_path_labels(self.path_storage, data["dp_roots"][lo:hi], data["dp_actions"][lo:hi], allr, set(mine))

    def dp_idle_step(self, data):
        """A rank whose shard of a (short) batch is empty: no forward, but it still takes part in the reward exchanges
        and keeps its tries in step with the other ranks'."""
        if self.use_policy and "dp_index" in data:
            chunk = max(1, int(getattr(self.config, "EPISODE_CHUNK", 16)))
            for c in range(self._dp_chunks(data, chunk)):
                self._dp_chunk_labels(data, c, chunk, [], [])

    def _forward_sequential(self, data):
        """The reference's own task-by-task schedule (kept for EPISODE_CHUNK: 0 and as the cross-check of the
        episode-batched path in tests)."""
        b, s, c, w, h = data["frames"].shape
        img, mask = data["frames"].view(b, s, c, w, h), data["masks"].view(b, s, w, h)
        det_losses, sup_losses, logits_out, boxes_out = [], [], [], []
        self._theta = theta = self._real_parameters()
        targets2 = self._second_order_targets()
        try:
            for task in range(b):
                labels = _labels(data, task)
                dtheta, steps, fusion_out, fast = self._adapt(img[task], mask[task], create_graph=True)
                set_parameters(self.detector, fast)
                post = self.detector(NestedTensor(img[task], mask[task]))
                sup = self.criterion(post, labels, background_c=0.1)
                if self.use_policy:
                    first = {k: v[[0]] for k, v in post.items() if k in ("pred_logits", "pred_boxes")}
                    gt = _weighted(self.criterion(first, [labels[0]], background_c=0.1))
                    actions = data["actions"][task][:4].tolist()
                    if "dp_index" in data:
                        label = self._dp_chunk_labels(data, task, 1, [task], [torch.mean(gt).item()])[0]
                    else:
                        label = best_path_labels(self.path_storage, [data["initial_image_path"][task]], [actions],
                                                 [torch.mean(gt).item()])[0]
                    best = torch.tensor(label, dtype=torch.long, device=gt.device)
                    weight = torch.ones(4, device=gt.device)
                    sup["loss_path"], _ = ops.WeightedCE.apply(fusion_out["actions"].reshape(4, 4), best, weight)
                    sup["policy_reward"] = gt
                sup_losses.append({k: v.detach() for k, v in sup.items()})
                total = _weighted(sup) + (sup["loss_path"] if self.use_policy else 0)
                with ops.skip_param_grads(frozenset(id(t) for t in dtheta)):   # (see the batched schedule)
                    torch.autograd.backward(total, inputs=targets2)

                # first-order detector update through the adapted weights (reference interactron.py:126-134)
                fast1 = theta
                for grads in steps:
                    fast1 = sgd_step(fast1, [None if g is None else g.detach() for g in grads], self.config.ADAPTIVE_LR)
                del steps, dtheta, fusion_out, post, sup, total, fast
                set_parameters(self.detector, fast1)
                ridx = random.randint(0, 4)
                post1 = self.detector(NestedTensor(img[task][ridx:ridx + 1], mask[task][ridx:ridx + 1]))
                dl = self.criterion(post1, labels[ridx:ridx + 1], background_c=0.1)
                det_losses.append({k: v.detach() for k, v in dl.items()})
                _weighted(dl).backward()
                logits_out.append(post1["pred_logits"].detach())
                boxes_out.append(post1["pred_boxes"].detach())
            if self.use_policy and "dp_index" in data:
                for c in range(b, self._dp_chunks(data, 1)):
                    self._dp_chunk_labels(data, c, 1, [], [])
        finally:
            set_parameters(self.detector, theta)
        predictions = {"pred_logits": torch.stack(logits_out, dim=0), "pred_boxes": torch.stack(boxes_out, dim=0)}
        losses = _mean_losses(det_losses, "loss_detector")
        losses.update(_mean_losses(sup_losses, "loss_supervisor"))
        return predictions, losses

    def train(self, mode=True):
        self.mode = "train" if mode else "test"
        self.detector.train(mode)
        self.fusion.train(mode)
        return self


class interactron(_Adaptive):
    use_policy = True

    def __init__(self, config):
        super().__init__()
        self.detector, self.criterion, self.postprocessor = build(config)
        _load_detector_weights(self.detector, config)
        self.fusion = Transformer(config)
        self.path_storage = {}
        self.config = config

    def _policy_logits(self, frames, masks):
        with torch.no_grad():
            pre = _lift(self.detector(NestedTensor(frames, masks)))
            return self.fusion(pre)["actions"]

    # The policy step is pure inference on fixed shapes (1..4 frames), ~1 500 launches of a few microseconds each: at one
    # episode it is bound by the host issuing them.  In eval mode the launch sequence of each frame count is captured
    # once into a HIP graph (static input buffers, weights by address) and replayed; any capture failure, training
    # mode or a weight re-allocation falls back to eager launches.  POLICY_GRAPH: false in the config turns it off.
    def _policy_graph(self, frames, masks):
        key = (tuple(frames.shape), frames.device.index)
        if "_graphs" not in self.__dict__:
            self._graphs = {}
        ent = self._graphs.get(key)
        stamp = self._graph_stamp()[0] if ent is not None else None
        if ent is None or ent[0] != stamp:
            sf, sm = frames.clone(), masks.clone()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):   # warm-up outside the capture (lazy initialisation, allocator pools, BN folds)
                self._policy_logits(sf, sm)
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            ops.capture_begin(None)   # (capture-local caches: key biases, weight planes)
            try:
                with torch.cuda.graph(graph):
                    out = self._policy_logits(sf, sm)
            finally:
                ops.capture_end()
            stamp, folds = self._graph_stamp()   # (after the warm-up: the folds exist now; held alive with the graph)
            ent = self._graphs[key] = (stamp, graph, sf, sm, out, folds)
        _, graph, sf, sm, out, _ = ent
        sf.copy_(frames)
        sm.copy_(masks)
        graph.replay()
        return out

    def get_next_action(self, data):
        b, s, c, w, h = data["frames"].shape
        frames, masks = data["frames"].view(b * s, c, w, h), data["masks"].view(b * s, w, h)
        use_graph = (frames.is_cuda and not self.fusion.training and not self.detector.training
                     and bool(getattr(self.config, "POLICY_GRAPH", True)) and self._graphs is not None)
        actions = None
        if use_graph:
            try:
                actions = self._policy_graph(frames, masks)
            except RuntimeError as e:   # capture not possible here: stay eager from now on, and say so once
                import warnings
                warnings.warn("get_next_action: HIP-graph capture of the policy step failed (%s); continuing with eager "
                              "launches for the rest of this process" % (str(e).splitlines()[0] if str(e) else type(e).__name__))
                self._graphs = None
                torch.cuda.synchronize()
        if actions is None:
            actions = self._policy_logits(frames, masks)
        return actions[s - 1].argmax(dim=-1).item()


class interactron_random(_Adaptive):
    use_policy = False

    def __init__(self, config):
        super().__init__()
        self.detector, self.criterion, self.postprocessor = build(config)
        _load_detector_weights(self.detector, config)
        self.fusion = DecoderTransformer(config)
        self.config = config


class detr_multiframe(_EpisodeModel):
    def __init__(self, config):
        super().__init__()
        self.detector, self.criterion, self.postprocessor = build(config)
        _load_detector_weights(self.detector, config)
        self.fusion = Transformer(config)
        self.config = config

    def predict(self, data):
        b, s, c, w, h = data["frames"].shape
        with torch.no_grad():
            out = self.fusion(_lift(self.detector(NestedTensor(data["frames"].view(b * s, c, w, h),
                                                               data["masks"].view(b * s, w, h)))))
        return {"pred_boxes": out["pred_boxes"].view(b, s, *out["pred_boxes"].shape[1:]),
                "pred_logits": out["pred_logits"].view(b, s, *out["pred_logits"].shape[1:])}

    def forward(self, data):
        """reference detr_multiframe.py:55-109 (criterion on the fusion outputs, one backward per episode).  The weights
        are shared by all episodes here, so a chunk of episodes is simply one batch: detector on E*5 frames, fusion
        with batch E, one matcher pass, the per-episode criteria summed into one backward (EPISODE_CHUNK: 0 keeps the
        task-by-task loop)."""
        chunk = int(getattr(self.config, "EPISODE_CHUNK", 16))
        b, s, c, w, h = data["frames"].shape
        img, mask = data["frames"].view(b, s, c, w, h), data["masks"].view(b, s, w, h)
        losses, lo, bo = [], [], []
        if chunk <= 0:
            for task in range(b):
                out = self.fusion(_lift(self.detector(NestedTensor(img[task], mask[task]))))
                loss = self.criterion(out, _labels(data, task), background_c=0.1)
                _weighted(loss).backward()
                losses.append({k: v.detach() for k, v in loss.items()})
                lo.append(out["pred_logits"][0:1].detach())
                bo.append(out["pred_boxes"][0:1].detach())
        rows_out = []
        for e0 in range(0, b if chunk > 0 else 0, max(chunk, 1)):
            E = min(chunk, b - e0)
            tg = ops.pack_targets([lab for t in range(e0, e0 + E) for lab in _labels(data, t)])
            det = self.detector(NestedTensor(img[e0:e0 + E].reshape(E * s, c, w, h), mask[e0:e0 + E].reshape(E * s, w, h)))
            out = self.fusion({k: v.reshape((E, s) + tuple(v.shape[1:])) for k, v in det.items()})
            o = {"pred_logits": out["pred_logits"].reshape(E * s, *out["pred_logits"].shape[-2:]),
                 "pred_boxes": out["pred_boxes"].reshape(E * s, *out["pred_boxes"].shape[-2:])}
            # matcher + per-episode criteria of the chunk on the device, their weighted sum into one backward
            (rows,) = self.criterion.grouped(o, tg, self.criterion.matcher.match(o, tg), ((s, s),), background_c=0.1)
            ops.Dot.apply(rows, _loss_weights(E, rows.device)).backward()
            rows_out.append(rows.detach())
            lo.append(o["pred_logits"].detach()[0::s].unsqueeze(1))
            bo.append(o["pred_boxes"].detach()[0::s].unsqueeze(1))
        if chunk > 0:
            return {"pred_logits": torch.cat(lo, dim=0), "pred_boxes": torch.cat(bo, dim=0)}, \
                _named_means(self.criterion, torch.cat(rows_out).mean(0), "loss_detector")
        return {"pred_logits": torch.stack(lo, dim=0), "pred_boxes": torch.stack(bo, dim=0)}, \
            _mean_losses(losses, "loss_detector")

    def train(self, mode=True):
        self.mode = "train" if mode else "test"
        self.detector.train(False)
        self.detector.transformer.decoder.train(mode)
        self.fusion.train(mode)
        return self


class detr(_EpisodeModel):
    def __init__(self, config):
        super().__init__()
        self.model, self.criterion, self.postprocessor = build(config)
        _load_detector_weights(self.model, config)
        self.config = config

    def _run(self, data):
        b, s, c, w, h = data["frames"].shape
        return self.model(NestedTensor(data["frames"].view(b * s, c, w, h), data["masks"].view(b * s, w, h))), b, s

    def predict(self, data):
        with torch.no_grad():
            out, b, s = self._run(data)
        return {k: v.reshape(b, s, *v.shape[1:]) for k, v in out.items()}

    def forward(self, data):
        out, b, s = self._run(data)
        labels = [l for i in range(b) for l in _labels(data, i)]
        losses = self.criterion(out, labels)
        (losses["loss_ce"] + 5 * losses["loss_bbox"] + 2 * losses["loss_giou"]).backward()
        return {k: v.detach().reshape(b, s, *v.shape[1:]) for k, v in out.items()}, \
            {k.replace("loss", "loss_detector"): v for k, v in losses.items()}

    def train(self, mode=True):
        self.mode = "train" if mode else "test"
        self.model.train(mode)
        return self
