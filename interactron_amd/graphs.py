"""HIP-graph replay of the sync-free segments of an episode chunk (second half).

The reference runs a meta-train step as ~6 000 small dependent launches per chunk of episodes, issued one by one from
Python (models/interactron.py:84-137 through torch.autograd).  With the reference's global batch of 16 sharded over 8
GPUs a rank sees 2 episodes per step: the kernels then take ~80 ms but the host needs ~100 ms to issue them -- the step is
bound by Python.  The launch sequence of a chunk is a pure function of its signature (episodes, frame shape, target
pitch), so it is captured ONCE into three HIP graphs (segments A, C, B of ``episode._Adaptive``) over static input
buffers and replayed with ~2 ms of host work per step:

    inputs -> static buffers,  salt += 1 (fresh dropout masks: ix_set_dropout_salt),  replay A,  replay C,
    wait for A's reward copy (the GPU is busy with C),  PathStorage on the host,  labels -> static buffer,  replay B

What makes a replay valid: every tensor a captured kernel reads by address is either a static buffer of this object, a
parameter / buffer / .grad of the model (their addresses are stamped; a change re-captures), or lives in the graphs'
private memory pool; dropout seeds come from a device word; the matcher runs on the device; nothing synchronises inside a
segment.  The first call of a signature runs eagerly (it also warms every host-side cache), the second captures, any
capture failure falls back to eager launches for good with one warning.
"""
import warnings

import torch

from . import hipops as ops
from .meta import set_parameters

# "thread_local": only the capturing thread's own unsafe calls invalidate a capture -- the RCCL watchdog thread of a multi-rank
# run polls events while a rank captures, which the default "global" mode treats as an error
CAPTURE_MODE = "thread_local"
GOLDEN = -0x61C8864680B583EB   # 0x9E3779B97F4A7C15 as a signed 64-bit step of the dropout salt
# Captured chunks address their ground truth through static CSR buffers with ONE fixed pitch (columns per image in the cost
# matrices, capacity I * GRAPH_PITCH): the kernels read the per-image counts from `off`, a pitch only has to be >= the
# largest count.  A signature therefore does not depend on how many boxes a batch happens to carry (real data: it changes
# batch to batch; keyed on it, every new pair cost an eager step, a capture and a private pool holding a step's activations);
# a batch with a fuller image runs eagerly.  At most GRAPH_SETS signatures stay captured (least recently used goes first).
GRAPH_PITCH = 64
GRAPH_SETS = 4


class ChunkState:
    """Working set of one chunk shared by the three segments (inputs, tensors handed from one segment to the next)."""

    def __init__(self, E, s, timer=None):
        self.E, self.s, self.timer = E, s, timer

    def mark(self, name):
        if self.timer is not None:
            self.timer.mark(name)

    def drop_tape(self):
        for k in ("dtheta", "grads", "nt", "actions_out", "sup_rows", "gts", "det_rows", "logits1", "boxes1", "path_ce", "c_grads", "c_params"):
            self.__dict__.pop(k, None)


def _results(st, policy, clone):
    f = (lambda t: t.clone()) if clone else (lambda t: t)
    res = {"sup_rows": f(st.sup_rows), "det_rows": f(st.det_rows), "logits1": f(st.logits1), "boxes1": f(st.boxes1)}
    if policy:
        res["path_ce"], res["gts"] = f(st.path_ce), f(st.gts)
    return res


def run_eager(model, E, s, inputs, policy_labels):
    """The three segments issued launch by launch (large chunks: the step is GPU-bound; also every signature's first call)."""
    from .episode import _PhaseTimer
    st = ChunkState(E, s, _PhaseTimer(model.phase_times))
    st.frames, st.masks = inputs["frames"], inputs["masks"]
    st.tg, st.tg1 = ops.pack_targets(inputs["targets"]), ops.pack_targets(inputs["targets1"])
    st.sel = ops.h2d_async(torch.tensor(inputs["sel"], dtype=torch.int64))
    policy = policy_labels is not None
    if policy:
        st.gts_host = torch.empty(E, dtype=torch.float32, pin_memory=True)
    model._seg_a(st)
    if policy:
        ready = torch.cuda.Event()
        ready.record()
    model._seg_c(st)
    if policy:   # the step's ONE host round trip; the GPU still has segment C queued behind the reward copy
        ready.synchronize()
        labels = policy_labels(st.gts_host.tolist())
        st.best_all = ops.h2d_async(torch.tensor(labels, dtype=torch.long).reshape(E * 4))
    model._seg_b(st)
    model._seg_d(st)
    res = _results(st, policy, clone=False)
    st.drop_tape()
    return res


def _static_targets(I, ldn, dev):
    cap = I * ldn
    tg = ops.Targets(torch.zeros(cap, dtype=torch.int64, device=dev), torch.full((cap, 4), 0.5, dtype=torch.float32, device=dev),
                     torch.zeros(I + 1, dtype=torch.int32, device=dev), [0] * I)
    tg.ldn = ldn
    tg.stage = torch.zeros(I + 1, dtype=torch.int32, pin_memory=True)
    return tg


def _load_targets(tg, targets):
    sizes = [int(t["labels"].shape[0]) for t in targets]
    total = sum(sizes)
    assert len(sizes) == tg.I and max(sizes + [0]) <= tg.ldn and total <= tg.ids.shape[0]
    if total:
        torch.cat([t["labels"] for t in targets], out=tg.ids[:total])
        torch.cat([t["boxes"] for t in targets], out=tg.boxes[:total])
    off = 0
    for i, n in enumerate(sizes):
        tg.stage[i] = off
        off += n
    tg.stage[tg.I] = off
    tg.off.copy_(tg.stage, non_blocking=True)
    tg.sizes, tg.targets = sizes, targets


class ChunkGraphs:
    """Static buffers + the three captured graphs of one chunk signature (E episodes of s frames [c, w, h], target pitches)."""

    def __init__(self, model, E, s, shape, ldn, ldn1, mask_dtype):
        c, w, h = shape
        dev = next(model.parameters()).device
        self.model, self.policy = model, model.use_policy
        self.st = st = ChunkState(E, s)
        st.frames = torch.zeros(E * s, c, w, h, device=dev)
        st.masks = torch.zeros(E * s, w, h, dtype=mask_dtype, device=dev)
        st.tg, st.tg1 = _static_targets(E * s, ldn, dev), _static_targets(E, ldn1, dev)
        st.sel = torch.zeros(E, dtype=torch.int64, device=dev)
        self.sel_stage = torch.zeros(E, dtype=torch.int64, pin_memory=True)
        if self.policy:
            st.gts_host = torch.zeros(E, dtype=torch.float32, pin_memory=True)
            st.best_all = torch.zeros(E * 4, dtype=torch.int64, device=dev)
            self.best_stage = torch.zeros(E * 4, dtype=torch.int64, pin_memory=True)
        self.salt = torch.zeros(1, dtype=torch.int64, device=dev)
        self.stream, self.side = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
        self.graphs, self.stamp, self.keep, self.loaded, self.alive = None, None, None, None, None

    def load(self, inputs):
        st = self.st
        # the pinned staging buffers are rewritten below: the uploads of the previous call must have run (a model without
        # the PathStorage round trip never waits for the GPU, the host may be a whole step ahead)
        if self.loaded is not None:
            self.loaded.synchronize()
        st.frames.copy_(inputs["frames"])
        st.masks.copy_(inputs["masks"])
        _load_targets(st.tg, inputs["targets"])
        _load_targets(st.tg1, inputs["targets1"])
        self.sel_stage.copy_(torch.tensor(inputs["sel"], dtype=torch.int64))
        st.sel.copy_(self.sel_stage, non_blocking=True)
        self.loaded = torch.cuda.Event()
        self.loaded.record()

    def capture(self):
        """Capture order A, C, B, D.  A, B and D share one memory pool (B's backward releases A's tape); C -- the first-order
        branch, which a replay runs on a second stream WHILE B runs -- allocates from its own pool and its own scratch slot, and
        reads of A only what this object keeps alive (the learned-loss gradients, the stem features, the static inputs)."""
        model, st = self.model, self.st
        theta = model._theta
        pool, pool_c = torch.cuda.graph_pool_handle(), torch.cuda.graph_pool_handle()
        graphs = []
        ops.prepare_scratch_slots((0, 1), st.frames.device)   # sized from the warm-up run, tickets zeroed eagerly
        ops.capture_begin(self.salt)
        try:
            for seg, pl, slot in ((model._seg_a, pool, 0), (model._seg_c, pool_c, 1), (model._seg_b, pool, 0), (model._seg_d, pool, 0)):
                g = torch.cuda.CUDAGraph()
                with ops.scratch_slot(slot), torch.cuda.graph(g, pool=pl, stream=self.stream, capture_error_mode=CAPTURE_MODE):
                    seg(st)
                graphs.append(g)
            # small results live in the pools: keep them (replays rewrite them in place), drop the autograd tape -- but keep
            # what C and D read of the other segments alive for as long as the graphs exist
            self.keep = _results(st, self.policy, clone=False)
            self.alive = (st.grads, st.nt, st.c_grads)
            if self.policy:
                self.keep["gts_host"] = st.gts_host
        finally:
            ops.capture_end()
            st.drop_tape()
            set_parameters(model.detector, theta)
        self.graphs, self.stamp = graphs, model._graph_stamp()[0]

    def __call__(self, inputs, policy_labels):
        self.load(inputs)
        self.salt.add_(GOLDEN)
        gA, gC, gB, gD = self.graphs
        main = torch.cuda.current_stream()
        gA.replay()
        self.side.wait_stream(main)
        if self.policy:
            ready = torch.cuda.Event()
            ready.record()
        with torch.cuda.stream(self.side):   # the first-order branch runs beside the supervisor backward
            gC.replay()
        if self.policy:
            ready.synchronize()
            labels = policy_labels(self.keep["gts_host"].tolist())
            self.best_stage.copy_(torch.tensor(labels, dtype=torch.long).reshape(-1))
            self.st.best_all.copy_(self.best_stage, non_blocking=True)
        gB.replay()
        main.wait_stream(self.side)
        gD.replay()
        return {k: v.clone() for k, v in self.keep.items() if k != "gts_host"}


def _wants_graph(model, E, shape):
    mode = getattr(model.config, "STEP_GRAPH", "auto")
    if isinstance(mode, str):
        mode = mode.strip().lower()
    if mode in (False, "false", "off", "0", 0):
        return False
    if mode in (True, "true", "on", "1", 1):
        return True
    # auto: small chunks are bound by the host issuing the launches; the reference batch of 16 x 300^2 episodes is GPU-bound but
    # still gains 2.7 % from a replay (no launch gaps, the first-order branch beside the second-order backward: 262.3 -> 255.3
    # ms, r4e).  A capture's private pools hold a whole step's working set (58 GB at 16 x 300^2, 152 GB at 8 x 800^2): the
    # eager warm-up step's cached blocks are handed back first and the capture only goes ahead when the device has room for
    # the warm-up's measured peak (chunk_runner); chunks beyond 8 x 800^2 stay eager
    return E * shape[1] * shape[2] <= 8 * 800 * 800


def chunk_runner(model, E, s, shape, ldn, ldn1):
    """-> callable(inputs, policy_labels) for this chunk signature (see the module docstring for when graphs are used)"""
    from .criterion import HungarianMatcher
    state = model.__dict__.setdefault("_chunk_graphs", {})
    eager = lambda inputs, policy_labels: run_eager(model, E, s, inputs, policy_labels)
    if state.get("disabled") or not _wants_graph(model, E, shape) or model.phase_times is not None:
        return eager
    matcher = model.criterion.matcher
    if type(matcher).assign is not HungarianMatcher._host_assign or max(ldn, ldn1) > min(GRAPH_PITCH, ops.LSAP_DEVICE_MAX):
        return eager   # (assignments pinned by a test, or an image with more boxes than the static buffers' pitch: this batch only)
    # gradients must accumulate IN PLACE into existing buffers (trainer.FlatBuffers provides them; after an eager step only
    # never-used parameters such as GPT.pos_emb are still without one).  zero_grad(set_to_none=True) between steps: eager.
    if sum(p.requires_grad and p.grad is None for p in model.parameters()) > 4:
        return eager
    key = (E, s, tuple(shape), model.detector.training, model.fusion.training)
    ent = state.get(key)
    if ent is not None:
        state[key] = state.pop(key)   # most recently used last
    if ent is None:   # first call of a signature: eager (warms BN folds, scratch, size caches); the next one captures
        state[key] = "warm"

        def warm(inputs, policy_labels):   # ... and measures what a step of THIS signature needs beyond what is already held.
            # The process-wide peak statistic is RESET for that (an earlier, larger signature -- 8 x 800^2 before 16 x 300^2 -- used to
            # inflate the need and pin the later one to eager launches); a reader of max_memory_allocated() sees the peak since the
            # latest first step of a signature, which is the step's own working set (bench.py resets it per workload anyway).
            base = torch.cuda.memory_allocated()
            torch.cuda.reset_peak_memory_stats()
            res = run_eager(model, E, s, inputs, policy_labels)
            model.__dict__.setdefault("_chunk_peaks", {})[key] = max(0, torch.cuda.max_memory_allocated() - base)
            return res
        return warm
    if ent == "eager" or (isinstance(ent, tuple) and ent[1] == model._graph_stamp()[0]):
        return eager
    if ent == "warm" or isinstance(ent, tuple) or ent.stamp != model._graph_stamp()[0]:
        def capture_then_run(inputs, policy_labels):
            fresh_grads = {id(p) for p in model.parameters() if p.grad is None}   # .grad tensors first bound inside the capture
            try:
                held = [k for k, v in state.items() if isinstance(v, ChunkGraphs) and k != key]
                for old in held[:max(0, len(held) - (GRAPH_SETS - 1))]:
                    del state[old]          # its graphs, static buffers and both private pools go with it
                need = model.__dict__.get("_chunk_peaks", {}).get(key, 0)
                if need > (8 << 30):   # a large working set: the pools must find it as FREE device memory
                    recapture = isinstance(state.get(key), ChunkGraphs)
                    if recapture:
                        state[key] = "warm"   # the stale graphs of this very signature hold the memory the new ones need
                    import gc
                    gc.collect()              # (evicted / stale ChunkGraphs sit in a reference cycle with the model: their pools only go now)
                    torch.cuda.synchronize()
                    torch.cuda.empty_cache()   # (the warm-up step's blocks sit in the allocator's cache, useless to a private pool)
                    free, _ = torch.cuda.mem_get_info()
                    if free < 1.15 * need + (4 << 30):
                        # not an error: eager launches.  A first capture that does not fit stays eager; a RE-capture (stamp moved)
                        # is tried again when the stamp moves NEXT (("retry", stamp): eager until then) -- a transient low reading
                        # must not pin the signature, and must not cost a collect + synchronize + empty_cache on every step either
                        state[key] = ("retry", model._graph_stamp()[0]) if recapture else "eager"
                        return run_eager(model, E, s, inputs, policy_labels)
                g = ChunkGraphs(model, E, s, shape, GRAPH_PITCH, GRAPH_PITCH, inputs["masks"].dtype)
                g.load(inputs)
                g.capture()
            except Exception as e:   # capture not possible here: stay eager from now on, and say so once
                warnings.warn("HIP-graph capture of the meta-train chunk failed (%s: %s); continuing with eager launches for the "
                              "rest of this process" % (type(e).__name__, (str(e).splitlines() or [""])[0]))
                state["disabled"] = True
                torch.cuda.synchronize()
                # nothing of the capture ran: its scratch tickets and any .grad buffer it bound are not to be trusted
                ops.reset_scratch_slots()
                for p in model.parameters():
                    if p.grad is not None and id(p) in fresh_grads:
                        p.grad = None
                return run_eager(model, E, s, inputs, policy_labels)
            state[key] = g
            return g(inputs, policy_labels)
        return capture_then_run
    return ent


class PredictGraph:
    """predict() of ONE episode (reference models/interactron.py:31-59: adapt on s frames, detect frame 0 through the adapted
    weights) as one captured graph per frame shape: ~3 000 launches, no host sync inside -- eager it is bound by the host
    (25 ms per episode), replayed by the kernels."""

    def __init__(self, model, frames, masks):
        self.model = model
        self.frames, self.masks = torch.zeros_like(frames), torch.zeros_like(masks)
        self.stream = torch.cuda.Stream(device=frames.device)
        self.graph, self.out, self.stamp = None, None, None

    def capture(self):
        g = torch.cuda.CUDAGraph()
        ops.prepare_scratch_slots((0,), self.frames.device)
        ops.capture_begin(None)
        try:
            with torch.cuda.graph(g, stream=self.stream, capture_error_mode=CAPTURE_MODE):
                out = self.model._predict_one(self.frames, self.masks)
        finally:
            ops.capture_end()
        self.graph, self.out, self.stamp = g, out, self.model._graph_stamp()[0]

    def __call__(self, frames, masks):
        self.frames.copy_(frames)
        self.masks.copy_(masks)
        self.graph.replay()
        return {k: v.clone() for k, v in self.out.items()}


def predict_runner(model, frames, masks):
    """-> callable(frames, masks) -> predict() result dict for one episode of this shape (eval mode): eager on the first call,
    captured on the second, replayed afterwards; PREDICT_GRAPH: false in the config keeps it eager."""
    state = model.__dict__.setdefault("_predict_graphs", {})
    eager = model._predict_one
    on = getattr(model.config, "PREDICT_GRAPH", True)
    if state.get("disabled") or on in (False, "false", "off", 0) or not frames.is_cuda or model.detector.training or model.fusion.training:
        return eager
    key = (tuple(frames.shape), masks.dtype)
    ent = state.get(key)
    if ent is None:
        state[key] = "warm"
        return eager
    if ent == "warm" or ent.stamp != model._graph_stamp()[0]:
        def capture_then_run(frames, masks):
            try:
                g = PredictGraph(model, frames, masks)
                g.capture()
            except Exception as e:
                warnings.warn("HIP-graph capture of predict() failed (%s: %s); continuing with eager launches for the rest of this "
                              "process" % (type(e).__name__, (str(e).splitlines() or [""])[0]))
                state["disabled"] = True
                torch.cuda.synchronize()
                ops.reset_scratch_slots()
                return eager(frames, masks)
            state[key] = g
            return g(frames, masks)
        return capture_then_run
    return ent
