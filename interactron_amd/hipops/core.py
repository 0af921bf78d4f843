"""hipops.core -- library handle, streams, the Function base class (fp32 / 16-bit dispatch), scratch, capture state, the compute
mode of a model, identities of weights and skipped gradients."""
import ctypes
import gc as _gc
import os
import os as _os
from collections import namedtuple

import torch
from torch.autograd import Function as _TorchFunction
from torch.autograd.function import once_differentiable

from .. import _lib

_c_void_p = ctypes.c_void_p


def _L():
    return _lib.load()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_dev = [None]


def _stream():
    """hipStream_t of torch's current stream.  (The raw accessor is ~10x cheaper than building a torch.cuda.Stream
    object per launch; a step issues ~13 000 launches and its backward is host-bound.)"""
    if _raw_stream is None:
        return torch.cuda.current_stream().cuda_stream
    if _dev[0] is None:
        _dev[0] = torch.cuda.current_device()   # one process per GPU: the device is fixed before the first launch
    return _raw_stream(_dev[0])


def h2d_async(t):
    """Host tensor -> current GPU through a pinned staging buffer, without blocking the host.  A plain ``.to(device)``
    of pageable memory is stream-ordered AND host-blocking: the host then sits out everything queued before it (the
    criterion's index tensors used to cost one such stall per image)."""
    if t.numel() == 0:
        return torch.empty(t.shape, dtype=t.dtype, device="cuda")
    stage = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    stage.copy_(t)
    return stage.to("cuda", non_blocking=True)


class _NullCtx:
    """Stand-in for the autograd context when a Function's forward is run without recording (see Function.call)."""
    needs_input_grad = (False,) * 64
    saved_tensors = ()

    def save_for_backward(self, *a):
        pass

    def mark_non_differentiable(self, *a):
        pass

    def set_materialize_grads(self, v):
        pass


_b16_seen = [False]   # a bf16 activation exists in this process (the 16-bit mode, b16.py); until then no call looks at dtypes
_B16_TWINS = os.environ.get("IX_B16_TWINS", "1") == "1"   # "0": every op without native bf16 support is ADAPTED (fp32 kernels between casts): A/B runs


def _any_b16(args):
    for a in args:
        if torch.is_tensor(a) and a.dtype == torch.bfloat16:
            return True
    return False


class Function(_TorchFunction):
    """torch.autograd.Function plus ``call``: inside backward passes that are not themselves recorded (grad mode off:
    the final second-order / first-order backward) the nested nodes skip the autograd bookkeeping and run their
    forward directly -- same kernels, roughly half the host time per node.

    ``b16``: what the op does with bf16 tensors of the 16-bit activation mode (b16.py) -- "native": its forward takes them;
    "adapt" (default): it is computed by its fp32 kernels between two conversion passes.  ``b16_out``: which fp32 results of an
    adapted call come back as bf16 (True: all; False: none -- reductions to scalars, parameter gradients; or one flag per output)."""
    b16 = "adapt"
    b16_out = True
    b16_twin = None   # a subclass with the same backward whose forward launches the 16-bit kernels (b16.py): calls with bf16 tensors go there

    @classmethod
    def apply(cls, *args):
        if _b16_seen[0] and cls.b16 != "native" and _any_b16(args):
            from .. import b16
            if cls.b16_twin is not None and _B16_TWINS:
                return cls.b16_twin.apply(*args)
            return b16.adapt(cls, args, lambda up: super(Function, cls).apply(*up))
        return super().apply(*args)

    @classmethod
    def call(cls, *args):
        if torch.is_grad_enabled():
            return cls.apply(*args)
        if _b16_seen[0] and cls.b16 != "native" and _any_b16(args):
            from .. import b16
            if cls.b16_twin is not None and _B16_TWINS:
                return cls.b16_twin.forward(_NullCtx(), *args)
            return b16.adapt(cls, args, lambda up: cls.forward(_NullCtx(), *up))
        return cls.forward(_NullCtx(), *args)


def _chk(rc, name):
    if rc != 0:
        _lib.check(rc, name)


def _req(t, name="tensor"):
    if not t.is_cuda:
        raise _lib.HipLibraryError("%s must live on the GPU: the HIP path has no CPU fallback" % name)
    if _dev[0] is not None and t.device.index != _dev[0]:
        # one process per GPU: every launch goes on the stream of the device bound at the first launch; a tensor of
        # another device would be dereferenced by kernels running there (memory fault), so it is refused here
        raise _lib.HipLibraryError("%s lives on cuda:%d but this process computes on cuda:%d (one process per GPU: bind "
                                   "the device with torch.cuda.set_device(LOCAL_RANK) before building the model)"
                                   % (name, t.device.index, _dev[0]))
    if t.dtype != torch.float32:
        raise TypeError("%s must be float32, got %s" % (name, t.dtype))
    return t if t.is_contiguous() else t.contiguous()


def _numel(shape):
    n = 1
    for s in shape:
        n *= s
    return n

# ---- the contraction scratch -----------------------------------------------------------------------------------------
# Split-K contractions write one partial plane per split into caller-provided scratch and add the planes in order
# (ix_gemm_f32_ws; deterministic, unlike the fp32 atomics of the workspace-free entry point).  ONE scratch tensor per
# process serves every call: launches are stream-ordered on the process's single compute stream, so the next contraction
# can only overwrite it after the previous reduction has read it.  It grows geometrically; superseded buffers are kept
# alive because captured HIP graphs hold their addresses.
_scratch = [None]
_scratch_retired = []
_scratch_slots = {0: None}   # slot -> scratch tensor; slot 0 is the default.  Launch sequences that may run CONCURRENTLY with the
_scratch_slot = [0]          # main one (graphs.ChunkGraphs replays the first-order branch on a second stream) take their own slot


class scratch_slot:
    """with scratch_slot(k): contractions / reductions issued inside use scratch buffer k (own split-K planes, own tickets)"""

    def __init__(self, slot):
        self.slot = slot

    def __enter__(self):
        _scratch_slots[_scratch_slot[0]] = _scratch[0]
        self.prev = _scratch_slot[0]
        _scratch_slot[0] = self.slot
        _scratch[0] = _scratch_slots.get(self.slot)
        return self

    def __exit__(self, *exc):
        _scratch_slots[self.slot] = _scratch[0]
        _scratch_slot[0] = self.prev
        _scratch[0] = _scratch_slots.get(self.prev)
        return False


TICKET_BYTES = 65536   # IX_TICKET_BYTES of csrc/common.h: the head of the scratch holds the reduction tickets


def _workspace(nbytes, device):
    """-> the process's scratch tensor, at least `nbytes` long.  Layout (include/interactron_hip.h): [TICKET_BYTES of
    tickets, zero when the buffer is created and left zero by every kernel][split-K planes / reduction partials]."""
    t = _scratch[0]
    if t is None or t.numel() < nbytes:
        # never inside a graph capture: the buffer would come from the graph's private pool and the zeroing of its tickets
        # would be RECORDED, not executed -- after a failed capture the module would keep a scratch whose tickets were never
        # cleared (reductions that then never fire, or fire early).  graphs.* sizes every slot before capture_begin.
        if _capture[0] is not None:
            raise _lib.HipLibraryError("the contraction scratch would have to grow inside a HIP-graph capture (%d > %d bytes): "
                                  "call prepare_scratch_slots() after a warm-up run first" % (nbytes, 0 if t is None else t.numel()))
        if t is not None:
            _scratch_retired.append(t)
        t = _scratch[0] = torch.empty(max(2 * nbytes, 64 << 20), dtype=torch.uint8, device=device)
        t[:TICKET_BYTES].zero_()
    return t


def prepare_scratch_slots(slots, device):
    """Before a capture: every scratch slot the captured launch sequences use exists, is as large as the largest scratch
    the warm-up run needed (the capture replays the same shapes) and has had its tickets zeroed EAGERLY."""
    _scratch_slots[_scratch_slot[0]] = _scratch[0]
    need = max([t.numel() for t in _scratch_slots.values() if t is not None] + [64 << 20])
    for k in slots:
        t = _scratch_slots.get(k)
        if t is None or t.numel() < need:
            if t is not None:
                _scratch_retired.append(t)
            t = _scratch_slots[k] = torch.empty(need, dtype=torch.uint8, device=device)
            t[:TICKET_BYTES].zero_()
    _scratch[0] = _scratch_slots.get(_scratch_slot[0])


def reset_scratch_slots():
    """After a FAILED capture: forget every scratch buffer (a captured-but-never-run kernel sequence may have left tickets
    half way); the next launch allocates and zeroes a fresh one.  The old buffers stay alive for earlier graphs."""
    for t in list(_scratch_slots.values()) + [_scratch[0]]:
        if t is not None and not any(t is r for r in _scratch_retired):
            _scratch_retired.append(t)
    for k in list(_scratch_slots):
        _scratch_slots[k] = None
    _scratch[0] = None


_red_ws = {}


def _reduce_ws(kind, rows, C, groups, device):
    """(pointer, bytes) of the scratch for one multi-workgroup reduction (sizes cached per signature)."""
    key = (kind, rows, C, groups)
    n = _red_ws.get(key)
    if n is None:
        out = ctypes.c_size_t(0)
        L = _L()
        if kind == "colsum":
            _chk(L.ix_workspace_bytes_colsum_f32(rows, C, groups, ctypes.byref(out)), "ix_workspace_bytes_colsum_f32")
        elif kind == "ln":
            _chk(L.ix_workspace_bytes_layernorm_bwd(rows, C, groups, ctypes.byref(out)), "ix_workspace_bytes_layernorm_bwd")
        elif kind == "wce":
            _chk(L.ix_workspace_bytes_weighted_ce(rows, ctypes.byref(out)), "ix_workspace_bytes_weighted_ce")
        else:   # scalar reductions: tickets + 4 KiB of partials
            out.value = TICKET_BYTES + 4096
        n = _red_ws[key] = out.value
    if n == 0:
        return None, 0
    return _workspace(n, device).data_ptr(), n
# Cached planes stand for (tensor object, its autograd version, its address, this epoch).  The epoch is bumped by everything that
# rewrites parameters BEHIND autograd's back: the fused Adam kernel and the flat re-homing of trainer.FlatBuffers (raw
# pointers), replica broadcasts, load_state_dict.  Code that edits `p.data` in place by other means calls weights_changed().
_wp_epoch = [0]


def weights_changed():
    _wp_epoch[0] += 1


def mark_weight(w):
    w._ix_weight = True
    return w


# ---- gradients nobody asked for -------------------------------------------------------------------------------------
# torch.autograd.grad(loss, inputs) prunes NODES that do not lead to `inputs`, but a custom Function's backward is a black
# box to the engine: it computes every input gradient and the engine drops the unused ones.  The MAML inner step asks for
# the gradient w.r.t. the per-episode copies of the detector's fast parameters only, yet every Linear of the fusion
# transformer (and the detector's in_proj blocks) would still run its weight-gradient contraction -- a third of all those
# contractions in a training step.  `skip_param_grads(params)` names the nn.Parameters whose gradient the running backward
# does not need; Gemm.backward consults it for its weight / bias operand (by object identity: the fast weights are other
# tensor objects, even where they share storage with a Parameter).
_unwanted = None
SKIP_UNUSED_GRADS = os.environ.get("IX_SKIP_UNUSED_GRADS", "1") == "1"   # "0": compute them all (A/B runs, tests)


class skip_param_grads:
    def __init__(self, ids):
        self.ids = ids if SKIP_UNUSED_GRADS else None

    def __enter__(self):
        global _unwanted
        self.prev, _unwanted = _unwanted, self.ids
        return self

    def __exit__(self, *exc):
        global _unwanted
        _unwanted = self.prev
        return False


def weight_view(w, *shape):
    """w.reshape(shape) that keeps standing for `w` in skip_param_grads (a view is a new tensor object)"""
    v = w.reshape(*shape)
    if v is not w:
        v._ix_of_param = _param_key(w)
        # 16-bit mode: the view of the parameter's bf16 copy goes with the view -- the flat shadow of a trainable parameter
        # (trainer.FlatBuffers.sync_b16) or the once-converted copy of a frozen one (b16.weight_b16); otherwise every call converts again
        fl = w.__dict__.get("_ix_b16_flat")
        if fl is not None and fl[1] == _wp_epoch[0] and fl[2] == w._version and fl[3] == w.data_ptr() and v.data_ptr() == w.data_ptr():
            v.__dict__["_ix_b16_flat"] = (fl[0].reshape(*shape), fl[1], v._version, v.data_ptr())
        elif _b16_seen[0] and isinstance(w, torch.nn.Parameter) and not w.requires_grad and v.data_ptr() == w.data_ptr():
            from .. import b16
            if COMPUTE_DTYPE in ("bf16", "bf16_fusion"):
                c = b16.weight_b16(w)
                v.__dict__["_ix_b16_flat"] = (c.reshape(*shape), _wp_epoch[0], v._version, v.data_ptr())
    return mark_weight(v)


def _is_unwanted(key, skip):
    """`key` stands for one Parameter (an id) or for several (a tuple of ids: cat_params) -- all of them have to be named"""
    if skip is None:
        return False
    if isinstance(key, tuple):
        return all(k in skip for k in key)
    return key in skip


def _param_key(t):
    """identity under which skip_param_grads knows a weight operand: the Parameter itself, or the Parameter a SplitRows view
    was cut from"""
    return getattr(t, "_ix_of_param", id(t))


def contraction_form():
    """1 = fp16x3 form of the 12-wave contraction kernel, 0 = bf16x6 (ix_gemm_set_x3 / IX_GEMM_KERNEL); + 2 in the
    single-pass 16-bit mode, + 4 in the 16-bit activation mode (a captured graph must not replay another form: episode._graph_stamp)"""
    lib = _L()
    cur = lib.ix_gemm_set_x3(1)
    lib.ix_gemm_set_x3(cur)
    return cur + {"f32": 0, "single_pass": 2, "bf16": 4, "bf16_fusion": 6}[COMPUTE_DTYPE]


# MODEL.COMPUTE_DTYPE -- a property of a MODEL (episode._EpisodeModel.compute_dtype), put in force for the duration of each of its
# entry points by `compute_mode` (round 5 set it process-wide at build time: a second model built without the key silently
# switched the first one).
#   "f32" (default): every contraction is fp32-grade (three fp16 / six bf16 matrix instructions per product) -- the parity path
#       and every headline number.
#   "bf16": the 16-bit ACTIVATION mode (BASELINE.json configs[1] "multi_frame_baseline ... bf16"; b16.py): activations live in HBM
#       as bf16, contractions run on csrc/gemm16.hip (operands by LDS-DMA, one bf16 matrix instruction per k-slice, fp32
#       accumulation), parameters / statistics / accumulations stay fp32.  Checked at SURVEY 8d's bf16 tolerances.
#   "single_pass" (round 4's "bf16", also "fp16"): fp32 STORAGE, contractions round each operand once to 16 bits (an fp16 value
#       of x * 2^-E with one exponent per 32 x 32 sub-block) and issue ONE matrix instruction per k-slice.
COMPUTE_DTYPE = "f32"
_DTYPE_NAMES = {"float32": "f32", "fp32": "f32", "f32": "f32", "bf16": "bf16", "bfloat16": "bf16", "fp16": "single_pass",
                "half": "single_pass", "f16": "single_pass", "single_pass": "single_pass", "bf16_fusion": "bf16_fusion"}


def normalize_compute_dtype(name):
    out = _DTYPE_NAMES.get(str(name).lower())
    if out is None:
        raise ValueError("MODEL.COMPUTE_DTYPE must be f32, bf16 (16-bit activations), bf16_fusion (16-bit activations in the fusion transformer only) "
                         "or single_pass / fp16 (fp32 storage, 16-bit single-pass contractions)")
    return out


def set_compute_dtype(name):
    """-> the previous mode.  Prefer `compute_mode` (scoped); models apply their own mode at every entry point."""
    global COMPUTE_DTYPE
    name = normalize_compute_dtype(name)
    _chk(0 if _L().ix_gemm_set_single_pass(1 if name == "single_pass" else 0) in (0, 1) else 1, "ix_gemm_set_single_pass")
    old, COMPUTE_DTYPE = COMPUTE_DTYPE, name
    return old


class compute_mode:
    """with compute_mode("bf16"): ...   -- the arithmetic mode of the launches issued inside (re-entrant, restores on exit)"""

    def __init__(self, name):
        self.name = normalize_compute_dtype(name)

    def __enter__(self):
        self.prev = COMPUTE_DTYPE
        if self.prev != self.name:
            set_compute_dtype(self.name)
        return self

    def __exit__(self, *exc):
        if COMPUTE_DTYPE != self.prev:
            set_compute_dtype(self.prev)
        return False


def b16_active():
    return COMPUTE_DTYPE == "bf16"


def b16_fusion_only():
    """MODEL.COMPUTE_DTYPE bf16_fusion: the detector computes fp32-grade, the fusion transformer casts its inputs to bf16 (fusion._token_inputs)"""
    return COMPUTE_DTYPE == "bf16_fusion"


_bias_cache = {}
_gc_was_on = [True]
_capture = [None]   # capture-local cache of additive key biases while a HIP-graph capture is running


def capture_begin(salt):
    """Called around a HIP-graph capture of launches from this module (graphs.ChunkGraphs).  `salt`: int64 device tensor every
    dropout kernel of the capture XORs into its seed at run time (ix_set_dropout_salt), so that a replay draws fresh masks.
    Tensors cached across calls must not be created inside a capture (their kernels only run at replay) nor evicted while a
    graph reads them by address: the key-bias cache is replaced by a private one for the duration.
    The cyclic garbage collector is emptied first and held off until capture_end: a dead ChunkGraphs <-> model cycle of an
    earlier signature owns CUDAGraph objects and pool memory, and torch 2.10's torch.cuda.graph no longer collects before a
    capture -- their destructors running on whichever thread trips the collector mid-capture abort the process
    (gpurun_out r4r: "Fatal Python error: Aborted / Garbage-collecting" inside graphs.capture, 5 of 5 runs on one box)."""
    _gc_was_on[0] = _gc.isenabled()
    _gc.collect()
    _gc.disable()
    _capture[0] = {}
    _chk(_L().ix_set_dropout_salt(salt.data_ptr() if salt is not None else None), "ix_set_dropout_salt")


def capture_end():
    _capture[0] = None
    if _gc_was_on[0]:
        _gc.enable()
    _chk(_L().ix_set_dropout_salt(None), "ix_set_dropout_salt")


# dropout: the mask is a pure function of (seed, element index), so the same Function is its own adjoint
_seed_state = {"base": 0x5EED, "counter": 0}


def manual_seed(seed):
    _seed_state["base"] = int(seed) & 0xFFFFFFFF
    _seed_state["counter"] = 0


def _next_seed():
    _seed_state["counter"] += 1
    return ((_seed_state["base"] << 32) ^ (_seed_state["counter"] * 0x9E3779B97F4A7C15)) & 0x7FFFFFFFFFFFFFFF   # (63 bits: torch.profiler cannot record larger Python ints)
