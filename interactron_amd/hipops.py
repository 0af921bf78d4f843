"""torch.autograd bindings of the HIP kernels (C-ABI in include/interactron_hip.h).

Every compute op of the hot path is a ``torch.autograd.Function`` whose forward *and* backward launch
hand-written gfx950 kernels through ctypes; PyTorch only provides device memory, streams and the autograd
tape.  The backward of each Function is itself expressed with Functions from this file, so the op set is closed
under differentiation: ``torch.autograd.grad(..., create_graph=True)`` followed by ``.backward()`` -- the MAML
meta-gradient of reference models/interactron.py:99-123 -- runs entirely on these kernels.

No CPU fallback exists: calling any op without the built library or with CPU tensors raises.
"""
import ctypes
import gc as _gc
import os
from collections import namedtuple

import torch
from torch.autograd import Function as _TorchFunction
from torch.autograd.function import once_differentiable

from . import _lib

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
            from . import b16
            if cls.b16_twin is not None:
                return cls.b16_twin.apply(*args)
            return b16.adapt(cls, args, lambda up: super(Function, cls).apply(*up))
        return super().apply(*args)

    @classmethod
    def call(cls, *args):
        if torch.is_grad_enabled():
            return cls.apply(*args)
        if _b16_seen[0] and cls.b16 != "native" and _any_b16(args):
            from . import b16
            if cls.b16_twin is not None:
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


# ---------------------------------------------------------------------------------------------------------
# GEMM
# ---------------------------------------------------------------------------------------------------------
# A strided matrix view into a flat tensor: elem(r, c) = base[offset + bo*so + bi*si + (c*ld + r if trans else r*ld + c)]
View = namedtuple("View", "offset ld trans so si")
# One batched contraction C = alpha * A(MxK) B(KxN): views for A, B and for C inside a fresh tensor `out_shape`
# (odt: dtype of C in the 16-bit mode -- None = bf16 when an operand is bf16; torch.float32: heads, parameter gradients)
GemmSpec = namedtuple("GemmSpec", "M N K bo bi A B C out_shape alpha odt", defaults=(None,))


def attn_pitch(S):
    """Row pitch (floats) of the [L, S] attention tensors: rows start on 128-byte lines.  Measured with the C-tile store
    pattern of the score product (tools/tile_fill_probe.py): rows that straddle cache lines (pitch 2060) drain at 3.4
    TB/s chip-wide, aligned rows at 5.6 -- the K = 64 score product sits exactly on that limit."""
    return (S + 31) // 32 * 32


def _flip(v):
    return View(v.offset, v.ld, not v.trans, v.so, v.si)


def _numel(shape):
    n = 1
    for s in shape:
        n *= s
    return n


# Pre-split fp16x3 contraction route (csrc/gemm_x3.hip), opt-in: IX_GEMM_X3=1.  Measured on the step's shapes
# (tools/x3_bench.py, profiles/r2b_x3_bench.json): 1.2-1.4x the bf16x6 kernel where both operands are stored with the
# contracted index contiguous and N >= 1024 (the operand-split passes amortise over N), slower elsewhere -- so only those
# shapes are routed, and the default stays bf16x6 for every contraction.
GEMM_X3 = os.environ.get("IX_GEMM_X3", "0") == "1"
_ws_bytes = {}   # contraction signature -> scratch bytes (split-K planes; fp16x3 operand planes on the opt-in route)

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


def _gemm_workspace_bytes(pa, pb, sp, presplit=True):
    """presplit=False: the caller's entry point never takes the opt-in pre-split route (row-sum / BN-fused contractions): size
    the scratch for ITS plan (split-K planes), not for the fp16 operand planes of a route it will not use."""
    x3 = presplit and GEMM_X3 and not sp.A.trans and sp.B.trans and sp.N >= 1024 and sp.M >= 1024
    key = (sp.M, sp.N, sp.K, sp.bo, sp.bi, sp.A.trans, sp.B.trans, sp.A.ld, sp.B.ld, sp.A.so, sp.B.so, pa & 15, pb & 15, x3)
    n = _ws_bytes.get(key)
    if n is None:
        out = ctypes.c_size_t(0)
        L = _L()
        L.ix_gemm_presplit_enable(1 if x3 else 0)
        _chk(L.ix_workspace_bytes_gemm_f32(sp.M, sp.N, sp.K, 0 if sp.A.trans else 1, 1 if sp.B.trans else 0, sp.A.ld,
                                           sp.B.ld, sp.bo, sp.bi, sp.A.so, sp.B.so, pa, pb, 0, 0, ctypes.byref(out)),
             "ix_workspace_bytes_gemm_f32")
        L.ix_gemm_presplit_enable(0)
        n = _ws_bytes[key] = (out.value, x3)
    return n


# ---- activation x WEIGHT PLANES (csrc/gemm_wp.hip) ---------------------------------------------------------------------------
# A Linear layer's forward and input gradient multiply an activation by a WEIGHT.  The 12-wave kernel converts both fp32
# operands to fp16 planes in its producer waves, once per output tile; a weight is read by every row tile of every
# contraction that uses it, several times per step.  Route: the weight is converted ONCE per (tensor, version, orientation) into
# the kernel's own LDS image (ix_wp_split_f32) -- cached on the tensor object, re-made when the tensor changes -- and the
# contraction runs on gemm_wp_kernel (weight planes and raw fp32 activation tiles by LDS-DMA, activation split in the
# consumers' registers, four workgroups per CU).  Same arithmetic class as the fp16x3 form (tests/test_ops_gpu.py::
# test_weight_planes_contraction_*).  Which operands are weights: `mark_weight` (set by linear() / weight_view()).
GEMM_WP = os.environ.get("IX_GEMM_WP", "1") == "1"
WP_MIN_ROWS = int(os.environ.get("IX_GEMM_WP_MIN_ROWS", "8192"))
_wp_stats = {"routed": 0, "splits": 0}
# Cached planes stand for (tensor object, its autograd version, its address, this epoch).  The epoch is bumped by everything that
# rewrites parameters BEHIND autograd's back: the fused Adam kernel and the flat re-homing of trainer.FlatBuffers (raw
# pointers), replica broadcasts, load_state_dict.  Code that edits `p.data` in place by other means calls weights_changed().
_wp_epoch = [0]


def weights_changed():
    _wp_epoch[0] += 1


def mark_weight(w):
    w._ix_weight = True
    return w


def _wp_plan(a, b, bias, sp):
    """-> (k_contig, ld, batch_stride, shared) when this contraction takes the weight-planes route, else None"""
    if not (GEMM_WP and getattr(b, "_ix_weight", False)) or COMPUTE_DTYPE != "f32":
        return None
    A, B, C = sp.A, sp.B, sp.C
    if A.trans or C.trans or sp.K % 32 or sp.M < 128 or sp.N < 128 or sp.alpha != 1.0:
        return None
    if (A.ld | A.offset | A.so | A.si) & 3 or sp.M * A.ld >= (1 << 29) or a.data_ptr() & 15:
        return None
    # the weight view: contiguous [N, K] rows (k-contiguous) or contiguous [K, N] (n-contiguous), one per outer slice or shared
    if B.offset & 3 or B.si != 0 and sp.bi > 1:
        return None
    if B.ld != (sp.K if B.trans else sp.N) or B.so not in (0, sp.N * sp.K):
        return None
    if bias is not None and bias.dim() == 2 and bias.shape[0] != sp.bo:
        return None
    # measured (tools/wp_bench.py, profiles/r4i_wp_bench.txt): 1.25-1.35x the 12-wave kernel on long activations (M >= 12 500
    # rows per slice), 1.15-1.28x at 1805 rows x 16 episodes when N >= 512 and a tie or a loss at N = 256 with K >= 1024 --
    # and a weight is used about once per orientation and weight set per step, so at 1805 rows the split (10-12 % of such a
    # contraction) eats the gain (r4g / r4h: routed launches 40.6 -> 35.9 ms, splits + 6.5 ms).  The route is taken where the
    # split is noise: long activations.  IX_GEMM_WP_MIN_ROWS moves the threshold (tests: 128).
    if sp.M < WP_MIN_ROWS or (sp.N <= 256 and sp.K >= 1024 and sp.M < 8192):
        return None
    if -(-sp.M // 128) * -(-sp.N // 128) * sp.bo * sp.bi < (512 if WP_MIN_ROWS > 128 else 96):
        return None   # too few tiles for four workgroups on each of 256 CUs (the 12-wave kernel is persistent and splits K)
    if torch.cuda.is_current_stream_capturing() and _capture[0] is None:
        return None   # a capture this module was not told about: nowhere safe to keep planes that only exist at replay
    lib = _L()
    if lib.ix_gemm_set_x3(1) != 1:   # the bf16x6 form was asked for (IX_GEMM_KERNEL=x6, the tests' kernel_form): this route is
        lib.ix_gemm_set_x3(0)        # the fp16x3 arithmetic -- leave the contraction to the 12-wave kernel's bf16x6 form
        return None
    return (1 if B.trans else 0, B.ld, B.so, B.so == 0 or sp.bo == 1)


def _wp_planes(b, sp, plan):
    """(planes, unscale) of weight operand `b` for this orientation: cached on the tensor object while its version stands.
    Inside a HIP-graph capture the cache is the capture's own (the split is part of the graph: a replay must redo it, weights
    change between replays while their addresses stay)."""
    kc, ld, so, shared = plan
    nb = 1 if shared else sp.bo
    key = (kc, ld, so, sp.B.offset, sp.N, sp.K, nb, _wp_epoch[0])
    if _capture[0] is not None:
        # (+ the scratch slot = the stream a captured segment replays on: segment C -- slot 1, the side stream -- and segment B
        #  run concurrently at replay, so neither may read planes that the other one's graph writes)
        store, tag = _capture[0], ("wp", _scratch_slot[0], id(b)) + key
    else:
        store = b.__dict__.setdefault("_ix_wp", {})
        tag = key
        if len(store) > 4:   # (superseded epochs / versions of this tensor)
            store.clear()
    hit = store.get(tag)
    if hit is not None and hit[2] == b._version and hit[3] == b.data_ptr():
        return hit[0], hit[1]
    L = _L()
    pb, ub = ctypes.c_size_t(), ctypes.c_size_t()
    _chk(L.ix_wp_planes_bytes(sp.N, sp.K, nb, ctypes.byref(pb), ctypes.byref(ub)), "ix_wp_planes_bytes")
    planes = torch.empty(pb.value, dtype=torch.uint8, device=b.device)
    unscale = torch.empty(ub.value // 4, dtype=torch.float32, device=b.device)
    _chk(L.ix_wp_split_f32(b.data_ptr() + sp.B.offset * 4, ld, so if not shared else 0, sp.N, sp.K, kc, nb, planes.data_ptr(),
                           unscale.data_ptr(), _stream()), "ix_wp_split_f32")
    store[tag] = (planes, unscale, b._version, b.data_ptr(), b if _capture[0] is not None else None)
    _wp_stats["splits"] += 1
    return planes, unscale


def _run_gemm(a, b, bias, sp, fill=True):
    """fill=False: the caller guarantees nobody reads the elements of `out` the product does not write (the pad columns
    of attention tensors: every consumer stops at the row length) -- saves a memset of the whole tensor."""
    covered = not fill or sp.bo * sp.bi * sp.M * sp.N == _numel(sp.out_shape)
    out = (torch.empty if covered else torch.zeros)(sp.out_shape, device=a.device, dtype=torch.float32)
    assert not sp.C.trans
    esz = 4
    pa, pb = a.data_ptr() + sp.A.offset * esz, b.data_ptr() + sp.B.offset * esz
    plan = _wp_plan(a, b, bias, sp)
    if plan is not None:
        planes, unscale = _wp_planes(b, sp, plan)
        _wp_stats["routed"] += 1
        _chk(_L().ix_gemm_wp_f32(pa, sp.A.ld, sp.A.so, sp.A.si, planes.data_ptr(), unscale.data_ptr(), 1 if plan[3] else 0,
                                 out.data_ptr() + sp.C.offset * esz, sp.C.ld, sp.C.so, sp.C.si,
                                 bias.data_ptr() if bias is not None else None,
                                 sp.N if (bias is not None and bias.dim() == 2) else 0, sp.M, sp.N, sp.K, sp.bo, sp.bi, sp.alpha,
                                 _stream()), "ix_gemm_wp_f32")
        return out
    nws, x3 = _gemm_workspace_bytes(pa, pb, sp)
    ws = _workspace(nws, a.device) if nws else None
    L = _L()
    if x3:
        L.ix_gemm_presplit_enable(1)
    rc = L.ix_gemm_f32_ws(pa, pb,
                          out.data_ptr() + sp.C.offset * esz, bias.data_ptr() if bias is not None else None,
                          sp.M, sp.N, sp.K, 0 if sp.A.trans else 1, 1 if sp.B.trans else 0,
                          sp.A.ld, sp.B.ld, sp.C.ld, sp.bo, sp.bi, sp.A.so, sp.A.si, sp.B.so, sp.B.si, sp.C.so, sp.C.si,
                          sp.N if (bias is not None and bias.dim() == 2) else 0, sp.alpha, 0, 0,
                          ws.data_ptr() if nws else None, nws, _stream())
    if x3:
        L.ix_gemm_presplit_enable(0)
    _chk(rc, "ix_gemm_f32_ws")
    return out


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
    return mark_weight(v)


def _is_unwanted(key, skip):
    """`key` stands for one Parameter (an id) or for several (a tuple of ids: cat_params) -- all of them have to be named"""
    if skip is None:
        return False
    if isinstance(key, tuple):
        return all(k in skip for k in key)
    return key in skip


class CatParams(Function):
    """torch.cat(params, 0) whose result keeps standing for its sources in skip_param_grads (fusion key / query / value
    projections evaluated as one contraction); the gradient goes back as row views."""

    @staticmethod
    def forward(ctx, *ws):
        ctx.sizes = [w.shape[0] for w in ws]
        out = torch.cat(ws, 0)
        keys = []
        for w in ws:
            k = _param_key(w)
            keys.extend(k if isinstance(k, tuple) else (k,))
        out._ix_of_param = tuple(keys)
        return out

    @staticmethod
    def backward(ctx, g):
        return tuple(g.split(ctx.sizes, 0))


def _param_key(t):
    """identity under which skip_param_grads knows a weight operand: the Parameter itself, or the Parameter a SplitRows view
    was cut from"""
    return getattr(t, "_ix_of_param", id(t))


class Gemm(Function):
    """out = alpha * A B (+ bias) for strided views A of `a` and B of `b`; see GemmSpec."""

    b16 = "native"

    @staticmethod
    def forward(ctx, a, b, bias, sp):
        ctx.a_key, ctx.b_key = _param_key(a), _param_key(b)
        ctx.bias_key = _param_key(bias) if bias is not None else None
        if a.dtype == torch.bfloat16 or b.dtype == torch.bfloat16:
            return Gemm._forward_b16(ctx, a, b, bias, sp)
        a, b = _req(a, "gemm A"), _req(b, "gemm B")
        ctx.sp = sp
        ctx.has_bias = bias is not None
        ctx.bias_groups = bias.shape[0] if (bias is not None and bias.dim() == 2) else 0
        if bias is not None:
            bias = _req(bias, "gemm bias")
            assert bias.numel() == (sp.bo if ctx.bias_groups else 1) * sp.N, (tuple(bias.shape), sp.bo, sp.N)
        ctx.a_shape, ctx.b_shape = tuple(a.shape), tuple(b.shape)
        ctx.save_for_backward(a, b)
        return _run_gemm(a, b, bias, sp)

    @staticmethod
    def _forward_b16(ctx, a, b, bias, sp):
        """the 16-bit mode: bf16 operands (an fp32 weight is read through its cached bf16 copy), C in sp.odt; the gradients come back
        in each operand's OWN dtype -- bf16 for activations, fp32 for parameters"""
        from . import b16
        wa, wb = getattr(a, "_ix_weight", False), getattr(b, "_ix_weight", False)
        a, b = b16._reqd(a, "gemm A"), b16._reqd(b, "gemm B")
        if wa:
            mark_weight(a)
        if wb:
            mark_weight(b)
        ctx.sp = sp
        ctx.has_bias = bias is not None
        ctx.bias_groups = bias.shape[0] if (bias is not None and bias.dim() == 2) else 0
        if bias is not None:
            bias = _req(bias, "gemm bias")
            assert bias.numel() == (sp.bo if ctx.bias_groups else 1) * sp.N, (tuple(bias.shape), sp.bo, sp.N)
        ctx.a_shape, ctx.b_shape = tuple(a.shape), tuple(b.shape)
        ctx.save_for_backward(a, b)
        return b16.run_gemm(a, b, bias, sp)

    @staticmethod
    def backward(ctx, dc):
        a, b = ctx.saved_tensors
        sp = ctx.sp
        dc = dc.contiguous()
        skip = _unwanted
        need_a = ctx.needs_input_grad[0] and not _is_unwanted(ctx.a_key, skip)
        need_b = ctx.needs_input_grad[1] and not _is_unwanted(ctx.b_key, skip)
        need_bias = ctx.has_bias and ctx.needs_input_grad[2] and not _is_unwanted(ctx.bias_key, skip)
        # the bias gradient colsum(dC) rides on the weight-gradient contraction dB^T = dC^T A (ix_gemm_rowsum_f32: the
        # A-producer waves of that launch sum the dC tiles they stream anyway) whenever dC is its plain m-contiguous A operand
        fuse = (GEMM_ROWSUM and need_bias and need_b and sp.B.trans and sp.bi == 1 and sp.C.offset == 0
                and (dc.dtype == a.dtype or b.dtype == torch.float32)   # (16-bit mode: bf16 dC and activation, fp32 weight)
                and sp.C.ld == sp.N and (ctx.bias_groups == sp.bo or (ctx.bias_groups == 0 and sp.bo == 1))
                and (sp.bo == 1 or sp.C.so == sp.M * sp.N))
        # (recorded backward: dC feeds up to three nodes -- one alias each, so that ITS gradient is one sum, Fanout)
        dcs = list(fanout(dc, int(need_a) + int(need_b) + int(need_bias and not fuse)))
        da, db = _gemm_backward(sp, a, b, ctx.a_shape, ctx.b_shape, dc, need_a, need_b and not fuse, dcs)
        dbias = None
        if fuse:
            s = GemmSpec(sp.N, sp.K, sp.M, sp.bo, sp.bi, _flip(sp.C), sp.A,
                         View(sp.B.offset, sp.B.ld, False, sp.B.so, sp.B.si), ctx.b_shape, sp.alpha)
            if dc.dtype == torch.bfloat16 or a.dtype == torch.bfloat16:
                from . import b16
                fused = b16.gemm_rowsum(dcs[-1], a, s, ctx.bias_groups)
                if fused is None:   # (rows not 16-byte aligned: the two separate nodes)
                    fuse = False
                    db = _gemm_backward(sp, a, b, ctx.a_shape, ctx.b_shape, dc, False, True, [dcs.pop()])[1]
                    d = dc
                    dbias = ColSum.call(d.reshape(ctx.bias_groups, -1, sp.N) if ctx.bias_groups else d.reshape(-1, sp.N))
                else:
                    dcs.pop()
                    db, dbias = fused
                return da, db, dbias, None
            db, dbias = GemmRowsum.call(dcs.pop(), a, s, ctx.bias_groups)
        elif need_bias:
            d = dcs.pop()
            dbias = ColSum.call(d.reshape(ctx.bias_groups, -1, sp.N) if ctx.bias_groups else d.reshape(-1, sp.N))
        return da, db, dbias, None


def _gemm_backward(sp, a, b, a_shape, b_shape, dc, need_a, need_b, dcs=None):
    """Gradients of C = alpha A B w.r.t. the storage of A and of B (each again one strided contraction).  `dcs`: aliases of dC
    to consume, one per node (hipops.fanout), or None."""
    da = db = None
    # (16-bit mode: a gradient has its operand's own dtype -- bf16 activations, fp32 parameters; None outside the mode)
    mixed = dc.dtype == torch.bfloat16 or a.dtype == torch.bfloat16 or b.dtype == torch.bfloat16
    oa, ob = (a.dtype, b.dtype) if mixed else (None, None)
    if need_a:
        dc = dcs.pop() if dcs else dc
        if not sp.A.trans:   # dA (MxK) = alpha * dC (MxN) * B^T (NxK)
            s = GemmSpec(sp.M, sp.K, sp.N, sp.bo, sp.bi, sp.C, _flip(sp.B),
                         View(sp.A.offset, sp.A.ld, False, sp.A.so, sp.A.si), a_shape, sp.alpha, oa)
            da = Gemm.call(dc, b, None, s)
        else:                # storage holds A^T (KxM): dA^T = alpha * B (KxN) * dC^T (NxM)
            s = GemmSpec(sp.K, sp.M, sp.N, sp.bo, sp.bi, sp.B, _flip(sp.C),
                         View(sp.A.offset, sp.A.ld, False, sp.A.so, sp.A.si), a_shape, sp.alpha, oa)
            da = Gemm.call(b, dc, None, s)
    if need_b:
        dc = dcs.pop() if dcs else dc
        if not sp.B.trans:   # dB (KxN) = alpha * A^T (KxM) * dC (MxN)
            s = GemmSpec(sp.K, sp.N, sp.M, sp.bo, sp.bi, _flip(sp.A), sp.C,
                         View(sp.B.offset, sp.B.ld, False, sp.B.so, sp.B.si), b_shape, sp.alpha, ob)
            db = Gemm.call(a, dc, None, s)
        else:                # storage holds B^T (NxK): dB^T = alpha * dC^T (NxM) * A (MxK)
            s = GemmSpec(sp.N, sp.K, sp.M, sp.bo, sp.bi, _flip(sp.C), sp.A,
                         View(sp.B.offset, sp.B.ld, False, sp.B.so, sp.B.si), b_shape, sp.alpha, ob)
            db = Gemm.call(dc, a, None, s)
    return da, db


GEMM_ROWSUM = os.environ.get("IX_GEMM_ROWSUM", "1") == "1"   # "0": bias gradients by a separate ix_colsum_f32 launch (A/B runs)


class GemmRowsum(Function):
    """(alpha A B, rowsum(A)) for a contiguous m-fastest A view of `a` (spec A.trans, offset 0): one launch of
    ix_gemm_rowsum_f32.  rowsum: [M], or [groups, M] for per-episode operands."""

    @staticmethod
    def forward(ctx, a, b, sp, groups):
        a, b = _req(a, "gemm A"), _req(b, "gemm B")
        assert sp.A.trans and sp.A.offset == 0 and sp.A.ld == sp.M and sp.bi == 1 and not sp.C.trans
        ctx.sp, ctx.groups = sp, groups
        ctx.a_shape, ctx.b_shape = tuple(a.shape), tuple(b.shape)
        ctx.save_for_backward(a, b)
        covered = sp.bo * sp.M * sp.N == _numel(sp.out_shape)
        out = (torch.empty if covered else torch.zeros)(sp.out_shape, device=a.device, dtype=torch.float32)
        rs = torch.empty((groups, sp.M) if groups else (sp.M,), device=a.device, dtype=torch.float32)
        pa, pb = a.data_ptr(), b.data_ptr() + sp.B.offset * 4
        nws, _ = _gemm_workspace_bytes(pa, pb, sp, presplit=False)
        ws = _workspace(nws, a.device) if nws else None
        _chk(_L().ix_gemm_rowsum_f32(pa, pb, out.data_ptr() + sp.C.offset * 4,
                                     sp.M, sp.N, sp.K, 0, 1 if sp.B.trans else 0, sp.A.ld, sp.B.ld, sp.C.ld, sp.bo, sp.A.so,
                                     sp.B.so, sp.C.so, sp.alpha, rs.data_ptr(), sp.M, ws.data_ptr() if nws else None, nws,
                                     _stream()), "ix_gemm_rowsum_f32")
        return out, rs

    @staticmethod
    def backward(ctx, gc, gr):
        a, b = ctx.saved_tensors
        sp = ctx.sp
        da = db = None
        if gc is not None:
            da, db = _gemm_backward(sp, a, b, ctx.a_shape, ctx.b_shape, gc.contiguous(), ctx.needs_input_grad[0],
                                    ctx.needs_input_grad[1])
        if gr is not None and ctx.needs_input_grad[0]:   # rowsum[m] = sum_k A(m, k): every k line of A's storage gets gr
            if da is None:
                da = torch.zeros(ctx.a_shape, device=a.device, dtype=torch.float32)
            da = AddRowVec.call(da, gr.contiguous(), max(ctx.groups, 1))
        return da, db, None, None


def linear(x, weight, bias=None, out_dtype=None):
    """y[..., o] = sum_i x[..., i] * weight[o, i] + bias[o]   (nn.Linear semantics).

    Episode-batched form: weight [E, N, K] (+ bias [E, N]) holds one set of MAML fast weights per episode and the
    leading dim of x is E * (rows per episode); episode e's rows meet episode e's weights in ONE batched launch.
    out_dtype (16-bit mode only): torch.float32 for a result that leaves the 16-bit part of the graph (heads)."""
    mark_weight(weight)
    if x.dtype != torch.bfloat16:
        out_dtype = None
    if weight.dim() == 3:
        E, N, K = weight.shape
        assert x.shape[-1] == K and x.numel() % (E * K) == 0, (tuple(x.shape), tuple(weight.shape))
        R = x.numel() // (E * K)
        sp = GemmSpec(R, N, K, E, 1, View(0, K, False, R * K, 0), View(0, K, True, N * K, 0), View(0, N, False, R * N, 0),
                      tuple(x.shape[:-1]) + (N,), 1.0, out_dtype)
        return Gemm.call(x, weight, bias, sp)
    K = x.shape[-1]
    N = weight.shape[0]
    R = x.numel() // K
    out_shape = tuple(x.shape[:-1]) + (N,)
    sp = GemmSpec(R, N, K, 1, 1, View(0, K, False, 0, 0), View(0, K, True, 0, 0), View(0, N, False, 0, 0),
                  out_shape, 1.0, out_dtype)
    return Gemm.call(x, weight, bias, sp)


def matmul_nn(a, b):
    """[M,K] @ [K,N] for plain contiguous 2-D tensors."""
    M, K = a.shape
    N = b.shape[1]
    sp = GemmSpec(M, N, K, 1, 1, View(0, K, False, 0, 0), View(0, N, False, 0, 0), View(0, N, False, 0, 0), (M, N), 1.0)
    return Gemm.call(a, b, None, sp)


def attention_scores(q, k, nbatch, heads, L, S, hd, q_ld, k_ld, q_off, k_off, scale):
    """scores[b,h,l,s] = scale * sum_d q[b,l,h*hd+d] * k[b,s,h*hd+d]  -> [nbatch, heads, L, Sp], Sp = attn_pitch(S).

    q / k are [nbatch, L|S, ld] activations (possibly packed side by side: offsets q_off / k_off inside a row)."""
    Sp = attn_pitch(S)
    sp = GemmSpec(L, S, hd, nbatch, heads, View(q_off, q_ld, False, L * q_ld, hd), View(k_off, k_ld, True, S * k_ld, hd),
                  View(0, Sp, False, heads * L * Sp, L * Sp), (nbatch, heads, L, Sp), scale)
    return Gemm.call(q, k, None, sp)


def attention_apply(p, v, nbatch, heads, L, S, hd, v_ld, v_off):
    """out[b,l,h*hd+d] = sum_s p[b,h,l,s] * v[b,s,h*hd+d]  -> [nbatch, L, heads*hd]."""
    Sp = p.shape[-1]
    E = heads * hd
    sp = GemmSpec(L, hd, S, nbatch, heads, View(0, Sp, False, heads * L * Sp, L * Sp),
                  View(v_off, v_ld, False, S * v_ld, hd), View(0, E, False, L * E, hd), (nbatch, L, E), 1.0)
    return Gemm.call(p, v, None, sp)


# ---------------------------------------------------------------------------------------------------------
# attention core as ONE autograd node (and its backward as one node)
# ---------------------------------------------------------------------------------------------------------
def _spec_dA(sp, a_shape):
    """spec of dA for C = alpha A B with A stored untransposed: dA (MxK) = alpha dC (MxN) B^T; operands (dC, b)."""
    assert not sp.A.trans
    return GemmSpec(sp.M, sp.K, sp.N, sp.bo, sp.bi, sp.C, _flip(sp.B), View(sp.A.offset, sp.A.ld, False, sp.A.so, sp.A.si),
                    a_shape, sp.alpha)


def _spec_dB(sp, b_shape):
    """spec of dB: B stored untransposed -> dB (KxN) = alpha A^T dC, operands (a, dC); B stored transposed (NxK rows)
    -> dB^T = alpha dC^T A, operands (dC, a).  Returns (spec, dc_first)."""
    out = View(sp.B.offset, sp.B.ld, False, sp.B.so, sp.B.si)
    if not sp.B.trans:
        return GemmSpec(sp.K, sp.N, sp.M, sp.bo, sp.bi, _flip(sp.A), sp.C, out, b_shape, sp.alpha), False
    return GemmSpec(sp.N, sp.K, sp.M, sp.bo, sp.bi, _flip(sp.C), sp.A, out, b_shape, sp.alpha), True


def _attn_specs(g):
    Sp = attn_pitch(g.S)
    E = g.heads * g.hd
    tt = View(0, Sp, False, g.heads * g.L * Sp, g.L * Sp)
    scores = GemmSpec(g.L, g.S, g.hd, g.n, g.heads, View(g.q_off, g.q_ld, False, g.L * g.q_ld, g.hd),
                      View(g.k_off, g.k_ld, True, g.S * g.k_ld, g.hd), tt, (g.n, g.heads, g.L, Sp), g.scale)
    apply_ = GemmSpec(g.L, g.hd, g.S, g.n, g.heads, tt, View(g.v_off, g.v_ld, False, g.S * g.v_ld, g.hd),
                      View(0, E, False, g.L * E, g.hd), (g.n, g.L, E), 1.0)
    return scores, apply_, Sp


AttnGeom = namedtuple("AttnGeom", "n heads L S hd q_ld k_ld q_off k_off v_ld v_off scale")


def _sum2(a, b):
    if a is None:
        return b
    if b is None:
        return a
    return Axpby.call(a, b, 1.0, 1.0)


ATTN_LEAN_BYTES = 4 << 30   # [L, S] tensors at least this large: AttentionCore keeps two of them per layer instead of four


def _dropout_of(y, p, seed):
    """d = dropout(y) with the mask of (seed, flat index) -- what ix_attn_prob_fwd_f32 wrote as its second output."""
    if p <= 0.0:
        return y
    d = torch.empty_like(y)
    _chk(_L().ix_dropout_f32(y.data_ptr(), d.data_ptr(), y.numel(), p, seed, _stream()), "ix_dropout_f32")
    return d


class AttentionCore(Function):
    """out[b,l,h*hd+:] = dropout(softmax(scale q k^T [+ key mask])) v  per (batch, head), as one node.

    Same arithmetic as attention_scores -> Softmax -> dropout -> attention_apply, but the [L, S] tensors meet exactly
    one elementwise kernel per pass (softmax+dropout fused, mask regenerated from the seed), autograd never sums
    [L, S]-sized gradients, and the double backward (AttentionCoreBwd.backward) is written out by hand."""

    @staticmethod
    def forward(ctx, q, k, v, g, mask, p, seed):
        q, k, v = _req(q, "attention q"), _req(k, "attention k"), _req(v, "attention v")
        sp_s, sp_a, Sp = _attn_specs(g)
        y = _run_gemm(q, k, None, sp_s, fill=False)
        d = torch.empty_like(y) if p > 0.0 else None
        _chk(_L().ix_attn_prob_fwd_f32(y.data_ptr(), y.data_ptr(), d.data_ptr() if d is not None else None,
                                       g.n * g.heads * g.L, g.S, Sp, mask.data_ptr() if mask is not None else None,
                                       g.heads * g.L, mask.shape[-1] if mask is not None else 0, p, seed, _stream()),
             "ix_attn_prob_fwd_f32")
        if d is None:
            d = y
        out = _run_gemm(d, v, None, sp_a)
        # lean mode (800x800 frames: one [L, S] tensor is 5 GB per episode): keep y only, regenerate d = dropout(y)
        # and gs = softmax_bwd(y, dropout_bwd(gd)) where they are needed -- two saved [L, S] tensors per layer, not four
        ctx.lean = ATTN_LEAN_BYTES is not None and y.numel() * 4 >= ATTN_LEAN_BYTES
        ctx.g, ctx.p, ctx.seed = g, p, seed
        if ctx.lean and d is not y:
            ctx.save_for_backward(q, k, v, y)
        else:
            ctx.save_for_backward(q, k, v, y, d)
        return out

    @staticmethod
    def backward(ctx, do):
        if len(ctx.saved_tensors) == 4:
            q, k, v, y = ctx.saved_tensors
            d = None
        else:
            q, k, v, y, d = ctx.saved_tensors
        gq, gk, gv = AttentionCoreBwd.call(q, k, v, y, d, do, ctx.g, ctx.p, ctx.seed, ctx.lean)
        return gq, gk, gv, None, None, None, None


class AttentionCoreBwd(Function):
    @staticmethod
    def forward(ctx, q, k, v, y, d, do, g, p, seed, lean=False):
        do = _req(do.contiguous(), "attention dO")
        sp_s, sp_a, Sp = _attn_specs(g)
        rows = g.n * g.heads * g.L
        if d is None:
            d = _dropout_of(y, p, seed)
        gd = _run_gemm(do, v, None, _spec_dA(sp_a, tuple(y.shape)), fill=False)                 # dO v^T            [n,H,L,Sp]
        gs = torch.empty_like(y)
        _chk(_L().ix_attn_prob_bwd_f32(y.data_ptr(), gd.data_ptr(), gs.data_ptr(), rows, g.S, Sp, p, seed, _stream()),
             "ix_attn_prob_bwd_f32")
        gq = _run_gemm(gs, k, None, _spec_dA(sp_s, tuple(q.shape)))                 # scale gs k
        s_k, dc_first = _spec_dB(sp_s, tuple(k.shape))
        gk = _run_gemm(gs, q, None, s_k) if dc_first else _run_gemm(q, gs, None, s_k)   # scale gs^T q
        s_v, dc_first_v = _spec_dB(sp_a, tuple(v.shape))
        gv = _run_gemm(do, d, None, s_v) if dc_first_v else _run_gemm(d, do, None, s_v)  # d^T dO
        ctx.g, ctx.p, ctx.seed, ctx.lean = g, p, seed, lean
        if lean:
            ctx.save_for_backward(q, k, v, y, do, gd)
        else:
            ctx.save_for_backward(q, k, v, y, do, gd, d, gs)
        return gq, gk, gv

    @staticmethod
    @once_differentiable
    def backward(ctx, hq, hk, hv):
        g, p, seed = ctx.g, ctx.p, ctx.seed
        sp_s, sp_a, Sp = _attn_specs(g)
        rows = g.n * g.heads * g.L
        if ctx.lean:
            q, k, v, y, do, gd = ctx.saved_tensors
            d = gs = None
        else:
            q, k, v, y, do, gd, d, gs = ctx.saved_tensors
        hq = _req(hq.contiguous()) if hq is not None else None
        hk = _req(hk.contiguous()) if hk is not None else None
        hv = _req(hv.contiguous()) if hv is not None else None
        # G = dL/d gs = scale (hq k^T + q hk^T);  HD = dL/d d = dO hv^T
        G2 = None
        if hq is not None and hk is not None:
            # both terms as ONE product over a doubled head dim, [hq | q] [k | hk]^T: the two small operand copies
            # replace a whole [L, S] tensor written by one GEMM and read again by the kernel below
            heads = lambda t, off, ld, rows: t.view(g.n, rows, ld)[..., off:off + g.heads * g.hd].reshape(g.n, rows, g.heads, g.hd)
            a2 = torch.cat([heads(hq, g.q_off, g.q_ld, g.L), heads(q, g.q_off, g.q_ld, g.L)], -1)
            b2 = torch.cat([heads(k, g.k_off, g.k_ld, g.S), heads(hk, g.k_off, g.k_ld, g.S)], -1)
            E2 = 2 * g.heads * g.hd
            sp2 = GemmSpec(g.L, g.S, 2 * g.hd, g.n, g.heads, View(0, E2, False, g.L * E2, 2 * g.hd),
                           View(0, E2, True, g.S * E2, 2 * g.hd), sp_s.C, sp_s.out_shape, g.scale)
            G1 = _run_gemm(a2, b2, None, sp2, fill=False)
            del a2, b2
        else:
            G1 = _run_gemm(hq, k, None, sp_s, fill=False) if hq is not None else (_run_gemm(q, hk, None, sp_s, fill=False) if hk is not None else None)
        HD = _run_gemm(do, hv, None, _spec_dA(sp_a, tuple(y.shape)), fill=False) if hv is not None else None
        HgD, HS = torch.empty_like(y), torch.empty_like(y)
        nul = lambda t: t.data_ptr() if t is not None else None
        _chk(_L().ix_attn_prob_bwd_bwd_f32(nul(G1), nul(G2), y.data_ptr(), gd.data_ptr(), nul(HD), HgD.data_ptr(),
                                           HS.data_ptr(), rows, g.S, Sp, p, seed, _stream()), "ix_attn_prob_bwd_bwd_f32")
        del G1, G2, HD
        s_q = _spec_dA(sp_s, tuple(q.shape))
        s_k, kf = _spec_dB(sp_s, tuple(k.shape))
        s_v, vf = _spec_dB(sp_a, tuple(v.shape))
        rk = lambda dc, a: _run_gemm(dc, a, None, s_k) if kf else _run_gemm(a, dc, None, s_k)
        need = ctx.needs_input_grad
        grad_q = grad_k = grad_v = grad_do = None
        if gs is None and (need[0] or need[1]):
            gs = torch.empty_like(y)
            _chk(_L().ix_attn_prob_bwd_f32(y.data_ptr(), gd.data_ptr(), gs.data_ptr(), rows, g.S, Sp, p, seed, _stream()),
                 "ix_attn_prob_bwd_f32")
        if need[0]:   # scale (gs hk + HS k)
            grad_q = _sum2(_run_gemm(gs, hk, None, s_q) if hk is not None else None, _run_gemm(HS, k, None, s_q))
        if need[1]:   # scale (gs^T hq + HS^T q)
            grad_k = _sum2(rk(gs, hq) if hq is not None else None, rk(HS, q))
        if need[2]:   # HgD^T dO
            grad_v = _run_gemm(do, HgD, None, s_v) if vf else _run_gemm(HgD, do, None, s_v)
        del gs, HS
        if need[5]:   # d hv + HgD v
            if d is None and hv is not None:
                d = _dropout_of(y, p, seed)
            grad_do = _sum2(_run_gemm(d, hv, None, sp_a) if hv is not None else None, _run_gemm(HgD, v, None, sp_a))
        return grad_q, grad_k, grad_v, None, None, grad_do, None, None, None, None


# ---------------------------------------------------------------------------------------------------------
# flash-style attention core (csrc/flash.hip): no [L, S] tensor in HBM
# ---------------------------------------------------------------------------------------------------------
# "flash": csrc/flash.hip (no [L, S] tensor in HBM; head dims 32 / 64); "materialised": the AttentionCore node above (scores
# and probabilities as [n, H, L, S] tensors).  IX_ATTENTION in the environment overrides the default (A/B runs).
import os as _os
ATTENTION_IMPL = _os.environ.get("IX_ATTENTION", "flash")
# "fp32" (default, the parity path): fp32-grade arithmetic everywhere.  "fp8": the two products of the flash FORWARD kernel
# on OCP e4m3 operands (BASELINE.json configs[4], the 1600 / 200-query stress configuration); derivatives stay fp32-grade.
ATTENTION_DTYPE = _os.environ.get("IX_ATTENTION_DTYPE", "fp32")


def _pad128(R):
    return (R + 127) // 128 * 128


class _PlanesC(ctypes.Structure):   # struct ix_attn_planes of include/interactron_hip.h
    _fields_ = [("row", ctypes.c_void_p), ("unscale", ctypes.c_void_p), ("tr", ctypes.c_void_p), ("tr_form", ctypes.c_int)]


# How the flash kernels run their token-contracting products (P v, dS k, ... and the second-order ones): "f16" = tr form 1, two
# fp16 planes and three matrix instructions per k-slice, the [L, S] intermediates scaled into fp16 range in registers (default
# since round 3); "bf16" = tr form 0, three bf16 planes and six instructions.  Both carry the parity record (tests/conftest.py
# kernel_form).  Read when an operand is split; the derivative passes follow the form their forward was split with.
FLASH_TR = _os.environ.get("IX_FLASH_TR", "f16")
FLASH_SPLIT_DOT = _os.environ.get("IX_FLASH_SPLIT_DOT", "1") == "1"   # "0": delta = dO . O by its own launch (A/B runs)
FLASH_NOBIAS = _os.environ.get("IX_FLASH_NOBIAS", "1") == "1"   # "0": always hand the kernels a key-bias tensor (A/B runs)
_TR_FORMS = {"bf16": 0, "f16": 1}


def flash_m16(on=None):
    """Head dim 64 in the fp16 form runs the 16x16x32 passes of csrc/flash16.hip (row planes only, no tr planes are written);
    ``flash_m16(False)`` switches back to the 32x32x16 passes of csrc/flash.hip (A/B runs, and the tests pin both).  Returns
    the previous setting.  Operands split under one setting must be consumed under the same one."""
    return bool(_L().ix_flash_set_m16(-1 if on is None else int(bool(on))))


def _rows_only(hd, form):
    return hd == 64 and form == 1 and flash_m16()


class AttnPlanes:
    """One attention operand as the flash kernels read it: fp16 row planes [2][n*H][Rp][hd] with their block unscale factors
    [n*H][Rp/32], and tr planes -- bf16 [3][n*H][hd][Rp] (form 0) or fp16 [2][n*H][hd][Rp] (form 1) (csrc/flash.hip);
    ``ref`` is the C view handed to the library."""

    def __init__(self, row, unscale, tr, tr_form=0):
        self.row, self.unscale, self.tr, self.tr_form = row, unscale, tr, tr_form
        self.c = _PlanesC(row.data_ptr() if row is not None else None, unscale.data_ptr() if unscale is not None else None,
                          tr.data_ptr() if tr is not None else None, tr_form)
        self.ref = ctypes.byref(self.c)


def attn_split(x, n, R, ld, off, H, hd, row=True, tr=True, tr_form=None, dot=None):
    """fp32 activations [n, R, ld] (head h at columns off + h*hd) -> AttnPlanes (Rp = R rounded up to 128).
    dot = (y, ldy, offy): also t[n*H, Rp] = sum_d x[.., h, d] * y[.., h, d] from the same read of x -> (AttnPlanes, t)."""
    in16 = x.dtype == torch.bfloat16   # (16-bit activation mode: the same planes from bf16 values, ix_attn_split_*_b16)
    x = _req(x, "attention operand") if not in16 else (x if x.is_contiguous() else x.contiguous())
    Rp = _pad128(R)
    dev = x.device
    form = _TR_FORMS[FLASH_TR] if tr_form is None else tr_form
    if _rows_only(hd, form):
        row, tr = row or tr, False
    rowp = torch.empty(2 * n * H * Rp * hd, dtype=torch.float16, device=dev) if row else None
    us = torch.empty(n * H * (Rp // 32), dtype=torch.float32, device=dev) if row or (tr and form == 1) else None
    trp = None
    if tr:
        trp = (torch.empty(2 * n * H * Rp * hd, dtype=torch.float16, device=dev) if form == 1 else
               torch.empty(3 * n * H * Rp * hd, dtype=torch.bfloat16, device=dev))
    if dot is not None:
        y, ldy, offy = dot
        t = torch.empty(n * H, Rp, dtype=torch.float32, device=dev)
        fn = _L().ix_attn_split_dot_b16 if in16 else _L().ix_attn_split_dot_f32
        _chk(fn(x.data_ptr(), rowp.data_ptr() if row else None, us.data_ptr() if us is not None else None,
                trp.data_ptr() if tr else None, form, n, R, Rp, ld, off, H, hd, _req(y).data_ptr(), ldy, offy,
                t.data_ptr(), _stream()), "ix_attn_split_dot")
        return AttnPlanes(rowp, us, trp, form), t
    if in16:
        return attn_split_multi([(x, R, ld, off, row, tr)], n, H, hd, tr_form=form)[0]
    _chk(_L().ix_attn_split_f32(x.data_ptr(), rowp.data_ptr() if row else None, us.data_ptr() if us is not None else None,
                                trp.data_ptr() if tr else None, form, n, R, Rp, ld, off, H, hd, _stream()), "ix_attn_split_f32")
    return AttnPlanes(rowp, us, trp, form)


def contraction_form():
    """1 = fp16x3 form of the 12-wave contraction kernel, 0 = bf16x6 (ix_gemm_set_x3 / IX_GEMM_KERNEL); + 2 in the
    single-pass 16-bit mode, + 4 in the 16-bit activation mode (a captured graph must not replay another form: episode._graph_stamp)"""
    lib = _L()
    cur = lib.ix_gemm_set_x3(1)
    lib.ix_gemm_set_x3(cur)
    return cur + {"f32": 0, "single_pass": 2, "bf16": 4}[COMPUTE_DTYPE]


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
                "half": "single_pass", "f16": "single_pass", "single_pass": "single_pass"}


def normalize_compute_dtype(name):
    out = _DTYPE_NAMES.get(str(name).lower())
    if out is None:
        raise ValueError("MODEL.COMPUTE_DTYPE must be f32, bf16 (16-bit activations) or single_pass / fp16 (fp32 storage, 16-bit single-pass contractions)")
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


def attn_split_multi(ops, n, H, hd, tr_form=None):
    """attn_split for up to three operands of one attention call in ONE launch.  ops: [(x, R, ld, off, row, tr), ...]."""
    form = _TR_FORMS[FLASH_TR] if tr_form is None else tr_form
    cnt = len(ops)
    xs, rows, uss, trs = [], [], [], []
    rows_only = _rows_only(hd, form)
    in16 = ops[0][0].dtype == torch.bfloat16
    assert all((o[0].dtype == torch.bfloat16) == in16 for o in ops), "the operands of one attention call share a storage dtype"
    for x, R, ld, off, row, tr in ops:
        if rows_only:
            row, tr = row or tr, False
        x = _req(x, "attention operand") if not in16 else (x if x.is_contiguous() else x.contiguous())
        Rp, dev = _pad128(R), x.device
        xs.append(x)
        rows.append(torch.empty(2 * n * H * Rp * hd, dtype=torch.float16, device=dev) if row else None)
        uss.append(torch.empty(n * H * (Rp // 32), dtype=torch.float32, device=dev) if row or (tr and form == 1) else None)
        trs.append(None if not tr else torch.empty(2 * n * H * Rp * hd, dtype=torch.float16, device=dev) if form == 1 else
                   torch.empty(3 * n * H * Rp * hd, dtype=torch.bfloat16, device=dev))
    ptr = lambda ts: (ctypes.c_void_p * cnt)(*[t.data_ptr() if t is not None else None for t in ts])
    ints = lambda vs: (ctypes.c_int * cnt)(*vs)
    fn = _L().ix_attn_split_multi_b16 if in16 else _L().ix_attn_split_multi_f32
    _chk(fn(cnt, ptr(xs), ptr(rows), ptr(uss), ptr(trs), form, n, ints([o[1] for o in ops]),
            ints([_pad128(o[1]) for o in ops]), (ctypes.c_int64 * cnt)(*[o[2] for o in ops]),
            ints([o[3] for o in ops]), H, hd, _stream()), "ix_attn_split_multi")
    return [AttnPlanes(r, u, t, form) for r, u, t in zip(rows, uss, trs)]


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


def attn_bias(mask, n, S, device):
    """additive key bias [n, Sp]: 0 valid / -inf masked or tail (Sp = S rounded up to 128); mask uint8 [n, S] or None.
    Without a mask the bias depends on (n, S) only and is kept (the GPT fusion asks for the same one in every layer of
    every step); with a mask it is kept for as long as the SAME mask tensor (address, version) is presented -- the six
    encoder and six decoder layers of one detector pass share one key_padding_mask."""
    Sb = _pad128(S)
    key = (device.index, n, S) if mask is None else (device.index, n, S, mask.data_ptr(), mask._version)
    if _capture[0] is not None:   # inside a graph capture: a private cache that dies with the capture (see capture_begin)
        hit = _capture[0].get(key)
        if hit is None:
            bias = torch.empty(n, Sb, dtype=torch.float32, device=device)
            _chk(_L().ix_attn_bias_f32(mask.data_ptr() if mask is not None else None, bias.data_ptr(), n, S, Sb,
                                       mask.shape[-1] if mask is not None else 0, _stream()), "ix_attn_bias_f32")
            hit = _capture[0][key] = (bias, mask)
        return hit[0]
    hit = _bias_cache.get(key)
    if hit is not None and (mask is None or hit[1] is mask):
        return hit[0]
    bias = torch.empty(n, Sb, dtype=torch.float32, device=device)
    _chk(_L().ix_attn_bias_f32(mask.data_ptr() if mask is not None else None, bias.data_ptr(), n, S, Sb,
                               mask.shape[-1] if mask is not None else 0, _stream()), "ix_attn_bias_f32")
    if mask is not None:   # one masked entry at a time (masks change with every batch)
        for k in [k for k in _bias_cache if len(k) == 5]:
            del _bias_cache[k]
    _bias_cache[key] = (bias, mask)
    return bias


def flash_dropmask(BH, L, S, p, seed, device="cuda"):
    """The flash kernels' dropout mask as a tensor [BH, L, S] (1/keep or 0) -- for tests."""
    m = torch.empty(BH, L, S, dtype=torch.float32, device=device)
    _chk(_L().ix_flash_dropmask_f32(m.data_ptr(), BH, L, S, p, seed, _stream()), "ix_flash_dropmask_f32")
    return m


def attn_split_fp8(x, n, R, ld, off, H, hd, row=True, tr=False):
    """fp32 activations -> one OCP e4m3 plane in row and / or tr layout + block unscale factors (csrc/flash.hip, fp8 forward)."""
    x = _req(x, "attention operand")
    Rp = _pad128(R)
    rowp = torch.empty(n * H * Rp * hd, dtype=torch.uint8, device=x.device) if row else None
    trp = torch.empty(n * H * Rp * hd, dtype=torch.uint8, device=x.device) if tr else None
    us = torch.empty(n * H * (Rp // 32), dtype=torch.float32, device=x.device)
    _chk(_L().ix_attn_split_fp8_f32(x.data_ptr(), rowp.data_ptr() if row else None, trp.data_ptr() if tr else None,
                                    us.data_ptr(), n, R, Rp, ld, off, H, hd, _stream()), "ix_attn_split_fp8_f32")
    return rowp, trp, us


def _bias_ptr(pl):
    return pl["bias"].data_ptr() if pl["bias"] is not None else None


def flash_forward(q, k, v, g, mask, p, seed, need_backward=True, dtype=None):
    """-> (out [n, L, H*hd], lse [n*H, Lp] (+inf beyond L), operand planes) for geometry g (AttnGeom).
    dtype "fp8": the two forward products on e4m3 operands (ATTENTION_DTYPE); the planes for the derivative kernels are
    the fp16 / bf16 ones either way."""
    dev = q.device
    dtype = dtype or ATTENTION_DTYPE
    fp8 = dtype == "fp8"
    # no key mask + the head-dim-64 fp16-form kernels (csrc/flash16.hip): no bias tensor at all (NULL: they blank the keys beyond
    # S of the last tile themselves and skip the bias loads / adds of every other tile)
    no_bias = FLASH_NOBIAS and mask is None and not fp8 and _rows_only(g.hd, _TR_FORMS[FLASH_TR])
    pl = {"bias": None if no_bias else attn_bias(mask, g.n, g.S, dev)}
    if need_backward or not fp8:
        pl["q"], pl["k"], pl["v"] = attn_split_multi([(q, g.L, g.q_ld, g.q_off, True, need_backward),
                                                      (k, g.S, g.k_ld, g.k_off, True, need_backward),
                                                      (v, g.S, g.v_ld, g.v_off, need_backward, True)], g.n, g.heads, g.hd)
    Lp, Sp, E = _pad128(g.L), _pad128(g.S), g.heads * g.hd
    out = torch.empty(g.n, g.L, E, dtype=torch.float32, device=dev)
    lse = torch.empty(g.n * g.heads, Lp, dtype=torch.float32, device=dev)   # (rows L..Lp come back as +inf: P = 0 there)
    if fp8:
        q8, _, qus = attn_split_fp8(q, g.n, g.L, g.q_ld, g.q_off, g.heads, g.hd)
        k8, _, kus = attn_split_fp8(k, g.n, g.S, g.k_ld, g.k_off, g.heads, g.hd)
        _, v8, vus = attn_split_fp8(v, g.n, g.S, g.v_ld, g.v_off, g.heads, g.hd, row=False, tr=True)
        _chk(_L().ix_flash_fwd_fp8_f32(q8.data_ptr(), qus.data_ptr(), k8.data_ptr(), kus.data_ptr(), v8.data_ptr(), vus.data_ptr(),
                                       pl["bias"].data_ptr(), out.data_ptr(), lse.data_ptr(), g.n, g.heads, g.L, Lp, g.S, Sp,
                                       g.hd, E, 0, g.scale, p, seed, _stream()), "ix_flash_fwd_fp8_f32")
        if need_backward:   # the derivative kernels recompute the scores from the fp16 planes: give them normalisers of those scores
            # (an lse-only pass, out == NULL: always the 32 x 32 forward kernel of csrc/flash.hip -- at head dim 64 the derivative passes
            #  are flash16's, the same fp16x3 products in another accumulation order, so P sums to 1 up to fp32 rounding rather than
            #  exactly; rows-only planes suffice, that kernel reads no tr planes without v.  tests/test_ops_gpu.py::
            #  test_flash_fp8_forward_with_the_derivative_passes_of_either_family, tests/test_b800_gpu.py::test_stress_config_training_step_*)
            _chk(_L().ix_flash_fwd_f32(pl["q"].ref, pl["k"].ref, None, pl["bias"].data_ptr(), None, lse.data_ptr(), g.n, g.heads,
                                       g.L, Lp, g.S, Sp, g.hd, E, 0, g.scale, 0.0, 0, _stream()), "ix_flash_fwd_f32")
        return out, lse, pl
    _chk(_L().ix_flash_fwd_f32(pl["q"].ref, pl["k"].ref, pl["v"].ref, _bias_ptr(pl), out.data_ptr(), lse.data_ptr(),
                               g.n, g.heads, g.L, Lp, g.S, Sp, g.hd, E, 0, g.scale, p, seed, _stream()), "ix_flash_fwd_f32")
    return out, lse, pl


def flash_supported(g):
    return g.hd in (32, 64) and g.q_ld % 4 == 0 and g.k_ld % 4 == 0 and g.v_ld % 4 == 0 and g.q_off % 4 == 0 \
        and g.k_off % 4 == 0 and g.v_off % 4 == 0 and g.n * g.heads <= 65535


class FlashAttention(Function):
    """out[b,l,h*hd+:] = dropout(softmax(scale q k^T [+ key mask])) v per (batch, head) without [L, S] tensors in HBM
    (csrc/flash.hip).  Saves q, k, v, out, the row normalisers and the 16-bit operand planes."""

    @staticmethod
    def forward(ctx, q, k, v, g, mask, p, seed):
        q, k, v = _req(q, "attention q"), _req(k, "attention k"), _req(v, "attention v")
        return FlashAttention._forward(ctx, q, k, v, g, mask, p, seed)

    @staticmethod
    def _forward(ctx, q, k, v, g, mask, p, seed):
        """(shared with the 16-bit twin, b16.FlashAttention16: q / k / v fp32 or bf16; the kernels write the output in fp32)"""
        out, lse, pl = flash_forward(q, k, v, g, mask, p, seed, need_backward=not isinstance(ctx, _NullCtx))
        ctx.g, ctx.p, ctx.seed, ctx.pl = g, p, seed, pl
        # packed projection buffers: [q | k] (nn.MultiheadAttention self-attention) or [k | q | v] (fusion blocks) in one tensor.
        # ONE gradient buffer serves the operands of a shared tensor only when their column ranges are disjoint and cover the
        # rows (the two packed layouts); the same tensor passed with overlapping columns (attention(x, x, x) with equal
        # offsets) gets separate buffers, which autograd then sums.
        E_ = g.heads * g.hd
        alias_qk = q.data_ptr() == k.data_ptr() and q.shape == k.shape
        alias_qv = q.data_ptr() == v.data_ptr() and q.shape == v.shape
        packed3 = alias_qk and alias_qv and g.q_ld == 3 * E_ and g.k_ld == 3 * E_ and g.v_ld == 3 * E_ \
            and sorted((g.q_off, g.k_off, g.v_off)) == [0, E_, 2 * E_]
        packed2 = alias_qk and not alias_qv and g.q_ld == 2 * E_ and g.k_ld == 2 * E_ and sorted((g.q_off, g.k_off)) == [0, E_]
        ctx.same_qk = (packed2 or packed3, packed3)
        ctx.save_for_backward(q, k, v, out, lse)
        return out

    @staticmethod
    def backward(ctx, do):
        q, k, v, out, lse = ctx.saved_tensors
        gq, gk, gv = FlashAttentionBwd.call(q, k, v, out, lse, do, ctx.g, ctx.p, ctx.seed, ctx.pl, ctx.same_qk)
        return gq, gk, gv, None, None, None, None


def _grad_buffers(g, q, k, v, same):
    """Gradient buffers in the operands' own (packed) layouts; columns that belong to other tensors stay zero.
    same = (k shares q's tensor, v shares q's tensor): shared tensors get ONE buffer."""
    same_qk, same_qv = same
    E, dev = g.heads * g.hd, q.device
    full = lambda ld, off: ld == E and off == 0
    packed2 = same_qk and not same_qv and g.q_ld == 2 * E and sorted((g.q_off, g.k_off)) == [0, E]   # [q | k]: fully covered
    packed3 = same_qk and same_qv and g.q_ld == 3 * E and sorted((g.q_off, g.k_off, g.v_off)) == [0, E, 2 * E]
    gq = (torch.empty if packed2 or packed3 or (full(g.q_ld, g.q_off) and not (same_qk or same_qv)) else torch.zeros)(
        q.shape, dtype=torch.float32, device=dev)
    gk = gq if same_qk else (torch.empty if full(g.k_ld, g.k_off) else torch.zeros)(k.shape, dtype=torch.float32, device=dev)
    gv = gq if same_qv else (torch.empty if full(g.v_ld, g.v_off) else torch.zeros)(v.shape, dtype=torch.float32, device=dev)
    return gq, gk, gv


class FlashAttentionBwd(Function):
    """(gq, gk, gv) of FlashAttention; gq / gk come back in the layout of the packed q / k projection buffers (one shared
    buffer when q and k are the same tensor: autograd then has nothing to add)."""

    @staticmethod
    def forward(ctx, q, k, v, out, lse, do, g, p, seed, pl, same_qk):
        do = _req(do.contiguous(), "attention dO")
        return FlashAttentionBwd._forward(ctx, q, k, v, out, lse, do, g, p, seed, pl, same_qk)

    @staticmethod
    def _forward(ctx, q, k, v, out, lse, do, g, p, seed, pl, same_qk):
        """(shared with the 16-bit twin: q / k / v / dO fp32 or bf16, `out` and the gradients fp32)"""
        dev = q.device
        Lp, Sp, E = _pad128(g.L), _pad128(g.S), g.heads * g.hd
        if FLASH_SPLIT_DOT:   # the planes of dO and delta = dO . O (per query and head) from one read of dO
            dop, delta = attn_split(do, g.n, g.L, E, 0, g.heads, g.hd, tr_form=pl["q"].tr_form, dot=(out, E, 0))
        else:
            dop = attn_split(do, g.n, g.L, E, 0, g.heads, g.hd, tr_form=pl["q"].tr_form)
            delta = torch.empty(g.n * g.heads, Lp, dtype=torch.float32, device=dev)
            _chk(_L().ix_attn_rowdot_f32(do.data_ptr(), out.data_ptr(), delta.data_ptr(), g.n, g.heads, g.L, Lp, g.hd, E, 0, E, 0,
                                         _stream()), "ix_attn_rowdot_f32")
        gq, gk, gv = _grad_buffers(g, q, k, v, same_qk)
        _chk(_L().ix_flash_bwd_f32(pl["q"].ref, pl["k"].ref, pl["v"].ref, dop.ref, _bias_ptr(pl), lse.data_ptr(),
                                   delta.data_ptr(), gq.data_ptr(), gk.data_ptr(), gv.data_ptr(), g.n, g.heads, g.L, Lp, g.S, Sp,
                                   g.hd, g.q_ld, g.q_off, g.k_ld, g.k_off, g.v_ld, g.v_off, g.scale, p, seed, _stream()),
             "ix_flash_bwd_f32")
        ctx.g, ctx.p, ctx.seed, ctx.same_qk = g, p, seed, same_qk
        ctx.pl = dict(pl, do=dop, delta=delta)
        ctx.save_for_backward(q, k, v, out, lse, do)
        # a shared buffer carries the gradients of everything packed in it: hand it to q, nothing to the others
        return gq, (None if same_qk[0] else gk), (None if same_qk[1] else gv)

    @staticmethod
    @once_differentiable
    def backward(ctx, hq, hk, hv):
        """Double backward: (dq, dk, dv, ddO) for the cotangents of (gq, gk, gv); three passes of csrc/flash.hip."""
        q, k, v, out, lse, do = ctx.saved_tensors
        g, pl, dev = ctx.g, ctx.pl, q.device
        Lp, Sp, E = _pad128(g.L), _pad128(g.S), g.heads * g.hd
        zeros = lambda t: torch.zeros(t.shape, dtype=torch.float32, device=dev)
        hq = _req(hq.contiguous()) if hq is not None else zeros(q)
        if ctx.same_qk[0]:
            hk = hq          # one cotangent buffer for the packed gradient buffer
        else:
            hk = _req(hk.contiguous()) if hk is not None else zeros(k)
        if ctx.same_qk[1]:
            hv = hq
        else:
            hv = _req(hv.contiguous()) if hv is not None else zeros(v)
        return FlashAttentionBwd._backward_impl(ctx, hq, hk, hv)

    @staticmethod
    def _backward_impl(ctx, hq, hk, hv):
        """(shared with the 16-bit twin: cotangents fp32 or bf16, all of one dtype; the results are fp32)"""
        q, k, v, out, lse, do = ctx.saved_tensors
        g, pl, dev = ctx.g, ctx.pl, q.device
        Lp, Sp, E = _pad128(g.L), _pad128(g.S), g.heads * g.hd
        hqp, hkp, hvp = attn_split_multi([(hq, g.L, g.q_ld, g.q_off, True, True), (hk, g.S, g.k_ld, g.k_off, True, True),
                                          (hv, g.S, g.v_ld, g.v_off, True, True)], g.n, g.heads, g.hd, tr_form=pl["q"].tr_form)
        dq, dk, dv = _grad_buffers(g, q, k, v, ctx.same_qk)
        ddo = torch.empty(g.n, g.L, E, dtype=torch.float32, device=dev)
        need = ctypes.c_size_t()
        _chk(_L().ix_workspace_bytes_flash_bwd_bwd(g.n, g.heads, g.L, ctypes.byref(need)), "ix_workspace_bytes_flash_bwd_bwd")
        ws = torch.empty(need.value // 4, dtype=torch.float32, device=dev)
        _chk(_L().ix_flash_bwd_bwd_f32(pl["q"].ref, pl["k"].ref, pl["v"].ref, pl["do"].ref, hqp.ref, hkp.ref, hvp.ref,
                                       _bias_ptr(pl), lse.data_ptr(), pl["delta"].data_ptr(), dq.data_ptr(),
                                       dk.data_ptr(), dv.data_ptr(), ddo.data_ptr(), g.n, g.heads, g.L, Lp, g.S, Sp, g.hd,
                                       g.q_ld, g.q_off, g.k_ld, g.k_off, g.v_ld, g.v_off, E, 0, g.scale, ctx.p, ctx.seed,
                                       ws.data_ptr(), need.value, _stream()), "ix_flash_bwd_bwd_f32")
        need_in = ctx.needs_input_grad
        return (dq if need_in[0] else None, (None if ctx.same_qk[0] else dk) if need_in[1] else None,
                (None if ctx.same_qk[1] else dv) if need_in[2] else None,
                None, None, ddo if need_in[5] else None, None, None, None, None, None)


def attention(q, k, v, nbatch, heads, L, S, hd, q_ld, k_ld, q_off, k_off, v_ld, v_off, scale, mask, p, training):
    """Scaled-dot-product attention out of packed projection buffers (see attention_scores / attention_apply for the
    layouts); `mask`: optional uint8 key-padding mask [nbatch, S]."""
    p = float(p) if training else 0.0
    g = AttnGeom(nbatch, heads, L, S, hd, q_ld, k_ld, q_off, k_off, v_ld, v_off, float(scale))
    seed = _next_seed() if p > 0.0 else 0
    if ATTENTION_IMPL == "flash" and flash_supported(g):
        return FlashAttention.call(q, k, v, g, mask, p, seed)
    return AttentionCore.call(q, k, v, g, mask, p, seed)


# ---------------------------------------------------------------------------------------------------------
# elementwise / broadcast
# ---------------------------------------------------------------------------------------------------------
class ColSum(Function):
    """[rows, C] -> [C], or grouped [G, rows, C] -> [G, C]."""
    b16_out = False   # (a bias gradient: fp32)

    @staticmethod
    def forward(ctx, x):
        x = _req(x)
        if x.dim() == 3:
            G, rows, C = x.shape
            out = torch.empty(G, C, device=x.device, dtype=torch.float32)
        else:
            (rows, C), G = x.shape, 1
            out = torch.empty(C, device=x.device, dtype=torch.float32)
        ctx.rows = rows
        wp, wn = _reduce_ws("colsum", rows, C, G, x.device)
        _chk(_L().ix_colsum_f32(x.data_ptr(), out.data_ptr(), rows, C, G, wp, wn, _stream()), "ix_colsum_f32")
        return out

    @staticmethod
    def backward(ctx, g):
        return BcastRows.call(g, ctx.rows)


class BcastRows(Function):
    """[C] -> [rows, C], or grouped [G, C] -> [G, rows, C]."""

    @staticmethod
    def forward(ctx, v, rows):
        v = _req(v)
        if v.dim() == 2:
            G, C = v.shape
            out = torch.empty(G, rows, C, device=v.device, dtype=torch.float32)
        else:
            G, C = 1, v.numel()
            out = torch.empty(rows, C, device=v.device, dtype=torch.float32)
        _chk(_L().ix_bcast_rows_f32(v.data_ptr(), out.data_ptr(), rows, C, G, _stream()), "ix_bcast_rows_f32")
        return out

    @staticmethod
    def backward(ctx, g):
        return ColSum.call(g), None


class Axpby(Function):
    """alpha*a + beta*b (same shapes)."""

    @staticmethod
    def forward(ctx, a, b, alpha, beta):
        a, b = _req(a), _req(b)
        assert a.shape == b.shape, (a.shape, b.shape)
        ctx.alpha, ctx.beta = alpha, beta
        out = torch.empty_like(a)
        _chk(_L().ix_axpby_f32(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), alpha, beta, _stream()),
             "ix_axpby_f32")
        return out

    @staticmethod
    def backward(ctx, g):
        ga = g if ctx.alpha == 1.0 else Scale.call(g, ctx.alpha)
        gb = g if ctx.beta == 1.0 else Scale.call(g, ctx.beta)
        return (ga if ctx.needs_input_grad[0] else None), (gb if ctx.needs_input_grad[1] else None), None, None


def add(a, b):
    return Axpby.call(a, b, 1.0, 1.0)


# ---- tensors with several consumers --------------------------------------------------------------------------------------
# The autograd engine sums the gradients of a tensor that feeds n nodes with n - 1 two-operand aten::add launches (each
# reads two tensors and writes one): 800 launches / 13 ms of a 16-episode step were the last stock kernels on the path.
# `fanout(x, n)` hands out n aliases of x whose gradients come back TOGETHER and are summed by one hand-written pass
# (ix_sum_n_f32: n reads, one write, left to right).  Closed under differentiation: the sum's own backward hands its
# cotangent to every operand, no kernel.
FANOUT = os.environ.get("IX_FANOUT", "1") == "1"   # "0": plain aliases, autograd sums (A/B runs)


def sum_n(tensors):
    """((t0 + t1) + t2) + ... over 2..8 tensors of one shape, one launch; longer lists in groups of 8"""
    ts = [_req(t) for t in tensors]
    while len(ts) > 1:
        head, ts = ts[:8], ts[8:]
        if len(head) == 1:
            ts.insert(0, head[0])
            break
        out = torch.empty_like(head[0])
        arr = (ctypes.c_void_p * len(head))(*[t.data_ptr() for t in head])
        _chk(_L().ix_sum_n_f32(arr, len(head), out.data_ptr(), out.numel(), _stream()), "ix_sum_n_f32")
        ts.insert(0, out)
    return ts[0]


class SumN(Function):
    @staticmethod
    def forward(ctx, *xs):
        assert all(x.shape == xs[0].shape for x in xs), [tuple(x.shape) for x in xs]
        return sum_n(xs)

    @staticmethod
    def backward(ctx, g):
        return tuple(g if need else None for need in ctx.needs_input_grad)


class Fanout(Function):
    b16 = "native"   # (aliases: no arithmetic)
    @staticmethod
    def forward(ctx, x, n):
        ctx.set_materialize_grads(False)
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *gs):
        gs = [g for g in gs if g is not None]
        if not gs:
            return None, None
        return (gs[0] if len(gs) == 1 else SumN.call(*gs)), None


def fanout(x, n):
    """n aliases of x for n consumers (x itself n times when nothing is recorded or x needs no gradient)"""
    if n <= 1 or not FANOUT or not torch.is_grad_enabled() or not x.requires_grad:
        return (x,) * n
    return Fanout.apply(x, n)


class Scale(Function):
    @staticmethod
    def forward(ctx, x, alpha):
        x = _req(x)
        ctx.alpha = alpha
        out = torch.empty_like(x)
        _chk(_L().ix_scale_f32(x.data_ptr(), out.data_ptr(), x.numel(), alpha, _stream()), "ix_scale_f32")
        return out

    @staticmethod
    def backward(ctx, g):
        return Scale.call(g, ctx.alpha), None


class AddRowVec(Function):
    """a [R, C] + v [C] broadcast over rows (learned query / position tables shared by all frames)."""

    @staticmethod
    def forward(ctx, a, v, groups=1):
        """groups > 1: v is [groups, C] (one vector per episode) and a is [groups, rows, C] flattened any way."""
        a, v = _req(a), _req(v)
        C = v.numel() // groups
        assert a.numel() % (C * groups) == 0
        out = torch.empty_like(a)
        _chk(_L().ix_add_rowvec_f32(a.data_ptr(), v.data_ptr(), out.data_ptr(), a.numel() // (C * groups), C, groups,
                                    _stream()), "ix_add_rowvec_f32")
        ctx.vshape, ctx.groups = tuple(v.shape), groups
        return out

    @staticmethod
    def backward(ctx, g):
        gv = None
        if ctx.needs_input_grad[1]:
            C = _numel(ctx.vshape) // ctx.groups
            gg = g.reshape(ctx.groups, -1, C) if ctx.groups > 1 else g.reshape(-1, C)
            gv = ColSum.call(gg).reshape(ctx.vshape)
        return g, gv, None


class SplitRows(Function):
    """Row blocks (views, no copies) of a packed parameter -- nn.MultiheadAttention's in_proj_weight / in_proj_bias -- whose
    gradient comes back as ONE concatenation instead of a zero-fill + slice copy per block + an accumulation of the
    full-size pieces (five small launches per block pair in autograd's slice backward)."""

    @staticmethod
    def forward(ctx, w, *sizes):
        ctx.sizes, ctx.tail = sizes, tuple(w.shape[1:])
        parts = tuple(w.split(list(sizes), 0))
        for t in parts:
            t._ix_of_param = id(w)   # (skip_param_grads: the blocks stand for the Parameter they were cut from)
        return parts

    @staticmethod
    def backward(ctx, *gs):
        parts = [g if g is not None else gs_zero(ctx, n, gs) for g, n in zip(gs, ctx.sizes)]
        return (torch.cat(parts, 0),) + (None,) * len(ctx.sizes)


def gs_zero(ctx, n, gs):
    ref = next(g for g in gs if g is not None)
    return torch.zeros((n,) + ctx.tail, device=ref.device, dtype=ref.dtype)


class Dot(Function):
    """sum(a*b) -> 0-d tensor."""
    b16_out = False   # (a scalar)

    @staticmethod
    def forward(ctx, a, b):
        a, b = _req(a), _req(b)
        ctx.save_for_backward(a, b)
        out = torch.empty((), device=a.device, dtype=torch.float32)
        wp, wn = _reduce_ws("scalar", 0, 0, 0, a.device)
        _chk(_L().ix_dot_f32(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), wp, wn, _stream()), "ix_dot_f32")
        return out

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        return ScaleDev.call(b, g), ScaleDev.call(a, g)


class ScaleDev(Function):
    """x * s with s a 0-d device tensor."""

    @staticmethod
    def forward(ctx, x, s):
        x, s = _req(x), _req(s)
        ctx.save_for_backward(x, s)
        out = torch.empty_like(x)
        _chk(_L().ix_scale_dev_f32(x.data_ptr(), s.data_ptr(), out.data_ptr(), x.numel(), _stream()), "ix_scale_dev_f32")
        return out

    @staticmethod
    def backward(ctx, g):
        x, s = ctx.saved_tensors
        return ScaleDev.call(g, s), Dot.call(g, x)


def l2_norm(x):
    """torch.norm(x): sqrt(sum x^2).  The 1-element sqrt stays a torch scalar op (plumbing)."""
    return torch.sqrt(Dot.call(x, x))


class RowNormSum(Function):
    """sum_e ||x_e||_2 over the rows of x [E, n] -> 0-d tensor, ONE launch (the learned loss of a chunk of episodes: reference
    models/interactron.py:96 per task).  Closed under the differentiation MAML needs: its backward is RowNormSumBwd, whose
    own backward is one more kernel."""
    b16_out = False   # (a scalar)

    @staticmethod
    def forward(ctx, x):
        x = _req(x)
        E, n = x.shape
        norms = torch.empty(E, device=x.device, dtype=torch.float32)
        total = torch.empty((), device=x.device, dtype=torch.float32)
        _chk(_L().ix_rownorm_sum_f32(x.data_ptr(), norms.data_ptr(), total.data_ptr(), E, n, _stream()), "ix_rownorm_sum_f32")
        ctx.save_for_backward(x, norms)
        return total

    @staticmethod
    def backward(ctx, g):
        x, norms = ctx.saved_tensors
        return RowNormSumBwd.call(x, norms, g)


class RowNormSumBwd(Function):
    """y = g x_e / ||x_e|| (g a 0-d tensor); norms are a function of x kept as a constant operand: the backward below carries
    their derivative (the - x <H, x> / n^3 term)."""

    @staticmethod
    def forward(ctx, x, norms, g):
        x, g = _req(x), _req(g)
        ctx.save_for_backward(x, norms, g)
        out = torch.empty_like(x)
        _chk(_L().ix_rownorm_sum_bwd_f32(x.data_ptr(), norms.data_ptr(), g.data_ptr(), out.data_ptr(), x.shape[0], x.shape[1],
                                         _stream()), "ix_rownorm_sum_bwd_f32")
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, H):
        x, norms, g = ctx.saved_tensors
        H = _req(H)
        Gx, Gg = torch.empty_like(x), torch.empty((), device=x.device, dtype=torch.float32)
        _chk(_L().ix_rownorm_sum_bwd_bwd_f32(x.data_ptr(), norms.data_ptr(), g.data_ptr(), H.data_ptr(), Gx.data_ptr(),
                                             Gg.data_ptr(), x.shape[0], x.shape[1], _stream()), "ix_rownorm_sum_bwd_bwd_f32")
        return Gx, None, Gg


def rownorm_sum(x):
    """sum of the L2 norms of the rows of x [E, n]"""
    return RowNormSum.call(x)


class Relu(Function):
    @staticmethod
    def forward(ctx, x):
        x = _req(x)
        out = torch.empty_like(x)
        _chk(_L().ix_relu_f32(x.data_ptr(), out.data_ptr(), x.numel(), _stream()), "ix_relu_f32")
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        return ReluBwd.call(g, y)


class ReluBwd(Function):
    """dy * [y > 0]; linear in dy, piecewise constant in y."""

    @staticmethod
    def forward(ctx, dy, y):
        dy, y = _req(dy), _req(y)
        ctx.save_for_backward(y)
        out = torch.empty_like(dy)
        _chk(_L().ix_relu_bwd_f32(dy.data_ptr(), y.data_ptr(), out.data_ptr(), dy.numel(), _stream()), "ix_relu_bwd_f32")
        return out

    @staticmethod
    def backward(ctx, G):
        (y,) = ctx.saved_tensors
        return ReluBwd.call(G, y), None


class ReluBwdSum(Function):
    """(ga + gb) * [y > 0]: ReluBwd of an activation with two consumers, the sum of their gradients in the same pass."""

    @staticmethod
    def forward(ctx, ga, gb, y):
        ga, gb, y = _req(ga), _req(gb), _req(y)
        ctx.save_for_backward(y)
        out = torch.empty_like(ga)
        _chk(_L().ix_relu_bwd_sum_f32(ga.data_ptr(), gb.data_ptr(), y.data_ptr(), out.data_ptr(), ga.numel(), _stream()),
             "ix_relu_bwd_sum_f32")
        return out

    @staticmethod
    def backward(ctx, G):
        (y,) = ctx.saved_tensors
        t = ReluBwd.call(G, y)
        return t, t, None


def _two_gradients(gs, y, relu):
    """the gradient(s) of a fused contraction + BN node's output(s) -> the ReLU-masked (if relu) single gradient.  Two outputs
    (fan = 2: the node handed out two aliases of its result) with two gradients: summed inside the ReLU derivative's pass."""
    gs = [g for g in gs if g is not None]
    if len(gs) == 2:
        if relu:
            return ReluBwdSum.call(gs[0].contiguous(), gs[1].contiguous(), y), True
        return SumN.call(gs[0], gs[1]), False
    return gs[0].contiguous(), False


class ReluBwdScaled(Function):
    """dy * [y > 0] * scale; linear in dy."""

    @staticmethod
    def forward(ctx, dy, y, scale):
        dy, y = _req(dy), _req(y)
        ctx.save_for_backward(y)
        ctx.scale = scale
        out = torch.empty_like(dy)
        _chk(_L().ix_relu_bwd_scaled_f32(dy.data_ptr(), y.data_ptr(), out.data_ptr(), dy.numel(), scale, _stream()),
             "ix_relu_bwd_scaled_f32")
        return out

    @staticmethod
    def backward(ctx, G):
        (y,) = ctx.saved_tensors
        return ReluBwdScaled.call(G, y, ctx.scale), None, None


class ReluDropout(Function):
    """dropout(relu(x)) as one pass; y > 0 exactly where the relu and the mask both pass, so the backward is one pass
    over (dy, y) with neither the mask hash nor x."""

    @staticmethod
    def forward(ctx, x, p, seed):
        x = _req(x)
        out = torch.empty_like(x)
        _chk(_L().ix_relu_dropout_f32(x.data_ptr(), out.data_ptr(), x.numel(), p, seed, _stream()), "ix_relu_dropout_f32")
        ctx.save_for_backward(out)
        ctx.scale = 1.0 / (1.0 - p)
        return out

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        return ReluBwdScaled.call(g, y, ctx.scale), None, None


class AddDropout(Function):
    """x + dropout(a) as one pass (residual connections)."""

    @staticmethod
    def forward(ctx, x, a, p, seed):
        x, a = _req(x), _req(a)
        assert x.shape == a.shape, (x.shape, a.shape)
        ctx.p, ctx.seed = p, seed
        out = torch.empty_like(x)
        _chk(_L().ix_add_dropout_f32(x.data_ptr(), a.data_ptr(), out.data_ptr(), x.numel(), p, seed, _stream()),
             "ix_add_dropout_f32")
        return out

    @staticmethod
    def backward(ctx, g):
        if ctx.needs_input_grad[0] and ctx.needs_input_grad[1] and torch.is_grad_enabled() and g.requires_grad:
            gx, ga = _AddDropoutBwd.apply(g, ctx.p, ctx.seed)   # (recorded: its own backward is ONE add_dropout pass)
            return gx, ga, None, None
        return (g if ctx.needs_input_grad[0] else None), \
            (_Dropout.call(g, ctx.p, ctx.seed) if ctx.needs_input_grad[1] else None), None, None


class _AddDropoutBwd(Function):
    """g -> (g, dropout(g)): AddDropout's backward as one node, so that the gradient of g in the outer backward is
    G_x + dropout(G_a) in one pass (add_dropout) instead of a dropout pass and an autograd sum."""

    @staticmethod
    def forward(ctx, g, p, seed):
        ctx.set_materialize_grads(False)
        ctx.p, ctx.seed = p, seed
        g = _req(g)
        return g.view_as(g), _Dropout.forward(_NullCtx(), g, p, seed)

    @staticmethod
    def backward(ctx, Gx, Ga):
        if Gx is None and Ga is None:
            return None, None, None
        if Ga is None:
            return Gx, None, None
        if Gx is None:
            return _Dropout.call(Ga.contiguous(), ctx.p, ctx.seed), None, None
        return AddDropout.call(Gx.contiguous(), Ga.contiguous(), ctx.p, ctx.seed), None, None


def add_dropout(x, a, p, training):
    if not training or p <= 0.0:
        return add(x, a)
    return AddDropout.call(x, a, float(p), _next_seed())


def relu_dropout(x, p, training):
    if not training or p <= 0.0:
        return Relu.call(x)
    return ReluDropout.call(x, float(p), _next_seed())


class Gelu(Function):
    @staticmethod
    def forward(ctx, x):
        x = _req(x)
        ctx.save_for_backward(x)
        out = torch.empty_like(x)
        _chk(_L().ix_gelu_f32(x.data_ptr(), out.data_ptr(), x.numel(), _stream()), "ix_gelu_f32")
        return out

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return GeluBwd.call(g, x)


class GeluBwd(Function):
    @staticmethod
    def forward(ctx, dy, x):
        dy, x = _req(dy), _req(x)
        ctx.save_for_backward(dy, x)
        out = torch.empty_like(dy)
        _chk(_L().ix_gelu_bwd_f32(dy.data_ptr(), x.data_ptr(), out.data_ptr(), dy.numel(), _stream()), "ix_gelu_bwd_f32")
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, G):
        dy, x = ctx.saved_tensors
        G = _req(G)
        gdy, gx = torch.empty_like(dy), torch.empty_like(x)
        _chk(_L().ix_gelu_bwd_bwd_f32(G.data_ptr(), dy.data_ptr(), x.data_ptr(), gdy.data_ptr(), gx.data_ptr(),
                                      dy.numel(), _stream()), "ix_gelu_bwd_bwd_f32")
        return gdy, gx


class Sigmoid(Function):
    @staticmethod
    def forward(ctx, x):
        x = _req(x)
        out = torch.empty_like(x)
        _chk(_L().ix_sigmoid_f32(x.data_ptr(), out.data_ptr(), x.numel(), _stream()), "ix_sigmoid_f32")
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        return SigmoidBwd.call(g, y)


class SigmoidBwd(Function):
    @staticmethod
    def forward(ctx, dy, y):
        dy, y = _req(dy), _req(y)
        ctx.save_for_backward(dy, y)
        out = torch.empty_like(dy)
        _chk(_L().ix_sigmoid_bwd_f32(dy.data_ptr(), y.data_ptr(), out.data_ptr(), dy.numel(), _stream()),
             "ix_sigmoid_bwd_f32")
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, G):
        dy, y = ctx.saved_tensors
        G = _req(G)
        gdy, gy = torch.empty_like(dy), torch.empty_like(y)
        _chk(_L().ix_sigmoid_bwd_bwd_f32(G.data_ptr(), dy.data_ptr(), y.data_ptr(), gdy.data_ptr(), gy.data_ptr(),
                                         dy.numel(), _stream()), "ix_sigmoid_bwd_bwd_f32")
        return gdy, gy


# dropout: the mask is a pure function of (seed, element index), so the same Function is its own adjoint
_seed_state = {"base": 0x5EED, "counter": 0}


def manual_seed(seed):
    _seed_state["base"] = int(seed) & 0xFFFFFFFF
    _seed_state["counter"] = 0


def _next_seed():
    _seed_state["counter"] += 1
    return ((_seed_state["base"] << 32) ^ (_seed_state["counter"] * 0x9E3779B97F4A7C15)) & 0x7FFFFFFFFFFFFFFF   # (63 bits: torch.profiler cannot record larger Python ints)


class _Dropout(Function):
    @staticmethod
    def forward(ctx, x, p, seed):
        x = _req(x)
        ctx.p, ctx.seed = p, seed
        out = torch.empty_like(x)
        _chk(_L().ix_dropout_f32(x.data_ptr(), out.data_ptr(), x.numel(), p, seed, _stream()), "ix_dropout_f32")
        return out

    @staticmethod
    def backward(ctx, g):
        return _Dropout.call(g, ctx.p, ctx.seed), None, None


def dropout(x, p, training):
    if not training or p <= 0.0:
        return x
    return _Dropout.call(x, float(p), _next_seed())


# ---------------------------------------------------------------------------------------------------------
# FrozenBatchNorm2d affine (+ residual, + ReLU), NHWC
# ---------------------------------------------------------------------------------------------------------
def bn_fold(weight, bias, running_mean, running_var, eps=1e-5):
    C = weight.numel()
    scale = torch.empty(C, device=weight.device, dtype=torch.float32)
    shift = torch.empty_like(scale)
    _chk(_L().ix_bn_fold_f32(_req(weight).data_ptr(), _req(bias).data_ptr(), _req(running_mean).data_ptr(),
                             _req(running_var).data_ptr(), scale.data_ptr(), shift.data_ptr(), C, eps, _stream()),
         "ix_bn_fold_f32")
    return scale, shift


def _channel_affine(x, scale, shift, residual, relu):
    out = torch.empty_like(x)
    _chk(_L().ix_channel_affine_f32(x.data_ptr(), scale.data_ptr(), shift.data_ptr() if shift is not None else None,
                                    residual.data_ptr() if residual is not None else None, out.data_ptr(), x.numel(),
                                    scale.numel(), 1 if relu else 0, _stream()), "ix_channel_affine_f32")
    return out


class ChannelScale(Function):
    """x * scale[c] (channel = last dim); scale is a constant buffer."""

    @staticmethod
    def forward(ctx, x, scale):
        x = _req(x)
        ctx.save_for_backward(scale)
        return _channel_affine(x, scale, None, None, False)

    @staticmethod
    def backward(ctx, g):
        (scale,) = ctx.saved_tensors
        return ChannelScale.call(g, scale), None


class RowScale(Function):
    """w * scale[n] along the output-channel dim of a weight tensor -- linear [(E,) N, K] (tail = 1), convolution
    [(E,) Cout, KH, KW, Cin] (tail = 3) -- or of its gradient; scale is a constant buffer.  The result keeps standing for the
    Parameter in skip_param_grads."""

    @staticmethod
    def forward(ctx, w, scale, tail):
        key = _param_key(w)
        w = _req(w, "weight")
        N = w.shape[-(tail + 1)]
        R = _numel(w.shape[-tail:])
        out = torch.empty_like(w)
        _chk(_L().ix_row_scale_f32(w.data_ptr(), scale.data_ptr(), out.data_ptr(), w.numel() // (N * R), N, R, _stream()),
             "ix_row_scale_f32")
        ctx.tail = tail
        ctx.save_for_backward(scale)
        # the scaled copy keeps standing for its Parameter in skip_param_grads, but it is NOT marked for the weight-planes route:
        # it is fresh in every backward and read by exactly one contraction, so its planes could never be reused -- an eager
        # ix_wp_split_f32 launch per call for nothing, and inside a capture (planes, copy) pinned in the graph's pool for good
        out._ix_of_param = key
        return out

    @staticmethod
    def backward(ctx, g):
        (scale,) = ctx.saved_tensors
        return RowScale.call(g.contiguous(), scale, ctx.tail), None, None


# IX_BN_SCALE_ON_WEIGHTS: "1" (default) = the backward of a fused contraction + frozen-BN node applies the BN scale to the weights
# (dx = g (W o scale), dW = scale o (g^T x): two passes over a weight tensor) where it would otherwise run a pass over the
# activation-sized gradient (no ReLU, or ReLU behind a residual: the bottleneck tails and the downsample branches); "0" = g o scale
BN_SCALE_ON_WEIGHTS = os.environ.get("IX_BN_SCALE_ON_WEIGHTS", "1") == "1"


def _bn_scale_on_weights(relu, has_res, w, tail):
    return BN_SCALE_ON_WEIGHTS and (has_res or not relu) and _numel(w.shape[-tail:]) % 4 == 0


class ReluBwdChannelScale(Function):
    """[y > 0] * g * scale[c]: linear in g, so it is its own second-order form."""

    @staticmethod
    def forward(ctx, g, y, scale):
        g, y = _req(g), _req(y)
        ctx.save_for_backward(y, scale)
        out = torch.empty_like(g)
        _chk(_L().ix_relu_bwd_channel_scale_f32(g.data_ptr(), y.data_ptr(), scale.data_ptr(), out.data_ptr(), g.numel(),
                                                g.shape[-1], _stream()), "ix_relu_bwd_channel_scale_f32")
        return out

    @staticmethod
    def backward(ctx, G):
        y, scale = ctx.saved_tensors
        return ReluBwdChannelScale.call(G, y, scale), None, None


class BnAct(Function):
    """y = [relu](x*scale[c] + shift[c] (+ residual)) on NHWC activations (FrozenBatchNorm2d folded)."""

    @staticmethod
    def forward(ctx, x, scale, shift, residual, relu):
        x = _req(x)
        if residual is not None:
            residual = _req(residual)
        y = _channel_affine(x, scale, shift, residual, relu)
        ctx.relu = relu
        ctx.has_res = residual is not None
        ctx.save_for_backward(scale, y if relu else None)
        return y

    @staticmethod
    def backward(ctx, g):
        scale, y = ctx.saved_tensors
        gx, gres = _bn_act_backward(g, y, scale, ctx.relu, ctx.has_res, ctx.needs_input_grad[0], ctx.needs_input_grad[3])
        return gx, None, None, gres, None


def _bn_act_backward(g, y, scale, relu, has_res, need_x, need_res):
    """(gradient of the pre-affine tensor, gradient of the residual) of y = [relu](z * scale + shift (+ res)) -- BnAct.backward's
    arithmetic on differentiable nodes, shared with the fused contraction + affine Functions"""
    g = g.contiguous()
    if relu and not has_res:   # one pass instead of relu-backward + channel scale
        return (ReluBwdChannelScale.call(g, y, scale) if need_x else None), None
    if relu:
        g = ReluBwd.call(g, y)
    return (ChannelScale.call(g, scale) if need_x else None), (g if (has_res and need_res) else None)


# IX_FUSE_CONV_BN: "1" = backbone convolutions carry their frozen-BN affine (+ residual) (+ ReLU) on the contraction call
# (ix_gemm_bn_act_f32 / ix_conv_gemm_bn_act_f32) instead of a separate elementwise launch; "0" = separate launches
FUSE_CONV_BN = os.environ.get("IX_FUSE_CONV_BN", "1") == "1"


class GemmBnAct(Function):
    """y = [relu]((A B) * scale[n] + shift[n] (+ residual)) for the plain row-major product of Gemm (1 x 1 convolutions);
    backward = BnAct's backward followed by Gemm's (all differentiable nodes: closed under the MAML double backward)."""

    b16 = "native"

    @staticmethod
    def forward(ctx, a, b, scale, shift, residual, relu, sp, fan=1):
        ctx.set_materialize_grads(False)
        ctx.a_key, ctx.b_key = _param_key(a), _param_key(b)
        assert sp.bi == 1 and sp.alpha == 1.0 and sp.C.offset == 0 and sp.C.ld == sp.N and not sp.C.trans
        assert sp.A.offset == 0 and sp.B.offset == 0 and (sp.bo == 1 or sp.C.so == sp.M * sp.N)
        if a.dtype == torch.bfloat16:   # 16-bit mode: the affine (+ residual) (+ ReLU) in the bf16 GEMM's own store (csrc/gemm16.hip)
            from . import b16
            wb = getattr(b, "_ix_weight", False)
            a, b, scale, shift = b16._reqd(a, "gemm A"), b16._reqd(b, "gemm B"), _req(scale), _req(shift)
            if wb:
                mark_weight(b)
            if residual is not None:
                residual = b16._reqd(residual)
                assert residual.dtype == torch.bfloat16
            out = b16.run_gemm(a, b, None, sp, scale=scale, shift=shift, residual=residual, act=1 if relu else 0)
            ctx.sp, ctx.relu, ctx.has_res = sp, relu, residual is not None
            ctx.a_shape, ctx.b_shape = tuple(a.shape), tuple(b.shape)
            ctx.save_for_backward(a, b, scale, out if relu else None)
            return out if fan == 1 else (out, out.view_as(out))
        a, b, scale, shift = _req(a, "gemm A"), _req(b, "gemm B"), _req(scale), _req(shift)
        if residual is not None:
            residual = _req(residual)
        out = torch.empty(sp.out_shape, device=a.device, dtype=torch.float32)
        nws, _ = _gemm_workspace_bytes(a.data_ptr(), b.data_ptr(), sp, presplit=False)
        ws = _workspace(nws, a.device) if nws else None
        _chk(_L().ix_gemm_bn_act_f32(a.data_ptr(), b.data_ptr(), out.data_ptr(), sp.M, sp.N, sp.K, 0 if sp.A.trans else 1,
                                     1 if sp.B.trans else 0, sp.A.ld, sp.B.ld, sp.bo, sp.A.so, sp.B.so, scale.data_ptr(),
                                     shift.data_ptr(), residual.data_ptr() if residual is not None else None,
                                     1 if relu else 0, ws.data_ptr() if nws else None, nws, _stream()), "ix_gemm_bn_act_f32")
        ctx.sp, ctx.relu, ctx.has_res = sp, relu, residual is not None
        ctx.a_shape, ctx.b_shape = tuple(a.shape), tuple(b.shape)
        ctx.save_for_backward(a, b, scale, out if relu else None)
        return out if fan == 1 else (out, out.view_as(out))   # (fan = 2: two aliases for two consumers, see _two_gradients)

    @staticmethod
    def backward(ctx, *gs):
        a, b, scale, y = ctx.saved_tensors
        if all(g is None for g in gs):
            return (None,) * 8
        need_a = ctx.needs_input_grad[0] and not _is_unwanted(ctx.a_key, _unwanted)
        need_b = ctx.needs_input_grad[1] and not _is_unwanted(ctx.b_key, _unwanted)
        da = db = None
        if _bn_scale_on_weights(ctx.relu, ctx.has_res, b, 1):
            g, masked = _two_gradients(gs, y, ctx.relu)
            g1 = ReluBwd.call(g, y) if (ctx.relu and not masked) else g
            want_res = ctx.has_res and ctx.needs_input_grad[4]
            gs = list(fanout(g1, int(need_a) + int(need_b) + int(want_res)))   # (one sum of its consumers' gradients in the outer backward)
            gres = gs.pop() if want_res else None
            if need_a:
                da = _gemm_backward(ctx.sp, a, RowScale.call(b, scale, 1), ctx.a_shape, ctx.b_shape, gs.pop(), True, False)[0]
            if need_b:
                db = RowScale.call(_gemm_backward(ctx.sp, a, b, ctx.a_shape, ctx.b_shape, gs.pop(), False, True)[1], scale, 1)
            return da, db, None, None, gres, None, None, None
        g, _ = _two_gradients(gs, None, False)
        gz, gres = _bn_act_backward(g, y, scale, ctx.relu, ctx.has_res, need_a or need_b, ctx.needs_input_grad[4])
        if gz is not None:
            da, db = _gemm_backward(ctx.sp, a, b, ctx.a_shape, ctx.b_shape, gz, need_a, need_b)
        return da, db, None, None, gres, None, None, None


def linear_bn_act(x, weight, scale, shift, residual, relu, fan=1):
    """[relu](linear(x, weight) * scale + shift (+ residual)) -- `linear` without bias, episode-batched weights included"""
    if weight.dim() == 3:
        E, N, K = weight.shape
        R = x.numel() // (E * K)
        sp = GemmSpec(R, N, K, E, 1, View(0, K, False, R * K, 0), View(0, K, True, N * K, 0), View(0, N, False, R * N, 0),
                      tuple(x.shape[:-1]) + (N,), 1.0)
    else:
        K, N = x.shape[-1], weight.shape[0]
        sp = GemmSpec(x.numel() // K, N, K, 1, 1, View(0, K, False, 0, 0), View(0, K, True, 0, 0), View(0, N, False, 0, 0),
                      tuple(x.shape[:-1]) + (N,), 1.0)
    return GemmBnAct.call(x, weight, scale, shift, residual, relu, sp, fan)


# ---------------------------------------------------------------------------------------------------------
# convolution pieces (NHWC)
# ---------------------------------------------------------------------------------------------------------
ConvGeom = namedtuple("ConvGeom", "n H W C KH KW stride pad dil OH OW Kp")


def conv_geom(n, H, W, C, KH, KW, stride, pad, dil):
    OH = (H + 2 * pad - dil * (KH - 1) - 1) // stride + 1
    OW = (W + 2 * pad - dil * (KW - 1) - 1) // stride + 1
    K = KH * KW * C
    return ConvGeom(n, H, W, C, KH, KW, stride, pad, dil, OH, OW, (K + 3) // 4 * 4)


def _im2col(x, g, strides):
    cols = torch.empty(g.n * g.OH * g.OW, g.Kp, device=x.device, dtype=torch.float32)
    sxn, sxh, sxw, sxc = strides
    _chk(_L().ix_im2col_f32(x.data_ptr(), cols.data_ptr(), g.n, g.H, g.W, g.C, sxn, sxh, sxw, sxc, g.KH, g.KW, g.stride,
                            g.pad, g.dil, g.Kp, _stream()), "ix_im2col_f32")
    return cols


def im2col_any_layout(x_nchw_or_nhwc, g, channels_last):
    """Non-differentiable patch extraction for the frozen stem; accepts the NCHW input frames directly."""
    x = _req(x_nchw_or_nhwc)
    if channels_last:
        strides = (g.H * g.W * g.C, g.W * g.C, g.C, 1)
    else:
        strides = (g.C * g.H * g.W, g.W, 1, g.H * g.W)
    return _im2col(x, g, strides)


class Im2Col(Function):
    @staticmethod
    def forward(ctx, x, g):
        x = _req(x)
        ctx.g = g
        return _im2col(x, g, (g.H * g.W * g.C, g.W * g.C, g.C, 1))

    @staticmethod
    def backward(ctx, dcols):
        return Col2Im.call(dcols, ctx.g), None


class Col2Im(Function):
    @staticmethod
    def forward(ctx, cols, g):
        cols = _req(cols)
        ctx.g = g
        dx = torch.empty(g.n, g.H, g.W, g.C, device=cols.device, dtype=torch.float32)
        _chk(_L().ix_col2im_f32(cols.data_ptr(), dx.data_ptr(), g.n, g.H, g.W, g.C, g.KH, g.KW, g.stride, g.pad, g.dil,
                                g.Kp, _stream()), "ix_col2im_f32")
        return dx

    @staticmethod
    def backward(ctx, G):
        return Im2Col.call(G, ctx.g), None


def maxpool_nhwc(x, k, stride, pad):
    x = _req(x)
    n, H, W, C = x.shape
    OH, OW = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    y = torch.empty(n, OH, OW, C, device=x.device, dtype=torch.float32)
    _chk(_L().ix_maxpool_nhwc_f32(x.data_ptr(), y.data_ptr(), n, H, W, C, k, stride, pad, _stream()), "ix_maxpool_nhwc_f32")
    return y


# ---- implicit-GEMM convolution (csrc/gemm.hip ix_conv_gemm_f32): no patch matrix in HBM ----------------------------------
ConvGemmGeom = namedtuple("ConvGemmGeom", "E imgs H W Cin OH OW Cout KH KW stride pad dil")
CONV_IMPL = os.environ.get("IX_CONV", "implicit")   # "im2col": keep every convolution on the patch-matrix path (A/B runs)
_conv_ok = {}


def conv_gemm_supported(cg):
    ok = _conv_ok.get(cg)
    if ok is None:
        ok = _conv_ok[cg] = CONV_IMPL == "implicit" and bool(_L().ix_conv_gemm_supported(
            cg.E, cg.imgs, cg.H, cg.W, cg.Cin, cg.OH, cg.OW, cg.Cout, cg.KH, cg.KW, cg.stride, cg.pad, cg.dil))
    return ok


_conv_ws = {}


def _conv_gemm(kind, src, other, out_shape, cg):
    out = torch.empty(out_shape, device=src.device, dtype=torch.float32)
    nws = _conv_ws.get((kind, cg))
    if nws is None:
        n = ctypes.c_size_t(0)
        _chk(_L().ix_workspace_bytes_conv_gemm_f32(kind, cg.E, cg.imgs, cg.H, cg.W, cg.Cin, cg.OH, cg.OW, cg.Cout, cg.KH, cg.KW,
                                                   cg.stride, cg.pad, cg.dil, ctypes.byref(n)), "ix_workspace_bytes_conv_gemm_f32")
        nws = _conv_ws[(kind, cg)] = n.value
    ws = _workspace(nws, src.device) if nws else None
    _chk(_L().ix_conv_gemm_f32(kind, src.data_ptr(), other.data_ptr(), out.data_ptr(), cg.E, cg.imgs, cg.H, cg.W, cg.Cin,
                               cg.OH, cg.OW, cg.Cout, cg.KH, cg.KW, cg.stride, cg.pad, cg.dil, ws.data_ptr() if nws else None,
                               nws, _stream()), "ix_conv_gemm_f32")
    return out


class ConvFwd(Function):
    """y = conv(x, w): x [E*imgs, H, W, Cin], w [(E,) Cout, KH, KW, Cin] -> [E*imgs, OH, OW, Cout].  With ConvBwdData and
    ConvBwdWeight the three implicit-GEMM kinds are closed under differentiation (each one's backward is the other two)."""

    @staticmethod
    def forward(ctx, x, w, cg):
        ctx.w_key = _param_key(w)
        x, w = _req(x, "conv x"), _req(w, "conv weight")
        ctx.cg = cg
        ctx.save_for_backward(x, w)
        return _conv_gemm(0, x, w, (cg.E * cg.imgs, cg.OH, cg.OW, cg.Cout), cg)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        need_w = ctx.needs_input_grad[1] and not _is_unwanted(ctx.w_key, _unwanted)
        dx = ConvBwdData.call(dy, w, ctx.cg) if ctx.needs_input_grad[0] else None
        dw = ConvBwdWeight.call(dy, x, ctx.cg, tuple(w.shape)) if need_w else None
        return dx, dw, None


class ConvBwdData(Function):
    @staticmethod
    def forward(ctx, dy, w, cg):
        ctx.w_key = _param_key(w)
        dy, w = _req(dy, "conv dy"), _req(w, "conv weight")
        ctx.cg = cg
        ctx.save_for_backward(dy, w)
        return _conv_gemm(1, dy, w, (cg.E * cg.imgs, cg.H, cg.W, cg.Cin), cg)

    @staticmethod
    def backward(ctx, g):
        dy, w = ctx.saved_tensors
        g = g.contiguous()
        need_w = ctx.needs_input_grad[1] and not _is_unwanted(ctx.w_key, _unwanted)
        ddy = ConvFwd.call(g, w, ctx.cg) if ctx.needs_input_grad[0] else None
        dw = ConvBwdWeight.call(dy, g, ctx.cg, tuple(w.shape)) if need_w else None
        return ddy, dw, None


class ConvBwdWeight(Function):
    @staticmethod
    def forward(ctx, dy, x, cg, w_shape):
        dy, x = _req(dy, "conv dy"), _req(x, "conv x")
        ctx.cg = cg
        ctx.save_for_backward(dy, x)
        return _conv_gemm(2, dy, x, w_shape, cg)

    @staticmethod
    def backward(ctx, g):
        dy, x = ctx.saved_tensors
        g = g.contiguous()
        ddy = ConvFwd.call(x, g, ctx.cg) if ctx.needs_input_grad[0] else None
        dx = ConvBwdData.call(dy, g, ctx.cg) if ctx.needs_input_grad[1] else None
        return ddy, dx, None, None


class ConvFwdBnAct(Function):
    """y = [relu](conv(x, w) * scale[c] + shift[c] (+ residual)): ConvFwd with the frozen-BN affine in the contraction's store"""

    @staticmethod
    def forward(ctx, x, w, scale, shift, residual, relu, cg, fan=1):
        ctx.set_materialize_grads(False)
        ctx.w_key = _param_key(w)
        x, w, scale, shift = _req(x, "conv x"), _req(w, "conv weight"), _req(scale), _req(shift)
        if residual is not None:
            residual = _req(residual)
        out = torch.empty((cg.E * cg.imgs, cg.OH, cg.OW, cg.Cout), device=x.device, dtype=torch.float32)
        nws = _conv_ws.get((0, cg))
        if nws is None:
            n = ctypes.c_size_t(0)
            _chk(_L().ix_workspace_bytes_conv_gemm_f32(0, cg.E, cg.imgs, cg.H, cg.W, cg.Cin, cg.OH, cg.OW, cg.Cout, cg.KH, cg.KW,
                                                       cg.stride, cg.pad, cg.dil, ctypes.byref(n)), "ix_workspace_bytes_conv_gemm_f32")
            nws = _conv_ws[(0, cg)] = n.value
        ws = _workspace(nws, x.device) if nws else None
        _chk(_L().ix_conv_gemm_bn_act_f32(x.data_ptr(), w.data_ptr(), out.data_ptr(), cg.E, cg.imgs, cg.H, cg.W, cg.Cin, cg.OH,
                                          cg.OW, cg.Cout, cg.KH, cg.KW, cg.stride, cg.pad, cg.dil, scale.data_ptr(),
                                          shift.data_ptr(), residual.data_ptr() if residual is not None else None,
                                          1 if relu else 0, ws.data_ptr() if nws else None, nws, _stream()),
             "ix_conv_gemm_bn_act_f32")
        ctx.cg, ctx.relu, ctx.has_res = cg, relu, residual is not None
        ctx.save_for_backward(x, w, scale, out if relu else None)
        return out if fan == 1 else (out, out.view_as(out))

    @staticmethod
    def backward(ctx, *gs):
        x, w, scale, y = ctx.saved_tensors
        if all(g is None for g in gs):
            return (None,) * 8
        need_w = ctx.needs_input_grad[1] and not _is_unwanted(ctx.w_key, _unwanted)
        dx = dw = None
        if _bn_scale_on_weights(ctx.relu, ctx.has_res, w, 3):
            g, masked = _two_gradients(gs, y, ctx.relu)
            g1 = ReluBwd.call(g, y) if (ctx.relu and not masked) else g
            want_res = ctx.has_res and ctx.needs_input_grad[4]
            gs = list(fanout(g1, int(bool(ctx.needs_input_grad[0])) + int(need_w) + int(want_res)))
            gres = gs.pop() if want_res else None
            if ctx.needs_input_grad[0]:
                dx = ConvBwdData.call(gs.pop(), RowScale.call(w, scale, 3), ctx.cg)
            if need_w:
                dw = RowScale.call(ConvBwdWeight.call(gs.pop(), x, ctx.cg, tuple(w.shape)), scale, 3)
            return dx, dw, None, None, gres, None, None, None
        g, _ = _two_gradients(gs, None, False)
        gz, gres = _bn_act_backward(g, y, scale, ctx.relu, ctx.has_res, ctx.needs_input_grad[0] or need_w, ctx.needs_input_grad[4])
        if gz is not None:
            dx = ConvBwdData.call(gz, w, ctx.cg) if ctx.needs_input_grad[0] else None
            dw = ConvBwdWeight.call(gz, x, ctx.cg, tuple(w.shape)) if need_w else None
        return dx, dw, None, None, gres, None, None, None


def conv2d_nhwc_bn_act(x, weight, scale, shift, residual=None, relu=False, stride=1, pad=0, dil=1, fan=1):
    """[relu](conv2d_nhwc(x, weight) * scale + shift (+ residual)): one launch where the contraction kernel takes the affine
    (1 x 1 / stride 1 and implicit-GEMM geometries, output channels % 4 == 0), else the two separate nodes"""
    n, H, W, C = x.shape
    batched = weight.dim() == 5
    Cout, KH, KW = weight.shape[-4], weight.shape[-3], weight.shape[-2]
    E = weight.shape[0] if batched else 1
    if FUSE_CONV_BN and Cout % 4 == 0:
        if KH == 1 and KW == 1 and stride == 1 and pad == 0:
            wv = weight_view(weight, E, Cout, C) if batched else weight_view(weight, Cout, C)
            return linear_bn_act(x, wv, scale, shift, residual, relu, fan)
        g = conv_geom(n, H, W, C, KH, KW, stride, pad, dil)
        cg = ConvGemmGeom(E, n // E, H, W, C, g.OH, g.OW, Cout, KH, KW, stride, pad, dil)
        if conv_gemm_supported(cg):
            return ConvFwdBnAct.call(x, weight, scale, shift, residual, relu, cg, fan)
    y = BnAct.apply(conv2d_nhwc(x, weight, stride, pad, dil), scale, shift, residual, relu)
    return y if fan == 1 else fanout(y, fan)


def conv2d_nhwc(x, weight, stride=1, pad=0, dil=1):
    """x [n,H,W,Cin] NHWC, weight [Cout,KH,KW,Cin] (nn.Conv2dNHWC's storage layout: the patch-matrix column order
    (kh, kw, cin), so it is the contraction's k-contiguous operand as stored) -> [n,OH,OW,Cout]."""
    n, H, W, C = x.shape
    if weight.dim() == 5:   # episode-batched fast weights [E, Cout, KH, KW, Cin]; frames of episode e are x[e*n/E:(e+1)*n/E]
        E, Cout, KH, KW, Cin = weight.shape
        assert Cin == C and n % E == 0
        if KH == 1 and KW == 1 and stride == 1 and pad == 0:
            return linear(x, weight_view(weight, E, Cout, Cin))
        g = conv_geom(n, H, W, C, KH, KW, stride, pad, dil)
        cg = ConvGemmGeom(E, n // E, H, W, Cin, g.OH, g.OW, Cout, KH, KW, stride, pad, dil)
        if conv_gemm_supported(cg):
            return ConvFwd.call(x, weight, cg)
        cols = Im2Col.call(x, g)
        assert g.Kp == KH * KW * Cin, "episode-batched convs need KH*KW*Cin % 4 == 0"
        return linear(cols.reshape(E, -1, g.Kp), weight_view(weight, E, Cout, KH * KW * Cin)).reshape(n, g.OH, g.OW, Cout)
    Cout, KH, KW, Cin = weight.shape
    assert Cin == C
    if KH == 1 and KW == 1 and stride == 1 and pad == 0:
        return linear(x, weight_view(weight, Cout, Cin))
    g = conv_geom(n, H, W, C, KH, KW, stride, pad, dil)
    cg = ConvGemmGeom(1, n, H, W, Cin, g.OH, g.OW, Cout, KH, KW, stride, pad, dil)
    if conv_gemm_supported(cg):
        return ConvFwd.call(x, weight, cg)
    cols = Im2Col.call(x, g)
    return linear(cols, weight_view(weight, Cout, KH * KW * Cin)).reshape(n, g.OH, g.OW, Cout)


# ---------------------------------------------------------------------------------------------------------
# softmax / LayerNorm
# ---------------------------------------------------------------------------------------------------------
class Softmax(Function):
    """softmax over the first `length` entries of the last dim (row pitch = last dim size); optional uint8
    key-padding mask [nmask, length] with `rows_per_mask` consecutive rows sharing one mask row."""

    @staticmethod
    def forward(ctx, x, length, mask, rows_per_mask):
        x = _req(x)
        ld = x.shape[-1]
        rows = x.numel() // ld
        y = torch.empty_like(x) if ld == length else torch.zeros_like(x)
        _chk(_L().ix_softmax_fwd_f32(x.data_ptr(), y.data_ptr(), rows, length, ld,
                                     mask.data_ptr() if mask is not None else None, rows_per_mask,
                                     mask.shape[-1] if mask is not None else 0, _stream()), "ix_softmax_fwd_f32")
        ctx.length = length
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        return SoftmaxBwd.call(y, g, ctx.length), None, None, None


class SoftmaxBwd(Function):
    @staticmethod
    def forward(ctx, y, dy, length):
        y, dy = _req(y), _req(dy)
        ld = y.shape[-1]
        dx = torch.empty_like(y) if ld == length else torch.zeros_like(y)
        _chk(_L().ix_softmax_bwd_f32(y.data_ptr(), dy.data_ptr(), dx.data_ptr(), y.numel() // ld, length, ld, _stream()),
             "ix_softmax_bwd_f32")
        ctx.length = length
        ctx.save_for_backward(y, dy)
        return dx

    @staticmethod
    @once_differentiable
    def backward(ctx, G):
        y, dy = ctx.saved_tensors
        G = _req(G)
        ld = y.shape[-1]
        alloc = torch.empty_like if ld == ctx.length else torch.zeros_like
        gy, gdy = alloc(y), alloc(y)
        _chk(_L().ix_softmax_bwd_bwd_f32(G.data_ptr(), y.data_ptr(), dy.data_ptr(), gy.data_ptr(), gdy.data_ptr(),
                                         y.numel() // ld, ctx.length, ld, _stream()), "ix_softmax_bwd_bwd_f32")
        return gy, gdy, None


class LayerNorm(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        x, gamma, beta = _req(x), _req(gamma), _req(beta)
        D = x.shape[-1]
        G = gamma.shape[0] if gamma.dim() == 2 else 1      # per-episode affine [G, D]: x is [G * rows, D]
        rows = x.numel() // D
        assert rows % G == 0
        y = torch.empty_like(x)
        mean = torch.empty(rows, device=x.device, dtype=torch.float32)
        rstd = torch.empty_like(mean)
        _chk(_L().ix_layernorm_fwd_f32(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), mean.data_ptr(),
                                       rstd.data_ptr(), rows // G, D, eps, G, _stream()), "ix_layernorm_fwd_f32")
        ctx.save_for_backward(x, gamma, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, g):
        x, gamma, mean, rstd = ctx.saved_tensors
        dx, dgamma, dbeta = LayerNormBwd.call(g, x, gamma, mean, rstd)
        return dx, dgamma, dbeta, None


class LayerNormBwd(Function):
    """(dy, x, gamma) -> (dx, dgamma, dbeta); mean/rstd are recomputable statistics of x (handled analytically)."""
    b16_out = (True, False, False)   # (dx is an activation; the parameter gradients stay fp32)

    @staticmethod
    def forward(ctx, dy, x, gamma, mean, rstd):
        dy = _req(dy)
        D = x.shape[-1]
        G = gamma.shape[0] if gamma.dim() == 2 else 1
        rows = x.numel() // D
        dx = torch.empty_like(x)
        both = torch.empty((2,) + tuple(gamma.shape), device=gamma.device, dtype=torch.float32)   # (one fill for the two)
        dgamma, dbeta = both[0], both[1]
        wp, wn = _reduce_ws("ln", rows // G, D, G, x.device)
        _chk(_L().ix_layernorm_bwd_f32(dy.data_ptr(), x.data_ptr(), gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                       dx.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), rows // G, D, G, wp, wn, _stream()),
             "ix_layernorm_bwd_f32")
        ctx.save_for_backward(dy, x, gamma, mean, rstd)
        return dx, dgamma, dbeta

    @staticmethod
    @once_differentiable
    def backward(ctx, Gx, Gg, Gb):
        dy, x, gamma, mean, rstd = ctx.saved_tensors
        D = x.shape[-1]
        G = gamma.shape[0] if gamma.dim() == 2 else 1
        rows = x.numel() // D
        Gx = _req(Gx) if Gx is not None else None
        Gg = _req(Gg) if Gg is not None else None
        Gb = _req(Gb) if Gb is not None else None
        gdy, gx = torch.empty_like(x), torch.empty_like(x)
        ggamma = torch.empty_like(gamma)
        wp, wn = _reduce_ws("ln", rows // G, D, G, x.device)
        _chk(_L().ix_layernorm_bwd_bwd_f32(Gx.data_ptr() if Gx is not None else None,
                                           Gg.data_ptr() if Gg is not None else None,
                                           Gb.data_ptr() if Gb is not None else None, dy.data_ptr(), x.data_ptr(),
                                           gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gdy.data_ptr(),
                                           gx.data_ptr(), ggamma.data_ptr(), rows // G, D, G, wp, wn, _stream()),
             "ix_layernorm_bwd_bwd_f32")
        return gdy, gx, ggamma, None, None


def layer_norm(x, gamma, beta, eps=1e-5):
    return LayerNorm.call(x, gamma, beta, eps)


# ---------------------------------------------------------------------------------------------------------
# criterion kernels
# ---------------------------------------------------------------------------------------------------------
def match_cost(logits, boxes, tgt_ids, tgt_boxes, w_class, w_bbox, w_giou):
    """Hungarian cost matrix [rows, T] (no grad)."""
    logits, boxes, tgt_boxes = _req(logits.detach()), _req(boxes.detach()), _req(tgt_boxes)
    rows, C = logits.shape
    T = tgt_ids.numel()
    cost = torch.empty(rows, T, device=logits.device, dtype=torch.float32)
    tgt_ids = tgt_ids.contiguous()
    assert tgt_ids.dtype == torch.int64 and tgt_ids.is_cuda
    _chk(_L().ix_match_cost_f32(logits.data_ptr(), boxes.data_ptr(), tgt_ids.data_ptr(), tgt_boxes.data_ptr(),
                                cost.data_ptr(), rows, C, T, w_class, w_bbox, w_giou, _stream()), "ix_match_cost_f32")
    return cost


def lsap(cost_cpu):
    """Host rectangular assignment on a CPU float32 [nr, nc] tensor -> (rows int64[k], cols int64[k])."""
    cost_cpu = cost_cpu.contiguous()
    assert not cost_cpu.is_cuda and cost_cpu.dtype == torch.float32
    nr, nc = cost_cpu.shape
    k = min(nr, nc)
    r = torch.empty(k, dtype=torch.int64)
    c = torch.empty(k, dtype=torch.int64)
    _chk(_L().ix_lsap_f32(cost_cpu.data_ptr(), nr, nc, r.data_ptr(), c.data_ptr()), "ix_lsap_f32")
    return r, c


# ---- device-resident matcher + set criterion (csrc/criterion.hip, second half) --------------------------------------------
class Targets:
    """Ground truth of I images as one CSR list on the device: ids int64 [T], boxes [T, 4], off int32 [I + 1]; ``sizes`` is
    the host copy of the per-image counts, ``ldn`` the column pitch of the cost matrices (max count, rounded up to 8)."""

    def __init__(self, ids, boxes, off, sizes):
        self.ids, self.boxes, self.off, self.sizes = ids, boxes, off, list(sizes)
        self.I = len(self.sizes)
        self.ldn = (max(self.sizes + [1]) + 7) // 8 * 8



def pack_targets(targets):
    """list of {"labels": int64 [n_i], "boxes": [n_i, 4]} (device tensors) -> Targets; one cat per field + one small upload"""
    sizes = [int(t["labels"].shape[0]) for t in targets]
    dev = targets[0]["boxes"].device if targets else torch.device("cuda")
    if sum(sizes) == 0:
        ids = torch.zeros(1, dtype=torch.int64, device=dev)
        boxes = torch.full((1, 4), 0.5, dtype=torch.float32, device=dev)
    else:
        ids = torch.cat([t["labels"] for t in targets]).contiguous()
        boxes = _req(torch.cat([t["boxes"] for t in targets]), "target boxes")
    off = [0]
    for n in sizes:
        off.append(off[-1] + n)
    tg = Targets(ids, boxes, h2d_async(torch.tensor(off, dtype=torch.int32)), sizes)
    tg.targets = targets   # (the per-image dicts: the host assignment route and the tests' pinning hook read them)
    return tg


LSAP_DEVICE_MAX = 256


def match_cost_csr(logits, boxes, tg, w_class, w_bbox, w_giou):
    """[I, Q, ldn] cost matrices of all images (columns beyond an image's target count are not written)"""
    I, Q, C = logits.shape
    logits, boxes = _req(logits.detach()), _req(boxes.detach())
    cost = torch.empty(I, Q, tg.ldn, device=logits.device, dtype=torch.float32)
    _chk(_L().ix_match_cost_csr_f32(logits.data_ptr(), boxes.data_ptr(), tg.ids.data_ptr(), tg.boxes.data_ptr(), tg.off.data_ptr(),
                                    cost.data_ptr(), I, Q, C, tg.ldn, w_class, w_bbox, w_giou, _stream()), "ix_match_cost_csr_f32")
    return cost


def lsap_device(cost, tg):
    """-> (tgt_of_q int32 [I, Q], q_of_tgt int32 [T]) -- scipy's assignment per image, computed on the GPU"""
    I, Q, ldn = cost.shape
    toq = torch.empty(I, Q, dtype=torch.int32, device=cost.device)
    qot = torch.empty(max(int(tg.ids.shape[0]), 1), dtype=torch.int32, device=cost.device)
    _chk(_L().ix_lsap_device_f32(cost.data_ptr(), tg.off.data_ptr(), I, Q, ldn, toq.data_ptr(), qot.data_ptr(), _stream()),
         "ix_lsap_device_f32")
    return toq, qot


class SetLoss(Function):
    """DETR set criterion of image groups: apply(logits [I, Q, C], boxes [I, Q, 4], tg, tgt_of_q, w_noobj, specs) with
    specs = ((stride, len), ...) -> one [G, 5] tensor per spec, G = I // stride, columns (loss_ce, class_error, loss_bbox,
    loss_giou, cardinality_error) of the group's images g * stride .. g * stride + len - 1, each with its own normalisers
    (reference detr.py:220-265 called once per group).  Only the FIRST spec is differentiable (the others are bookkeeping:
    the frame-0 reward of interactron.py:104-108)."""

    @staticmethod
    def forward(ctx, logits, boxes, tg, tgt_of_q, w_noobj, specs):
        logits, boxes = _req(logits, "criterion logits"), _req(boxes, "criterion boxes")
        I, Q, C = logits.shape
        dev = logits.device
        rowstat = torch.empty(I * Q, 4, dtype=torch.float32, device=dev)
        lse = torch.empty(I * Q, dtype=torch.float32, device=dev)
        flags = torch.empty(I * Q, dtype=torch.int32, device=dev)
        L = _L()
        _chk(L.ix_set_loss_rows_f32(logits.data_ptr(), boxes.data_ptr(), tg.ids.data_ptr(), tg.boxes.data_ptr(), tg.off.data_ptr(),
                                    tgt_of_q.data_ptr(), rowstat.data_ptr(), lse.data_ptr(), flags.data_ptr(), I, Q, C, w_noobj,
                                    _stream()), "ix_set_loss_rows_f32")
        outs, norm0 = [], None
        for k, (stride, ln) in enumerate(specs):
            assert I % stride == 0 and 1 <= ln <= stride, (I, stride, ln)
            G = I // stride
            out = torch.empty(G, 5, dtype=torch.float32, device=dev)
            norm = torch.empty(G, 2, dtype=torch.float32, device=dev)
            _chk(L.ix_set_loss_groups_f32(rowstat.data_ptr(), flags.data_ptr(), tg.off.data_ptr(), stride, ln, G, Q, out.data_ptr(),
                                          norm.data_ptr(), _stream()), "ix_set_loss_groups_f32")
            outs.append(out)
            if k == 0:
                norm0 = norm
        ctx.tg, ctx.w_noobj, ctx.spec0 = tg, w_noobj, specs[0]
        ctx.save_for_backward(logits, boxes, tgt_of_q, lse, norm0)
        for o in outs[1:]:
            ctx.mark_non_differentiable(o)
        return tuple(outs)

    @staticmethod
    @once_differentiable
    def backward(ctx, g0, *_):
        logits, boxes, tgt_of_q, lse, norm0 = ctx.saved_tensors
        tg = ctx.tg
        I, Q, C = logits.shape
        stride, ln = ctx.spec0
        g0 = _req(g0.contiguous())
        dl, db = torch.empty_like(logits), torch.empty_like(boxes)
        _chk(_L().ix_set_loss_bwd_f32(logits.data_ptr(), boxes.data_ptr(), tg.ids.data_ptr(), tg.boxes.data_ptr(), tg.off.data_ptr(),
                                      tgt_of_q.data_ptr(), lse.data_ptr(), g0.data_ptr(), norm0.data_ptr(), stride, ln, I, Q, C,
                                      ctx.w_noobj, dl.data_ptr(), db.data_ptr(), _stream()), "ix_set_loss_bwd_f32")
        return dl, db, None, None, None, None


class WeightedCE(Function):
    """F.cross_entropy(logits [R,C], target [R], weight [C]) with mean reduction; also returns per-row argmax."""

    @staticmethod
    def forward(ctx, logits, target, weight):
        logits, weight = _req(logits), _req(weight)
        R, C = logits.shape
        lse = torch.empty(R, device=logits.device, dtype=torch.float32)
        argmax = torch.empty(R, device=logits.device, dtype=torch.int64)
        sums = torch.empty(2, device=logits.device, dtype=torch.float32)
        target = target.contiguous()
        wp, wn = _reduce_ws("wce", R, 0, 0, logits.device)
        _chk(_L().ix_weighted_ce_fwd_f32(logits.data_ptr(), target.data_ptr(), weight.data_ptr(), lse.data_ptr(),
                                         argmax.data_ptr(), sums.data_ptr(), R, C, wp, wn, _stream()), "ix_weighted_ce_fwd_f32")
        ctx.save_for_backward(logits, target, weight, lse, sums)
        ctx.mark_non_differentiable(argmax)
        return sums[0] / sums[1], argmax

    @staticmethod
    @once_differentiable
    def backward(ctx, g, _):
        logits, target, weight, lse, sums = ctx.saved_tensors
        R, C = logits.shape
        d = torch.empty_like(logits)
        g = g.contiguous()
        _chk(_L().ix_weighted_ce_bwd_f32(logits.data_ptr(), target.data_ptr(), weight.data_ptr(), lse.data_ptr(),
                                         sums.data_ptr(), g.data_ptr(), d.data_ptr(), R, C, _stream()),
             "ix_weighted_ce_bwd_f32")
        return d, None, None


class BoxLoss(Function):
    """Sums of L1 and (1 - GIoU) over matched (prediction row, target box) pairs -> tensor [2]."""

    @staticmethod
    def forward(ctx, pred, src_idx, tgt):
        pred, tgt = _req(pred), _req(tgt)
        out = torch.empty(2, device=pred.device, dtype=torch.float32)
        src_idx = src_idx.contiguous()
        _chk(_L().ix_box_loss_fwd_f32(pred.data_ptr(), src_idx.data_ptr(), tgt.data_ptr(), out.data_ptr(),
                                      src_idx.numel(), _stream()), "ix_box_loss_fwd_f32")
        ctx.save_for_backward(pred, src_idx, tgt)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        pred, src_idx, tgt = ctx.saved_tensors
        d = torch.empty_like(pred)
        g = g.contiguous()
        _chk(_L().ix_box_loss_bwd_f32(pred.data_ptr(), src_idx.data_ptr(), tgt.data_ptr(), g.data_ptr(), d.data_ptr(),
                                      pred.shape[0], src_idx.numel(), _stream()), "ix_box_loss_bwd_f32")
        return d, None, None


def sine_position(mask_u8, num_pos_feats=128, temperature=10000.0, scale=6.283185307179586):
    """mask uint8 [n,h,w] (1 = padded) -> [n, h*w, 2*num_pos_feats] token-major position embedding."""
    n, h, w = mask_u8.shape
    pos = torch.empty(n, h * w, 2 * num_pos_feats, device=mask_u8.device, dtype=torch.float32)
    _chk(_L().ix_sine_pos_f32(mask_u8.data_ptr(), pos.data_ptr(), n, h, w, num_pos_feats, temperature, scale, _stream()),
         "ix_sine_pos_f32")
    return pos


def mask_nearest(mask_u8, h, w):
    n, H, W = mask_u8.shape
    out = torch.empty(n, h, w, device=mask_u8.device, dtype=torch.uint8)
    _chk(_L().ix_mask_nearest_u8(mask_u8.data_ptr(), out.data_ptr(), n, H, W, h, w, _stream()), "ix_mask_nearest_u8")
    return out


# ---------------------------------------------------------------------------------------------------------
# MAML fast weights and the outer step
# ---------------------------------------------------------------------------------------------------------
def _ptr_array(tensors):
    arr = (ctypes.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr() if t is not None else None
    return arr


def _size_array(tensors):
    arr = (ctypes.c_int64 * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = t.numel() if t is not None else 0
    return arr


class ExpandEpisodes(Function):
    """apply(E, p_1..p_n) -> ([E, *p_1.shape], ..): the per-episode copies of a parameter list in one multi-tensor launch
    set (199 single-tensor launches before).  Backward = ReduceEpisodes: the sum over the E copies, i.e. the reference's
    gradient accumulation over the tasks of a batch, in a fixed order (no atomics)."""

    @staticmethod
    def forward(ctx, E, *ps):
        pc = [_req(p) for p in ps]
        outs = [torch.empty((E,) + tuple(p.shape), device=p.device, dtype=torch.float32) for p in pc]
        _chk(_L().ix_expand_multi_f32(_ptr_array(pc), _ptr_array(outs), _size_array(pc), len(pc), E, _stream()),
             "ix_expand_multi_f32")
        ctx.E = E
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        idx = [i for i, g in enumerate(gs) if g is not None and ctx.needs_input_grad[1 + i]]
        res = [None] * len(gs)
        if idx:
            for i, r in zip(idx, ReduceEpisodes.call(ctx.E, *[gs[i] for i in idx])):
                res[i] = r
        return (None,) + tuple(res)


def expand_episodes(E, tensors, groups=8):
    """ExpandEpisodes over consecutive groups of the parameter list (the detector's stages in module order) instead of one
    node for all ~200 tensors: a group's per-episode gradients are reduced and RELEASED as soon as the backward has passed
    its stage, instead of all E-copy gradients staying alive until the end of the backward (E x |theta| x 4 bytes of peak
    memory at 800x800)."""
    tensors = list(tensors)
    n = max(1, (len(tensors) + groups - 1) // groups)
    out = []
    for i in range(0, len(tensors), n):
        out.extend(ExpandEpisodes.apply(E, *tensors[i:i + n]))
    return out


class ReduceEpisodes(Function):
    """apply(E, g_1..g_n) with g_i [E, ...] -> (sum over the leading dim, ..) in one multi-tensor launch set."""

    @staticmethod
    def forward(ctx, E, *gs):
        gc = [_req(g) for g in gs]
        outs = [torch.empty(tuple(g.shape[1:]), device=g.device, dtype=torch.float32) for g in gc]
        _chk(_L().ix_reduce_multi_f32(_ptr_array(gc), _ptr_array(outs), _size_array(outs), len(gc), E, _stream()),
             "ix_reduce_multi_f32")
        ctx.E = E
        return tuple(outs)

    @staticmethod
    def backward(ctx, *hs):
        idx = [i for i, h in enumerate(hs) if h is not None and ctx.needs_input_grad[1 + i]]
        res = [None] * len(hs)
        if idx:
            for i, r in zip(idx, ExpandEpisodes.call(ctx.E, *[hs[i] for i in idx])):
                res[i] = r
        return (None,) + tuple(res)


class ClippedSGD(Function):
    """fast_i = p_i - clamp(lr*g_i, +-clip) for all tensors in one multi-tensor launch set.

    apply(lr, clip, n, p_1..p_n, g_1..g_n) -> (fast_1..fast_n).  g_i may be None (tensor passes through)."""

    @staticmethod
    def forward(ctx, lr, clip, n, *tensors):
        ps, gs = tensors[:n], tensors[n:]
        idx = [i for i in range(n) if gs[i] is not None]
        pc = [_req(ps[i]) for i in idx]
        gc = [_req(gs[i]) for i in idx]
        outs = [torch.empty_like(p) for p in pc]
        if idx:
            _chk(_L().ix_sgd_clip_multi_f32(_ptr_array(pc), _ptr_array(gc), _ptr_array(outs), _size_array(pc), len(pc),
                                            lr, clip, _stream()), "ix_sgd_clip_multi_f32")
        ctx.lr, ctx.clip, ctx.n, ctx.idx = lr, clip, n, idx
        ctx.save_for_backward(*gc)
        res = list(ps)
        for j, i in enumerate(idx):
            res[i] = outs[j]
        # pass-through tensors must not alias the inputs for autograd
        return tuple(r if i in set(idx) else r.view_as(r) for i, r in enumerate(res))

    @staticmethod
    def backward(ctx, *G):
        gs = ctx.saved_tensors
        n, idx = ctx.n, ctx.idx
        grad_p = [G[i] if ctx.needs_input_grad[3 + i] else None for i in range(n)]
        grad_g = [None] * n
        need = [j for j, i in enumerate(idx) if ctx.needs_input_grad[3 + n + i] and G[i] is not None]
        if need:
            res = _ClippedSGDBwd.call(ctx.lr, ctx.clip, len(need), *([G[idx[j]] for j in need] + [gs[j] for j in need]))
            for j, r in zip(need, res):
                grad_g[idx[j]] = r
        return (None, None, None) + tuple(grad_p) + tuple(grad_g)


class _ClippedSGDBwd(Function):
    """out_i = -lr * G_i * [|lr*g_i| <= clip]."""

    @staticmethod
    def forward(ctx, lr, clip, n, *tensors):
        Gs = [_req(t) for t in tensors[:n]]
        gs = [_req(t) for t in tensors[n:]]
        outs = [torch.empty_like(t) for t in Gs]
        _chk(_L().ix_sgd_clip_bwd_multi_f32(_ptr_array(Gs), _ptr_array(gs), _ptr_array(outs), _size_array(Gs), n, lr,
                                            clip, _stream()), "ix_sgd_clip_bwd_multi_f32")
        ctx.lr, ctx.clip, ctx.n = lr, clip, n
        ctx.save_for_backward(*gs)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *GG):
        # linear in G (the indicator is piecewise constant in g)
        gs = ctx.saved_tensors
        n = ctx.n
        res = _ClippedSGDBwd.call(ctx.lr, ctx.clip, n, *(list(GG) + list(gs)))
        return (None, None, None) + tuple(res) + (None,) * n


def accumulate_multi(dst, src):
    """dst_i += src_i for two lists of tensors in ONE multi-tensor launch (the clipped-SGD kernel with lr = -1 and no clip:
    p - clamp(-g) = p + g)."""
    dst, src = [_req(d) for d in dst], [_req(s) for s in src]
    if dst:
        _chk(_L().ix_sgd_clip_multi_f32(_ptr_array(dst), _ptr_array(src), _ptr_array(dst), _size_array(dst), len(dst), -1.0, 3.0e38,
                                        _stream()), "ix_sgd_clip_multi_f32")


def sumsq_accum(x_flat, out_scalar):
    wp, wn = _reduce_ws("scalar", 0, 0, 0, x_flat.device)
    _chk(_L().ix_sumsq_accum_f32(x_flat.data_ptr(), x_flat.numel(), out_scalar.data_ptr(), wp, wn, _stream()), "ix_sumsq_accum_f32")


def adam_step(p, g, m, v, lr, beta1, beta2, eps, step, sumsq=None, max_norm=0.0, zero_grad=False):
    _chk(_L().ix_adam_step_f32(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), lr, beta1, beta2, eps,
                               step, sumsq.data_ptr() if sumsq is not None else None, max_norm, 1 if zero_grad else 0,
                               _stream()), "ix_adam_step_f32")
    weights_changed()   # (raw-pointer update: no autograd version moves; cached weight planes are void)
