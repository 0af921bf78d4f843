"""Outer (meta) optimisation step and multi-GPU episode sharding.

reference engine/interactron_trainer.py:93-111: after ``model(data)`` has accumulated gradients, the trainer runs
``clip_grad_norm_(model.parameters(), GRAD_NORM_CLIP)`` and steps two Adams (detector lr 1e-5, fusion lr 1e-4).
The reference's multi-GPU story is ``nn.DataParallel`` (interactron_trainer.py:44-46); here it is one process per
GPU: every rank runs the episodes ``rank::world_size`` of the global batch, all parameters and their gradients live
in two flat fp32 buffers, and ONE RCCL all-reduce (SUM, matching the reference's un-normalised accumulation over
the batch) of the flat gradient buffer over xGMI precedes the clip + Adam, which run identically on every rank as
two fused HIP launches (``ix_sumsq_accum_f32`` + ``ix_adam_step_f32``).
"""
import os

import torch
import torch.distributed as dist

ALIGN = 64  # elements; keeps every parameter view 256-byte aligned inside the flat buffer (float4 loads stay legal)


def shard_batch(data, rank, world_size, by_root=False):
    """Episodes ``rank::world_size`` of a collated batch (reference collate layout, utils/storage_utils.py:53-64).

    ``by_root``: also attach what keeps ``model.path_storage`` identical to a single process (reference
    models/interactron.py:109-115 keys it by root image and fills it episode after episode): the root paths and actions
    of the WHOLE batch plus this rank's positions in it (``dp_roots``, ``dp_actions``, ``dp_index``, ``dp_world``).  The
    model then exchanges the per-episode rewards (a few floats per chunk) and replays the global batch in global order."""
    if world_size == 1:
        return data
    total = data["frames"].shape[0]
    idx = list(range(rank, total, world_size))
    out = {}
    for k, v in data.items():
        if torch.is_tensor(v):
            out[k] = v[idx]
        elif isinstance(v, (list, tuple)):
            out[k] = [v[i] for i in idx]
        else:
            out[k] = v
    if by_root and "initial_image_path" in data and "actions" in data:
        out["dp_roots"] = list(data["initial_image_path"])
        out["dp_actions"] = data["actions"][:, :4].tolist()
        out["dp_index"] = idx
        out["dp_world"] = world_size
    return out


def init_distributed(backend=None):
    """(rank, local_rank, world_size) from the torchrun environment; initialises the process group when needed."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and torch.cuda.is_available():
        # bind this rank's GPU first, whether or not the group exists yet: RCCL communicators are created on the current
        # device, and everything built afterwards (evaluator, trainer, flat buffers) must land on cuda:LOCAL_RANK
        torch.cuda.set_device(local % torch.cuda.device_count())
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = backend or os.environ.get("IX_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local, world


class FlatBuffers:
    """Re-homes the trainable parameters of ``groups`` (list of parameter lists) into one flat buffer and gives every
    parameter a persistent ``.grad`` view into a second flat buffer.  Pure tensor plumbing: works on any device."""

    def __init__(self, groups):
        self.groups = [[p for p in g if p.requires_grad] for g in groups]
        offs, total = [], 0
        self.segments = []
        for g in self.groups:
            start = total
            for p in g:
                offs.append(total)
                total += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
            self.segments.append((start, total))
        params = [p for g in self.groups for p in g]
        dev = params[0].device
        self.params = torch.zeros(total, device=dev, dtype=torch.float32)
        self.grads = torch.zeros(total, device=dev, dtype=torch.float32)
        self.grad_views = []   # (parameter, its slot of the flat gradient buffer)
        for p, o in zip(params, offs):
            n = p.numel()
            self.params[o:o + n].copy_(p.data.reshape(-1))
            if p.grad is not None:
                self.grads[o:o + n].copy_(p.grad.reshape(-1))
            p.data = self.params[o:o + n].view(p.shape)
            p.grad = self.grads[o:o + n].view(p.shape)
            self.grad_views.append((p, p.grad))
        if dev.type == "cuda":
            from . import hipops
            hipops.weights_changed()

    def sync_b16(self):
        """16-bit activation mode: ONE conversion pass over the flat parameter buffer into a flat bf16 shadow, and every parameter's
        view of it handed to the weight cache (b16.weight_b16) -- instead of one pass per weight tensor after every optimiser step
        (286 launches, 1.6 ms per multi_frame_baseline step, profiles/r6g_mfb_bf16_kernel_stats.csv).  Valid until the parameters
        change again (the cache epoch, the tensor's version and address are recorded)."""
        from . import b16, hipops as ops
        if getattr(self, "params_b16", None) is None:
            self.params_b16 = torch.empty(self.params.numel(), device=self.params.device, dtype=torch.bfloat16)
            self._b16_views = []
            o = 0
            for g in self.groups:
                for p in g:
                    n = p.numel()
                    self._b16_views.append((p, self.params_b16[o:o + n].view(p.shape)))
                    o += (n + ALIGN - 1) // ALIGN * ALIGN
        b16.cast_b16_into(self.params, self.params_b16)
        epoch = ops._wp_epoch[0]
        for p, v in self._b16_views:
            p.__dict__["_ix_b16_flat"] = (v, epoch, p._version, p.data_ptr())

    def all_reduce_grads(self):
        """The path's single data-path collective: SUM of the flat meta-gradient buffer over all ranks."""
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(self.grads, op=dist.ReduceOp.SUM)


class FlatOuterStep:
    """clip_grad_norm_(all, max_norm) + Adam(detector) + Adam(fusion) on the flat buffers (HIP kernels)."""

    def __init__(self, model, detector_lr=1e-5, fusion_lr=1e-4, max_norm=1.0, betas=(0.9, 0.999), eps=1e-8, groups=None,
                 lrs=None, steal_grads=None):
        """Default: Adam(detector, detector_lr) + Adam(fusion, fusion_lr) like the interactron trainers; ``groups`` /
        ``lrs`` give explicit parameter lists and learning rates (direct-supervision trainer: one group)."""
        if groups is not None:
            self.lrs = list(lrs)
        else:
            groups = [list(model.detector.parameters())]
            self.lrs = [detector_lr]
            if hasattr(model, "fusion"):
                groups.append(list(model.fusion.parameters()))
                self.lrs.append(fusion_lr)
        self.flat = FlatBuffers(groups)
        self.m = torch.zeros_like(self.flat.params)
        self.v = torch.zeros_like(self.flat.params)
        self.sumsq = torch.zeros((), device=self.flat.params.device, dtype=torch.float32)
        self.max_norm, self.betas, self.eps, self.t = max_norm, betas, eps, 0
        # Models whose step is ONE plain backward pass and no captured graph (detr, detr_multiframe): the parameters go into the backward
        # WITHOUT a .grad, so autograd's AccumulateGrad keeps the incoming gradient tensor instead of launching one at::add per parameter
        # into the flat views (316 launches, 1.4 ms of the 41 ms multi_frame_baseline step in the 16-bit mode); step() copies them into
        # the flat buffer with one multi-tensor launch set and drops them again.  The episode models (two backward passes, gradients
        # accumulated in place by captured graphs: graphs.chunk_runner) keep the persistent views.
        from .episode import _Adaptive
        if steal_grads is None:
            steal_grads = os.environ.get("IX_STEAL_GRADS", "1") == "1"
        self.steal_grads = bool(steal_grads) and (not isinstance(model, _Adaptive)) and self.flat.params.is_cuda
        if self.steal_grads:
            self._release_grads()
        self.shadow_b16 = getattr(model, "compute_dtype", "f32") in ("bf16", "bf16_fusion") and self.flat.params.is_cuda
        if self.shadow_b16:
            self.flat.sync_b16()

    def zero_grads(self):
        """Drop whatever gradients are pending (the reference's test epoch back-propagates inside model() too; those are discarded)."""
        self.flat.grads.zero_()
        if self.steal_grads:
            self._release_grads()

    def _release_grads(self):
        for p, _ in self.flat.grad_views:
            p.grad = None

    def _collect_grads(self):
        from . import hipops as ops
        src, dst = [], []
        for p, view in self.flat.grad_views:
            g = p.grad
            if g is None or g.data_ptr() == view.data_ptr():
                continue   # (no gradient this step: the slot is zero since the last optimiser step | already the flat view)
            src.append(g if g.is_contiguous() else g.contiguous())
            dst.append(view)
        ops.copy_multi(src, dst)

    def step(self, all_reduce=True):
        """``all_reduce=False``: the caller has already summed the gradients over the ranks (tests that inspect them)."""
        from . import hipops as ops
        f = self.flat
        if self.steal_grads:
            self._collect_grads()
        if all_reduce:
            f.all_reduce_grads()
        self.t += 1
        self.sumsq.zero_()
        ops.sumsq_accum(f.grads, self.sumsq)
        for (a, b), lr in zip(f.segments, self.lrs):
            ops.adam_step(f.params[a:b], f.grads[a:b], self.m[a:b], self.v[a:b], lr, self.betas[0], self.betas[1],
                          self.eps, self.t, self.sumsq, self.max_norm, zero_grad=True)
        if self.shadow_b16:
            f.sync_b16()
        if self.steal_grads:
            self._release_grads()
        return torch.sqrt(self.sumsq)
