"""MAML parameter plumbing with the reference's semantics (utils/meta_utils.py:5-142).

``get_parameters`` only descends into children: a module with both own parameters and children contributes only its
children's (so ``MultiheadAttention.in_proj_*`` are never adapted).  ``sgd_step`` runs the fused multi-tensor HIP
kernel ``ix_sgd_clip_multi_f32`` and stays differentiable w.r.t. the gradients (second-order path).
"""
import collections

from . import hipops as ops


def get_parameters(model):
    children = list(model.children())
    if not children:
        return [p for p in model._parameters.values() if p is not None and p.requires_grad]
    out = []
    for child in children:
        out.extend(get_parameters(child))
    return tuple(out)


def set_parameters(model, params):
    if not isinstance(params, collections.abc.Iterator):
        params = iter(params)
    children = list(model.children())
    if not children:
        for name, p in model._parameters.items():
            if p is not None and p.requires_grad:
                model._parameters[name] = next(params)
    else:
        for child in children:
            set_parameters(child, params)
    return []


def clone_parameters(params):
    return tuple(p.clone() for p in params)


def detach_parameters(params):
    out = []
    for p in params:
        d = p.clone().detach()
        d.requires_grad = True
        out.append(d)
    return tuple(out)


def detach_gradients(grads):
    return tuple(None if g is None else g.clone().detach() for g in grads)


def sgd_step(params, grads, lr, clip=0.01):
    """p - clip(lr*g, -clip, clip) per tensor; a ``None`` gradient passes the parameter through."""
    params, grads = list(params), list(grads)
    n = len(params)
    return tuple(ops.ClippedSGD.apply(float(lr), float(clip), n, *(params + grads)))
