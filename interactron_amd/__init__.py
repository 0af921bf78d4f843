"""interactron_amd -- MI355X-native (gfx950) implementation of Interactron's per-episode adaptive-detection hot
path behind the reference's Python surface (``build_model`` -> ``forward / predict / get_next_action``,
``SetCriterion``, ``HungarianMatcher``, ``NestedTensor``, ``get_config``).

All compute runs in the hand-written HIP kernels of ``interactron_amd/csrc`` reached through the C-ABI declared in
``include/interactron_hip.h``; importing the package is cheap, the kernel library is loaded on first use and there
is no CPU fallback.
"""
from .config import Config, build_evaluator, build_model, build_trainer, get_args, get_config  # noqa: F401

__all__ = ["Config", "build_model", "build_trainer", "build_evaluator", "get_config", "get_args", "SetCriterion", "HungarianMatcher", "NestedTensor", "PathStorage",
           "collate_fn", "manual_seed"]


def __getattr__(name):
    if name in ("SetCriterion", "HungarianMatcher"):
        from . import criterion
        return getattr(criterion, name)
    if name == "NestedTensor":
        from .detector import NestedTensor
        return NestedTensor
    if name in ("PathStorage", "collate_fn"):
        from . import storage
        return getattr(storage, name)
    if name == "manual_seed":
        from .hipops import manual_seed
        return manual_seed
    raise AttributeError(name)
