"""torch.autograd bindings of the HIP kernels (C-ABI in include/interactron_hip.h).

Every compute op of the hot path is a ``torch.autograd.Function`` whose forward *and* backward launch hand-written gfx950 kernels
through ctypes; PyTorch only provides device memory, streams and the autograd tape.  The backward of each Function is itself
expressed with Functions of this package, so the op set is closed under differentiation: ``torch.autograd.grad(..., create_graph=True)``
followed by ``.backward()`` -- the MAML meta-gradient of reference models/interactron.py:99-123 -- runs entirely on these kernels.
No CPU fallback exists: calling any op without the built library or with CPU tensors raises.

The package is split by concern (round 6; it was one 2 800-line module):

    core           library handle, streams, the Function base class (fp32 / 16-bit dispatch), scratch, capture state, the
                   compute mode of a model, identities of weights and skipped gradients
    contraction    strided batched contractions and their routing (12-wave kernels, weight planes, bf16 GEMM), Linear layers
    elementwise    elementwise / broadcast / reduction Functions, dropout, softmax, LayerNorm
    convolution    frozen-BN affines, fused contraction + BN, implicit-GEMM convolutions
    attn           materialised and flash attention, operand planes, key biases
    criterion_ops  matcher cost, assignment, set criterion, policy cross-entropy, position embedding
    meta_ops       multi-tensor MAML plumbing and the fused outer step
    (../b16.py     the 16-bit activation mode: conversions, the adapter, the twins of the ops above)

``interactron_amd.hipops`` stays the one name the rest of the package (and the tests) use: functions, classes and containers of the
parts are re-exported here; the parts' SWITCHES (module-level strings / numbers / flags such as FLASH_TR, ATTENTION_DTYPE, GEMM_WP,
COMPUTE_DTYPE) live in exactly one part each and are read and written THROUGH this module -- ``hipops.FLASH_TR = "bf16"`` sets
``hipops.attn.FLASH_TR``, the only place the kernels' callers read it -- and a function replaced here (tests pin
``_next_seed``, tools wrap ``_run_gemm``) is replaced in its part as well.
"""
import sys
import types

from . import core, elementwise, contraction, convolution, attn, criterion_ops, meta_ops

_PARTS = (core, elementwise, contraction, convolution, attn, criterion_ops, meta_ops)
_SWITCH = {}    # name -> the part that owns a rebindable scalar
_OWNER = {}     # name -> the part that defines a function / class / container
for _m in _PARTS:
    for _k, _v in vars(_m).items():
        if _k.startswith("__") or isinstance(_v, types.ModuleType):
            continue
        if getattr(_v, "__module__", _m.__name__) != _m.__name__ and not isinstance(_v, (str, bool, int, float, type(None), list, dict, tuple)):
            continue   # (something a part imported from another part or from torch: its owner exports it)
        if isinstance(_v, (str, bool, int, float, type(None))):
            _SWITCH.setdefault(_k, _m)
        else:
            _OWNER.setdefault(_k, _m)
for _k, _m in _OWNER.items():
    globals()[_k] = getattr(_m, _k)


class _Facade(types.ModuleType):
    def __getattr__(self, name):   # (only reached for names that are not attributes of this module: the switches)
        m = _SWITCH.get(name)
        if m is None:
            raise AttributeError("module %r has no attribute %r" % (__name__, name))
        return getattr(m, name)

    def __setattr__(self, name, value):
        m = _SWITCH.get(name)
        if m is not None:
            setattr(m, name, value)
            return
        m = _OWNER.get(name)
        if m is not None:
            setattr(m, name, value)
        super().__setattr__(name, value)


sys.modules[__name__].__class__ = _Facade
