"""ctypes binding of ``libinteractron_hip.so`` (the C-ABI in ``include/interactron_hip.h``).

The argument types are derived from the header itself, so the header stays the single source of truth for the
boundary.  There is no CPU fallback: if the library (or a symbol) is missing every op raises.
"""
import ctypes
import os
import re

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("IX_LIB_PATH") or os.path.join(_PKG, "lib", "libinteractron_hip.so")   # (IX_LIB_PATH: diagnostic builds)
HEADER_PATH = os.path.join(os.path.dirname(_PKG), "include", "interactron_hip.h")

_SCALARS = {"int": ctypes.c_int, "int64_t": ctypes.c_int64, "uint64_t": ctypes.c_uint64, "float": ctypes.c_float,
            "size_t": ctypes.c_size_t}


class HipLibraryError(RuntimeError):
    pass


def parse_header(path=HEADER_PATH):
    """-> {name: (restype, [argtypes])} for every function declared in the header."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    out = {}
    for m in re.finditer(r"(const char\*|int)\s+(ix_\w+)\s*\(([^)]*)\)\s*;", text):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        argtypes = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a or a.startswith("ix_stream_t"):
                    argtypes.append(ctypes.c_void_p)
                else:
                    argtypes.append(_SCALARS[a.split()[0]])
        out[name] = (ctypes.c_char_p if ret.startswith("const char") else ctypes.c_int, argtypes)
    return out


_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryError(
            "HIP kernel library not built: %s is missing. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C interactron_amd/csrc`). There is no CPU fallback." % LIB_PATH)
    # PyTorch first: its wheel bundles its own HIP runtime (torch/lib/libamdhip64.so).  Loaded after it, this library binds
    # to that same runtime instance; loaded BEFORE torch it would pull in /opt/rocm's copy as a second runtime in the
    # process, and kernels launched on torch's streams then fail with "no ROCm-capable device is detected".
    import torch  # noqa: F401
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise HipLibraryError("cannot load %s: %s" % (LIB_PATH, e))
    for name, (restype, argtypes) in parse_header().items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise HipLibraryError("%s does not export %s declared in %s" % (LIB_PATH, name, HEADER_PATH))
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().ix_last_error()
        raise HipLibraryError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else "?"))
