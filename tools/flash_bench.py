"""Times the flash attention kernels (forward / backward / double backward) against the materialised attention node on
the training step's shapes (tuning aid).  usage: python tools/flash_bench.py [fusion|encoder|cross|b800]"""
import math
import sys
import time

import torch

sys.path.insert(0, ".")
from interactron_amd import hipops as ops  # noqa: E402

SHAPES = {"fusion": (16, 8, 2060, 2060, 64), "encoder": (80, 8, 361, 361, 32), "cross": (80, 8, 50, 361, 32),
          "self50": (80, 8, 50, 50, 32), "b800": (2, 8, 12755, 12755, 64), "enc800": (10, 8, 2500, 2500, 32)}


def timeit(fn, it=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(it):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / it * 1e3


def run(name, p=0.1):
    n, H, L, S, hd = SHAPES[name]
    E = H * hd
    prod = 2.0 * n * H * L * S * hd   # one [L, S] x hd product
    g = ops.AttnGeom(n, H, L, S, hd, E, E, 0, 0, E, 0, 1.0 / math.sqrt(hd))
    res = {}
    for impl in ("flash", "materialised"):
        if impl == "materialised" and n * H * L * S * 4 > 6e9:
            continue
        ops.ATTENTION_IMPL = impl
        q, k, v = (torch.randn(n, R, E, device="cuda", requires_grad=True) for R in (L, S, S))
        gy = torch.randn(n, L, E, device="cuda", requires_grad=True)
        ws = [torch.randn_like(t) for t in (q, k, v)]
        node = ops.FlashAttention if impl == "flash" else ops.AttentionCore
        out = node.apply(q, k, v, g, None, p, 77)
        t_f = timeit(lambda: node.apply(q, k, v, g, None, p, 77))
        g1 = torch.autograd.grad(out, [q, k, v], gy, create_graph=True)
        t_b = timeit(lambda: torch.autograd.grad(out, [q, k, v], gy, retain_graph=True))
        s = sum((a * w).sum() for a, w in zip(g1, ws))
        t_bb = timeit(lambda: torch.autograd.grad(s, [q, k, v, gy], retain_graph=True))
        res[impl] = (t_f, t_b, t_bb)
        del out, g1, s
        torch.cuda.empty_cache()
    f = res["flash"]
    line = "%-8s p=%.1f flash: fwd %.3f ms (%.0f TF/s of 2 products), bwd %.3f (%.0f of 7), bwd_bwd %.3f (%.0f of 22)" % (
        name, p, f[0], 2 * prod / f[0] / 1e9, f[1], 7 * prod / f[1] / 1e9, f[2], 22 * prod / f[2] / 1e9)
    if "materialised" in res:
        m = res["materialised"]
        line += " | materialised: %.3f / %.3f / %.3f" % m
    print(line, flush=True)


import os
for name in (sys.argv[1:] or ["fusion", "encoder", "cross", "self50"]):
    run(name, float(os.environ.get("IX_BENCH_PDROP", "0.1")))
