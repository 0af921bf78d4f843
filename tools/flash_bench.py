"""Times the flash attention kernels against the materialised attention node on the step's shapes (tuning aid)."""
import math
import sys
import time

import torch

sys.path.insert(0, ".")
from interactron_amd import hipops as ops  # noqa: E402


def timeit(fn, it=10):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(it):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / it * 1e3


for n, H, L, S, hd, p in [(16, 8, 2060, 2060, 64, 0.0), (16, 8, 2060, 2060, 64, 0.1), (80, 8, 361, 361, 32, 0.1),
                          (80, 8, 50, 361, 32, 0.1), (2, 8, 12755, 12755, 64, 0.1)]:
    E = H * hd
    q, k, v = (torch.randn(n, R, E, device="cuda") for R in (L, S, S))
    g = ops.AttnGeom(n, H, L, S, hd, E, E, 0, 0, E, 0, 1.0 / math.sqrt(hd))
    flops = 4.0 * n * H * L * S * hd
    t_split = timeit(lambda: ops.attn_split(q, n, L, E, 0, H, hd))
    qr, qt = ops.attn_split(q, n, L, E, 0, H, hd)
    kr, kt = ops.attn_split(k, n, S, E, 0, H, hd)
    vr, vt = ops.attn_split(v, n, S, E, 0, H, hd)
    bias = ops.attn_bias(None, n, S, q.device)
    Lp, Sp = ops._pad128(L), ops._pad128(S)
    out = torch.empty(n, L, E, device="cuda")
    lse = torch.empty(n * H, Lp, device="cuda")

    def fwd():
        ops._chk(ops._L().ix_flash_fwd_f32(qr.data_ptr(), kr.data_ptr(), vt.data_ptr(), bias.data_ptr(), out.data_ptr(),
                                           lse.data_ptr(), n, H, L, Lp, S, Sp, hd, E, 0, g.scale, p, 1234, ops._stream()), "fwd")
    t_f = timeit(fwd)
    line = "n=%d H=%d L=%d S=%d hd=%d p=%.1f: split %.3f ms, flash fwd %.3f ms = %.1f TFLOP/s fp32-eq" % (
        n, H, L, S, hd, p, t_split, t_f, flops / t_f / 1e9)
    if n * H * L * S * 4 < 8e9:
        with torch.no_grad():
            t_m = timeit(lambda: ops.AttentionCore.forward(ops._NullCtx(), q, k, v, g, None, p, 1234))
        line += "; materialised fwd %.3f ms" % t_m
    print(line, flush=True)
