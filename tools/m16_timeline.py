"""Clock-stamp timeline of flash16_bb_q_kernel (diagnostic build libix_m16_diag.so, -DM16_DIAG): per tile, the cycles
between the stamps of wave 0 and wave 4 (the two waves of one SIMD) of workgroup (0, 0), tiles 8..15.
usage: IX_LIB_PATH=interactron_amd/lib/libix_m16_diag.so python tools/m16_timeline.py"""
import ctypes
import math
import sys

import torch

sys.path.insert(0, ".")
from interactron_amd import _lib, hipops as ops  # noqa: E402

n, H, L, S, hd = 2, 8, 12755, 12755, 64
E = H * hd
g = ops.AttnGeom(n, H, L, S, hd, E, E, 0, 0, E, 0, 1.0 / math.sqrt(hd))
q, k, v = (torch.randn(n, R, E, device="cuda", requires_grad=True) for R in (L, S, S))
gy = torch.randn(n, L, E, device="cuda", requires_grad=True)
ws = [torch.randn_like(t) for t in (q, k, v)]
out = ops.FlashAttention.apply(q, k, v, g, None, 0.1, 77)
g1 = torch.autograd.grad(out, [q, k, v], gy, create_graph=True)
s = sum((a * w).sum() for a, w in zip(g1, ws))
for _ in range(2):
    torch.autograd.grad(s, [q, k, v, gy], retain_graph=True)
torch.cuda.synchronize()
lib = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * 256)()
assert lib.ix_diag_m16_read(buf) == 0
names = ["V start", "V end", "barrier passed", "M end", "barrier passed"]
for w in range(2):
    print("wave %d: cycles per segment, tiles 8..15" % (4 * w))
    rows = []
    for t in range(8):
        st = [buf[(w * 8 + t) * 16 + i] for i in range(5)]
        rows.append([st[i + 1] - st[i] for i in range(4)] + [st[4] - st[0]])
    for i in range(4):
        print("  %-14s -> %-14s %s" % (names[i], names[i + 1], " ".join("%6d" % r[i] for r in rows)))
    print("  %-32s %s" % ("whole tile", " ".join("%6d" % r[4] for r in rows)))
print("wave 4 V start minus wave 0 V start per tile:", [buf[(8 + t) * 16] - buf[t * 16] for t in range(8)])
