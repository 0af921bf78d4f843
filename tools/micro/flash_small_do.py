import math, sys
import torch
sys.path.insert(0, ".")
from interactron_amd import hipops as ops

def ref(q, k, v, H, scale):
    n, L, E = q.shape
    hd = E // H
    qh, kh, vh = (t.view(n, -1, H, hd).transpose(1, 2) for t in (q, k, v))
    p = torch.softmax(qh @ kh.transpose(-1, -2) * scale, -1)
    return (p @ vh).transpose(1, 2).reshape(n, L, E)

torch.manual_seed(0)
for (n, H, L, S, hd) in ((1, 8, 2060, 2060, 64), (1, 2, 300, 517, 64), (5, 8, 50, 361, 32)):
    E = H * hd
    q, k, v = torch.randn(n, L, E), torch.randn(n, S, E), torch.randn(n, S, E)
    for gs in (1.0, 1e-3, 1e-6):
        gy = torch.randn(n, L, E) * gs
        xr = [t.double().requires_grad_(True) for t in (q, k, v)]
        gr = torch.autograd.grad(ref(*xr, H, 1 / math.sqrt(hd)), xr, gy.double())
        for form in ("bf16", "f16"):
            ops.FLASH_TR = form
            g = ops.AttnGeom(n, H, L, S, hd, E, E, 0, 0, E, 0, 1 / math.sqrt(hd))
            xh = [t.cuda().requires_grad_(True) for t in (q, k, v)]
            oh = ops.FlashAttention.apply(xh[0], xh[1], xh[2], g, None, 0.0, 0)
            gh = torch.autograd.grad(oh, xh, gy.cuda())
            print((n, H, L, S, hd), "gy x%g" % gs, form, ["%.1e" % float((a.cpu().double() - b).abs().max() / b.abs().max()) for a, b in zip(gh, gr)])
