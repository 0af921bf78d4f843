"""Which attention site makes query_embed's meta-gradient move when it runs in the f16 tr form?  (E = 2, 128 x 160)"""
import random
import sys

import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from interactron_amd import hipops as ops  # noqa: E402
from test_parity_gpu import make, synthetic_episodes, to_gpu  # noqa: E402

orig = ops.flash_forward
sel = [lambda g: False]


def fwd(q, k, v, g, mask, p, seed, **kw):
    ops.FLASH_TR = "f16" if sel[0](g) else "bf16"
    return orig(q, k, v, g, mask, p, seed, **kw)


ops.flash_forward = fwd
data = to_gpu(synthetic_episodes(2, height=128, width=160, tag="dp"))


def run(f):
    sel[0] = f
    m = make("interactron")
    m.config.STEP_GRAPH = False
    random.seed(11)
    m.zero_grad()
    m(data)
    return {k: p.grad.detach().double().cpu() for k, p in m.named_parameters() if p.grad is not None}


base = run(lambda g: False)
sites = {"none (repeat)": lambda g: False, "all": lambda g: True, "fusion hd64": lambda g: g.hd == 64,
         "encoder self": lambda g: g.hd == 32 and g.L == g.S and g.L != 50, "decoder self": lambda g: g.hd == 32 and g.L == 50 and g.S == 50,
         "decoder cross": lambda g: g.hd == 32 and g.L == 50 and g.S != 50}
for name, f in sites.items():
    r = run(f)
    k = "detector.query_embed.weight"
    rel = sorted(((float((r[n] - base[n]).norm() / base[n].norm().clamp_min(1e-300)), n) for n in base if base[n].norm() > 1e-4), reverse=True)
    print("%-14s query_embed %.2e; worst others: %s" % (name, float((r[k] - base[k]).norm() / base[k].norm()),
                                                       ", ".join("%s %.1e" % (n.split("detector.")[-1][-40:], v) for v, n in rel[:3])))
r1, r2 = run(lambda g: True), run(lambda g: True)
k = "detector.query_embed.weight"
print("all twice identical:", all(torch.equal(r1[n], r2[n]) for n in r1))
d = (r1[k] - base[k])
print("rows |d| / |base row|:", ["%.1e" % float(x) for x in (d.norm(dim=1) / base[k].norm(dim=1))])
print("base row norms:", ["%.1e" % float(x) for x in base[k].norm(dim=1)])
print("cols: diff norm by 32-col block", ["%.1e" % float(x) for x in d.view(50, 8, 32).norm(dim=(0, 2))])
