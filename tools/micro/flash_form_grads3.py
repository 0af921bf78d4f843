"""How far does query_embed's meta-gradient (E = 2, 128 x 160) move under perturbations that are NOT the flash tr form?"""
import random
import sys

import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from interactron_amd import _lib, hipops as ops  # noqa: E402
from test_parity_gpu import make, synthetic_episodes, to_gpu  # noqa: E402

lib = _lib.load()
data = to_gpu(synthetic_episodes(2, height=128, width=160, tag="dp"))


def run(x3, tr, chunk=None):
    lib.ix_gemm_set_x3(x3)
    ops.FLASH_TR = tr
    m = make("interactron")
    m.config.STEP_GRAPH = False
    if chunk:
        m.config.EPISODE_CHUNK = chunk
    random.seed(11)
    m.zero_grad()
    _, losses = m(data)
    return {k: p.grad.detach().double().cpu() for k, p in m.named_parameters() if p.grad is not None}, {k: float(v.sum()) for k, v in losses.items()}


base, lb = run(1, "bf16")
k = "detector.query_embed.weight"
for name, args in (("x6 bf16", (0, "bf16")), ("x3 f16", (1, "f16")), ("x6 f16", (0, "f16")), ("x3 bf16 chunk1", (1, "bf16", 1)), ("x3 f16 chunk1", (1, "f16", 1))):
    r, l = run(*args)
    rel = sorted((float((r[n] - base[n]).norm() / base[n].norm().clamp_min(1e-300)) for n in base if base[n].norm() > 1e-4), reverse=True)
    print("%-16s query_embed %.2e  rows10/27 %.2e %.2e  median tensor %.2e  2nd worst %.2e  losses max rel %.1e" % (
        name, float((r[k] - base[k]).norm() / base[k].norm()), float((r[k][10] - base[k][10]).norm() / base[k][10].norm()),
        float((r[k][27] - base[k][27]).norm() / base[k][27].norm()), rel[len(rel) // 2], rel[1],
        max(abs(l[n] - lb[n]) / max(abs(lb[n]), 1e-12) for n in lb)))
