"""Do the assignments differ between the two states of the E = 2 / 128 x 160 step?"""
import random
import sys

import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from interactron_amd import _lib, criterion as cr, hipops as ops  # noqa: E402
from test_parity_gpu import make, synthetic_episodes, to_gpu  # noqa: E402

lib = _lib.load()
data = to_gpu(synthetic_episodes(2, height=128, width=160, tag="dp"))
orig = cr.HungarianMatcher.assign


def run(x3, tr):
    lib.ix_gemm_set_x3(x3)
    ops.FLASH_TR = tr
    m = make("interactron")
    m.config.STEP_GRAPH = False
    calls = []

    def spy(matcher, costs, targets):
        res = orig(matcher, costs, targets)
        calls.append(([c.clone() for c in costs], [(a.clone(), b.clone()) for a, b in res]))
        return res

    cr.HungarianMatcher.assign = spy
    random.seed(11)
    m.zero_grad()
    try:
        m(data)
    finally:
        cr.HungarianMatcher.assign = orig
    return calls, m.detector.query_embed.weight.grad.detach().cpu().double()


a, ga = run(1, "bf16")
b, gb = run(0, "bf16")
print("query_embed moved by %.2e; %d / %d assign calls" % (float((ga - gb).norm() / ga.norm()), len(a), len(b)))
for ci, ((ca, ra), (cb, rb)) in enumerate(zip(a, b)):
    for ii, ((qa, ta), (qb, tb)) in enumerate(zip(ra, rb)):
        if not (torch.equal(qa, qb) and torch.equal(ta, tb)):
            ca_i = ca[ii]
            tot_a = float(ca_i[qa, ta].sum())
            tot_b = float(ca_i[qb, tb].sum())
            print("call %d image %d differs: queries %s vs %s; targets %s vs %s; cost of A's / B's assignment under A's matrix: %.6f / %.6f" % (
                ci, ii, qa.tolist(), qb.tolist(), ta.tolist(), tb.tolist(), tot_a, tot_b))
