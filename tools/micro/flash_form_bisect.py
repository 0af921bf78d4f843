"""Where do the two tr forms of the flash kernels part ways inside predict()?  Records every flash forward / backward result of
two eager predict() calls (bf16 form, then f16 form) and prints the relative difference call by call."""
import sys

import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from interactron_amd import hipops as ops  # noqa: E402
from test_parity_gpu import make, synthetic_episodes, to_gpu  # noqa: E402

rec = {"bf16": [], "f16": []}
cur = ["bf16"]
orig_fwd = ops.flash_forward
orig_bwd = ops.FlashAttentionBwd.forward


def fwd(q, k, v, g, mask, p, seed, **kw):
    r = orig_fwd(q, k, v, g, mask, p, seed, **kw)
    rec[cur[0]].append(("fwd n%d H%d L%d S%d hd%d" % (g.n, g.heads, g.L, g.S, g.hd), [r[0].detach().clone(), r[1].detach().clone()]))
    return r


def bwd(ctx, q, k, v, out, lse, do, g, p, seed, pl, same_qk):
    r = orig_bwd(ctx, q, k, v, out, lse, do, g, p, seed, pl, same_qk)
    rec[cur[0]].append(("bwd n%d H%d L%d S%d hd%d same=%s do=%.2e" % (g.n, g.heads, g.L, g.S, g.hd, same_qk, float(do.abs().max())),
                        [t.detach().clone() for t in r if t is not None]))
    return r


ops.flash_forward = fwd
ops.FlashAttentionBwd.forward = staticmethod(bwd)
m = make("interactron")
m.__dict__.setdefault("_predict_graphs", {})["disabled"] = True
ep = to_gpu(synthetic_episodes(1, tag="golden"))
for form in ("bf16", "f16"):
    cur[0] = ops.FLASH_TR = form
    m.predict(ep)
for (na, ta), (nb, tb) in zip(rec["bf16"], rec["f16"]):
    d = ["%.1e" % float((a - b).abs().max() / a.abs().max().clamp_min(1e-30)) for a, b in zip(ta, tb)]
    print(na, d)
