import os, sys, random, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.test_parity_gpu import make, to_gpu
from interactron_amd.synthetic import synthetic_episodes
from interactron_amd import criterion as cr
ep = to_gpu(synthetic_episodes(1, tag="golden"))
m = make("detr")
orig = cr.HungarianMatcher.forward
log = []
def spy(self, outputs, targets):
    r = orig(self, outputs, targets)
    log.append(hashlib.md5(b"".join(a.numpy().tobytes() + b.numpy().tobytes() for a, b in r)).hexdigest()[:8])
    return r
cr.HungarianMatcher.forward = spy
for i in range(8):
    m.zero_grad(); log.clear()
    _, losses = m(ep)
    g = dict(m.model.named_parameters())["query_embed.weight"].grad
    print(i, "idx", log, "qe grad norm %.9g" % float(g.double().norm()), "loss_ce %.7f" % float(losses["loss_detector_ce"]),
          "cls.w %.8g" % float(dict(m.model.named_parameters())["class_embed.weight"].grad.double().norm()))
