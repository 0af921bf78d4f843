"""Run-to-run spread of the ill-conditioned scalar gradient fusion.loss_decoder.layers.2.bias (config 3 and G13)."""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.test_parity_gpu import make, to_gpu
from tests.helpers import ReferenceMatching
from interactron_amd.synthetic import synthetic_episodes
G = torch.load("tests/golden/golden_indices.pt", weights_only=False)
ep = to_gpu(synthetic_episodes(1, tag="golden"))
m = make("interactron_random")
vals = []
for i in range(8):
    m.zero_grad(); random.seed(7)
    with ReferenceMatching(G["random_forward"]):
        m(ep)
    vals.append(float(dict(m.fusion.named_parameters())["loss_decoder.layers.2.bias"].grad))
print("config3 loss_decoder.layers.2.bias grads:", ["%.4e" % v for v in vals], "(reference -6.7501e-05 / float64 6.7522e-05 in norm)")
