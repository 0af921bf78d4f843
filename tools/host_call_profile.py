import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from interactron_amd import hipops as ops
x = torch.randn(64, 64, device="cuda"); w = torch.randn(64, 64, device="cuda"); b = torch.randn(64, device="cuda")
xg = x.clone().requires_grad_(True); wg = w.clone().requires_grad_(True); bg = b.clone().requires_grad_(True)
out = torch.empty(64, 64, device="cuda")
def fb():
    y = ops.linear(xg, wg, bg)
    torch.autograd.grad(y, [xg, wg, bg], out)
for _ in range(300): fb()
torch.cuda.synchronize()
t=time.perf_counter()
for _ in range(2000): fb()
print("fb: %.1f us" % ((time.perf_counter()-t)/2000*1e6))
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(2000): fb()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
