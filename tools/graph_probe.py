"""Feasibility probe: capture detector forward+backward (custom ctypes HIP launches + autograd) into a HIP graph."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from interactron_amd import Config, build_model, NestedTensor
from interactron_amd import hipops as ops
from interactron_amd.synthetic import load_procedural, synthetic_episodes
from interactron_amd.trainer import FlatOuterStep

cfg, _ = bench.model_cfg(300, 50)
model = build_model(Config(**cfg))
load_procedural(model.fusion, "fusion.")
model = model.cuda().eval()
outer = FlatOuterStep(model)   # persistent .grad views
data = bench.to_gpu(synthetic_episodes(1, tag="bench-r0"), torch.device("cuda"))
frames = data["frames"][0].clone(); masks = data["masks"][0].clone()
w = torch.randn(5, 50, 1236, device="cuda")

def fwd_bwd():
    out = model.detector(NestedTensor(frames, masks))
    loss = ops.Dot.apply(out["pred_logits"], w)
    loss.backward()
    return out["pred_logits"]

s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        outer.flat.grads.zero_()
        ref = fwd_bwd().clone()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
gref = outer.flat.grads.clone()
t = time.perf_counter()
for _ in range(5):
    fwd_bwd()
torch.cuda.synchronize()
print("eager fwd+bwd ms", (time.perf_counter() - t) / 5 * 1e3, flush=True)

outer.flat.grads.zero_()
g = torch.cuda.CUDAGraph()
t = time.perf_counter()
with torch.cuda.graph(g):
    static_out = fwd_bwd()
torch.cuda.synchronize()
print("capture s", time.perf_counter() - t, flush=True)
outer.flat.grads.zero_()
t = time.perf_counter(); g.replay(); torch.cuda.synchronize()
print("first replay ms", (time.perf_counter() - t) * 1e3, flush=True)
print("out max diff", float((static_out - ref).abs().max()), "grad rel diff",
      float((outer.flat.grads - gref).norm() / gref.norm()), flush=True)
t = time.perf_counter()
for _ in range(10):
    g.replay()
torch.cuda.synchronize()
print("replay ms", (time.perf_counter() - t) / 10 * 1e3, flush=True)
