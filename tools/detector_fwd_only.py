"""Detector forward of 80 frames with per-episode weights (phase 1 of the step), alone, for rocprofv3 --stats."""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from interactron_amd import Config, build_model, hipops as ops
from interactron_amd.detector import NestedTensor
from interactron_amd.meta import get_parameters, set_parameters
from interactron_amd.synthetic import load_procedural, synthetic_episodes
cfg, _ = bench.model_cfg(300, 50, 16)
model = build_model(Config(**cfg)); load_procedural(model.fusion, "fusion."); model = model.cuda().train()
data = bench.to_gpu(synthetic_episodes(16, height=300, width=300, tag="bench-r0"), torch.device("cuda"))
E, s = 16, 5
frames = data["frames"].reshape(E * s, 3, 300, 300); masks = data["masks"].reshape(E * s, 300, 300)
theta = get_parameters(model.detector)
dtheta = [ops.BcastRows.apply(p.detach().reshape(-1), E).reshape((E,) + tuple(p.shape)).requires_grad_(True) for p in theta]
set_parameters(model.detector, dtheta)
for i in range(6):
    out = model.detector(NestedTensor(frames, masks))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(5):
    out = model.detector(NestedTensor(frames, masks))
e1.record(); torch.cuda.synchronize()
print("detector forward (80 frames, graph recorded): %.1f ms" % (e0.elapsed_time(e1) / 5))
