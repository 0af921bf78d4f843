"""Per-phase wall clock of one meta-train step (synchronising diagnostic).  Usage: python tools/phase_times.py [E] [size]"""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from interactron_amd import Config, build_model
from interactron_amd.synthetic import load_procedural, synthetic_episodes
from interactron_amd.trainer import FlatOuterStep
E = int(sys.argv[1]) if len(sys.argv) > 1 else 16
size = int(sys.argv[2]) if len(sys.argv) > 2 else 300
cfg, _ = bench.model_cfg(size, 50, E)
model = build_model(Config(**cfg)); load_procedural(model.fusion, "fusion."); model = model.cuda().train()
outer = FlatOuterStep(model)
data = bench.to_gpu(synthetic_episodes(E, height=size, width=size, tag="bench-r0"), torch.device("cuda"))
random.seed(0)
for _ in range(2):
    model(data); outer.step()
model.phase_times = {}
for _ in range(3):
    model(data); outer.step()
tot = sum(model.phase_times.values())
for k, v in sorted(model.phase_times.items()):
    print("%-44s %8.1f ms  %5.1f%%" % (k, v / 3, 100 * v / tot))
print("%-44s %8.1f ms" % ("sum (per step, synchronised phases)", tot / 3))
