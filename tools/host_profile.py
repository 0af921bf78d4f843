"""Host-side profile of one meta-train step (torch.profiler): which ops are launched how often, and host time.
Usage (GPU box): python tools/host_profile.py [size] [episodes]"""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from interactron_amd import Config, build_model
from interactron_amd.synthetic import load_procedural, synthetic_episodes
from interactron_amd.trainer import FlatOuterStep

size = int(sys.argv[1]) if len(sys.argv) > 1 else 300
eps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
cfg, _ = bench.model_cfg(size, 50)
model = build_model(Config(**cfg))
load_procedural(model.fusion, "fusion.")
model = model.cuda().train()
outer = FlatOuterStep(model)
data = bench.to_gpu(synthetic_episodes(eps, height=size, width=size, tag="bench-r0"), torch.device("cuda"))
random.seed(0)

def step():
    model(data); outer.step()

for _ in range(2):
    step()
torch.cuda.synchronize()
t = time.perf_counter(); step(); t_host = time.perf_counter() - t; torch.cuda.synchronize(); t_all = time.perf_counter() - t
print("size %d episodes %d: host-issue time %.1f ms, wall (incl. GPU drain) %.1f ms" % (size, eps, t_host * 1e3, t_all * 1e3))
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU]) as prof:
    step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=60, max_name_column_width=60))
