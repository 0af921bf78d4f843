"""Asynchronous phase trace of one meta-train step: per phase, how long the host took to issue it, how long the GPU took
to run it, and how far the GPU is behind the host at the boundary (lag ~ 0 = the GPU is starved there).
Usage: python tools/phase_lag.py [E] [size]"""
import os, random, sys, time
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
torch.cuda.synchronize()
for rep in range(2):
    model.phase_times = []
    base = torch.cuda.Event(enable_timing=True); base.record(); t_base = time.perf_counter()
    model(data)
    ev_end = torch.cuda.Event(enable_timing=True); ev_end.record(); t_fwd = time.perf_counter()
    outer.step()
    ev_o = torch.cuda.Event(enable_timing=True); ev_o.record(); t_o = time.perf_counter()
    torch.cuda.synchronize(); t_done = time.perf_counter()
    tr = model.phase_times + [("(return of model(data))", t_fwd, ev_end), ("outer step", t_o, ev_o)]
    model.phase_times = None
    print("step %d: wall %.1f ms" % (rep, (t_done - t_base) * 1e3))
    print("%-46s %10s %10s %10s %10s" % ("phase (ends at)", "host ms", "gpu ms", "host at", "gpu lag"))
    ph, pg = 0.0, 0.0
    for name, t, ev in tr:
        h = (t - t_base) * 1e3
        g = base.elapsed_time(ev)
        print("%-46s %10.1f %10.1f %10.1f %10.1f" % (name, h - ph, g - pg, h, g - h))
        ph, pg = h, g
