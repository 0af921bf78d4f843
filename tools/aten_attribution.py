"""Which host code launches the ATen kernels of one meta-training step?  Runs the headline workload under torch.profiler
(python stacks on) and prints, per ATen op family, the total device time grouped by (input shapes, innermost frame of
this package) -- the work list for DESIGN.md's "ATen in the step" table."""
import collections
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench

args = bench.parse_args([]) if hasattr(bench, "parse_args") else None
from interactron_amd import Config, build_model
from interactron_amd.synthetic import load_procedural, synthetic_episodes
from interactron_amd.trainer import FlatOuterStep

size, episodes, chunk = int(os.environ.get("SIZE", 300)), int(os.environ.get("EPISODES", 16)), int(os.environ.get("CHUNK", 8))
cfg, _ = bench.model_cfg(size, 50, chunk, "interactron")
cfg["STEP_GRAPH"] = "false"   # (attribute the eager launches: a graph replay has no Python stacks)
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
model = build_model(Config(**cfg))
load_procedural(model.fusion, "fusion.")
model = model.to(dev).train()
outer = FlatOuterStep(model, detector_lr=1e-5, fusion_lr=1e-4, max_norm=1.0)
data = bench.to_gpu(synthetic_episodes(episodes, height=size, width=size, tag="attr"), dev)


def step():
    model(data)
    outer.step()


step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()

PKG = os.sep + "interactron_amd" + os.sep
agg = collections.defaultdict(lambda: [0.0, 0])
fam = collections.defaultdict(lambda: [0.0, 0])
for ev in prof.events():
    if not ev.name.startswith("aten::") or ev.device_time_total <= 0 or ev.cpu_children:
        # leaf aten ops only (the op that owns the kernel)
        if not (ev.name.startswith("aten::") and ev.self_device_time_total > 0):
            continue
    t = ev.self_device_time_total
    if t <= 0:
        continue
    site = "?"
    for fr in ev.stack or []:
        if PKG in fr:
            site = fr.split(PKG)[-1]
            break
    else:
        if ev.stack:
            site = "(autograd engine)" if any("backward" in f or "autograd" in f for f in ev.stack) else ev.stack[0][-60:]
        else:
            # autograd engine thread: name the node being evaluated (the enclosing "autograd::engine::evaluate_function: X")
            par, site = ev.cpu_parent, "(no stack: autograd engine thread)"
            while par is not None:
                if "evaluate_function" in par.name or par.name.endswith("Backward") or "AccumulateGrad" in par.name:
                    site = "engine: " + par.name.split("evaluate_function: ")[-1]
                    break
                par = par.cpu_parent
    shapes = str([s for s in (ev.input_shapes or []) if s])[:70]
    agg[(ev.name, shapes, site)][0] += t
    agg[(ev.name, shapes, site)][1] += 1
    fam[ev.name][0] += t
    fam[ev.name][1] += 1
print("== ATen op families (self device time, one step)")
for k, (t, n) in sorted(fam.items(), key=lambda kv: -kv[1][0])[:20]:
    print("%-28s %8.2f ms %6d calls" % (k, t / 1e3, n))
print("== by (op, shapes, site)")
rows = sorted(agg.items(), key=lambda kv: -kv[1][0])[:70]
for (name, shapes, site), (t, n) in rows:
    print("%-22s %7.2f ms %5d  %-70s %s" % (name, t / 1e3, n, shapes, site))
if len(sys.argv) > 1:
    json.dump([{"op": k[0], "shapes": k[1], "site": k[2], "ms": v[0] / 1e3, "calls": v[1]} for k, v in rows], open(sys.argv[1], "w"), indent=1)
