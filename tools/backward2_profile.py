"""Contraction launches inside the second-order backward only (per-shape time), via the library's event profiler."""
import collections, csv, ctypes, os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from interactron_amd import Config, build_model, _lib
from interactron_amd.synthetic import load_procedural, synthetic_episodes
from interactron_amd.trainer import FlatOuterStep
lib = _lib.load()
E = 16
cfg, _ = bench.model_cfg(300, 50, E)
model = build_model(Config(**cfg)); load_procedural(model.fusion, "fusion."); model = model.cuda().train()
outer = FlatOuterStep(model)
data = bench.to_gpu(synthetic_episodes(E, tag="bench-r0"), torch.device("cuda"))
random.seed(0)
for _ in range(2):
    model(data); outer.step()
orig = torch.autograd.backward
state = {"n": 0}
def spy(*a, **k):
    state["n"] += 1
    if state["n"] == 1:      # the first backward call of the step = the second-order one
        torch.cuda.synchronize(); lib.ix_gemm_prof_enable(1)
        r = orig(*a, **k)
        torch.cuda.synchronize()
        lib.ix_gemm_prof_dump(b"/tmp/bw2.csv")
        ms, pairs = ctypes.c_double(), ctypes.c_int64(); lib.ix_gemm_prof_read(ctypes.byref(ms), ctypes.byref(pairs)); lib.ix_gemm_prof_enable(0)
        print("second-order backward: %d contraction launches, %.1f ms" % (pairs.value, ms.value))
        return r
    return orig(*a, **k)
torch.autograd.backward = spy
model(data)
rows = list(csv.DictReader(open("/tmp/bw2.csv")))
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for r in rows:
    k = (int(r["M"]), int(r["N"]), int(r["K"]), int(r["batch"]), r["a_kc"], r["b_kc"], r["tile"], r["split"])
    a = agg[k]; a[0] += 1; a[1] += float(r["ms"]); a[2] += 2.0 * k[0] * k[1] * k[2] * k[3]
tot = sum(a[1] for a in agg.values())
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print("%6d %6d %6d %5d %s %s tile %-4s split %-3s cnt %3d us %8.1f TF/s %6.1f ms %6.2f" % (k + (a[0], 1e3 * a[1] / a[0], a[2] / a[1] / 1e9, a[1])))
print("total %.1f ms, %.1f TFLOP/s" % (tot, sum(a[2] for a in agg.values()) / tot / 1e9))
