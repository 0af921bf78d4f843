import os, sys, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "torch threads", torch.get_num_threads(), flush=True)
os.system("lscpu | grep -E 'Model name|^CPU\\(s\\)|Thread|Socket|NUMA node\\(s\\)'; cat /sys/fs/cgroup/cpu.max 2>/dev/null")
from interactron_amd.synthetic import procedural_state_dict, synthetic_episodes
from oracle import detector as od, episode as oe, fusion as of
import bench
cfg,_ = bench.model_cfg(300, 50)
t=time.time()
det = {k[len("detector."):]: v for k, v in procedural_state_dict({"detector." + k: v for k, v in od.detr_state_shapes().items()}).items()}
fus = {k[len("fusion."):]: v for k, v in procedural_state_dict({"fusion." + k: v for k, v in of.fusion_state_shapes(cfg, "gpt").items()}).items()}
print("weights", time.time()-t, flush=True)
data = synthetic_episodes(1, tag="bench-r0")
for nt in (int(sys.argv[1]),):
    torch.set_num_threads(nt)
    a=torch.randn(2048,2048); t=time.time(); 
    for _ in range(10): a@a
    print("threads", nt, "matmul 2048 x10", time.time()-t, flush=True)
    random.seed(0); t=time.time()
    oe.interactron_forward(det, fus, data, cfg, {}, "gpt")
    print("threads", nt, "episode", time.time()-t, flush=True)
