#!/usr/bin/env python3
"""Benchmark of the Interactron per-episode hot path on MI355X (contract: see the task description / DESIGN.md).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--size 300] [--episodes 2]

One *step* = one outer (meta-train) step of ``configs/interactron.yaml`` on this rank's batch of synthetic 5-frame
episodes: ``model(data)`` (reference models/interactron.py:61-151 -- 3 detector forwards, GPT fusion, learned-loss
gradient with create_graph, clipped-SGD inner step, second-order backward, first-order detector backward) followed
by the trainer's update (reference engine/interactron_trainer.py:93-111 -- here: ONE RCCL all-reduce of the flat
meta-gradient buffer, clip_grad_norm_ and both Adams as fused HIP launches).  Nothing is skipped inside the timed
region; inputs are resident in HBM when it starts.

Multi-GPU: one process per GPU (torchrun env), every rank runs its own ``--episodes`` episodes (weak scaling), the
only collective is the meta-gradient all-reduce.

``python bench.py --gpus N`` works with or without torchrun: without a RANK in the environment the process starts the N
ranks itself (before it touches the GPU) and relays rank 0's line.

Output: ONE JSON line on rank 0 with frames/sec (= 5 * episodes * ranks * steps / seconds), a ``roofline`` object for
the dominant kernel (the MFMA contraction kernel behind every Linear / conv product; the flash attention kernels are
listed beside it), a ``cpu_baseline`` object (the CPU oracle, oracle/episode.py, on this box's host cores: 1 warm-up + 3
timed episodes, median), an ``n800`` object (the north-star 5 x 3x800x800 episodes, same step, measured in the same
process after the headline) and, for N > 1, ``rccl_ranks`` / ``allreduce`` (the meta-gradient all-reduce timed alone).
"""
import argparse
import ctypes
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# /opt/skills/guides/MI355X_MICROARCH.md dense peaks
FP32_MFMA_PEAK_TFLOPS = 157.3    # v_mfma_f32_32x32x2_f32
BF16_MFMA_PEAK_TFLOPS = 2500.0   # v_mfma_f32_32x32x16_bf16
X6_PRODUCTS = 6                  # bf16 MFMAs the bf16x6 kernel issues per fp32 multiply-add (3-way split, 6 kept terms)


CONFIGS = {   # --config -> (reference yaml, MODEL.TYPE, oracle fusion style)
    "interactron": ("configs/interactron.yaml", "interactron", "gpt"),
    "interactron_random": ("configs/interactron_random.yaml", "interactron_random", "decoder"),
    "multi_frame_baseline": ("configs/multi_frame_baseline.yaml", "detr_multiframe", "gpt"),
    "single_frame_baseline": ("configs/single_frame_baseline.yaml", "detr", None),
}


def model_cfg(size, queries, chunk=8, model_type="interactron", step_graph=None, compute_dtype="f32", inner_steps=1):
    # stride-16 backbone: h = w = ceil(size / 16) after the stem/maxpool/strided stages (19 at 300, 50 at 800)
    h = size
    for k, s, p in ((7, 2, 3), (3, 2, 1), (3, 2, 1), (3, 2, 1)):
        h = (h + 2 * p - k) // s + 1
    tokens = h * h
    return dict(TYPE=model_type, WEIGHTS="procedural", NUM_CLASSES=1235, SET_COST_CLASS=1.0, SET_COST_BBOX=5.0,
                SET_COST_GIOU=2.0, NUM_LAYERS=4, NUM_HEADS=8, EMBEDDING_DIM=512, BLOCK_SIZE=5 * (tokens + queries) + 5,
                IMG_FEATURE_SIZE=256, OUTPUT_SIZE=512, BOX_EMB_SIZE=256, EMBEDDING_PDROP=0.1, RESIDUAL_PDROP=0.1,
                ATTENTION_PDROP=0.1, ADAPTIVE_LR=1e-3, NUM_QUERIES=queries, EPISODE_CHUNK=chunk,
                COMPUTE_DTYPE=compute_dtype, INNER_STEPS=inner_steps,
                **({} if step_graph in (None, "auto") else {"STEP_GRAPH": step_graph})), tokens


def to_gpu(data, dev):
    out = dict(data)
    out["frames"], out["masks"] = data["frames"].to(dev), data["masks"].to(dev)
    out["category_ids"] = [[t.to(dev) for t in ep] for ep in data["category_ids"]]
    out["boxes"] = [[t.to(dev) for t in ep] for ep in data["boxes"]]
    return out


def pmc_traffic(kernel, size):
    """HBM bytes per launch of the fp32-on-16-bit contraction kernels (`kernel` and the kernels that share its launches' slot in
    the live counters: both forms of the 12-wave kernel and the weight-planes kernel), launch-weighted, from the committed
    rocprofv3 PMC passes of this same command at this frame size (profiles/*_pmc_hbm_traffic_<size>.json: FETCH_SIZE x2 +
    WRITE_SIZE, separate passes; the 300x300 files of earlier rounds carry no suffix); None when no profile of that shape
    exists.  PMC counters cannot be collected from inside the process, so this is the profile's figure, not a live one."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_hbm_traffic_%d.json" % size)))
    if not files and size == 300:
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_hbm_traffic.json")))
    group = ("gemm_f32_f16x3_p12_kernel", "gemm_f32_f16x3_w256_kernel", "gemm_f32_bf16x6_p12_kernel", "gemm_wp_kernel")
    for f in reversed(files):
        try:
            ks = json.load(open(f))["kernels"]
        except (KeyError, ValueError, OSError):
            continue
        tot = n = 0.0
        for name, v in ks.items():
            if name.startswith(group):
                tot += v["hbm_bytes_per_launch"] * v["launches"]
                n += v["launches"]
        if n:
            return tot / n
    return None


def pmc_traffic_b16():
    """HBM bytes per launch of the bf16 GEMM (gemm16_kernel + its 256-tile form + split-K tails), launch-weighted over the GEMM launches, from
    the committed PMC passes of `bench.py --config multi_frame_baseline --compute-dtype bf16` (profiles/*_pmc_hbm_traffic_mfb_bf16.json)"""
    import glob
    for f in reversed(sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_hbm_traffic_mfb_bf16.json")))):
        try:
            ks = json.load(open(f))["kernels"]
        except (KeyError, ValueError, OSError):
            continue
        tot = sum(v["hbm_bytes_per_launch"] * v["launches"] for k, v in ks.items() if k.startswith("gemm16_"))
        n = sum(v["launches"] for k, v in ks.items() if k.startswith("gemm16_") and "reduce" not in k)
        if n:
            return tot / n
    return None


def usable_cores():
    """Host cores this process may actually use: min(affinity, cgroup CPU quota).  The GPU boxes expose 256 logical
    CPUs behind a 16-CPU cgroup quota; running 256 threads against that quota throttles the oracle ~100x."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return n


def cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def cpu_baseline(cfg, size, config="interactron", timed=3):
    """The CPU oracle (a from-scratch PyTorch-CPU restatement of the reference path, pinned to fixtures captured from
    the imported reference) on synthetic episodes of the same workload, all usable host cores: one warm-up episode, then
    `timed` timed ones (one episode = one meta-train pass of the same model, 5 frames); the median is reported."""
    import torch
    from interactron_amd.synthetic import procedural_state_dict, synthetic_episodes
    from oracle import detector as od, episode as oe, fusion as of
    cores = usable_cores()
    torch.set_num_threads(cores)
    det = {k[len("detector."):]: v for k, v in
           procedural_state_dict({"detector." + k: v for k, v in od.detr_state_shapes().items()}).items()}
    style = CONFIGS[config][2]
    fus = None
    if style is not None:
        fus = {k[len("fusion."):]: v for k, v in
               procedural_state_dict({"fusion." + k: v for k, v in of.fusion_state_shapes(cfg, style).items()}).items()}
        if style == "decoder" and "pos_embed" not in fus:
            fus["pos_embed"] = of.decoder_fusion_pos_embed()
    random.seed(0)
    times = []
    for i in range(1 + timed):
        data = synthetic_episodes(1, height=size, width=size, tag="bench-cpu%d" % i)
        t0 = time.perf_counter()
        if config == "single_frame_baseline":
            oe.detr_train_forward(det, data)
            fn = "detr_train_forward"
        elif config == "multi_frame_baseline":
            oe.multiframe_forward(det, fus, data, cfg)
            fn = "multiframe_forward"
        else:
            oe.interactron_forward(det, fus, data, cfg, {}, style)
            fn = "interactron_forward"
        times.append(time.perf_counter() - t0)
    timed_s = sorted(times[1:])
    med = timed_s[len(timed_s) // 2]
    return {"value": 5.0 / med, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": "%d training episodes (5 frames, %dx%d, fp32) through oracle/episode.py:%s after 1 warm-up episode, "
                      "median %.2f s (each: %s) on %d threads of %s"
                      % (timed, size, size, fn, med, ", ".join("%.2f" % t for t in times[1:]), cores, cpu_model_name())}


def cpu_detector_800(timed=2):
    """the CPU figure next to the 800x800 number is the detector forward alone (oracle/detector.py on one
    3x800x800 frame, all usable host cores) -- the fusion at T = 12 755 materialises 8 x T^2 fp32 attention and is not
    practical on the host."""
    import torch
    from interactron_amd.synthetic import procedural_state_dict, synthetic_episodes
    from oracle import detector as od
    cores = usable_cores()
    torch.set_num_threads(cores)
    det = {k[len("detector."):]: v for k, v in
           procedural_state_dict({"detector." + k: v for k, v in od.detr_state_shapes().items()}).items()}
    data = synthetic_episodes(1, frames=1, height=800, width=800, tag="bench-cpu800")
    frames, masks = data["frames"][0], data["masks"][0]
    times = []
    with torch.no_grad():
        for _ in range(1 + timed):
            t0 = time.perf_counter()
            od.detr_forward(det, frames, masks)
            times.append(time.perf_counter() - t0)
    med = sorted(times[1:])[len(times[1:]) // 2]
    return {"value": 1.0 / med, "unit": "frames/s (detector forward only, 1 frame of 3x800x800)", "cores": cores, "kind": "port",
            "sample": "%d detector forwards through oracle/detector.py:detr_forward after 1 warm-up, median %.2f s on %d threads of %s"
                      % (timed, med, cores, cpu_model_name())}


def two_frame_pair(dev, frames=2, size=800, gpu_steps=5):
    """The like-for-like CPU pair beside the north-star shape, MEASURED inside this run: ONE meta-train step of `interactron` on a
    TWO-frame 800 x 800 episode (fusion T = 2 x (2500 + 50) + 5 = 5 105) through the HIP path (median of `gpu_steps` after a
    warm-up) and through the CPU oracle (oracle/episode.py:interactron_forward, all usable host cores, one episode ~ 35 s), same
    inputs, same weights, losses compared.  The five-frame episode (T = 12 755) is not practical on the host: reference
    models/gpt.py:48-52 materialises 8 x T^2 fp32 attention per layer (5.2 GB at T = 12 755 with a double backward over it;
    834 MB at T = 5 105).  The first-order branch draws its frame from the two that exist (the reference draws
    random.randint(0, 4) for its five: models/interactron.py:126)."""
    import torch
    from interactron_amd import Config, build_model
    from interactron_amd.synthetic import load_procedural, procedural_state_dict, synthetic_episodes
    from oracle import detector as od, episode as oe, fusion as of
    s = frames
    cfg, tokens = model_cfg(size, 50, 1, "interactron")
    cfg["BLOCK_SIZE"] = s * (tokens + 50) + 5
    data = synthetic_episodes(1, frames=s, height=size, width=size, tag="two-frame")
    # the policy head always scores four moves (reference models/interactron.py:116-118): four scripted actions, as in a 5-frame episode
    data["actions"] = synthetic_episodes(1, frames=5, height=16, width=16, tag="two-frame")["actions"]
    real_randint = random.randint
    random.randint = lambda a, b: real_randint(a, min(b, s - 1))   # the first-order branch's frame: one of the s that exist
    try:
        model = build_model(Config(**cfg))
        load_procedural(model.fusion, "fusion.")
        model = model.to(dev).eval()
        gpu = to_gpu(data, dev)
        times = []
        for i in range(1 + gpu_steps):
            for p_ in model.parameters():
                p_.grad = None
            random.seed(11)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            _, losses = model(gpu)
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
        hip_s = sorted(times[1:])[len(times[1:]) // 2]
        hip_losses = {k: float(v) for k, v in losses.items()}
        del model, gpu
        torch.cuda.empty_cache()
        cores = usable_cores()
        torch.set_num_threads(cores)
        det = {k[len("detector."):]: v for k, v in
               procedural_state_dict({"detector." + k: v for k, v in od.detr_state_shapes().items()}).items()}
        fus = {k[len("fusion."):]: v for k, v in
               procedural_state_dict({"fusion." + k: v for k, v in of.fusion_state_shapes(cfg, "gpt").items()}).items()}
        random.seed(11)
        t0 = time.perf_counter()
        _, ref_losses, _ = oe.interactron_forward(det, fus, data, cfg, {}, "gpt")
        cpu_s = time.perf_counter() - t0
    finally:
        random.randint = real_randint
    worst = max(abs(hip_losses[k] - float(v)) / max(abs(float(v)), 1.0) for k, v in ref_losses.items())
    return {"workload": "one interactron meta-train episode of %d frames x 3x%dx%d, Q=50, fusion T=%d, eval mode (dropout off), procedural weights"
                        % (s, size, size, cfg["BLOCK_SIZE"]),
            "hip": {"seconds_per_episode": hip_s, "frames_per_s": s / hip_s, "sample": "median of %d steps after 1 warm-up" % gpu_steps},
            "cpu_baseline": {"seconds_per_episode": cpu_s, "value": s / cpu_s, "unit": "frames/s", "cores": cores, "kind": "port",
                             "sample": "1 episode through oracle/episode.py:interactron_forward on %d threads of %s (no warm-up)"
                                       % (cores, cpu_model_name())},
            "gpu_over_cpu": cpu_s / hip_s, "worst_relative_loss_difference": worst, "comparable": True,
            "source": "measured inside this bench.py run"}


def launch_ranks(args):
    """``python bench.py --gpus N`` without a torchrun environment: start the N ranks as child processes (one per GPU,
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment) BEFORE anything in this process touches the GPU,
    relay rank 0's JSON line and fail if any rank fails."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    codes = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    for ln in out.decode().splitlines():   # exactly the JSON line (libraries print to stdout too: "[Gloo] Rank 0 is connected ...")
        if ln.startswith('{"metric"'):
            sys.stdout.write(ln + "\n")
        else:
            sys.stderr.write(ln + "\n")
    sys.stdout.flush()
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    if bad:
        sys.stderr.write("bench.py: ranks failed (rank, exit code): %s\n" % bad)
        sys.exit(1)


def run_workload(args, size, episodes, chunk, steps, warmup, ctx, want_roofline, tag, inner_steps=None):
    """Build the model of `--config` at frame size `size`, run `warmup` + `steps` steps of `--mode`, return the metrics
    of the timed steps (max over ranks) and, from one extra profiled step, the per-kernel roofline numbers."""
    import torch
    import torch.distributed as dist
    from interactron_amd import Config, build_model
    from interactron_amd.synthetic import load_procedural, synthetic_episodes
    from interactron_amd.trainer import FlatOuterStep
    lib, dev, rank, world, local = ctx["lib"], ctx["dev"], ctx["rank"], ctx["world"], ctx["local"]

    torch.cuda.reset_peak_memory_stats()   # `peak_memory_GB` is this workload's
    cfg, tokens = model_cfg(size, args.queries, chunk, CONFIGS[args.config][1], args.step_graph, args.compute_dtype,
                            inner_steps or args.inner_steps)
    model = build_model(Config(**cfg))
    if hasattr(model, "fusion"):
        load_procedural(model.fusion, "fusion.")
    model = model.to(dev).train()
    if hasattr(model, "detector"):
        outer = FlatOuterStep(model, detector_lr=1e-5, fusion_lr=1e-4, max_norm=1.0)
    else:   # configs/single_frame_baseline.yaml: one Adam over the whole DETR (direct-supervision trainer)
        outer = FlatOuterStep(model, max_norm=1.0, groups=[list(model.parameters())], lrs=[1e-5])
    data = to_gpu(synthetic_episodes(episodes, height=size, width=size, tag="%s-r%d" % (tag, rank)), dev)
    random.seed(1234 + rank)
    last = {}

    def episode(i, frames=None):
        return {"frames": data["frames"][i:i + 1, :frames], "masks": data["masks"][i:i + 1, :frames],
                "category_ids": [data["category_ids"][i][:frames]], "boxes": [data["boxes"][i][:frames]],
                "actions": data["actions"][i:i + 1], "initial_image_path": [data["initial_image_path"][i]]}

    if args.mode != "train":   # evaluation modes: one episode at a time, as the reference evaluators call them
        assert args.config in ("interactron", "interactron_random") or args.mode == "predict"
        model.eval()

    def step():
        if args.mode == "train":
            _, losses = model(data)
            last["losses"] = losses
            outer.step()
            return
        for i in range(episodes if args.mode != "predict-batched" else 0):
            if args.mode == "interactive":   # the policy looks at 1..4 frames, then the adapted prediction on all 5
                for s in range(1, 5):
                    model.get_next_action(episode(i, s))
            out = model.predict(episode(i))
            last["losses"] = {"pred_logits": out["pred_logits"]}
        if args.mode == "predict-batched":
            last["losses"] = {"pred_logits": model.predict(data)["pred_logits"]}

    def fence():
        if world > 1:
            if dist.get_backend() == "nccl":
                dist.barrier(device_ids=[local])
            else:
                dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        step()
    fence()
    lib.ix_gemm_stats(None, None, 1)
    lib.ix_flash_stats(None, None, 1)
    t0 = time.perf_counter()
    issue = 0.0
    for _ in range(steps):
        ti = time.perf_counter()
        step()
        issue += time.perf_counter() - ti   # host time until the step's launches are queued (no synchronisation added)
    fence()
    dt = time.perf_counter() - t0
    flops, launches, fflops, flaunches = ctypes.c_double(), ctypes.c_int64(), ctypes.c_double(), ctypes.c_int64()
    lib.ix_gemm_stats(ctypes.byref(flops), ctypes.byref(launches), 1)
    lib.ix_flash_stats(ctypes.byref(fflops), ctypes.byref(flaunches), 1)
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    res = {"peak_memory_GB": torch.cuda.max_memory_allocated() / 1e9, "seconds": dt, "steps": steps, "episodes": episodes, "frames_per_s": 5.0 * episodes * world * steps / dt,
           "ms_per_step": dt * 1e3 / steps, "host_issue_ms_per_step": issue * 1e3 / steps, "block_size": cfg["BLOCK_SIZE"], "cfg": cfg,
           "step_graphs": sorted(type(v).__name__ for v in getattr(model, "__dict__", {}).get("_chunk_graphs", {}).values()),
           "gemm_gflop_per_step": flops.value / 1e9 / steps, "gemm_launches_per_step": launches.value / steps,
           "attention_gflop_per_step": fflops.value / 1e9 / steps, "attention_launches_per_step": flaunches.value / steps,
           "allreduce": None, "roofline": None}

    if world > 1 and args.mode == "train":
        # the step's single data-path collective, timed by itself: 5 all-reduces of the flat gradient buffer, fenced
        fence()
        t0 = time.perf_counter()
        for _ in range(5):
            outer.flat.all_reduce_grads()
        fence()
        ar = (time.perf_counter() - t0) / 5
        nbytes = outer.flat.grads.numel() * 4
        outer.flat.grads.zero_()
        res["allreduce"] = {"backend": "rccl" if dist.get_backend() == "nccl" else dist.get_backend(), "ranks": world,
                            "bytes": nbytes, "ms": ar * 1e3,
                            "bus_GBps": 2.0 * (world - 1) / world * nbytes / ar / 1e9}

    if want_roofline:
        # HIP-event pair around every launch of the contraction and attention kernels on their own stream, one extra step
        # of the same workload (kept out of the timed region so the events do not perturb `value`).
        # (issued launch by launch: a replayed graph passes no launch through the library's event brackets)
        had = getattr(model.config, "STEP_GRAPH", None) if hasattr(model, "config") else None
        if hasattr(model, "config"):
            model.config.STEP_GRAPH = "off"
        lib.ix_gemm_prof_enable(1)
        step()
        torch.cuda.synchronize()
        if hasattr(model, "config"):
            if had is None:
                del model.config.STEP_GRAPH
            else:
                model.config.STEP_GRAPH = had
        if args.gemm_csv and rank == 0:
            lib.ix_gemm_prof_dump(args.gemm_csv.encode())
        cms, cfl, cmf, cn = (ctypes.c_double * 3)(), (ctypes.c_double * 3)(), (ctypes.c_double * 3)(), (ctypes.c_int64 * 3)()
        lib.ix_prof_contractions(cms, cfl, cmf, cn)
        cby = (ctypes.c_double * 3)()
        lib.ix_prof_contraction_bytes(cby)
        fms, ffl, fmf, fn = (ctypes.c_double * 7)(), (ctypes.c_double * 7)(), (ctypes.c_double * 7)(), (ctypes.c_int64 * 7)()
        lib.ix_prof_flash(fms, ffl, fmf, fn)
        bms, bfl, bby, bn = ctypes.c_double(), ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
        lib.ix_prof_b16(ctypes.byref(bms), ctypes.byref(bfl), ctypes.byref(bby), ctypes.byref(bn))
        ms, pairs = ctypes.c_double(), ctypes.c_int64()
        lib.ix_gemm_prof_read(ctypes.byref(ms), ctypes.byref(pairs))
        lib.ix_gemm_prof_enable(0)
        res["roofline"] = build_roofline(list(cms), list(cfl), list(cmf), list(cn), list(fms), list(ffl), list(fmf), list(fn), size)
        if bn.value:
            # the 16-bit activation mode: the bf16 GEMM (csrc/gemm16.hip) is the dominant kernel -- ONE matrix instruction per multiply-add,
            # so executed = algorithmic FLOPs; priced against the same dense 16-bit MFMA peak
            r = res["roofline"]
            b16 = {"kernel": "gemm16_kernel (bf16 operands by LDS-DMA, v_mfma_f32_16x16x32_bf16, fp32 accumulation)", "bound": "mfma",
                   "achieved": bfl.value / (bms.value * 1e-3) / 1e12, "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                   "frac": bfl.value / (bms.value * 1e-3) / 1e12 / BF16_MFMA_PEAK_TFLOPS, "launches_per_step": int(bn.value),
                   "kernel_ms_per_step": bms.value, "gflop_per_step": bfl.value / 1e9, "avg_launch_us": bms.value * 1e3 / bn.value,
                   "algorithmic_bytes_per_launch": bby.value / bn.value, "algorithmic_GBps": bby.value / (bms.value * 1e-3) / 1e9,
                   "traffic": pmc_traffic_b16() if args.config == "multi_frame_baseline" else None}
            r["bf16_gemm"] = b16
            if bms.value > r["kernel_ms_per_step"]:   # it IS the dominant kernel of this run: the top-level fields describe it
                r["fp32_on_16bit_kernels"] = {k: r[k] for k in ("kernel", "achieved", "frac", "launches_per_step", "gflop_per_step",
                                                                "kernel_ms_per_step", "algorithmic_tflops", "avg_launch_us")}
                for k in ("kernel", "achieved", "frac", "launches_per_step", "gflop_per_step", "kernel_ms_per_step", "avg_launch_us", "traffic"):
                    r[k] = b16[k]
                r["algorithmic_tflops"] = b16["achieved"]
                r["executed_gflop_per_step"] = b16["gflop_per_step"]
                r["note"] = ("achieved = 2MNK FLOP/s of the bf16 GEMM over its launches (one matrix instruction per multiply-add: executed = "
                             "algorithmic) against the dense 16-bit MFMA peak; the fp32-on-16-bit kernels that still run (adapted ops, "
                             "convolution gathers) are listed under fp32_on_16bit_kernels")
        en = cn[1] + cn[2]
        res["roofline"]["algorithmic_bytes_per_launch"] = (cby[1] + cby[2]) / max(1, en)
        t = res["roofline"]["traffic"]
        res["roofline"]["traffic_over_algorithmic"] = (t / res["roofline"]["algorithmic_bytes_per_launch"]) if t and en else None
        # what the matrix pipe of THIS box sustains (ix_diag_mfma_rate_f16: back-to-back v_mfma_f32_32x32x16_f16 on every SIMD,
        # measured live): the data-sheet `peak` assumes 2.4 GHz, an all-CU matrix load runs at ~ 1.8.  `frac` stays against `peak`.
        sus, scratch = ctypes.c_double(0.0), torch.zeros(4, device=dev)
        if lib.ix_diag_mfma_rate_f16(ctypes.byref(sus), scratch.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0 and sus.value > 0:
            res["roofline"]["sustained_mfma_tflops_measured"] = sus.value
            res["roofline"]["frac_of_sustained"] = res["roofline"]["achieved"] / sus.value
        if res["gemm_launches_per_step"] == 0:
            # the timed steps were replayed from HIP graphs, which pass no launch through the library's counters: the per-step
            # work figures then come from the profiled step (the same launches, issued one by one)
            r = res["roofline"]
            res["gemm_gflop_per_step"] = r["gflop_per_step"] + (cfl[0] / 1e9)
            res["gemm_launches_per_step"] = float(r["launches_per_step"] + cn[0])
            res["attention_gflop_per_step"] = r["attention_kernels"]["gflop_per_step"]
            res["attention_launches_per_step"] = float(r["attention_kernels"]["launches_per_step"])
            res["work_counted_in"] = "the profiled eager step (the timed steps replay captured HIP graphs)"

    bad = [k for k, v in last["losses"].items() if not bool(torch.isfinite(v).all())]
    assert not bad, "non-finite losses after the timed steps: %s" % bad
    del model, outer, data
    torch.cuda.empty_cache()
    return res


def hbm_kernels(torch, hipops, dev, rows=32960, seq=(16, 2060)):
    """The HBM-bound kernels of the path, timed alone on step-sized buffers (HIP events on the current stream, 5 launches
    after a warm-up): achieved GB/s = ALGORITHMIC bytes per launch / average launch time, against the 8 TB/s HBM3E peak
    (/opt/skills/guides/MI355X_MICROARCH.md; a float4 copy reaches 6.3).  Sizes: the 58.4 M-element flat parameter buffer
    (Adam, clip norm), theta's 38.0 M elements (clipped SGD), and at THIS run's shape the fusion activation [episodes x T, 512]
    (LayerNorm) and projection [episodes, T, 512] (operand split): [32960, 512] at the headline, [51005, 512] for the stress
    configuration (--size 1600 --queries 200 --episodes 1)."""
    import math
    def timed(fn, it=5):
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(it):
            fn()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / it * 1e-3
    out = {}
    n = 58_400_000
    pbuf, g, m, v = (torch.randn(n, device=dev) * 1e-3 for _ in range(4))
    v.abs_()
    sq = torch.zeros((), device=dev)
    t = timed(lambda: hipops.adam_step(pbuf, g, m, v, 1e-5, 0.9, 0.999, 1e-8, 1, sq, 1.0, zero_grad=False))
    out["adam_kernel"] = {"bytes": 28 * n, "GBps": 28 * n / t / 1e9}
    t = timed(lambda: hipops.sumsq_accum(g, sq))
    out["sumsq_accum_kernel"] = {"bytes": 4 * n, "GBps": 4 * n / t / 1e9}
    n2 = 38_030_808
    ps, gs = [torch.randn(n2, device=dev)], [torch.randn(n2, device=dev) * 10]
    t = timed(lambda: hipops.ClippedSGD.forward(hipops._NullCtx(), 1e-3, 0.01, 1, *(ps + gs)))
    out["multi_map2_kernel<sgd_clip>"] = {"bytes": 12 * n2, "GBps": 12 * n2 / t / 1e9}
    x = torch.randn(rows, 512, device=dev)   # the fusion activation of this run: (episodes per chunk x T) rows of 512
    w, b = torch.ones(512, device=dev), torch.zeros(512, device=dev)
    t = timed(lambda: hipops.LayerNorm.forward(hipops._NullCtx(), x, w, b, 1e-5))
    out["ln_fwd_kernel"] = {"bytes": 8 * x.numel(), "GBps": 8 * x.numel() / t / 1e9}
    y = torch.randn(seq[0], seq[1], 512, device=dev)
    t = timed(lambda: hipops.attn_split(y, seq[0], seq[1], 512, 0, 8, 64))
    # 4 B read + 4 B of fp16 row planes written per element; + tr planes (4 B fp16 / 6 B bf16) unless the head-dim-64 passes read
    # the row planes both ways (csrc/flash16.hip: no tr planes at all)
    per = 8 + (0 if hipops._rows_only(64, hipops._TR_FORMS[hipops.FLASH_TR]) else 4 if hipops.FLASH_TR == "f16" else 6)
    out["attn_split_kernel"] = {"bytes": per * y.numel(), "GBps": per * y.numel() / t / 1e9, "bytes_per_element": per}
    for k in out:
        out[k]["frac_of_8TBps"] = out[k]["GBps"] / 8000.0
    return out


# (head dim 64 in the fp16 form runs the 16x16x32 twins of csrc/flash16.hip -- flash16_fwd / bwd_q / bwd_kv / bb_stats / bb_q / bb_kv --
#  under the same slots; head dim 32 the 32x32x16 kernels of csrc/flash.hip)
FLASH_KERNELS = ["all", "flash_fwd_kernel", "flash_bwd_q_kernel", "flash_bwd_kv_kernel", "flash_bb_q_kernel<stats>",
                 "flash_bb_q_kernel", "flash_bb_kv_kernel"]


def build_roofline(cms, cfl, cmf, cn, fms, ffl, fmf, fn, size):
    """`roofline` object for the dominant kernel = the 12-wave contraction kernel that evaluates fp32 products on the 16-bit
    matrix cores (fp16x3 form: two fp16 planes + a sub-block exponent, 3 MFMAs per fp32 multiply-add -- the default; bf16x6
    form: three bf16 planes, 6 MFMAs -- narrow tiles and IX_GEMM_KERNEL=x6).  `achieved` = matrix-instruction FLOP/s it
    EXECUTES over its launches (HIP events per launch), against the dense 16-bit MFMA peak; the algorithmic
    (fp32-equivalent) rate is listed beside it.  The exact-fp32 MFMA kernel and the flash attention kernels follow."""
    def rate(fl, ms):
        return fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    forms = {}
    for i, name in ((2, "gemm_f32_f16x3_p12_kernel"), (1, "gemm_f32_bf16x6_p12_kernel")):
        if cn[i]:
            forms[name] = {"launches_per_step": int(cn[i]), "kernel_ms_per_step": cms[i], "gflop_per_step": cfl[i] / 1e9,
                           "algorithmic_tflops": rate(cfl[i], cms[i]), "executed_mfma_tflops": rate(cmf[i], cms[i]),
                           "mfma_frac": rate(cmf[i], cms[i]) / BF16_MFMA_PEAK_TFLOPS}
    if "gemm_f32_f16x3_p12_kernel" in forms:
        forms["gemm_f32_f16x3_p12_kernel"]["includes"] = ("the 256 x 128 tiles of the same form (gemm_f32_f16x3_w256_kernel, bit-identical results) "
                                                         "where the cost model takes them, and the weight-planes launches (gemm_wp_kernel)")
    ems, efl, emf, en = cms[1] + cms[2], cfl[1] + cfl[2], cmf[1] + cmf[2], cn[1] + cn[2]
    dominant = "gemm_f32_f16x3_p12_kernel" if cms[2] >= cms[1] else "gemm_f32_bf16x6_p12_kernel"
    attn = {"kernel_ms_per_step": fms[0], "launches_per_step": int(fn[0]), "gflop_per_step": ffl[0] / 1e9,
            "achieved": rate(ffl[0], fms[0]),
            "unit": "TFLOP/s (fp32-equivalent, algorithmic: the reference graph's products -- 2 forward, 4 backward, 10 double backward)",
            "executed_mfma_tflops": rate(fmf[0], fms[0]), "mfma_peak": BF16_MFMA_PEAK_TFLOPS,
            "mfma_frac": rate(fmf[0], fms[0]) / BF16_MFMA_PEAK_TFLOPS,
            "kernels": {FLASH_KERNELS[i]: {"ms_per_step": fms[i], "launches": int(fn[i]), "achieved": rate(ffl[i], fms[i]),
                                           "mfma_frac": rate(fmf[i], fms[i]) / BF16_MFMA_PEAK_TFLOPS}
                        for i in range(1, 7) if fms[i] > 0}}
    total_ms = cms[0] + ems + fms[0]
    total_fl = cfl[0] + efl + ffl[0]
    return {"bound": "mfma", "kernel": dominant + (" (+ the other form of the same kernel, see forms)" if len(forms) > 1 else ""),
            "achieved": rate(emf, ems), "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": rate(emf, ems) / BF16_MFMA_PEAK_TFLOPS,
            "traffic": pmc_traffic(dominant, size), "launches_per_step": int(en), "gflop_per_step": efl / 1e9,
            "executed_gflop_per_step": emf / 1e9, "algorithmic_tflops": rate(efl, ems),
            "avg_launch_us": ems * 1e3 / max(1, en), "kernel_ms_per_step": ems, "forms": forms,
            "note": "achieved = matrix-instruction FLOP/s executed by the fp32-on-16-bit contraction kernel (3 per fp32 "
                    "multiply-add in the fp16x3 form, 6 in the bf16x6 form) against the dense 16-bit MFMA peak; "
                    "algorithmic_tflops = fp32-equivalent 2MNK rate.  The peak assumes 2.4 GHz: back to back this kernel holds "
                    "the package at its 1400 W cap and the shader clock at 1.4-1.8 GHz (profiles/README.md, power probe)",
            "fp32_mfma_kernel": {"achieved": rate(cfl[0], cms[0]), "peak": FP32_MFMA_PEAK_TFLOPS,
                                 "frac": rate(cfl[0], cms[0]) / FP32_MFMA_PEAK_TFLOPS,
                                 "launches_per_step": int(cn[0]), "kernel_ms_per_step": cms[0]},
            "attention_kernels": attn,
            "all_mfma_kernels": {"achieved": rate(total_fl, total_ms), "gflop_per_step": total_fl / 1e9, "kernel_ms_per_step": total_ms,
                                 "unit": "TFLOP/s (algorithmic, fp32-equivalent)"}}


JSON_FD = 1
N800_STEPS = 10        # timed steps of the north-star sub-measurement (after 2 warm-up steps)
ALLREDUCE_MS_ASSUMED = 2.0   # 234 MB flat gradient over xGMI at ~200 GB/s bus bandwidth (used only where no RCCL timing exists)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", type=int, default=300, help="frame height = width (300 = reference data, 800 = north-star)")
    ap.add_argument("--queries", type=int, default=50)
    ap.add_argument("--episodes", type=int, default=16, help="episodes per GPU per step (reference configs/interactron.yaml BATCH_SIZE: 16)")
    ap.add_argument("--config", default="interactron", choices=sorted(CONFIGS),
                    help="which reference config's training step to run (default: the headline meta-train step)")
    ap.add_argument("--mode", default="train", choices=["train", "predict", "interactive", "predict-batched"],
                    help="train (default, the headline): meta-train step; predict: model.predict per episode (eval adapt, "
                         "reference interactron.py:31-59); interactive: predict + 4 x get_next_action per episode (SURVEY 8d iii)")
    ap.add_argument("--chunk", type=int, default=16, help="EPISODE_CHUNK: episodes run together as one batched pass (0 = sequential)")
    ap.add_argument("--n800-episodes", type=int, default=8,
                    help="episodes per GPU per step of the north-star sub-measurement (5 x 3x800x800 frames each; 0 = skip it)")
    ap.add_argument("--attention-dtype", default="fp32", choices=["fp32", "fp8"],
                    help="fp8: the forward attention products on OCP e4m3 MFMA (BASELINE.json configs[4], the 1600 / 200-query "
                         "stress configuration: --size 1600 --queries 200 --attention-dtype fp8); fp32 = the parity path")
    ap.add_argument("--global-batch", type=int, default=0,
                    help="strong scaling: ONE batch of this many episodes per step, rank r runs episodes r::N of it (what train.py + "
                         "shard_batch do with the reference config, BATCH_SIZE 16); 0 = weak scaling, --episodes per GPU")
    ap.add_argument("--small-e", type=int, default=2,
                    help="episodes per GPU of the `small_e` sub-measurement at N = 1 (the per-GPU share of the reference's global "
                         "batch on 8 GPUs; replayed from HIP graphs); 0 = skip it")
    ap.add_argument("--step-graph", default="auto", choices=["auto", "on", "off"],
                    help="MODEL.STEP_GRAPH: replay the chunk's launch sequence from captured HIP graphs (auto: whenever the capture fits in free device memory)")
    ap.add_argument("--compute-dtype", default="f32", choices=["f32", "bf16", "bf16_fusion", "single_pass"],
                    help="MODEL.COMPUTE_DTYPE: f32 = fp32-grade contractions (the parity path, every headline); bf16 = the 16-bit ACTIVATION "
                         "mode (activations stored as bf16, bf16 GEMM with LDS-DMA operands; BASELINE.json configs[1]: --config "
                         "multi_frame_baseline --compute-dtype bf16); single_pass = fp32 storage, single-pass 16-bit contractions (round 4) "
                         "-- their own lines, `dtype` says so, never the headline")
    ap.add_argument("--inner-steps", type=int, default=1,
                    help="MODEL.INNER_STEPS: learned-loss SGD steps per episode (reference: 1; BASELINE.json's '5-step adapt loop' = 5, "
                         "a stress setting -- the default run reports it as the `inner5` sub-line)")
    ap.add_argument("--inner5-episodes", type=int, default=4, help="episodes per GPU of the `inner5` sub-measurement (0 = skip it)")
    ap.add_argument("--stress-steps", type=int, default=2,
                    help="timed steps of the `stress` sub-measurement (BASELINE.json configs[4]: 1600 long edge, 200 queries, fp8 MFMA "
                         "attention forward; one episode per step, after two warm-up steps; 0 = skip it)")
    ap.add_argument("--bf16-steps", type=int, default=10,
                    help="timed steps of the `bf16_mode` sub-measurement (multi_frame_baseline in fp32-grade and in the 16-bit activation mode; 0 = skip it)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-two-frame", action="store_true", help="skip the two-frame 800 x 800 CPU / HIP pair of the north-star object (~ 40 s)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--gemm-csv", default=None, help="write one line per contraction launch of the profiled step (tuning aid)")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:   # plain `python bench.py --gpus N`: become the launcher (no GPU call yet)
        return launch_ranks(args)

    # stdout carries exactly ONE line, the JSON: everything libraries write to file descriptor 1 from here on (gloo prints
    # "[Gloo] Rank 0 is connected ..." there) goes to stderr, and the line is written to the saved descriptor at the end
    global JSON_FD
    sys.stdout.flush()
    JSON_FD = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    from interactron_amd import _lib
    from interactron_amd.trainer import init_distributed

    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
    # bind this rank's GPU BEFORE the process group exists: RCCL communicators are created on the current device
    local = int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count()   # (gloo smoke runs: several ranks per GPU)
    torch.cuda.set_device(local)
    rank, _, world = init_distributed()
    assert world == args.gpus, "--gpus %d but WORLD_SIZE=%d (torchrun --nproc-per-node must match)" % (args.gpus, world)
    dev = torch.device("cuda", local)
    lib = _lib.load()
    ctx = {"lib": lib, "dev": dev, "rank": rank, "world": world, "local": local}
    from interactron_amd import hipops as _ops
    _ops.ATTENTION_DTYPE = args.attention_dtype

    strong = args.global_batch > 0
    if strong:   # this rank's share of ONE global batch (rank r: episodes r::N), the rest of the line as usual
        assert args.global_batch % world == 0, "--global-batch must be a multiple of --gpus"
        args.episodes = args.global_batch // world
        args.chunk = min(args.chunk, args.episodes)
    head = run_workload(args, args.size, args.episodes, args.chunk, args.steps, args.warmup, ctx, not args.no_roofline, "bench")
    headline_cfg = (args.size == 300 and args.mode == "train" and args.config == "interactron" and not strong
                    and args.compute_dtype == "f32" and args.inner_steps == 1)
    # The per-GPU share of the reference's global batch of 16 on 8 GPUs (engine/interactron_trainer.py:78-84 + SURVEY 8e): what
    # a rank of the strong-scaling run executes per step.  At N = 1 as `small_e`; at N > 1 the same global batch as `strong`.
    small = strong_run = None
    if headline_cfg and world == 1 and args.small_e > 0:
        small = run_workload(args, 300, args.small_e, args.small_e, 10, 3, ctx, False, "bench")
    if headline_cfg and world > 1 and 16 % world == 0:
        strong_run = run_workload(args, 300, 16 // world, 16 // world, 10, 3, ctx, False, "bench")
    # BASELINE.json's "5-step adapt loop": the same step with MODEL.INNER_STEPS = 5 (five learned-loss gradients with the
    # second-order graph through all of them; the fast weights never leave the device) -- a stress setting, never the headline
    inner5 = None
    if headline_cfg and world == 1 and args.inner5_episodes > 0:
        inner5 = run_workload(args, 300, args.inner5_episodes, args.inner5_episodes, 3, 2, ctx, False, "bench", inner_steps=5)   # (2 warm-up steps: eager, then the capture)
    # BASELINE.json configs[1] "multi_frame_baseline ... 1 x MI355X bf16": the 16-bit ACTIVATION mode (b16.py) on that configuration's
    # training step, next to the same step in fp32-grade arithmetic -- its own sub-line with its own dtype, never the headline
    b16_line = None
    if headline_cfg and world == 1 and args.bf16_steps > 0:
        import copy
        sub = copy.copy(args)
        sub.config, sub.mode = "multi_frame_baseline", "train"
        pair = {}
        for dt in ("f32", "bf16"):
            sub.compute_dtype = dt
            r = run_workload(sub, 300, 16, 16, args.bf16_steps, 3, ctx, dt == "bf16" and not args.no_roofline, "bench-mfb")
            pair[dt] = {"value": r["frames_per_s"], "unit": "frames/s", "ms_per_step": r["ms_per_step"], "steps": args.bf16_steps, "warmup": 3,
                        "peak_memory_GB": r["peak_memory_GB"]}
            if dt == "bf16" and r["roofline"]:
                pair[dt]["roofline"] = {k: r["roofline"].get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic",
                                                                           "launches_per_step", "kernel_ms_per_step", "gflop_per_step",
                                                                           "avg_launch_us", "bf16_gemm", "fp32_on_16bit_kernels")}
                pair[dt]["attention_ms_per_step"] = r["roofline"]["attention_kernels"]["kernel_ms_per_step"]
        # ... and the headline configuration itself in the mode (configs/interactron.yaml: second-order backward through bf16 activations,
        # fp32 fast weights), at the reference's 300 x 300 shape, the north-star 800 x 800 shape and the 2-episode share of an 8-GPU step
        sub.config = "interactron"
        inter = {}
        for name, (size, eps, steps, warm) in (("p300_e16", (300, 16, 10, 3)), ("small_e", (300, 2, 10, 3)), ("n800_e8", (800, 8, 5, 2))):
            if (name == "n800_e8" and args.n800_episodes <= 0) or (name == "small_e" and args.small_e <= 0):
                continue
            try:
                r = run_workload(sub, size, eps, eps, steps, warm, ctx, False, "bench-b16-" + name)
                inter[name] = {"workload": "%d episodes/GPU x 5 frames x 3x%dx%d, same meta-train step" % (eps, size, size), "value": r["frames_per_s"],
                               "unit": "frames/s", "ms_per_step": r["ms_per_step"], "steps": steps, "warmup": warm,
                               "peak_memory_GB": r["peak_memory_GB"], "step_graphs": r["step_graphs"]}
            except torch.cuda.OutOfMemoryError as e:
                inter[name] = {"error": "out of memory: %s" % str(e)[:200]}
                torch.cuda.empty_cache()
        # ... and with only the FUSION transformer in the mode (MODEL.COMPUTE_DTYPE bf16_fusion: the detector stays fp32-grade): the
        # interactron step keeps whole-gradient cosine 0.979 / losses 0.35 % against the oracle there (0.935 / 5 % with everything in bf16)
        sub.compute_dtype = "bf16_fusion"
        inter_f = {}
        for name, (size, eps, steps, warm) in (("p300_e16", (300, 16, 10, 3)), ("n800_e8", (800, 8, 5, 2))):
            if name == "n800_e8" and args.n800_episodes <= 0:
                continue
            try:
                r = run_workload(sub, size, eps, eps, steps, warm, ctx, False, "bench-b16f-" + name)
                inter_f[name] = {"workload": "%d episodes/GPU x 5 frames x 3x%dx%d, same meta-train step" % (eps, size, size), "value": r["frames_per_s"],
                                 "unit": "frames/s", "ms_per_step": r["ms_per_step"], "steps": steps, "warmup": warm,
                                 "peak_memory_GB": r["peak_memory_GB"], "step_graphs": r["step_graphs"]}
            except torch.cuda.OutOfMemoryError as e:
                inter_f[name] = {"error": "out of memory: %s" % str(e)[:200]}
                torch.cuda.empty_cache()
        b16_line = {"interactron": inter,
                    "interactron_bf16_fusion": dict(inter_f, dtype="fp32-grade detector + bf16 fusion transformer (MODEL.COMPUTE_DTYPE bf16_fusion); gradient "
                                                    "fidelity: tests/test_parity_gpu.py::test_interactron_step_with_the_fusion_transformer_in_the_16_bit_mode"),
                    "workload": "configs/multi_frame_baseline.yaml training step (detr_multiframe.forward + clip + Adam), 16 episodes/GPU x 5 frames x "
                                "3x300x300, Q=50, fusion T=2060, procedural weights, train mode",
                    "dtype": "bf16 (activations stored as bf16, bf16 matrix instructions with LDS-DMA operands, fp32 accumulation / statistics / "
                             "parameters; parity at SURVEY 8d's bf16 row: tests/test_parity_gpu.py::test_config2_multiframe_bf16_activations)",
                    "f32": pair["f32"], "bf16": pair["bf16"], "speedup_over_f32": pair["bf16"]["value"] / pair["f32"]["value"]}
    # BASELINE.json configs[4], the bandwidth-bound stress configuration: 1600 x 1600 frames, 200 queries (T = 51 005), one episode
    # per step.  The training step runs fp32-grade attention (a differentiated call never takes the fp8 forward: hipops/attn.py
    # flash_forward); the configuration's fp8 MFMA attention is what predict() runs -- measured beside it, with
    # the fp32-grade predict for comparison.
    stress = None
    if headline_cfg and world == 1 and args.stress_steps > 0 and args.attention_dtype == "fp32":
        q0, args.queries, mode0 = args.queries, 200, args.mode
        try:
            stress = run_workload(args, 1600, 1, 1, args.stress_steps, 2, ctx, False, "bench1600")   # (2 warm-up steps: eager, then the capture)
            args.mode = "predict"
            for dt in ("fp32", "fp8"):
                _ops.ATTENTION_DTYPE = dt
                r = run_workload(args, 1600, 1, 1, args.stress_steps, 1, ctx, False, "bench1600")
                stress["predict_" + dt] = {"value": r["frames_per_s"], "unit": "frames/s", "ms_per_step": r["ms_per_step"],
                                           "peak_memory_GB": r["peak_memory_GB"]}
        except torch.cuda.OutOfMemoryError as e:
            stress = {"error": "out of memory: %s" % str(e)[:200]}
            torch.cuda.empty_cache()
        finally:
            _ops.ATTENTION_DTYPE, args.queries, args.mode = args.attention_dtype, q0, mode0
    # The north-star shape (BASELINE.json: synthetic 5 x 3x800x800 episodes; fusion BLOCK_SIZE = 12 755),
    # measured in the same process after the headline: same step definition, fewer episodes per pass, its own warm-up.
    n800 = r8 = None
    if headline_cfg and args.n800_episodes > 0:
        try:
            r8 = run_workload(args, 800, args.n800_episodes, args.n800_episodes, N800_STEPS, 2, ctx, not args.no_roofline, "bench800")
        except torch.cuda.OutOfMemoryError as e:   # (8 episodes per pass need 149 GB: never lose the headline over it)
            r8 = None
            n800 = {"error": "out of memory at %d episodes per pass: %s" % (args.n800_episodes, str(e)[:200])}
            torch.cuda.empty_cache()
    if r8 is not None:
        n800 = {"workload": "%d episodes/GPU x 5 frames x 3x800x800, Q=%d, fusion T=%d, same step as the headline"
                            % (args.n800_episodes, args.queries, r8["block_size"]),
                "value": r8["frames_per_s"], "unit": "frames/s", "steps": N800_STEPS, "warmup": 2, "ms_per_step": r8["ms_per_step"],
                "peak_memory_GB": r8["peak_memory_GB"],
                "episodes_per_gpu": args.n800_episodes, "gemm_gflop_per_step": r8["gemm_gflop_per_step"],
                "attention_gflop_per_step": r8["attention_gflop_per_step"], "roofline": r8["roofline"]}

    if rank == 0:
        from interactron_amd import hipops
        cfg = head["cfg"]
        line = {
            "metric": "frames/sec (5-frame episodes of 3x%dx%d frames)" % (args.size, args.size), "value": head["frames_per_s"], "unit": "frames/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": head["ms_per_step"],
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": {"f32": "f32", "bf16": "bf16 (activations stored as bf16, bf16 matrix instructions, fp32 accumulation / statistics / parameters)",
                      "bf16_fusion": "f32 detector (fp32-grade) + bf16 fusion transformer (activations stored as bf16, bf16 matrix instructions, single-term "
                                     "attention; fp32 accumulation / statistics / parameters)",
                      "single_pass": "bf16-class (single-pass 16-bit contractions: fp16 x 2^E per 32x32 block, fp32 accumulation; fp32 storage, "
                                     "fp32-grade attention / norms)"}[args.compute_dtype],
            "data": "synthetic",
            "config": {"workload": ("%s training step (%s.forward + all-reduce + clip + Adam), "
                                    "%d episodes/GPU x 5 frames x 3x%dx%d, Q=%d, fusion T=%d, procedural weights, train mode "
                                    "(dropout on; model-level parity is pinned in eval mode AND, with the kernels' own dropout masks handed to the oracle, "
                                    "in train mode: tests/test_parity_gpu.py::test_train_mode_step_with_the_kernels_own_dropout_masks_against_the_oracle)"
                                    % (CONFIGS[args.config][0], CONFIGS[args.config][1], args.episodes, args.size, args.size,
                                       args.queries, cfg["BLOCK_SIZE"])) if args.mode == "train" else
                                   ("%s %s, one episode at a time (%d episodes/GPU x 5 frames x 3x%dx%d), eval mode"
                                    % (CONFIGS[args.config][0], {"predict": "predict() (adapt + frame-0 prediction)",
                                                                 "predict-batched": "predict() on all episodes of the batch at once",
                                                                 "interactive": "interactive episode (4 x get_next_action + predict)"}
                                       [args.mode], args.episodes, args.size, args.size)),
                       "mode": args.mode, "episodes_per_gpu": args.episodes, "frame_size": args.size, "inner_steps": args.inner_steps,
                       "attention": hipops.ATTENTION_IMPL, "attention_dtype": hipops.ATTENTION_DTYPE,
                       "peak_memory_GB": head.get("peak_memory_GB"), "parallelism": "dp%d" % world,
                       "global_batch": args.global_batch if strong else args.episodes * world,
                       "host_issue_ms_per_step": head["host_issue_ms_per_step"], "step_graphs": head["step_graphs"],
                       "contraction_kernel": os.environ.get("IX_GEMM_KERNEL", "x3 (default: fp16x3 form, bf16x6 for narrow tiles)"),
                       "flash_tr": hipops.FLASH_TR + (" (two fp16 planes, 3 MFMAs per k-slice in the token-contracting products)"
                                                   if hipops.FLASH_TR == "f16" else " (three bf16 planes, 6 MFMAs)")},
            "gemm_gflop_per_step": head["gemm_gflop_per_step"], "gemm_launches_per_step": head["gemm_launches_per_step"],
            "attention_gflop_per_step": head["attention_gflop_per_step"],
            "roofline": head["roofline"],
            "cpu_baseline": None,
            "n800": n800,
            # BASELINE.json's north_star quotes frames/s on synthetic 5 x 3x800x800 episodes: that figure, at top level (the
            # headline `value` is configs/interactron.yaml at the reference's real 300 x 300 shapes)
            "north_star": ({"metric": "frames/sec (5-frame episodes of 3x800x800 frames)", "value": n800["value"], "unit": "frames/s",
                            "workload": n800["workload"], "ms_per_step": n800["ms_per_step"], "steps": n800["steps"],
                            "roofline": ({k: n800["roofline"][k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic",
                                                                           "algorithmic_tflops", "kernel_ms_per_step")}
                                         if n800.get("roofline") else None),
                            "attention_ms_per_step": (n800["roofline"]["attention_kernels"]["kernel_ms_per_step"] if n800.get("roofline") else None),
                            "cpu_baseline": None} if n800 is not None and "error" not in n800 else None),
            "small_e": None, "strong": None,
            "stress": (stress if stress is None or "error" in stress else
                       {"workload": "BASELINE.json configs[4]: 1 episode/GPU x 5 frames x 3x1600x1600, Q=200, fusion T=%d, same training step "
                                    "(fp32-grade attention: differentiated calls never take the fp8 forward, its output breaks the derivative "
                                    "passes' dO . O identity -- tests/test_ops_gpu.py::test_fp8_attention_is_for_calls_that_are_not_differentiated); "
                                    "predict_fp8 / predict_fp32: predict() on the same episode with the forward attention products on fp8 "
                                    "(OCP e4m3) MFMA / fp32-grade" % stress["block_size"],
                        "value": stress["frames_per_s"], "unit": "frames/s", "steps": args.stress_steps, "warmup": 2,
                        "ms_per_step": stress["ms_per_step"], "peak_memory_GB": stress["peak_memory_GB"], "attention_dtype": "fp32 (training step)",
                        "predict_fp8": stress.get("predict_fp8"), "predict_fp32": stress.get("predict_fp32"),
                        "step_graphs": stress["step_graphs"]}),
            "bf16_mode": b16_line,
            "inner5": ({"workload": "%d episodes/GPU x 5 frames x 3x300x300, MODEL.INNER_STEPS = 5 (BASELINE.json north_star's 5-step adapt loop; "
                                    "the reference and the headline take 1 step)" % args.inner5_episodes, "inner_steps": 5,
                        "episodes_per_gpu": args.inner5_episodes, "steps": 3, "warmup": 2, "ms_per_step": inner5["ms_per_step"],
                        "value": inner5["frames_per_s"], "unit": "frames/s", "peak_memory_GB": inner5.get("peak_memory_GB"),
                        "step_graphs": inner5["step_graphs"]} if inner5 is not None else None),
            "hbm_kernels": hbm_kernels(torch, hipops, dev, min(args.chunk, args.episodes) * cfg["BLOCK_SIZE"],
                                       (min(args.chunk, args.episodes), cfg["BLOCK_SIZE"])) if not args.no_roofline else None,
            "rccl_ranks": world if world > 1 and head["allreduce"] and head["allreduce"]["backend"] == "rccl" else 0,
            "allreduce": head["allreduce"],
        }
        if small is not None:
            ar = ALLREDUCE_MS_ASSUMED
            line["small_e"] = {"workload": "%d episodes/GPU (the per-GPU share of the reference's global batch of 16 on 8 GPUs), same step, "
                                           "replayed from captured HIP graphs" % args.small_e, "episodes_per_gpu": args.small_e,
                               "steps": 10, "warmup": 3, "ms_per_step": small["ms_per_step"], "host_issue_ms_per_step": small["host_issue_ms_per_step"],
                               "value": small["frames_per_s"], "unit": "frames/s", "step_graphs": small["step_graphs"],
                               "allreduce_ms_assumed": ar,
                               "strong_scaling_projection": head["ms_per_step"] * (args.small_e * 8.0 / args.episodes) / (small["ms_per_step"] + ar),
                               "projection": "t(E = %d) x (8 x %d / %d) / (t(E = %d) + allreduce): speed-up of 8 GPUs over 1 GPU on one global "
                                             "batch of %d episodes" % (args.episodes, args.small_e, args.episodes, args.small_e, 8 * args.small_e)}
        if strong_run is not None:
            line["strong"] = {"workload": "global batch 16 (reference BATCH_SIZE), rank r runs episodes r::%d: %d per GPU" % (world, 16 // world),
                              "global_batch": 16, "episodes_per_gpu": 16 // world, "steps": 10, "warmup": 3, "scaling": "strong",
                              "ms_per_step": strong_run["ms_per_step"], "host_issue_ms_per_step": strong_run["host_issue_ms_per_step"],
                              "value": strong_run["frames_per_s"], "unit": "frames/s", "step_graphs": strong_run["step_graphs"]}
        if not args.no_cpu_baseline and world == 1 and args.mode == "train" and args.size <= 800 and args.attention_dtype == "fp32":
            line["cpu_baseline"] = cpu_baseline(cfg, args.size, args.config)
            if n800 is not None and "error" not in n800:
                n800["cpu_baseline"] = cpu_detector_800()
                n800["cpu_baseline"]["comparable"] = False
                n800["cpu_baseline"]["note"] = ("NOT a like-for-like step: the detector forward alone on one 800 x 800 frame.  A full "
                                                "meta-train step at T = 12 755 needs 8 x T^2 fp32 attention tensors with double backward "
                                                "on the host (BASELINE.md 3); never divide the n800 value by this figure -- the "
                                                "like-for-like pair is `two_frame_episode`")
                # the comparable pair: ONE meta-train episode of two 800 x 800 frames (T = 5 105) through the oracle and through HIP,
                # measured here (~ 35 s of host time for the oracle's episode)
                if not args.no_two_frame:
                    n800["two_frame_episode"] = two_frame_pair(dev)
                    line["north_star"]["cpu_baseline"] = dict(n800["two_frame_episode"]["cpu_baseline"],
                                                              workload=n800["two_frame_episode"]["workload"],
                                                              hip_frames_per_s_same_workload=n800["two_frame_episode"]["hip"]["frames_per_s"],
                                                              gpu_over_cpu=n800["two_frame_episode"]["gpu_over_cpu"])
        sys.stdout.flush()
        os.write(JSON_FD, (json.dumps(line) + "\n").encode())   # the one line on the real stdout
    if world > 1:
        dist.barrier(device_ids=[local]) if dist.get_backend() == "nccl" else dist.barrier()
        torch.cuda.synchronize()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
