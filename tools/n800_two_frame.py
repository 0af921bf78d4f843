"""A like-for-like CPU figure beside the north-star shape: ONE meta-train step of `interactron` on a TWO-frame 800 x 800 episode
(fusion T = 2 x (2500 + 50) + 5 = 5 105) through the CPU oracle (oracle/episode.py:interactron_forward, all usable host cores)
and through the HIP path, same inputs, same weights, losses compared.  The five-frame episode (T = 12 755) is not practical on
the host: reference models/gpt.py:48-52 materialises 8 x T^2 fp32 attention per layer (5.2 GB at T = 12 755, with a double
backward over it; 834 MB at T = 5 105).  The first-order branch draws its frame from the two that exist (the reference draws
random.randint(0, 4) for its five: models/interactron.py:126).

    python tools/n800_two_frame.py [--frames 2] [--out profiles/<name>.json]      (GPU box; ~1-2 minutes of host time)
"""
import argparse
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=2)
    ap.add_argument("--size", type=int, default=800)
    ap.add_argument("--gpu-steps", type=int, default=5)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    from interactron_amd import Config, build_model
    from interactron_amd.synthetic import load_procedural, procedural_state_dict, synthetic_episodes
    from oracle import detector as od, episode as oe, fusion as of

    s = args.frames
    cfg, tokens = bench.model_cfg(args.size, 50, 1, "interactron")
    cfg["BLOCK_SIZE"] = s * (tokens + 50) + 5
    data = synthetic_episodes(1, frames=s, height=args.size, width=args.size, tag="two-frame")
    # the policy head always scores four moves (reference models/interactron.py:116-118): four scripted actions, as in a 5-frame episode
    data["actions"] = synthetic_episodes(1, frames=5, height=16, width=16, tag="two-frame")["actions"]
    real_randint = random.randint
    random.randint = lambda a, b: real_randint(a, min(b, s - 1))   # the first-order branch's frame: one of the s that exist

    # ---- HIP
    dev = torch.device("cuda", 0)
    model = build_model(Config(**cfg))
    load_procedural(model.fusion, "fusion.")
    model = model.to(dev).eval()
    gpu = bench.to_gpu(data, dev)
    times = []
    for i in range(1 + args.gpu_steps):
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

    # ---- oracle
    cores = bench.usable_cores()
    torch.set_num_threads(cores)
    det = {k[len("detector."):]: v for k, v in
           procedural_state_dict({"detector." + k: v for k, v in od.detr_state_shapes().items()}).items()}
    fus = {k[len("fusion."):]: v for k, v in
           procedural_state_dict({"fusion." + k: v for k, v in of.fusion_state_shapes(cfg, "gpt").items()}).items()}
    random.seed(11)
    t0 = time.perf_counter()
    _, ref_losses, _ = oe.interactron_forward(det, fus, data, cfg, {}, "gpt")
    cpu_s = time.perf_counter() - t0
    worst = max(abs(hip_losses[k] - float(v)) / max(abs(float(v)), 1.0) for k, v in ref_losses.items())
    out = {"workload": "one interactron meta-train episode of %d frames x 3x%dx%d, Q=50, fusion T=%d, eval mode (dropout off), procedural weights"
                       % (s, args.size, args.size, cfg["BLOCK_SIZE"]),
           "hip": {"seconds_per_episode": hip_s, "frames_per_s": s / hip_s, "sample": "median of %d steps after 1 warm-up" % args.gpu_steps},
           "cpu_baseline": {"seconds_per_episode": cpu_s, "value": s / cpu_s, "unit": "frames/s", "cores": cores, "kind": "port",
                            "sample": "1 episode through oracle/episode.py:interactron_forward on %d threads of %s (no warm-up)"
                                      % (cores, bench.cpu_model_name())},
           "gpu_over_cpu": cpu_s / hip_s, "worst_relative_loss_difference": worst, "comparable": True}
    print(json.dumps(out, indent=1))
    if args.out:
        json.dump(out, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
