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
    out = bench.two_frame_pair(torch.device("cuda", 0), args.frames, args.size, args.gpu_steps)   # (bench.py measures the same pair in every default run)
    print(json.dumps(out, indent=1))
    if args.out:
        json.dump(out, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
