"""Which contractions carry the step's GEMM time: one EAGER meta-train step with every contraction call (hipops._run_gemm,
hipops._conv_gemm) bracketed by events on the compute stream, grouped by shape.  The eager step serialises what the
captured graphs overlap, so the per-shape times are kernel + split / reduce passes of that call, not step shares.

    python tools/gemm_census.py [--size 800] [--episodes 8] [--top 40] [--out profiles/<name>.json]     (GPU box)
"""
import argparse
import collections
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=800)
    ap.add_argument("--episodes", type=int, default=8)
    ap.add_argument("--top", type=int, default=40)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    os.environ.setdefault("IX_STEP_GRAPH", "0")
    from interactron_amd import Config, build_model, hipops
    from interactron_amd.synthetic import load_procedural, synthetic_episodes
    from interactron_amd.trainer import FlatOuterStep

    dev = torch.device("cuda", 0)
    cfg, _ = bench.model_cfg(args.size, 50, 8, "interactron", step_graph="off")
    model = build_model(Config(**cfg))
    load_procedural(model.fusion, "fusion.")
    model = model.to(dev).train()
    outer = FlatOuterStep(model, detector_lr=1e-5, fusion_lr=1e-4, max_norm=1.0)
    data = bench.to_gpu(synthetic_episodes(args.episodes, height=args.size, width=args.size, tag="census"), dev)
    random.seed(5)

    def step():
        model(data)
        outer.step()

    step()
    torch.cuda.synchronize()
    log = []
    run_gemm, conv_gemm = hipops._run_gemm, hipops._conv_gemm

    def timed(key, flop, fn, *a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn(*a, **k)
        e1.record()
        log.append((key, flop, e0, e1))
        return out

    def run_gemm_t(a, b, bias, sp, fill=True):
        routed = hipops._wp_plan(a, b, bias, sp) is not None
        key = ("gemm", sp.M, sp.N, sp.K, sp.bo * sp.bi, 0 if sp.A.trans else 1, 1 if sp.B.trans else 0, "wp" if routed else "")
        return timed(key, 2.0 * sp.M * sp.N * sp.K * sp.bo * sp.bi, run_gemm, a, b, bias, sp, fill)

    def conv_gemm_t(kind, src, other, out_shape, cg):
        M = cg.E * cg.imgs * cg.OH * cg.OW
        key = ("conv" + "fdw"[kind], M, cg.Cout, cg.KH * cg.KW * cg.Cin, 1, cg.KH, cg.stride, "%dx%d" % (cg.H, cg.W))
        return timed(key, 2.0 * M * cg.Cout * cg.KH * cg.KW * cg.Cin, conv_gemm, kind, src, other, out_shape, cg)

    hipops._run_gemm, hipops._conv_gemm = run_gemm_t, conv_gemm_t
    step()
    torch.cuda.synchronize()
    hipops._run_gemm, hipops._conv_gemm = run_gemm, conv_gemm
    agg = collections.OrderedDict()
    for key, flop, e0, e1 in log:
        a = agg.setdefault(key, [0, 0.0, 0.0])
        a[0] += 1
        a[1] += e0.elapsed_time(e1)
        a[2] += flop
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    total = sum(v[1] for _, v in rows)
    print("%d contraction calls, %.1f ms in all (eager, serialised)" % (len(log), total))
    print("%-6s %8s %6s %8s %5s  a b  %-8s | %5s %9s %8s %7s" % ("kind", "M", "N", "K", "batch", "note", "calls", "ms", "us/call", "TF/s"))
    out = []
    for key, (n, ms, fl) in rows[:args.top]:
        print("%-6s %8d %6d %8d %5d  %s %s  %-8s | %5d %9.2f %8.1f %7.1f" % (key[0], key[1], key[2], key[3], key[4], key[5], key[6], key[7],
                                                                             n, ms, ms * 1e3 / n, fl / ms / 1e9))
    for key, (n, ms, fl) in rows:
        out.append({"kind": key[0], "M": key[1], "N": key[2], "K": key[3], "batch": key[4], "a": key[5], "b": key[6], "note": key[7],
                    "calls": n, "ms": ms, "tflops": fl / ms / 1e9})
    if args.out:
        json.dump({"workload": "%d episodes x 5 frames x 3x%dx%d, one eager meta-train step" % (args.episodes, args.size, args.size),
                   "calls": len(log), "ms": total, "shapes": out}, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
