"""The stride-2 data gradients of the backbone (layer2.0 / layer3.0 conv2 3x3 and the layer3.0 1x1 downsample; reference
models/detr_models/backbone.py:88-90 -> torchvision resnet50) at the 300^2 x 16-episode and 800^2 x 8-episode steps: the
parity-class form (csrc/gemm.hip conv_bwd_data_s2) with the classes written in place against the same form through scratch +
the interleaving pass, alternating best-of-3 (IX_S2_BENCH_MODES="1 0": the class form against the one-launch gather).
usage (GPU box): python tools/conv_s2_bench.py [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from interactron_amd import hipops as ops

MODES = tuple(int(m) for m in os.environ.get("IX_S2_BENCH_MODES", "2 6").split())
# imgs, H, W, Cin, Cout, k, pad
SHAPES = [(80, 75, 75, 128, 128, 3, 1), (80, 38, 38, 256, 256, 3, 1), (80, 38, 38, 512, 1024, 1, 0),
          (40, 200, 200, 128, 128, 3, 1), (40, 100, 100, 256, 256, 3, 1), (40, 100, 100, 512, 1024, 1, 0)]


def timed(fn, it):
    fn(); fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / it * 1e3


def main():
    it = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    g = torch.Generator().manual_seed(3)
    lib = ops._L()
    print("%-32s %12s %12s   useful TFLOP/s (second -> first)" % ("imgs H W Cin Cout k", "mode %d us" % MODES[0], "mode %d us" % MODES[1]))
    for (imgs, H, W, Cin, Cout, k, pad) in SHAPES:
        geo = ops.conv_geom(imgs, H, W, Cin, k, k, 2, pad, 1)
        cg = ops.ConvGemmGeom(1, imgs, H, W, Cin, geo.OH, geo.OW, Cout, k, k, 2, pad, 1)
        w = (torch.randn(Cout, k, k, Cin, generator=g) * 0.05).cuda()
        dy = torch.randn(imgs, geo.OH, geo.OW, Cout, generator=g).cuda()
        fn = lambda: ops._conv_gemm(1, dy, w, (imgs, H, W, Cin), cg)
        t = [[], []]
        for _ in range(3):
            for mode in MODES:
                lib.ix_conv_set_s2_split(mode)
                ops._conv_ws.clear()
                t[MODES.index(mode)].append(timed(fn, it))
        lib.ix_conv_set_s2_split(1)
        ops._conv_ws.clear()
        t = [min(v) for v in t]
        fl = 2.0 * imgs * geo.OH * geo.OW * Cout * k * k * Cin
        print("%-32s %12.1f %12.1f   %6.1f -> %6.1f" % ("%d %d %d %d %d %d" % (imgs, H, W, Cin, Cout, k), t[0], t[1], fl / t[1] / 1e6, fl / t[0] / 1e6))


if __name__ == "__main__":
    main()
