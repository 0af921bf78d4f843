"""Times the three implicit-GEMM convolution kinds (ix_conv_gemm_f32: forward, data gradient, weight gradient) on the backbone's
3 x 3 shapes of the 16-episode 300 x 300 step (ResNet-50 layer1-4 conv2, shared and per-episode weights; dilated last stage),
alternating best-of-3.  usage (GPU box): python tools/conv_bench.py    [IX_LIB_PATH=... for another build]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from interactron_amd import hipops as ops

# E (weight sets), images per set, H, W, Cin = Cout, stride, dilation      (80 frames of 16 episodes; layer4 of DETR-DC5 is dilated)
SHAPES = [(1, 80, 75, 75, 64, 1, 1), (1, 80, 75, 75, 128, 2, 1), (1, 80, 38, 38, 128, 1, 1), (1, 80, 38, 38, 256, 2, 1),
          (16, 5, 19, 19, 256, 1, 1), (16, 5, 19, 19, 512, 1, 2), (1, 80, 19, 19, 512, 1, 2)]


def timed(fn, it=10):
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
    g = torch.Generator().manual_seed(3)
    tot = [0.0, 0.0, 0.0]
    print("%-34s %9s %9s %9s   TFLOP/s fwd" % ("E imgs H W C stride dil", "fwd us", "dgrad us", "wgrad us"))
    for (E, imgs, H, W, C, stride, dil) in SHAPES:
        pad = dil
        geo = ops.conv_geom(E * imgs, H, W, C, 3, 3, stride, pad, dil)
        cg = ops.ConvGemmGeom(E, imgs, H, W, C, geo.OH, geo.OW, C, 3, 3, stride, pad, dil)
        assert ops.conv_gemm_supported(cg), cg
        x = torch.randn(E * imgs, H, W, C, generator=g).cuda()
        w = (torch.randn(*((E,) if E > 1 else ()), C, 3, 3, C, generator=g) * 0.05).cuda()
        dy = torch.randn(E * imgs, geo.OH, geo.OW, C, generator=g).cuda()
        fns = [lambda: ops._conv_gemm(0, x, w, (E * imgs, geo.OH, geo.OW, C), cg),
               lambda: ops._conv_gemm(1, dy, w, (E * imgs, H, W, C), cg),
               lambda: ops._conv_gemm(2, dy, x, tuple(w.shape), cg)]
        t = [[], [], []]
        for _ in range(3):
            for k in range(3):
                t[k].append(timed(fns[k]))
        t = [min(v) for v in t]
        fl = 2.0 * E * imgs * geo.OH * geo.OW * C * 9 * C
        print("%-34s %9.1f %9.1f %9.1f   %6.1f" % ("%d %d %d %d %d %d %d" % (E, imgs, H, W, C, stride, dil), t[0], t[1], t[2], fl / t[0] / 1e6))
        for k in range(3):
            tot[k] += t[k]
    print("sum: fwd %.1f us, dgrad %.1f us, wgrad %.1f us" % tuple(tot))


if __name__ == "__main__":
    main()
