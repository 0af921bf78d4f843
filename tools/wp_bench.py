"""ix_gemm_wp_f32 (activation x weight planes, csrc/gemm_wp.hip) against the 12-wave kernel (ix_gemm_f32_ws) on the step's
Linear-layer shapes: accuracy vs float64 and time per call (weight split excluded / listed beside it).

    python tools/wp_bench.py            # on the GPU box"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from interactron_amd import _lib

lib = _lib.load()
dev = torch.device("cuda")
stream = torch.cuda.current_stream().cuda_stream
SHAPES = [  # M, N, K, batch, b_kc   (profiles/r4b_gemm_launches_e16_300.csv, plain contractions with a weight as B)
    (1805, 2048, 256, 16, 1), (1805, 2048, 256, 16, 0), (1805, 256, 2048, 16, 1), (1805, 256, 2048, 16, 0),
    (1805, 2048, 512, 16, 1), (1805, 512, 2048, 16, 0), (32960, 512, 2048, 1, 1), (32960, 2048, 512, 1, 0),
    (32960, 1536, 512, 1, 1), (32960, 512, 512, 1, 1), (28880, 256, 256, 1, 1), (1805, 1024, 256, 16, 1),
    (250, 256, 256, 16, 1), (250, 2048, 256, 16, 1), (12500, 2048, 256, 8, 1), (12500, 256, 2048, 8, 0),
]


def timed(fn, it=20):
    fn()
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / it * 1e3


def main():
    g = torch.Generator().manual_seed(1)
    print("%-28s %9s %9s %9s %8s %8s %9s %9s" % ("M N K batch bkc", "old us", "wp us", "split us", "old TF", "wp TF", "err old", "err wp"))
    tot_old = tot_new = 0.0
    for (M, N, K, b, bkc) in SHAPES:
        A = torch.randn(b, M, K, generator=g).to(dev)
        W = (torch.randn(b, N, K, generator=g) if bkc else torch.randn(b, K, N, generator=g)).to(dev) * 0.05
        bias = torch.randn(b, N, generator=g).to(dev)
        C0, C1 = torch.empty(b, M, N, device=dev), torch.empty(b, M, N, device=dev)
        pb, ub = ctypes.c_size_t(), ctypes.c_size_t()
        lib.ix_wp_planes_bytes(N, K, b, ctypes.byref(pb), ctypes.byref(ub))
        planes = torch.empty(pb.value, dtype=torch.uint8, device=dev)
        us = torch.empty(ub.value // 4, dtype=torch.float32, device=dev)
        ws_n = ctypes.c_size_t()
        lib.ix_workspace_bytes_gemm_f32(M, N, K, 1, bkc, K, K if bkc else N, b, 1, M * K, N * K, A.data_ptr(), W.data_ptr(), 0, 0,
                                        ctypes.byref(ws_n))
        ws = torch.zeros(max(ws_n.value, 65536), dtype=torch.uint8, device=dev)

        def old():
            rc = lib.ix_gemm_f32_ws(A.data_ptr(), W.data_ptr(), C0.data_ptr(), bias.data_ptr(), M, N, K, 1, bkc, K, K if bkc else N, N,
                                    b, 1, M * K, 0, N * K, 0, M * N, 0, N, 1.0, 0, 0, ws.data_ptr(), ws.numel(), stream)
            assert rc == 0, lib.ix_last_error()

        def split():
            rc = lib.ix_wp_split_f32(W.data_ptr(), K if bkc else N, N * K, N, K, bkc, b, planes.data_ptr(), us.data_ptr(), stream)
            assert rc == 0, lib.ix_last_error()

        def new():
            rc = lib.ix_gemm_wp_f32(A.data_ptr(), K, M * K, 0, planes.data_ptr(), us.data_ptr(), 0, C1.data_ptr(), N, M * N, 0,
                                    bias.data_ptr(), N, M, N, K, b, 1, 1.0, stream)
            assert rc == 0, lib.ix_last_error()

        split()
        t_old, t_new, t_split = timed(old), timed(new), timed(split)
        Wm = W if not bkc else W.transpose(1, 2)
        ref = A[0].double() @ Wm[0].double() + bias[0].double()
        scale = A[0].double().abs() @ Wm[0].double().abs() + bias[0].double().abs() + 1e-300
        e0 = float(((C0[0].double() - ref).abs() / scale).max())
        e1 = float(((C1[0].double() - ref).abs() / scale).max())
        fl = 2.0 * M * N * K * b
        parts = []
        for flags in (1, 2, 3, 4, 7, 8 + (2 << 8), 8 + (6 << 8), 8 + (12 << 8), 8 + (24 << 8)):
            lib.ix_gemm_wp_debug(flags)
            parts.append("d%d %.1f" % (flags, timed(new)))
        lib.ix_gemm_wp_debug(0)
        print("   " + "  ".join(parts) + "   (1 no store, 2 no mma, 4 no dma)")
        print("%-28s %9.1f %9.1f %9.1f %8.1f %8.1f %9.2e %9.2e" % ("%d %d %d %d %d" % (M, N, K, b, bkc), t_old, t_new, t_split,
                                                                   fl / t_old / 1e6, fl / t_new / 1e6, e0, e1))
        tot_old += t_old
        tot_new += t_new
    print("sum: old %.1f us, wp %.1f us" % (tot_old, tot_new))


if __name__ == "__main__":
    main()
