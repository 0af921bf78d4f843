"""gemm_f32_f16x3_w256_kernel (256 x 128 tiles) against the 12-wave kernel's 128 x 128 tiles on the step's plain contraction
shapes (profiles/r4b_gemm_launches_e16_300.csv): time per call through ix_gemm_f32_ws with ix_gemm_set_w256(0 / 2 / 1).

    python tools/w256_bench.py            # on the GPU box"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from interactron_amd import _lib

lib = _lib.load()
dev = torch.device("cuda")
stream = torch.cuda.current_stream().cuda_stream
SHAPES = [  # M, N, K, batch, a_kc, b_kc
    (1805, 2048, 256, 16, 1, 1), (1805, 2048, 256, 16, 1, 0), (1805, 256, 2048, 16, 1, 1), (1805, 256, 2048, 16, 1, 0),
    (1805, 512, 2048, 16, 1, 0), (1805, 1024, 256, 16, 1, 1), (1805, 256, 256, 16, 1, 1), (1805, 256, 256, 16, 1, 0),
    (256, 2048, 1805, 16, 0, 0), (2048, 256, 1805, 16, 0, 0), (256, 256, 1805, 16, 0, 0), (512, 512, 32960, 1, 0, 0),
    (2048, 512, 32960, 1, 0, 0), (4120, 4120, 64, 8, 1, 1), (4120, 64, 4120, 8, 1, 0), (250, 256, 256, 16, 1, 1),
    (250, 2048, 256, 16, 1, 1), (12500, 256, 256, 8, 1, 1), (4120, 512, 512, 8, 1, 1), (1805, 768, 256, 16, 1, 1),
]


def _w256_count(lib):
    import ctypes
    n = ctypes.c_int64(0)
    assert lib.ix_gemm_w256_launches(ctypes.byref(n)) == 0
    return n.value


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
    print("%-30s %9s %9s %9s %7s %7s %6s %s" % ("M N K batch akc bkc", "128 us", "256 us", "auto us", "128 TF", "256 TF", "ratio", "auto"))
    tot = [0.0, 0.0, 0.0]
    for (M, N, K, b, akc, bkc) in SHAPES:
        A = (torch.randn(b, M, K, generator=g) if akc else torch.randn(b, K, M, generator=g)).to(dev)
        W = (torch.randn(b, N, K, generator=g) if bkc else torch.randn(b, K, N, generator=g)).to(dev) * 0.05
        C = torch.empty(b, M, N, device=dev)
        lda, ldb = (K if akc else M), (K if bkc else N)
        ws_n = ctypes.c_size_t()
        lib.ix_workspace_bytes_gemm_f32(M, N, K, akc, bkc, lda, ldb, b, 1, M * K, N * K, A.data_ptr(), W.data_ptr(), 0, 0, ctypes.byref(ws_n))
        ws = torch.zeros(max(ws_n.value, 65536), dtype=torch.uint8, device=dev)

        def run():
            rc = lib.ix_gemm_f32_ws(A.data_ptr(), W.data_ptr(), C.data_ptr(), None, M, N, K, akc, bkc, lda, ldb, N, b, 1, M * K, 0, N * K, 0,
                                    M * N, 0, 0, 1.0, 0, 0, ws.data_ptr(), ws.numel(), stream)
            assert rc == 0, lib.ix_last_error()

        t, outs, took = [[], [], []], [None, None, None], 0
        for rnd_ in range(3):   # alternate the modes (the first timing of a shape runs on colder clocks: 5-8 % slower) and keep the best
            for slot, mode in ((0, 0), (1, 2), (2, 1)) if rnd_ % 2 == 0 else ((2, 1), (1, 2), (0, 0)):
                old = lib.ix_gemm_set_w256(mode)
                before = _w256_count(lib)
                t[slot].append(timed(run, it=10))
                if mode == 1:
                    took = _w256_count(lib) > before
                outs[slot] = C.clone()
                lib.ix_gemm_set_w256(old)
        t = [min(x) for x in t]
        same = bool(torch.equal(outs[0], outs[1]))
        fl = 2.0 * M * N * K * b
        print("%-30s %9.1f %9.1f %9.1f %7.1f %7.1f %6.2f %s%s" % ("%d %d %d %d %d %d" % (M, N, K, b, akc, bkc), t[0], t[1], t[2], fl / t[0] / 1e6,
                                                                 fl / t[1] / 1e6, t[0] / t[1], "w256" if took else "128", "" if same else "  BITS DIFFER"))
        for i in range(3):
            tot[i] += t[i]
    print("sum: 128 %.1f us, 256 %.1f us, auto %.1f us" % tuple(tot))


if __name__ == "__main__":
    main()
