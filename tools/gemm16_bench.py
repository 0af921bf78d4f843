"""The bf16 GEMM (csrc/gemm16.hip) alone on the contraction shapes of a 16-episode multi_frame_baseline step at 300 x 300 (forward,
input gradient, weight gradient of the Linear layers and 1 x 1 convolutions): time per call (HIP events, median of 3 x 10 launches),
TFLOP/s against the 2.5 PFLOP/s dense bf16 peak, algorithmic GB/s.

    python tools/gemm16_bench.py [--json out.json]
"""
import argparse
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from interactron_amd import _lib  # noqa: E402

ROWS_ENC, ROWS_FUS, ROWS_DEC, PIX38 = 28880, 32960, 4000, 115520
LINEARS = [   # (rows, out, in, count per step: forward)
    (ROWS_ENC, 512, 256, 6), (ROWS_ENC, 256, 256, 12), (ROWS_ENC, 2048, 256, 6), (ROWS_ENC, 256, 2048, 7),
    (PIX38, 128, 512, 3), (PIX38, 512, 128, 4), (PIX38, 128, 256, 1), (ROWS_ENC, 256, 1024, 5), (ROWS_ENC, 1024, 256, 6),
    (ROWS_ENC, 512, 2048, 2), (ROWS_ENC, 2048, 512, 3), (ROWS_ENC, 512, 1024, 1), (ROWS_ENC, 2048, 1024, 1),
    (ROWS_FUS, 1536, 512, 4), (ROWS_FUS, 512, 512, 5), (ROWS_FUS, 2048, 512, 4), (ROWS_FUS, 512, 2048, 4),
    (ROWS_DEC, 256, 256, 30), (ROWS_DEC, 2048, 256, 6), (ROWS_DEC, 256, 2048, 6), (ROWS_DEC, 512, 1496, 1),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json", default=None)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--stages", type=int, default=0, help="1 / 2: force the kernel form (ix_gemm_b16_set_stages); 0: the library's default")
    args = ap.parse_args()
    lib = _lib.load()
    if args.stages:
        lib.ix_gemm_b16_set_stages(args.stages)
    st = torch.cuda.current_stream().cuda_stream
    ws = torch.zeros(512 << 20, dtype=torch.uint8, device="cuda")
    rows_out, tot = [], {"fwd": [0.0, 0.0], "dx": [0.0, 0.0], "dw": [0.0, 0.0]}

    def run(kind, M, N, K, a_kc, b_kc, A, B, lda, ldb, c_f32):
        C = torch.empty(M, N, dtype=torch.float32 if c_f32 else torch.bfloat16, device="cuda")

        def call():
            rc = lib.ix_gemm_b16(A.data_ptr(), B.data_ptr(), C.data_ptr(), None, M, N, K, a_kc, b_kc, lda, ldb, N, 1, 1, 0, 0, 0, 0, 0, 0, 0,
                                 1.0, int(c_f32), None, None, None, 0, ws.data_ptr(), ws.numel(), st)
            assert rc == 0, lib.ix_last_error()
        call()
        times = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                call()
            e1.record()
            e1.synchronize()
            times.append(e0.elapsed_time(e1) / args.iters * 1e-3)
        t = sorted(times)[1]
        fl = 2.0 * M * N * K
        by = 2.0 * (M * K + K * N) + (4.0 if c_f32 else 2.0) * M * N
        return t, fl, by

    for rows, out, inn, cnt in LINEARS:
        x = torch.randn(rows, inn, device="cuda").to(torch.bfloat16)
        w = (torch.randn(out, inn, device="cuda") * 0.05).to(torch.bfloat16)
        dy = torch.randn(rows, out, device="cuda").to(torch.bfloat16)
        for kind, (M, N, K, akc, bkc, A, B, lda, ldb, f32) in (
                ("fwd", (rows, out, inn, 1, 1, x, w, inn, inn, False)),
                ("dx", (rows, inn, out, 1, 0, dy, w, out, inn, False)),
                ("dw", (out, inn, rows, 0, 0, dy, x, out, inn, True))):
            t, fl, by = run(kind, M, N, K, akc, bkc, A, B, lda, ldb, f32)
            rows_out.append({"kind": kind, "M": M, "N": N, "K": K, "us": t * 1e6, "tflops": fl / t / 1e12, "GBps": by / t / 1e9, "count": cnt})
            tot[kind][0] += t * cnt
            tot[kind][1] += fl * cnt
            print("%-3s M %6d N %5d K %6d  %8.1f us  %7.1f TFLOP/s  %7.1f GB/s  x%d" % (kind, M, N, K, t * 1e6, fl / t / 1e12, by / t / 1e9, cnt))
    summary = {k: {"ms_per_step": v[0] * 1e3, "tflops": v[1] / v[0] / 1e12, "frac_of_2500": v[1] / v[0] / 2.5e15} for k, v in tot.items()}
    allt, allf = sum(v[0] for v in tot.values()), sum(v[1] for v in tot.values())
    summary["all"] = {"ms_per_step": allt * 1e3, "tflops": allf / allt / 1e12, "frac_of_2500": allf / allt / 2.5e15}
    print(json.dumps(summary, indent=1))
    if args.json:
        json.dump({"shapes": rows_out, "summary": summary}, open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
