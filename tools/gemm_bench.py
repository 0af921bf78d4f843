"""Micro-benchmark of ix_gemm_f32 over the contraction shapes of one meta-train step (tools/gemm_shapes_p300.csv:
M,N,K,batch,a_kc,b_kc,count per step of 2 episodes).  Prints per-shape time / TFLOP/s and the count-weighted total.
Usage (GPU box): python tools/gemm_bench.py [top_n] [tile_hint] [split_hint]"""
import csv, ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from interactron_amd import _lib

lib = _lib.load()
top = int(sys.argv[1]) if len(sys.argv) > 1 else 40
tile_hint = int(sys.argv[2]) if len(sys.argv) > 2 else 0
split_hint = int(sys.argv[3]) if len(sys.argv) > 3 else 0
shape_file = sys.argv[4] if len(sys.argv) > 4 else "gemm_shapes_p300_e8.csv"
shapes = [tuple(int(r[k]) for k in ("M", "N", "K", "batch", "a_kc", "b_kc", "count"))
          for r in csv.DictReader(open(os.path.join(os.path.dirname(__file__), shape_file)))]
shapes.sort(key=lambda s: -2.0 * s[0] * s[1] * s[2] * s[3] * s[6])
stream = torch.cuda.current_stream().cuda_stream
tot_ms = tot_fl = 0.0
allms = 0.0
print("%6s %6s %6s %4s akc bkc cnt |   us    TF/s | weighted ms" % ("M", "N", "K", "b"))
for i, (M, N, K, b, akc, bkc, cnt) in enumerate(shapes):
    A = torch.randn(b, M * K, device="cuda")
    B = torch.randn(b, K * N, device="cuda")
    C = torch.empty(b, M * N, device="cuda")
    lda = K if akc else M
    ldb = K if bkc else N
    need = ctypes.c_size_t(0)   # as the product calls it: split-K planes in a caller workspace + ordered reduction (no atomics)
    assert lib.ix_workspace_bytes_gemm_f32(M, N, K, akc, bkc, lda, ldb, b, 1, M * K, K * N, A.data_ptr(), B.data_ptr(), tile_hint,
                                           split_hint, ctypes.byref(need)) == 0
    ws = torch.zeros(max(need.value, 65536) // 4 + 4, device="cuda")
    def run():
        rc = lib.ix_gemm_f32_ws(A.data_ptr(), B.data_ptr(), C.data_ptr(), None, M, N, K, akc, bkc, lda, ldb, N, b, 1,
                                M * K, 0, K * N, 0, M * N, 0, 0, 1.0, tile_hint, split_hint, ws.data_ptr(), ws.numel() * 4, stream)
        assert rc == 0
    for _ in range(3):
        run()
    reps = 10 if 2.0 * M * N * K * b > 5e9 else 20
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    fl = 2.0 * M * N * K * b
    allms += us * cnt / 1e3
    if i < top:
        print("%6d %6d %6d %4d  %d   %d %4d | %7.1f %6.1f | %7.2f" % (M, N, K, b, akc, bkc, cnt, us, fl / us / 1e6, us * cnt / 1e3))
    tot_fl += fl * cnt
print("count-weighted total over %d shapes: %.1f ms per step, %.1f TFLOP/s" % (len(shapes), allms, tot_fl / allms / 1e9))
