"""Sweep tile/split hints of ix_gemm_f32 per shape: empirical best vs the library's own choice (hint 0/0)."""
import csv, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from interactron_amd import _lib
lib = _lib.load()
top = int(sys.argv[1]) if len(sys.argv) > 1 else 45
shapes = [tuple(int(r[k]) for k in ("M", "N", "K", "batch", "a_kc", "b_kc", "count"))
          for r in csv.DictReader(open(os.path.join(os.path.dirname(__file__), "gemm_shapes_p300.csv")))]
shapes.sort(key=lambda s: -2.0 * s[0] * s[1] * s[2] * s[3] * s[6])
stream = torch.cuda.current_stream().cuda_stream
def timeit(M, N, K, b, akc, bkc, A, B, C, th, sh, reps=12):
    lda = K if akc else M; ldb = K if bkc else N
    def run():
        assert lib.ix_gemm_f32(A.data_ptr(), B.data_ptr(), C.data_ptr(), None, M, N, K, akc, bkc, lda, ldb, N, b, 1,
                               M * K, 0, K * N, 0, M * N, 0, 0, 1.0, th, sh, stream) == 0
    for _ in range(2): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
auto_tot = best_tot = 0.0
print("%6s %6s %6s %4s akc bkc cnt | auto us | best us (tile,split) | all configs" % ("M", "N", "K", "b"))
for (M, N, K, b, akc, bkc, cnt) in shapes[:top]:
    A = torch.randn(b, M * K, device="cuda"); B = torch.randn(b, K * N, device="cuda"); C = torch.empty(b, M * N, device="cuda")
    auto = timeit(M, N, K, b, akc, bkc, A, B, C, 0, 0)
    res = {}
    for th in (64, 128):
        bk = 64 if th == 64 else 32
        for sh in (1, 2, 3, 4, 6, 8, 12, 16):
            if sh > 1 and K < 2 * bk * sh: continue
            res[(th, sh)] = timeit(M, N, K, b, akc, bkc, A, B, C, th, sh)
    bk_ = min(res, key=res.get)
    auto_tot += auto * cnt / 1e3; best_tot += min(res[bk_], auto) * cnt / 1e3
    print("%6d %6d %6d %4d  %d   %d %4d | %7.1f | %7.1f %s | %s" % (M, N, K, b, akc, bkc, cnt, auto, res[bk_], bk_,
          " ".join("%d/%d:%.0f" % (k[0], k[1], v) for k, v in sorted(res.items()))))
print("weighted: auto %.1f ms, best-of-sweep %.1f ms" % (auto_tot, best_tot))
