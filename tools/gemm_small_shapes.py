"""Small-M shapes the cost model sends to the exact-fp32 kernel (tile 64): would the bf16x6 kernel be faster?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from interactron_amd import _lib
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
def t(M, N, K, b, akc, bkc, th, sh=0, reps=20):
    A = torch.randn(b, M * K, device="cuda"); B = torch.randn(b, K * N, device="cuda"); C = torch.empty(b, M, N, device="cuda")
    lda = K if akc else M; ldb = K if bkc else N
    def run():
        rc = lib.ix_gemm_f32(A.data_ptr(), B.data_ptr(), C.data_ptr(), None, M, N, K, akc, bkc, lda, ldb, N, b, 1, M * K, 0, K * N, 0, M * N, 0, 0, 1.0, th, sh, st)
        assert rc == 0
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
for (M, N, K, b, akc, bkc) in [(256, 256, 4000, 1, 0, 0), (256, 256, 250, 16, 0, 0), (250, 256, 256, 16, 1, 0), (250, 256, 256, 16, 1, 1),
                               (4000, 256, 256, 1, 1, 0), (4000, 256, 256, 1, 1, 1), (256, 256, 5776, 1, 0, 0), (4000, 256, 512, 1, 1, 0),
                               (50, 256, 256, 16, 1, 0), (800, 256, 256, 1, 1, 0), (256, 256, 361, 16, 0, 0)]:
    print("M%d N%d K%d b%d kc%d%d: auto %6.1f us   fp32 tile64 %6.1f   fp32 tile128 %6.1f   bf16x6 %6.1f us"
          % (M, N, K, b, akc, bkc, t(M, N, K, b, akc, bkc, 0), t(M, N, K, b, akc, bkc, 64), t(M, N, K, b, akc, bkc, 128), t(M, N, K, b, akc, bkc, 1128)), flush=True)
