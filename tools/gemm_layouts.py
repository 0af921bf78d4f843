"""Layout experiment: same M,N,K for the four operand layouts."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from interactron_amd import _lib
lib = _lib.load()
stream = torch.cuda.current_stream().cuda_stream
def t(M, N, K, b, akc, bkc, th=0, sh=0, reps=10):
    A = torch.randn(b, M * K, device="cuda"); B = torch.randn(b, K * N, device="cuda"); C = torch.empty(b, M * N, device="cuda")
    lda = K if akc else M; ldb = K if bkc else N
    def run():
        assert lib.ix_gemm_f32(A.data_ptr(), B.data_ptr(), C.data_ptr(), None, M, N, K, akc, bkc, lda, ldb, N, b, 1,
                               M * K, 0, K * N, 0, M * N, 0, 0, 1.0, th, sh, stream) == 0
    for _ in range(2): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    return us, 2.0 * M * N * K * b / us / 1e6
for (M, N, K, b) in [(2048, 2048, 2048, 8), (1792, 4608, 512, 8), (512, 4608, 1792, 8), (4096, 4096, 4096, 1), (8192, 8192, 8192, 1)]:
    for th in (128, 64):
        print(M, N, K, b, "tile", th, " ".join("akc%d/bkc%d: %7.1f us %5.1f TF" % ((a, bb) + t(M, N, K, b, a, bb, th, 1)) for a in (1, 0) for bb in (1, 0)), flush=True)
