"""Short-K products (attention scores): is the bf16x6 kernel or the C write stream the limit?  Times the kernel on aligned
and ragged row pitches next to a plain fill of the same C tensor."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from interactron_amd import _lib
if os.environ.get("IX_LIB"):   # A/B runs: another build of the library
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), os.environ["IX_LIB"])
lib = _lib.load()
stream = torch.cuda.current_stream().cuda_stream
lib.ix_gemm_set_mode(int(os.environ.get("IX_MODE", "2")))

def timeit(fn, reps=10):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps

def gemm(M, N, K, b, th, ldc=None):
    ldc = ldc or N
    A = torch.randn(b, M * K, device="cuda"); B = torch.randn(b, K * N, device="cuda")
    C = torch.empty(b, M, ldc, device="cuda")
    def run():
        rc = lib.ix_gemm_f32(A.data_ptr(), B.data_ptr(), C.data_ptr(), None, M, N, K, 1, 1, K, K, ldc, b, 1, M * K, 0, K * N, 0,
                             M * ldc, 0, 0, 1.0, th, 1, stream)
        assert rc == 0
    us = timeit(run)
    return us, C

for (M, N, K, b, ldc) in [(2060, 2060, 64, 128, 2060), (2060, 2060, 64, 128, 2080), (2048, 2048, 64, 128, 2048), (2060, 2060, 128, 128, 2060),
                          (2060, 2060, 256, 128, 2060), (361, 361, 32, 640, 361), (384, 384, 32, 640, 384), (1805, 2048, 256, 16, 2048)]:
    us, C = gemm(M, N, K, b, 1128, ldc)
    us32, _ = gemm(M, N, K, b, 128, ldc)
    fill = timeit(lambda: C.fill_(1.0))
    gb = b * M * N * 4 / 1e9
    print("M%d N%d K%d b%d ldc%d: bf16x6 %7.1f us (%5.1f TF/s, C write %.2f TB/s)   fp32 %7.1f us   fill of C %7.1f us (%.2f TB/s)"
          % (M, N, K, b, ldc, us, 2.0 * M * N * K * b / us / 1e6, gb / us * 1e3, us32, fill, b * M * ldc * 4 / 1e9 / fill * 1e3), flush=True)
