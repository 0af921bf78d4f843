"""Fixed latency of the contraction kernels: back-to-back dependent launches of tiny / small problems (HIP events)."""
import sys, torch
sys.path.insert(0, ".")
from interactron_amd import _lib
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
def run(M, N, K, b, hint, it=200):
    A = torch.randn(b, M, K, device="cuda"); B = torch.randn(b, N, K, device="cuda"); C = torch.empty(b, M, N, device="cuda")
    f = lambda: lib.ix_gemm_f32(A.data_ptr(), B.data_ptr(), C.data_ptr(), None, M, N, K, 1, 1, K, K, N, b, 1, M * K, 0, N * K, 0, M * N, 0, 0, 1.0, hint, 1, st)
    for _ in range(5): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): f()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
x = torch.zeros(1024, device="cuda")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(5): x.add_(1.0)
e0.record()
for _ in range(200): x.add_(1.0)
e1.record(); e1.synchronize()
print("tiny ATen add_: %.1f us per launch" % (e0.elapsed_time(e1) / 200 * 1e3))
for (M, N, K, b) in [(128, 128, 32, 1), (128, 128, 256, 1), (128, 128, 2048, 1), (250, 256, 256, 2), (1805, 256, 256, 2), (1805, 256, 2048, 2), (3610, 512, 512, 1)]:
    print("%5d x %4d x %5d b%d:  12-wave %.1f us   fp32 64-tile %.1f us   fp32 128-tile %.1f us" % (M, N, K, b, run(M, N, K, b, 1128), run(M, N, K, b, 64), run(M, N, K, b, 128)))
