"""Timing of diagnostic builds of the bf16x6 kernel (tools/libdiag_*.so, numbers are WRONG by construction):
NOCONV = producers skip the split arithmetic, NOMMA = consumers issue 1 of 6 MFMA terms.  Build them first:
    cd interactron_amd/csrc && for v in NOCONV NOMMA; do hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -x hip \
        -DX6_DIAG_$v -shared gemm.hip api.cpp -o ../../tools/libdiag_$v.so; done
Result at r1d (4096^3, one box): full 820 us, NOCONV 620, NOMMA 529 -> neither side dominates; the common floor is the
per-CU operand load path (32 KB per K step per CU = ~15 B/clk/CU from L2)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
here = os.path.dirname(os.path.abspath(__file__))
from interactron_amd import _lib
libs = {"full": _lib.load()}
for v in ("NOCONV", "NOMMA"):
    l = ctypes.CDLL(os.path.join(here, "libdiag_%s.so" % v))
    l.ix_gemm_f32.restype = ctypes.c_int
    l.ix_gemm_f32.argtypes = libs["full"].ix_gemm_f32.argtypes
    libs[v] = l
stream = torch.cuda.current_stream().cuda_stream
for (M, N, K, b) in [(4096, 4096, 4096, 1), (1804, 2048, 256, 16), (2060, 64, 2060, 128)]:
    A = torch.randn(b, M * K, device="cuda"); B = torch.randn(b, K * N, device="cuda"); C = torch.empty(b, M * N, device="cuda")
    out = []
    for name, lib in libs.items():
        def run():
            assert lib.ix_gemm_f32(A.data_ptr(), B.data_ptr(), C.data_ptr(), None, M, N, K, 1, 1, K, K, N, b, 1, M * K, 0, K * N, 0, M * N, 0, 0, 1.0, 1128, 1, stream) == 0
        for _ in range(3): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        out.append("%s %.1f us" % (name, e0.elapsed_time(e1) * 100))
    print(M, N, K, b, " | ".join(out), flush=True)
