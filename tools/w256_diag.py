"""What bounds a K step of the 256 x 128 fp16x3 kernel (gemm_f32_f16x3_w256_kernel)?  The diagnostic builds of
tools/gemm_x3_diag.sh (wrong numbers by construction: NOCONV / NOMMA / NOLOAD / HALFBAR and combinations) timed on both tile
sizes:  python tools/w256_diag.py   (GPU box; build first on the CPU box: sh tools/gemm_x3_diag.sh)"""
import ctypes
import glob
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from interactron_amd import _lib

here = os.path.dirname(os.path.abspath(__file__))
full = _lib.load()
libs = {"full": full}
for f in sorted(glob.glob(os.path.join(here, "..", "interactron_amd", "lib", "libx3diag_*.so"))):
    l = ctypes.CDLL(f)
    l.ix_gemm_f32.restype = ctypes.c_int
    l.ix_gemm_f32.argtypes = full.ix_gemm_f32.argtypes
    libs[os.path.basename(f)[len("libx3diag_"):-3]] = l
stream = torch.cuda.current_stream().cuda_stream
shapes = [(1805, 2048, 256, 16, 1, 1), (1805, 512, 2048, 16, 1, 0), (2048, 256, 1805, 16, 0, 0), (4096, 4096, 4096, 1, 1, 1)]
print("%-30s %5s" % ("M N K b akc bkc", "tile") + " ".join("%20s" % n for n in libs))
for (M, N, K, b, akc, bkc) in shapes:
    A = torch.randn(b, M * K, device="cuda"); B = torch.randn(b, K * N, device="cuda"); C = torch.empty(b, M * N, device="cuda")
    lda, ldb = (K if akc else M), (K if bkc else N)
    for mode in (0, 2):
        out = []
        for name, lib in libs.items():
            lib.ix_gemm_set_w256(mode)
            def run():
                assert lib.ix_gemm_f32(A.data_ptr(), B.data_ptr(), C.data_ptr(), None, M, N, K, akc, bkc, lda, ldb, N, b, 1, M * K, 0,
                                       K * N, 0, M * N, 0, 0, 1.0, 0, 1, stream) == 0
            for _ in range(3):
                run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                run()
            e1.record(); torch.cuda.synchronize()
            out.append("%17.1f us" % (e0.elapsed_time(e1) * 100))
            lib.ix_gemm_set_w256(1)
        print("%-30s %5s" % ("%d %d %d %d %d %d" % (M, N, K, b, akc, bkc), "256" if mode else "128") + " ".join("%20s" % o for o in out), flush=True)
