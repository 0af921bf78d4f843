"""What bounds a K step of the fp16x3 contraction kernel?  Times diagnostic builds (numbers are WRONG by construction) next to
the product library on a few of the step's shapes:  NOCONV = producers write raw bits (no maximum / exponent / conversion),
NOMMA = consumers issue one of the three MFMA terms, NOLOAD = every operand request goes out of range (zeros, no memory
traffic), and combinations.  Build first (on the CPU box):  sh tools/gemm_x3_diag.sh
usage (GPU box): python tools/gemm_x3_diag.py"""
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
# (M, N, K, batch, a_kc, b_kc): encoder FFN of the 16-episode step, a weight gradient, a backbone 1x1, the fusion FFN, a 2-episode shape
shapes = [(28880, 2048, 256, 1, 1, 1), (2048, 256, 28880, 1, 0, 0), (115520, 512, 128, 1, 1, 1), (32960, 2048, 512, 1, 1, 1),
          (3610, 256, 2048, 1, 1, 1), (4096, 4096, 4096, 1, 1, 1)]
print("%-34s" % "M N K b akc bkc" + " ".join("%12s" % n for n in libs))
for (M, N, K, b, akc, bkc) in shapes:
    A = torch.randn(b, M * K, device="cuda"); B = torch.randn(b, K * N, device="cuda"); C = torch.empty(b, M * N, device="cuda")
    lda, ldb = (K if akc else M), (K if bkc else N)
    out = []
    for name, lib in libs.items():
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
        out.append("%9.1f us" % (e0.elapsed_time(e1) * 100))
    print("%-34s" % ("%d %d %d %d %d %d" % (M, N, K, b, akc, bkc)) + " ".join("%12s" % o for o in out), flush=True)
