"""bf16x6 kernel vs float64 reference and vs the fp32-MFMA kernel: accuracy for all layouts / ragged sizes / split-K,
then speed on the layout experiment shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from interactron_amd import _lib
if os.environ.get("IX_LIB"):   # A/B runs: another build of the library
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), os.environ["IX_LIB"])
lib = _lib.load()
stream = torch.cuda.current_stream().cuda_stream
torch.manual_seed(0)
lib.ix_gemm_set_mode(int(os.environ.get("IX_MODE", "1")))

def run(A, B, M, N, K, b, akc, bkc, th, sh, bias=None):
    C = torch.empty(b, M, N, device="cuda")
    lda = K if akc else M; ldb = K if bkc else N
    rc = lib.ix_gemm_f32(A.data_ptr(), B.data_ptr(), C.data_ptr(), bias.data_ptr() if bias is not None else None, M, N, K,
                         akc, bkc, lda, ldb, N, b, 1, M * K, 0, K * N, 0, M * N, 0, 0, 1.0, th, sh, stream)
    assert rc == 0, lib.ix_last_error()
    return C

worst = 0.0
for (M, N, K, b) in [] if os.environ.get('IX_SKIP_ACC') else [(130, 77, 256, 2), (128, 128, 32, 2), (200, 136, 64, 2), (200, 136, 96, 2), (200, 136, 100, 2), (300, 260, 1805, 3), (1805, 512, 256, 2), (257, 129, 4099, 1), (2060, 64, 2060, 3), (361, 32, 361, 5), (300, 50, 777, 2), (100, 20, 300, 2)]:
    for akc in (1, 0):
        for bkc in (1, 0):
            for sh in (1, 3):
                a = torch.randn(b, M, K, device="cuda") * torch.rand(b, M, 1, device="cuda").exp()
                bm = torch.randn(b, K, N, device="cuda")
                bias = torch.randn(N, device="cuda")
                ref = (a.double() @ bm.double() + bias.double()).float()
                A = a if akc else a.transpose(1, 2).contiguous()
                B = bm.transpose(1, 2).contiguous() if bkc else bm
                scale = (a.double().abs() @ bm.double().abs()).float() + 1e-30
                e6 = ((run(A, B, M, N, K, b, akc, bkc, 1128, sh, bias) - ref).abs() / scale).max().item()
                e32 = ((run(A, B, M, N, K, b, akc, bkc, 128, sh, bias) - ref).abs() / scale).max().item()
                worst = max(worst, e6)
                flag = "" if e6 < max(5e-7, 1.5 * e32) else "   <-- BAD"
                print("M%d N%d K%d b%d akc%d bkc%d split%d: bf16x6 err %.2e   fp32-mfma err %.2e%s" % (M, N, K, b, akc, bkc, sh, e6, e32, flag), flush=True)
print("worst bf16x6 error relative to sum|a||b|: %.3e" % worst)

def t(M, N, K, b, akc, bkc, th, reps=10):
    A = torch.randn(b, M * K, device="cuda"); B = torch.randn(b, K * N, device="cuda")
    for _ in range(2): run(A, B, M, N, K, b, akc, bkc, th, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): run(A, B, M, N, K, b, akc, bkc, th, 1)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    return us, 2.0 * M * N * K * b / us / 1e6
for (M, N, K, b) in [(2060, 64, 2060, 128), (1805, 2048, 256, 16), (1804, 2048, 256, 16), (4096, 4096, 4096, 1), (2060, 2060, 64, 128), (364, 364, 32, 640), (256, 256, 28880, 1)]:
    for th in (128, 1128):
        print(M, N, K, b, "tile", th, " ".join("akc%d/bkc%d: %7.1f us %5.1f TF" % ((a, bb) + t(M, N, K, b, a, bb, th)) for a in (1, 0) for bb in (1, 0)), flush=True)
