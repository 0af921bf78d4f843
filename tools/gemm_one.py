import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from interactron_amd import _lib
lib = _lib.load()
stream = torch.cuda.current_stream().cuda_stream
M, N, K, b, akc, bkc, th = [int(x) for x in sys.argv[1:8]]
A = torch.randn(b, M * K, device="cuda"); B = torch.randn(b, K * N, device="cuda"); C = torch.empty(b, M * N, device="cuda")
lda = K if akc else M; ldb = K if bkc else N
for _ in range(5):
    assert lib.ix_gemm_f32(A.data_ptr(), B.data_ptr(), C.data_ptr(), None, M, N, K, akc, bkc, lda, ldb, N, b, 1,
                           M * K, 0, K * N, 0, M * N, 0, 0, 1.0, th, 1, stream) == 0
torch.cuda.synchronize()
