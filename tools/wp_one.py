"""One contraction shape, N launches of the 12-wave kernel and N of the weight-planes kernel: the subject of a rocprofv3 --pmc
pass (SQ counters of both kernels on identical work).  python3 tools/wp_one.py [M N K batch]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from interactron_amd import _lib
lib = _lib.load()
stream = torch.cuda.current_stream().cuda_stream
M, N, K, b = [int(x) for x in (sys.argv[1:5] if len(sys.argv) > 4 else (1805, 2048, 256, 16))]
g = torch.Generator().manual_seed(1)
A = torch.randn(b, M, K, generator=g).cuda(); W = (torch.randn(b, N, K, generator=g) * 0.05).cuda()
C0, C1 = torch.empty(b, M, N, device="cuda"), torch.empty(b, M, N, device="cuda")
pb, ub = ctypes.c_size_t(), ctypes.c_size_t()
lib.ix_wp_planes_bytes(N, K, b, ctypes.byref(pb), ctypes.byref(ub))
planes = torch.empty(pb.value, dtype=torch.uint8, device="cuda"); us = torch.empty(ub.value // 4, device="cuda")
lib.ix_wp_split_f32(W.data_ptr(), K, N * K, N, K, 1, b, planes.data_ptr(), us.data_ptr(), stream)
ws_n = ctypes.c_size_t()
lib.ix_workspace_bytes_gemm_f32(M, N, K, 1, 1, K, K, b, 1, M * K, N * K, A.data_ptr(), W.data_ptr(), 0, 0, ctypes.byref(ws_n))
ws = torch.zeros(max(ws_n.value, 65536), dtype=torch.uint8, device="cuda")
for _ in range(5):
    assert lib.ix_gemm_f32_ws(A.data_ptr(), W.data_ptr(), C0.data_ptr(), None, M, N, K, 1, 1, K, K, N, b, 1, M * K, 0, N * K, 0, M * N, 0, 0,
                              1.0, 0, 0, ws.data_ptr(), ws.numel(), stream) == 0
    assert lib.ix_gemm_wp_f32(A.data_ptr(), K, M * K, 0, planes.data_ptr(), us.data_ptr(), 0, C1.data_ptr(), N, M * N, 0, None, 0, M, N, K,
                              b, 1, 1.0, stream) == 0
torch.cuda.synchronize()
print("max |difference| between the two kernels:", float((C0 - C1).abs().max()))
