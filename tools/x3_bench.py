"""Microbenchmark of the two fp32-grade contraction kernels on the step's dominant shapes (one GPU):
bf16x6 (ix_gemm_f32) vs pre-split fp16x3 (ix_gemm_f32_ws), TFLOP/s including the operand-split launches."""
import ctypes
import json
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from interactron_amd import _lib

lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
SHAPES = [  # (M, N, K, batch, a_kc, b_kc)  -- fusion linears (tokens x features), attention-free part of the step
    (32960, 512, 512, 1, 1, 1), (32960, 2048, 512, 1, 1, 1), (32960, 512, 2048, 1, 1, 1),
    (512, 512, 32960, 1, 0, 0), (2048, 512, 32960, 1, 0, 0), (32960, 512, 512, 1, 1, 0),
    (2060, 512, 512, 16, 1, 1), (2060, 2048, 512, 16, 1, 1), (7220, 256, 2304, 1, 1, 1), (28880, 64, 576, 1, 1, 1),
    (1805, 256, 256, 8, 1, 1), (4096, 4096, 4096, 1, 1, 1),
]
out = []
for (M, N, K, b, akc, bkc) in SHAPES:
    A = torch.randn(b, M, K, device="cuda") if akc else torch.randn(b, K, M, device="cuda")
    B = torch.randn(b, N, K, device="cuda") if bkc else torch.randn(b, K, N, device="cuda")
    C = torch.empty(b, M, N, device="cuda")
    lda, ldb = (K if akc else M), (K if bkc else N)
    nws = ctypes.c_size_t(0)
    lib.ix_workspace_bytes_gemm_f32(M, N, K, akc, bkc, lda, ldb, b, 1, M * K, K * N, A.data_ptr(), B.data_ptr(), 0, 0, ctypes.byref(nws))
    ws = torch.empty(max(nws.value, 16), dtype=torch.uint8, device="cuda")
    res = {"shape": [M, N, K, b, akc, bkc], "ws_MB": nws.value / 1e6}
    for name in ("x6", "x3"):
        def run():
            if name == "x6":
                return lib.ix_gemm_f32(A.data_ptr(), B.data_ptr(), C.data_ptr(), None, M, N, K, akc, bkc, lda, ldb, N, b, 1, M * K, 0,
                                       K * N, 0, M * N, 0, 0, 1.0, 0, 0, st)
            return lib.ix_gemm_f32_ws(A.data_ptr(), B.data_ptr(), C.data_ptr(), None, M, N, K, akc, bkc, lda, ldb, N, b, 1, M * K, 0,
                                      K * N, 0, M * N, 0, 0, 1.0, 0, 0, ws.data_ptr(), nws.value, st)
        for _ in range(3):
            assert run() == 0, lib.ix_last_error()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        res[name + "_ms"] = round(ms, 4)
        res[name + "_TFLOPs"] = round(2.0 * M * N * K * b / ms / 1e9, 1)
    print(json.dumps(res))
    out.append(res)
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
