"""Median clock budget of a K step of the fp16x3 12-wave kernel from the diagnostic build's stamps (make -C interactron_amd/csrc diag):
producer wave 4 and consumer wave 0 of workgroup 0.   python tools/gemm_kstep_summary.py M N K batch"""
import ctypes, os, statistics, sys
import torch
here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(here, "interactron_amd", "lib", "libix_diag_timing.so"))
P, I, L, F = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float
lib.ix_gemm_f32.argtypes = [P, P, P, P, I, I, I, I, I, L, L, L, I, I, L, L, L, L, L, L, L, F, I, I, P]
lib.ix_gemm_dbg_read.argtypes = [P]
lib.ix_gemm_set_w256.argtypes = [I]
lib.ix_gemm_set_w256(0)
M, N, K, b = [int(x) for x in (sys.argv[1:5] if len(sys.argv) > 4 else (1805, 2048, 256, 16))]
A = torch.randn(b, M * K, device="cuda"); B = torch.randn(b, K * N, device="cuda"); C = torch.empty(b, M, N, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    assert lib.ix_gemm_f32(A.data_ptr(), B.data_ptr(), C.data_ptr(), None, M, N, K, 1, 1, K, K, N, b, 1, M * K, 0, K * N, 0, M * N, 0, 0, 1.0, 1128, 1, stream) == 0
torch.cuda.synchronize()
buf = (ctypes.c_longlong * 2048)()
assert lib.ix_gemm_dbg_read(buf) == 0
seq = {0: [], 1: []}
for role in (0, 1):
    for i in range(511):
        tag, t = buf[role * 1024 + 2 * i], buf[role * 1024 + 2 * i + 1]
        if t == 0:
            break
        seq[role].append((tag, t))
def spans(role, a, b_):
    out = []
    s = seq[role]
    for i in range(len(s) - 1):
        if s[i][0] == a and s[i + 1][0] == b_:
            out.append(s[i + 1][1] - s[i][1])
    return out
def med(x):
    return statistics.median(x) if x else float("nan")
print("shape %d x %d x %d x %d; clocks (s_memtime) medians over the stamped K steps of workgroup 0" % (M, N, K, b))
names = {(1, 0, 1): "producer: wait for the ring stage", (1, 1, 2): "producer: convert + write planes", (1, 2, 3): "producer: advance + request refill",
         (1, 3, 4): "producer: at the barrier", (1, 4, 0): "producer: loop back",
         (0, 13, 15): "consumer: second half of the K tile (after the barrier)", (0, 15, 12): "consumer: first half of the next tile (to the barrier)",
         (0, 12, 13): "consumer: at the barrier", (0, 15, 14): "consumer: last tile -> epilogue start", (0, 14, 10): "consumer: epilogue (C tile out)",
         (0, 10, 12): "consumer: item start -> first barrier"}
for (role, a, b_), n in names.items():
    x = spans(role, a, b_)
    print("  %-62s n=%3d  median %6.0f  min %6.0f  max %6.0f" % (n, len(x), med(x), min(x) if x else 0, max(x) if x else 0))
ks = [seq[1][i + 5][1] - seq[1][i][1] for i in range(0, len(seq[1]) - 5) if seq[1][i][0] == 0 and seq[1][i + 5][0] == 0]
print("  producer K step (tag 0 -> tag 0): median %.0f" % med(ks))
