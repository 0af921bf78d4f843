"""Clock stamps of every workgroup of one gemm_wp_kernel launch (ix_gemm_wp_debug flag 16): how long a tile spends loading its
first stage, in its K loop, issuing its C stores and waiting for their acknowledgement, and how the workgroups of one CU
overlap.  100 MHz wall clock (s_memrealtime): 10 ns resolution."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from interactron_amd import _lib
lib = _lib.load()
stream = torch.cuda.current_stream().cuda_stream
M, N, K, b = [int(x) for x in (sys.argv[1:5] if len(sys.argv) > 4 else (1805, 2048, 256, 16))]
g = torch.Generator().manual_seed(1)
A = torch.randn(b, M, K, generator=g).cuda(); W = (torch.randn(b, N, K, generator=g) * 0.05).cuda(); C = torch.empty(b, M, N, device="cuda")
pb, ub = ctypes.c_size_t(), ctypes.c_size_t()
lib.ix_wp_planes_bytes(N, K, b, ctypes.byref(pb), ctypes.byref(ub))
planes = torch.empty(pb.value, dtype=torch.uint8, device="cuda"); us = torch.empty(ub.value // 4, device="cuda")
lib.ix_wp_split_f32(W.data_ptr(), K, N * K, N, K, 1, b, planes.data_ptr(), us.data_ptr(), stream)
tiles = ((M + 127) // 128) * ((N + 127) // 128) * b
buf = torch.zeros(tiles * 8, dtype=torch.int64, device="cuda")
def run():
    assert lib.ix_gemm_wp_f32(A.data_ptr(), K, M * K, 0, planes.data_ptr(), us.data_ptr(), 0, C.data_ptr(), N, M * N, 0, None, 0, M, N, K, b, 1, 1.0, stream) == 0
run(); torch.cuda.synchronize()
lib.ix_gemm_wp_debug_stamps(buf.data_ptr()); lib.ix_gemm_wp_debug(16)
run(); torch.cuda.synchronize()
lib.ix_gemm_wp_debug(0)
t = buf.cpu().numpy().reshape(tiles, 8).astype(np.int64)
t0 = t[:, 0].min()
us_ = lambda x: x / 100.0     # 100 MHz -> microseconds
print("launch: %d workgroups, first start -> last end %.1f us" % (tiles, us_(t[:, 4].max() - t0)))
for name, a, c in (("first stage (DMA latency)", 0, 1), ("K loop (remaining stages)", 1, 2), ("C stores issued", 2, 3), ("C stores acknowledged", 3, 4), ("whole tile", 0, 4)):
    d = us_(t[:, c] - t[:, a])
    print("%-28s median %7.2f us   p10 %7.2f   p90 %7.2f" % (name, np.median(d), np.percentile(d, 10), np.percentile(d, 90)))
hw, xcc = t[:, 5], t[:, 6] & 0xf
cu = (xcc << 20) | (((hw >> 13) & 7) << 12) | (((hw >> 12) & 1) << 8) | ((hw >> 8) & 0xf)     # XCC, SE, SH, CU
ids = np.unique(cu)
print("distinct CUs seen:", len(ids))
conc = []
for c in ids[:32]:
    rows = t[cu == c]
    ev = sorted([(r[0], 1) for r in rows] + [(r[4], -1) for r in rows])
    cur = area = 0; last = ev[0][0]
    for x, d in ev:
        area += cur * (x - last); last = x; cur += d
    conc.append(area / max(1, ev[-1][0] - ev[0][0]))
print("mean concurrent workgroups per CU (first 32 CUs): %.2f" % float(np.mean(conc)))
c = ids[0]
rows = t[cu == c]; rows = rows[rows[:, 0].argsort()][:12]
print("one CU, first 12 workgroups (us from launch): start, stage0, loop end, stores issued, stores acked")
for r in rows:
    print("   " + "  ".join("%7.2f" % us_(r[k] - t0) for k in range(5)))
