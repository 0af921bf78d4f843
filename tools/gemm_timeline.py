"""Clock-stamp timeline of workgroup 0 of the persistent bf16x6 kernel (diagnostic build).

Build:  cd interactron_amd/csrc && hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DX6_DIAG_TIMING -x hip -c gemm.hip -o /tmp/gemm_diag.o
        && hipcc -shared -fPIC --offload-arch=gfx950 ../lib/obj/{api.cpp,lsap.cpp,elementwise.hip,rowwise.hip,conv_aux.hip,criterion.hip,meta.hip}.o /tmp/gemm_diag.o -o ../lib/libix_diag_timing.so
Producer tags: 0 step start, 1 stage landed, 2 converted + stored, 3 cursors advanced + refill requested, 4 barrier passed.
Consumer tags: 10 item start, 11 item decoded, 12 at barrier, 13 barrier passed, 14 K loop done (epilogue starts)."""
import ctypes, os, sys
import torch
here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(here, "interactron_amd", "lib", "libix_diag_timing.so"))
P, I, L, F = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float
lib.ix_gemm_f32.argtypes = [P, P, P, P, I, I, I, I, I, L, L, L, I, I, L, L, L, L, L, L, L, F, I, I, P]
lib.ix_gemm_set_mode.argtypes = [I]
lib.ix_gemm_dbg_read.argtypes = [P]
lib.ix_gemm_set_mode(int(os.environ.get("IX_MODE", "3")))
M, N, K, b = [int(x) for x in (sys.argv[1:5] if len(sys.argv) > 4 else (2060, 2060, 64, 128))]
A = torch.randn(b, M * K, device="cuda"); B = torch.randn(b, K * N, device="cuda"); C = torch.empty(b, M, N, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    rc = lib.ix_gemm_f32(A.data_ptr(), B.data_ptr(), C.data_ptr(), None, M, N, K, 1, 1, K, K, N, b, 1, M * K, 0, K * N, 0, M * N, 0, 0, 1.0,
                         1128, 1, stream)
    assert rc == 0
torch.cuda.synchronize()
buf = (ctypes.c_longlong * 2048)()
assert lib.ix_gemm_dbg_read(buf) == 0
wall = (ctypes.c_longlong * 1024)()
lib.ix_gemm_dbg_read_wall.argtypes = [P]
assert lib.ix_gemm_dbg_read_wall(wall) == 0
for role in (0, 1):   # shader clock actually delivered while the kernel ran: s_memtime ticks per 10 ns s_memrealtime tick
    n = 0
    while n < 511 and buf[role * 1024 + 2 * n + 1]: n += 1
    if n > 2:
        dt = buf[role * 1024 + 2 * (n - 1) + 1] - buf[role * 1024 + 1]
        dw = wall[role * 512 + n - 1] - wall[role * 512]
        print("%s: %d stamps, %d s_memtime ticks over %d x 10 ns -> %.3f GHz" % ("producer" if role else "consumer", n, dt, dw, dt / max(dw, 1) / 10.0))
ev = []
for role in (0, 1):
    for i in range(511):
        tag, t = buf[role * 1024 + 2 * i], buf[role * 1024 + 2 * i + 1]
        if t == 0: break
        ev.append((t, role, tag))
ev.sort()
t0 = ev[0][0]
last = {0: t0, 1: t0}
print("clock ticks are s_memtime units (100 MHz on gfx9: 1 tick = 10 ns)" )
for t, role, tag in ev[:int(os.environ.get("N_EVENTS", "140"))]:
    print("%8d  %s tag %2d   (+%d)" % (t - t0, "            producer" if role else "consumer", tag, t - last[role]))
    last[role] = t
