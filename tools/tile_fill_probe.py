"""Write-pattern probe (diagnostic build, see tools/gemm_timeline.py): how fast can 256 workgroups store C in GEMM-tile
order, with no compute at all?"""
import ctypes, os
import torch
here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(here, "interactron_amd", "lib", "libix_diag_timing.so"))
P, I, L = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
lib.ix_diag_tile_fill.argtypes = [P, I, I, L, I, I, I, I, I, P]
stream = torch.cuda.current_stream().cuda_stream
def t(M, N, ldc, b, bm, bn, order, grid=256, reps=10):
    C = torch.empty(b, M, ldc, device="cuda")
    run = lambda: lib.ix_diag_tile_fill(C.data_ptr(), M, N, ldc, b, bm, bn, order, grid, stream)
    for _ in range(2): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    return us, b * M * N * 4 / us / 1e6
for (M, N, ldc, b) in [(2060, 2060, 2060, 128), (2048, 2048, 2048, 128)]:
    for (bm, bn) in [(128, 128), (128, 64), (64, 256), (32, 512), (16, 1024), (8, 2048)]:
        for order in (0, 1):
            us, tb = t(M, N, ldc, b, bm, bn, order)
            print("M%d N%d ldc%d b%d tile %dx%d order %d: %7.1f us  %.2f TB/s" % (M, N, ldc, b, bm, bn, order, us, tb), flush=True)
print("per-CU store rate against the number of active workgroups (tile 128x128, order 1):")
for (M, N, ldc) in [(2060, 2060, 2060), (2048, 2048, 2048)]:
    for grid in (8, 16, 32, 64, 128, 256):
        b = max(1, grid // 2)
        us, tb = t(M, N, ldc, b, 128, 128, 1, grid)
        print("ldc%d grid %3d: %7.1f us  %.2f TB/s  = %.1f B/clk/CU at 2.2 GHz" % (ldc, grid, us, tb, tb * 1e12 / grid / 2.2e9), flush=True)
