import ctypes, os, torch
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
for (M, N, ldc) in [(2048, 2048, 2048)]:
    for grid, b in ((8, 4), (256, 128)):
        for waves in (1, 2, 4, 8, 16):
            us, tb = t(M, N, ldc, b, 128, 128, waves << 8, grid)
            print("ldc%d grid %3d, %2d waves per CU storing: %7.1f us  %.2f TB/s = %.1f B/clk/CU" % (ldc, grid, waves, us, tb, tb * 1e12 / grid / 2.2e9), flush=True)
