"""Clocks / power reported by rocm-smi while one contraction kernel runs back to back (is the GPU power-limited?)."""
import os, subprocess, sys, time
here = os.path.dirname(os.path.abspath(__file__))
if len(sys.argv) > 1 and sys.argv[1] == "load":
    sys.path.insert(0, os.path.dirname(here))
    import torch
    from interactron_amd import _lib
    lib = _lib.load()
    lib.ix_gemm_set_mode(int(sys.argv[2]))   # test hook: 3 = bf16x6 kernel, 0 = exact-fp32 kernel
    th = int(sys.argv[3])
    M = N = K = 4096
    A = torch.randn(M * K, device="cuda"); B = torch.randn(K * N, device="cuda"); C = torch.empty(M, N, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    t0 = time.time(); n = 0
    while time.time() - t0 < 8:
        for _ in range(50):
            lib.ix_gemm_f32(A.data_ptr(), B.data_ptr(), C.data_ptr(), None, M, N, K, 1, 1, K, K, N, 1, 1, 0, 0, 0, 0, 0, 0, 0, 1.0, th, 1, st)
        torch.cuda.synchronize(); n += 50
    dt = time.time() - t0
    print("load mode %s tile %d: %.1f us per 4096^3 product = %.1f TFLOP/s" % (sys.argv[2], th, dt / n * 1e6, 2 * 4096.0 ** 3 / (dt / n) / 1e12), flush=True)
    sys.exit(0)
print(subprocess.run(["rocm-smi", "--showmaxpower", "--showperflevel", "--showclocks"], capture_output=True, text=True).stdout[-1500:])
for mode, th in (("3", 1128), ("0", 128)):
    p = subprocess.Popen([sys.executable, os.path.abspath(__file__), "load", mode, str(th)])
    time.sleep(4.0)
    for _ in range(3):
        out = subprocess.run(["rocm-smi", "-P", "-c", "-t", "-u"], capture_output=True, text=True).stdout
        keep = [l for l in out.splitlines() if any(k in l for k in ("Power", "sclk", "mclk", "Temperature (Sensor junction)", "GPU use"))]
        print(" | ".join(l.split("\t")[-1].strip() if "\t" in l else l.strip() for l in keep))
        time.sleep(0.7)
    p.wait()
