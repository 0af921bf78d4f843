"""Host cost per launch of the Python/ctypes layer (tiny shapes, so the GPU is never the limit)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from interactron_amd import hipops as ops

def bench(name, fn, n=3000):
    for _ in range(200): fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): fn()
    dt = time.perf_counter() - t
    torch.cuda.synchronize()
    print("%-44s %6.2f us per call" % (name, dt / n * 1e6), flush=True)

x = torch.randn(64, 64, device="cuda"); w = torch.randn(64, 64, device="cuda"); b = torch.randn(64, device="cuda")
xg = x.clone().requires_grad_(True); wg = w.clone().requires_grad_(True); bg = b.clone().requires_grad_(True)
lib = ops._L(); st = ops._stream()
out = torch.empty(64, 64, device="cuda")
bench("ctypes ix_gemm_f32 alone", lambda: lib.ix_gemm_f32(x.data_ptr(), w.data_ptr(), out.data_ptr(), None, 64, 64, 64, 1, 1, 64, 64, 64, 1, 1, 0, 0, 0, 0, 0, 0, 0, 1.0, 0, 0, st))
bench("ctypes ix_axpby_f32 alone", lambda: lib.ix_axpby_f32(x.data_ptr(), w.data_ptr(), out.data_ptr(), 4096, 1.0, 1.0, st))
bench("ops._stream()", ops._stream)
bench("torch.empty(64,64)", lambda: torch.empty(64, 64, device="cuda"))
bench("torch add (aten)", lambda: torch.add(x, w))
bench("ops.linear no grad", lambda: ops.linear(x, w, b))
bench("ops.linear with grad (fwd only)", lambda: ops.linear(xg, wg, bg))
def fb():
    y = ops.linear(xg, wg, bg)
    torch.autograd.grad(y, [xg, wg, bg], out)
bench("ops.linear fwd + grad (3 grads)", fb)
def fb2():
    y = ops.linear(xg, wg, bg)
    g = torch.autograd.grad(y, [xg, wg, bg], out, create_graph=True)
    torch.autograd.grad(g[0].sum() + g[1].sum(), [xg, wg])
bench("ops.linear fwd + grad(create_graph) + grad", fb2, 1000)
bench("ops.Relu fwd no grad", lambda: ops.Relu.apply(x))
bench("ops.add (Axpby)", lambda: ops.add(xg, wg))
bench("ops.layer_norm with grad", lambda: ops.layer_norm(xg, b, b))
