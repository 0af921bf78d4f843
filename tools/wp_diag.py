"""where does the weight-planes kernel lose accuracy?  (diagnostic)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from interactron_amd import _lib
lib = _lib.load()
stream = torch.cuda.current_stream().cuda_stream
def rnd(*shape, seed=0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))
def run(M, N, K, a, w):
    a, w = a.cuda().contiguous(), w.cuda().contiguous()
    pb, ub = ctypes.c_size_t(), ctypes.c_size_t()
    lib.ix_wp_planes_bytes(N, K, 1, ctypes.byref(pb), ctypes.byref(ub))
    planes = torch.empty(pb.value, dtype=torch.uint8, device="cuda"); us = torch.empty(ub.value // 4, device="cuda")
    assert lib.ix_wp_split_f32(w.data_ptr(), K, N * K, N, K, 1, 1, planes.data_ptr(), us.data_ptr(), stream) == 0
    C = torch.empty(M, N, device="cuda"); C0 = torch.empty(M, N, device="cuda")
    assert lib.ix_gemm_wp_f32(a.data_ptr(), K, M * K, 0, planes.data_ptr(), us.data_ptr(), 1, C.data_ptr(), N, M * N, 0, None, 0, M, N, K, 1, 1, 1.0, stream) == 0
    assert lib.ix_gemm_f32(a.data_ptr(), w.data_ptr(), C0.data_ptr(), None, M, N, K, 1, 1, K, K, N, 1, 1, 0, 0, 0, 0, 0, 0, 0, 1.0, 1128, 1, stream) == 0
    ref = a.double() @ w.double().t(); scale = a.double().abs() @ w.double().abs().t() + 1e-300
    e = ((C.double() - ref).abs() / scale); e0 = ((C0.double() - ref).abs() / scale)
    i = int(e.argmax()); r, c = i // N, i % N
    return float(e.max()), float(e0.max()), (r, c)
M, N, K = 1805, 512, 96
a, w = rnd(M, K, seed=1), rnd(N, K, seed=3)
sa, sw = (2.0 * rnd(M, 1, seed=2)).exp(), (2.0 * rnd(N, 1, seed=5)).exp()
for name, aa, ww in (("plain", a, w), ("a rows", a * sa, w), ("w rows", a, w * sw), ("both", a * sa, w * sw)):
    print(name, run(M, N, K, aa, ww))
for K2 in (32, 64, 256, 1024):
    a, w = rnd(M, K2, seed=1), rnd(N, K2, seed=3)
    print("K", K2, "w rows", run(M, N, K2, a, w * sw), "a rows", run(M, N, K2, a * sa, w))
print("---- batched case of the test")
M, N, K, bo = 1805, 512, 96, 3
a = rnd(bo, 1, M, K, seed=1) * (2.0 * rnd(bo, 1, M, 1, seed=2)).exp()
w = rnd(bo, N, K, seed=3) * (2.0 * rnd(bo, N, 1, seed=5)).exp()
a, w = a.cuda(), w.cuda()
pb, ub = ctypes.c_size_t(), ctypes.c_size_t()
lib.ix_wp_planes_bytes(N, K, bo, ctypes.byref(pb), ctypes.byref(ub))
planes = torch.empty(pb.value, dtype=torch.uint8, device="cuda"); us = torch.empty(ub.value // 4, device="cuda")
assert lib.ix_wp_split_f32(w.data_ptr(), K, N * K, N, K, 1, bo, planes.data_ptr(), us.data_ptr(), stream) == 0
C = torch.empty(bo, M, N, device="cuda"); C0 = torch.empty(bo, M, N, device="cuda")
assert lib.ix_gemm_wp_f32(a.data_ptr(), K, M * K, M * K, planes.data_ptr(), us.data_ptr(), 0, C.data_ptr(), N, M * N, M * N, None, 0, M, N, K, bo, 1, 1.0, stream) == 0
assert lib.ix_gemm_f32(a.data_ptr(), w.data_ptr(), C0.data_ptr(), None, M, N, K, 1, 1, K, K, N, bo, 1, M * K, 0, N * K, 0, M * N, 0, 0, 1.0, 1128, 1, stream) == 0
ref = a[:, 0].double() @ w.double().transpose(1, 2); scale = a[:, 0].double().abs() @ w.double().abs().transpose(1, 2) + 1e-300
for s in range(bo):
    e = ((C[s].double() - ref[s]).abs() / scale[s]); e0 = ((C0[s].double() - ref[s]).abs() / scale[s])
    i = int(e.argmax()); r, c = i // N, i % N
    print("slice", s, "wp", float(e.max()), "old", float(e0.max()), "at", (r, c), "us", us.view(bo, -1)[s, c // 32].item(),
          "row scale a", float(a[s, 0, r].abs().max()), "w", float(w[s, c].abs().max()), "w block max", float(w[s, c // 32 * 32:c // 32 * 32 + 32].abs().max()))
