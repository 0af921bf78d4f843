"""Times the fused attention-probability kernels at the fusion ([128, 2060, 2060]) and detector ([640, 361, 364]) sizes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from interactron_amd import _lib
if os.environ.get("IX_LIB"):
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), os.environ["IX_LIB"])
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
def timeit(fn, reps=5):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
for (b, L, S) in [(128, 2060, 2060), (640, 361, 361), (640, 50, 361)]:
    Sp = (S + 31) // 32 * 32 if os.environ.get("ALIGN", "1") == "1" else (S + 3) // 4 * 4
    rows = b * L
    t = [torch.randn(rows, Sp, device="cuda") for _ in range(7)]
    gb = rows * Sp * 4 / 1e9
    f = timeit(lambda: lib.ix_attn_prob_fwd_f32(t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), rows, S, Sp, None, 1, 0, 0.1, 123, st))
    bw = timeit(lambda: lib.ix_attn_prob_bwd_f32(t[1].data_ptr(), t[3].data_ptr(), t[4].data_ptr(), rows, S, Sp, 0.1, 123, st))
    bb = timeit(lambda: lib.ix_attn_prob_bwd_bwd_f32(t[0].data_ptr(), t[2].data_ptr(), t[1].data_ptr(), t[3].data_ptr(), t[4].data_ptr(), t[5].data_ptr(), t[6].data_ptr(), rows, S, Sp, 0.1, 123, st))
    print("[%d,%d,%d] %.2f GB/tensor: fwd %7.1f us (%.2f TB/s of 3 passes)  bwd %7.1f us (%.2f TB/s of 3)  bwd_bwd %7.1f us (%.2f TB/s of 7)"
          % (b, L, S, gb, f, 3 * gb / f * 1e3, bw, 3 * gb / bw * 1e3, bb, 7 * gb / bb * 1e3), flush=True)
