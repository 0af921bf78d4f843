"""Shared comparison helpers for the golden-fixture tests."""
import torch


def check_record(rec, t, atol=1e-5, rtol=1e-4, what=""):
    """Compare tensor ``t`` against a ``summarize`` record written by tests/golden/make_golden.py."""
    assert rec is not None, what
    t = t.detach().cpu()
    assert tuple(t.shape) == tuple(rec["shape"]), (what, tuple(t.shape), rec["shape"])
    if "full" in rec:
        torch.testing.assert_close(t.to(rec["full"].dtype), rec["full"], atol=atol, rtol=rtol, msg=lambda m: what + ": " + m)
        return
    flat = t.reshape(-1)
    torch.testing.assert_close(flat[rec["idx"]], rec["sample"], atol=atol, rtol=rtol, msg=lambda m: what + ": " + m)
    n = float(flat.double().norm())
    assert abs(n - rec["norm"]) <= rtol * 10 * max(rec["norm"], 1e-12) + atol, (what, n, rec["norm"])


def check_grad(rec, g, rel=1e-3, what="", noise=1e-6, norm64=None):
    """Compare a gradient against a ``grad_record`` entry (None flag, L2 norm, first 8 values).

    ``norm64``: the float64 (exact) norm of the same tensor from tests/golden/golden_train_f64.pt; when given, the
    tolerance is widened by 3x the reference's own float32 rounding error |rec.norm - norm64| on that tensor."""
    if rec is None:
        assert g is None, what + ": reference leaves .grad None"
        return
    assert g is not None, what + ": reference has a gradient here"
    g = g.detach().cpu()
    assert tuple(g.shape) == tuple(rec["shape"]), what
    n = float(g.double().norm())
    if max(n, rec["norm"]) < noise:
        return   # mathematically zero gradient (e.g. attention key bias: softmax is shift invariant); both are rounding noise
    ref_noise = 0.0 if norm64 is None else 3.0 * abs(rec["norm"] - norm64)
    assert abs(n - rec["norm"]) <= rel * max(rec["norm"], 1e-9) + ref_noise + 1e-9, (what, n, rec["norm"], norm64)
    scale = max(rec["norm"] / max(g.numel(), 1) ** 0.5, 1e-12)
    err = (g.reshape(-1)[:8] - rec["head"]).abs().max().item()
    assert err <= 20 * rel * scale + ref_noise + 1e-9, (what, err, scale)
