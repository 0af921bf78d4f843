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


_RATIOS = {}   # IX_TEST_RECORD=<file>: worst observed / allowed ratio per check kind instead of asserting (tolerance survey)


def _judge(kind, what, value, bound, detail):
    import os
    path = os.environ.get("IX_TEST_RECORD")
    if not path:
        assert value <= bound, (what, kind) + tuple(detail)
        return
    r = value / max(bound, 1e-300)
    if os.environ.get("IX_TEST_RECORD_ALL") and r > 0.2:   # every check above a fifth of its bound (per-tensor tolerance survey)
        with open(path, "a") as f:
            f.write("%s %.4f %s\n" % (kind, r, what))
        return
    if r > _RATIOS.get(kind, (0.0, ""))[0]:
        _RATIOS[kind] = (r, what)
        with open(path, "a") as f:
            f.write("%s %.4f %s\n" % (kind, r, what))


def check_grad(rec, g, rel=1e-3, what="", noise=1e-6, norm64=None, sample64=None):
    """Compare a gradient against a ``grad_record`` entry (None flag, L2 norm, first 8 values, 256 values on an even
    stride over the whole tensor -- a slice routed to the wrong place keeps the norm but not the strided sample).

    ``norm64``: the float64 (exact) norm of the same tensor from tests/golden/golden_train_f64.pt; when given, the
    tolerance is widened by 3x the reference's own float32 rounding error |rec.norm - norm64| on that tensor.
    ``sample64``: the float64 values at the 256 strided positions; when given, the sample-L2 bound is widened by 3x the
    reference's own float32 error on that sample, ||rec.sample - sample64|| (a few second-order backbone tensors have
    elements on ReLU / clip kinks: the reference's float32 is up to 0.7 % off the truth there, and so is any other
    float32 summation order)."""
    if rec is None:
        assert g is None, what + ": reference leaves .grad None"
        return
    assert g is not None, what + ": reference has a gradient here"
    g = g.detach().cpu()
    if g.dim() == 4 and tuple(g.shape) != tuple(rec["shape"]):
        g = g.permute(0, 3, 1, 2).contiguous()   # conv weights are stored [out, kh, kw, in]; the fixtures are [out, in, kh, kw]
    assert tuple(g.shape) == tuple(rec["shape"]), what
    n = float(g.double().norm())
    if max(n, rec["norm"]) < noise:
        return   # mathematically zero gradient (e.g. attention key bias: softmax is shift invariant); both are rounding noise
    ref_noise = 0.0 if norm64 is None else 3.0 * abs(rec["norm"] - norm64)
    if g.numel() == 1:
        # a one-element gradient (the scalar bias of the learned-loss head) is a single, heavily cancelling sum: float32
        # implementations scatter by 0.1-0.7 % around the float64 value on it (the reference itself is 0.56 % off in G13),
        # and run to run it is bimodal (eight runs of config 3: 6.77e-5 x5, 6.93e-5 x3 -- an element of the clipped inner
        # SGD step or a ReLU sitting on its kink flips with the summation order); there are no other elements to average
        # that out of the norm (tools/scalar_spread.py)
        rel = max(rel, 5e-2)
    # Norm within rel.  (Round 2 had to double this: split-K / row-sum atomics made two runs of one binary differ by up to
    # 0.7 % on single tensors of the second-order configurations.  Every multi-workgroup sum is ordered now -- split-K
    # planes, ticketed column sums -- and a step is bit-reproducible: tests/test_parity_gpu.py::test_step_is_bit_reproducible.)
    _judge("norm", what, abs(n - rec["norm"]), rel * max(rec["norm"], 1e-9) + ref_noise + 1e-9, (n, rec["norm"], norm64))
    scale = max(rec["norm"] / max(g.numel(), 1) ** 0.5, 1e-12)
    err = (g.reshape(-1)[:8] - rec["head"]).abs().max().item()
    _judge("head", what, err, 20 * rel * scale + ref_noise + 1e-9, (err, scale))
    if "sample" in rec:
        # 256 values on an even stride over the whole tensor.  Float32 summation-order noise is relative to the tensor's
        # RMS, not to each element, and in the second-order gradients single elements sit on kinks (an element of the
        # clipped inner step, a ReLU at zero) where the HIP and the reference summation orders may fall on different sides:
        # the sample must agree in L2 within 4 x rel, at most 2 % of its elements may be off by more than 20 x rel x RMS,
        # and none by more than 100 x rel x RMS.  A slice routed to the wrong place is off by O(RMS) on EVERY element it covers.
        got = g.reshape(-1)[rec["idx"]].double()
        ref = rec["sample"].double()
        diff = (got - ref).abs()
        bound = 20 * rel * scale + ref_noise + 1e-9
        outliers = int((diff > bound).sum())
        _judge("strided outliers", what, outliers, max(1, len(ref) // 50), (outliers, float(diff.max()), scale))
        _judge("strided worst element", what, float(diff.max()), 5 * bound, (float(diff.max()), scale))
        rn = float(ref.norm())
        samp_noise = 0.0 if sample64 is None else 3.0 * float((ref - sample64.double()).norm())
        _judge("strided sample L2", what, float(diff.norm()), 4 * rel * max(rn, scale * len(ref) ** 0.5) + ref_noise + samp_noise + 1e-9,
               (float(diff.norm()), rn, samp_noise))


def image_key(t):
    """Identifies one image's ground truth (labels + boxes); same function as tests/golden/make_indices.py."""
    import hashlib
    h = hashlib.md5()
    h.update(t["labels"].detach().cpu().to(torch.int64).numpy().tobytes())
    h.update(t["boxes"].detach().cpu().to(torch.float32).numpy().tobytes())
    return h.hexdigest()


class ReferenceMatching:
    """Pins the Hungarian assignment of an end-to-end run to the one the reference made on the same inputs
    (tests/golden/golden_indices.pt, recorded by tests/golden/make_indices.py: per image -- keyed by its ground truth --
    every assignment the reference made for it, one per criterion call that saw the image).

    RNG-free weights make many queries predict nearly the same box, so the cost matrix has near-ties and float32
    rounding noise decides which query is matched.  For every image the HIP path's own optimum is computed as usual;
    the recorded assignment that is cheapest under the HIP cost matrix must cost the same as that optimum to 1e-4
    relative -- i.e. any difference is a tie, not an error -- and is then returned, so that the downstream gradient
    comparison is about arithmetic, not about tie-breaking."""

    # Share of images whose own optimum may differ from the recorded assignment (always by a tie, see above) before the
    # run is declared a failure: measured 0-3 % on the RNG-free weights over the large fixtures; a matcher or cost-kernel
    # bug flips most images (and breaks the tie assertion first).  The small tests (5-18 images) see 0 or 1 proven ties --
    # a fixed number per test now that the step is deterministic, printed on exit -- hence an absolute slack of 1.
    MAX_FLIP_SHARE = 0.10
    FLIP_SLACK = 1

    def __init__(self, recorded, max_flip_share=None, ordered=False, check_ties=True, tie_tol=1e-4):
        """ordered: the k-th time an image is presented it gets the k-th assignment recorded for it (a run that replays another
        run of THIS code call by call -- the data-parallel tests), instead of the cheapest recorded one (the reference's
        recordings, whose criterion calls need not line up with ours).  With exact ties between two recorded assignments of one
        image the cheapest-candidate rule may pick the other call's, and the two runs then differ by a tie-break, not by
        arithmetic."""
        self.recorded = recorded
        self.flips = 0
        self.calls = 0
        self.ordered = ordered
        self.check_ties = check_ties   # False: the recorded assignment is taken unchecked (fp8 scores move the optimum itself)
        self.tie_tol = tie_tol         # what counts as "costs the same" (16-bit activations: the cost matrix itself carries 1e-3 of noise)
        self.seen = {}
        self.max_flip_share = self.MAX_FLIP_SHARE if max_flip_share is None else max_flip_share

    def __enter__(self):
        from interactron_amd import criterion as cr
        from interactron_amd import hipops as ops
        self._cls = cr.HungarianMatcher
        self._orig = cr.HungarianMatcher.assign
        outer = self

        @torch.no_grad()
        def assign(matcher, costs, targets):
            out = []
            for c, t in zip(costs, targets):
                outer.calls += 1
                cands = outer.recorded.get(image_key(t))
                assert cands, "matcher saw an image without a recorded reference assignment"
                if c.shape[1] == 0:
                    out.append(cands[0])
                    continue
                r, col = ops.lsap(c)
                own = float(c[r, col].double().sum())
                if outer.ordered:
                    k = outer.seen.get(image_key(t), 0)
                    outer.seen[image_key(t)] = k + 1
                    assert k < len(cands), "image presented more often than it was recorded"
                    cands = [cands[k]]
                cost, (rr, rc) = min(((float(c[a, b].double().sum()), (a, b)) for a, b in cands), key=lambda x: x[0])
                assert not outer.check_ties or abs(cost - own) <= outer.tie_tol * max(1.0, abs(own)), \
                    "assignment differs from the reference by more than a tie: %.7f vs %.7f" % (own, cost)
                if not (torch.equal(r, rr) and torch.equal(col, rc)):
                    outer.flips += 1
                out.append((rr, rc))
            return out

        cr.HungarianMatcher.assign = assign
        return self

    def __exit__(self, *exc):
        self._cls.assign = self._orig
        print("ReferenceMatching: %d of %d images matched differently from the reference (ties)" % (self.flips, self.calls))
        if exc[0] is None:
            assert self.flips <= self.max_flip_share * max(self.calls, 1) + self.FLIP_SLACK, \
                "HIP matcher disagrees with the reference's assignment on %d of %d images" % (self.flips, self.calls)
