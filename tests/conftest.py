import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: CPU test that takes more than a few seconds")


@pytest.fixture(scope="session")
def golden():
    import torch

    def load(name):
        return torch.load(os.path.join(GOLDEN, name), weights_only=False)

    return load


@pytest.fixture(params=["x3", "x6", "x3wp"])
def kernel_form(request):
    """Runs a test once per form of the contraction kernels: "x3" = two fp16 planes + a sub-block exponent, three MFMAs per
    k-slice (the default since round 3), "x6" = three bf16 planes, six MFMAs (IX_GEMM_KERNEL=x6), "x3wp" (round 4) = x3 with
    EVERY eligible Linear contraction on the activation x weight-planes kernel (csrc/gemm_wp.hip; the product routes only long
    activations there, hipops.WP_MIN_ROWS).  All three carry the parity record of the model-level tests."""
    from interactron_amd import _lib, hipops
    if request.param == "x3wp" and "conv" in request.node.name:
        pytest.skip("convolutions gather their activation operand: never on the weight-planes route")
    lib = _lib.load()
    old = lib.ix_gemm_set_x3(0 if request.param == "x6" else 1)
    old_tr, hipops.FLASH_TR = hipops.FLASH_TR, "bf16" if request.param == "x6" else "f16"   # the attention kernels' twin switch
    old_rows, old_wp = hipops.WP_MIN_ROWS, hipops.GEMM_WP
    if request.param == "x3wp":
        hipops.WP_MIN_ROWS, hipops.GEMM_WP = 128, True
        before = hipops._wp_stats["routed"]
    yield "x3" if request.param == "x3wp" else request.param
    if request.param == "x3wp" and any(k in request.node.name for k in ("g7", "g13", "config")):   # (full-size models)
        assert hipops._wp_stats["routed"] > before, "the weight-planes route was never taken"
    hipops.WP_MIN_ROWS, hipops.GEMM_WP = old_rows, old_wp
    lib.ix_gemm_set_x3(old)
    hipops.FLASH_TR = old_tr


@pytest.fixture(params=["f16", "bf16", "f16_32x32"])
def flash_form(request):
    """Runs a test once per form of the flash kernels' token-contracting products: "f16" = two fp16 planes, three MFMAs per
    k-slice, intermediates scaled into fp16 range in registers (default since round 3; at head dim 64 the 16x16x32 passes of
    csrc/flash16.hip since round 5); "bf16" = three bf16 planes, six MFMAs (IX_FLASH_TR=bf16); "f16_32x32" = the fp16 form on
    the 32x32x16 passes at every head dim (ix_flash_set_m16(0): the family the 16x16x32 passes replaced at head dim 64)."""
    from interactron_amd import hipops
    old, hipops.FLASH_TR = hipops.FLASH_TR, "f16" if request.param == "f16_32x32" else request.param
    old_m16 = hipops.flash_m16(request.param != "f16_32x32")
    yield request.param
    hipops.FLASH_TR = old
    hipops.flash_m16(old_m16)
