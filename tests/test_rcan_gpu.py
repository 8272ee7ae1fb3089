"""GPU tier: RCAN on MI355X through the C ABI vs the fp32 CPU oracle."""
import pytest

import rcan_checks as rc

pytestmark = pytest.mark.gpu


def test_rcan_small(hip_lib):
    rc.check_rcan(hip_lib, "cuda:0", 20, 36, n_resgroups=1, n_resblocks=2)
    rc.check_rcan(hip_lib, "cuda:0", 24, 20, n_resgroups=2, n_resblocks=1, n_feats=32, unshuffle=2, mean_shift=True)


def test_rcan_full_depth(hip_lib):
    """Full-depth RCAN (10 groups x 20 RCAB, 64 feats = 401 3x3 convs) on a 192x128 crop:
    the fp16 trunk must hold PSNR >= 40 dB through the whole residual stack."""
    p = rc.check_rcan(hip_lib, "cuda:0", 192, 128, n_resgroups=10, n_resblocks=20)
    print(f"full-depth RCAN PSNR vs fp32 oracle: {p:.1f} dB")
