"""GPU tier: RCAN on MI355X through the C ABI vs the fp32 CPU oracle."""
import pytest

import rcan_checks as rc
from parity_log import record

pytestmark = pytest.mark.gpu


def test_rcan_small(hip_lib):
    rc.check_rcan(hip_lib, "cuda:0", 20, 36, n_resgroups=1, n_resblocks=2)
    rc.check_rcan(hip_lib, "cuda:0", 24, 20, n_resgroups=2, n_resblocks=1, n_feats=32, unshuffle=2, mean_shift=True)


def test_rcan_full_depth(hip_lib):
    """Full-depth RCAN (10 groups x 20 RCAB, 64 feats = 401 3x3 convs) on a 192x128 crop:
    the fp16 trunk must hold PSNR >= 40 dB through the whole residual stack."""
    p = rc.check_rcan(hip_lib, "cuda:0", 192, 128, n_resgroups=10, n_resblocks=20)
    print(f"full-depth RCAN PSNR vs fp32 oracle: {p:.1f} dB")
    record("rcan.full_depth_10x20.192x128", psnr_db=p)


def test_rcan_odd_sizes(hip_lib):
    """pages and bubble crops come in any size: odd sides with the pixel-unshuffle (lite) variant are padded up and cropped back"""
    rc.check_rcan(hip_lib, "cuda:0", 25, 21, n_resgroups=1, n_resblocks=1, n_feats=32, unshuffle=2)
    rc.check_rcan(hip_lib, "cuda:0", 33, 20, n_resgroups=1, n_resblocks=2, unshuffle=2)
    rc.check_rcan(hip_lib, "cuda:0", 31, 47, n_resgroups=1, n_resblocks=1)


def test_rcan_page_2048x3072(hip_lib):
    """BASELINE config 5 page size through a shallow trunk (1 group x 2 RCAB, the CPU oracle pass at 6.3 MP takes ~30 s): tiling, halo and
    pixel-shuffle addressing at the large geometry"""
    p = rc.check_rcan(hip_lib, "cuda:0", 3072, 2048, n_resgroups=1, n_resblocks=2)
    print(f"RCAN 2048x3072 (shallow) PSNR vs fp32 oracle: {p:.1f} dB")
    record("rcan.shallow_1x2.2048x3072", psnr_db=p)
