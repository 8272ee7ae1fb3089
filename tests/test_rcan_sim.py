"""CPU tier: the RCAN graph end to end on the kernel simulator vs the fp32 oracle (small shapes)."""
import rcan_checks as rc


def test_rcan_small(emu_lib):
    assert rc.check_rcan(emu_lib, "cpu", 20, 36, n_resgroups=1, n_resblocks=2) > 60


def test_rcan_pixel_unshuffle_meanshift(emu_lib):
    assert rc.check_rcan(emu_lib, "cpu", 24, 20, n_resgroups=2, n_resblocks=1, n_feats=32, unshuffle=2, mean_shift=True) > 60


def test_rcan_batch2(emu_lib):
    rc.check_rcan(emu_lib, "cpu", 17, 19, n_resgroups=1, n_resblocks=1, n=2)
