"""CPU tier: the RCAN graph end to end on the kernel simulator vs the fp32 oracle (small shapes)."""
import rcan_checks as rc


def test_rcan_small(emu_lib):
    assert rc.check_rcan(emu_lib, "cpu", 20, 36, n_resgroups=1, n_resblocks=2) > 60


def test_rcan_pixel_unshuffle_meanshift(emu_lib):
    assert rc.check_rcan(emu_lib, "cpu", 24, 20, n_resgroups=2, n_resblocks=1, n_feats=32, unshuffle=2, mean_shift=True) > 60


def test_rcan_batch2(emu_lib):
    rc.check_rcan(emu_lib, "cpu", 17, 19, n_resgroups=1, n_resblocks=1, n=2)


def test_rcan_pixel_unshuffle_odd_sizes(emu_lib):
    """the lite (pixel-unshuffle) model — the reference's default upscaler — is called on pages and bubble crops of any
    size; odd sides are padded up by reflection and the output cropped back to (2h, 2w)"""
    rc.check_rcan(emu_lib, "cpu", 25, 21, n_resgroups=1, n_resblocks=1, n_feats=32, unshuffle=2)
    rc.check_rcan(emu_lib, "cpu", 16, 19, n_resgroups=1, n_resblocks=1, n_feats=32, unshuffle=2, mean_shift=True)


def test_rcan_plan_cache_is_bounded(emu_lib):
    """one plan per distinct crop size used to pin device memory for ever.  Small images (bubble crops) now share
    masked bucket plans; larger ones keep exact-size plans in an LRU that destroys what it evicts."""
    import torch
    from mangatranslator_amd.core.ml.rcan import RCANUpscaler
    from oracle.rcan_ref import load_ref, make_state_dict
    sd = make_state_dict(n_feats=32, n_resgroups=1, n_resblocks=1, unshuffle=2)
    m = RCANUpscaler(sd, device="cpu", lib=emu_lib)
    ref = load_ref(sd)
    g = torch.Generator().manual_seed(0)
    for (h, w) in [(30, 40), (9, 8), (40, 12), (9, 8)]:          # big first: smaller crops then meet its left-overs in the buffers
        x = torch.rand(1, 3, h, w, generator=g)
        y = m(x)
        assert y.shape == (1, 3, 2 * h, 2 * w)
        assert (y.cpu() - ref(x)).abs().max() < 2e-2, (h, w)                      # the masked canvas run == the image's own zero padding
    assert len(m._buckets) == 1 and len(m._plans) == 0                             # one 64 x 64 canvas served every size
    u8 = m.upscale_u8(torch.zeros(7, 9, 3, dtype=torch.uint8))
    assert tuple(u8.shape) == (14, 18, 3)
    m.BUCKET_MAX = m.BIG_BUCKET_MAX = 0                                            # exact-size plans (what pages use): bounded LRU
    m._plans.capacity = 3
    first = None
    for (h, w) in [(8, 8), (10, 8), (10, 12), (12, 10), (14, 8), (8, 8)]:
        assert m(torch.rand(1, 3, h, w)).shape == (1, 3, 2 * h, 2 * w)
        first = first if first is not None else m._plans[(1, 8, 8)]
    assert len(m._plans) <= 3 and first._h is None             # the oldest plan was destroyed, and (8, 8) was simply rebuilt
