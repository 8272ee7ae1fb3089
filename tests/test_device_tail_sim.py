"""CPU tier: the page-tail kernels on the simulator against Pillow / numpy (see device_tail_checks.py)."""
import numpy as np
from PIL import Image

import device_tail_checks as dc
from mangatranslator_amd.core.image.device_tail import pil_resample_tables


def test_tap_tables_reproduce_pillow_on_an_impulse():
    """the taps themselves: resizing a 1-pixel-high impulse row with Pillow spreads 255 * tap / 2^22 (rounded) around each output"""
    for n_in, n_out, filt, pil_f in ((40, 17, "lanczos", Image.Resampling.LANCZOS), (23, 61, "lanczos", Image.Resampling.LANCZOS), (50, 20, "bilinear", Image.Resampling.BILINEAR)):
        bounds, taps, ksize = pil_resample_tables(n_in, n_out, filt)
        for pos in (0, n_in // 3, n_in - 1):
            row = np.zeros((1, n_in), np.uint8)
            row[0, pos] = 255
            ref = np.asarray(Image.fromarray(row, "L").resize((n_out, 1), pil_f))[0]
            mine = np.zeros(n_out, np.int64)
            for o in range(n_out):
                lo, n = bounds[o]
                if lo <= pos < lo + n:
                    mine[o] = np.clip(((1 << 21) + 255 * int(taps[o, pos - lo])) >> 22, 0, 255)
            assert np.array_equal(mine, ref), (n_in, n_out, filt, pos)


def test_tap_tables_over_many_geometries():
    """the tables alone (no kernel): a numpy evaluation of one resampling pass — int32 accumulation from 2^21, >> 22, clip — over 120
    random (in, out) size pairs and the three filters equals Pillow's horizontal and vertical passes on a random strip; catches the
    rounding of the window bounds and of the normalised taps at sizes the kernel tests do not visit (1-pixel axes, 3x up, 7.3x down)"""
    rng = np.random.default_rng(11)
    filters = (("lanczos", Image.Resampling.LANCZOS), ("bilinear", Image.Resampling.BILINEAR), ("bicubic", Image.Resampling.BICUBIC))
    pairs = [(1, 1), (1, 7), (7, 1), (2, 3), (3, 2), (100, 300), (300, 41), (41, 300), (1023, 1024), (1024, 1023), (640, 88)]
    pairs += [(int(a), int(b)) for a, b in zip(rng.integers(1, 700, 109), rng.integers(1, 700, 109))]
    for i, (n_in, n_out) in enumerate(pairs):
        name, pil_f = filters[i % 3]
        bounds, taps, ksize = pil_resample_tables(n_in, n_out, name)
        strip = rng.integers(0, 256, (3, n_in), dtype=np.uint8)
        strip[1] = np.where(np.arange(n_in) % 5 < 2, 255, 0)                 # hard edges: ringing into the clip
        idx = np.clip(bounds[:, :1] + np.arange(ksize)[None, :], 0, n_in - 1)                       # [out, ksize]
        valid = np.arange(ksize)[None, :] < bounds[:, 1:2]
        acc = (strip[:, idx].astype(np.int64) * np.where(valid, taps, 0)[None]).sum(-1) + (1 << 21)
        mine = np.clip(acc >> 22, 0, 255).astype(np.uint8)
        ref_h = np.asarray(Image.fromarray(strip, "L").resize((n_out, 3), pil_f))
        assert np.array_equal(mine, ref_h), (n_in, n_out, name, "horizontal")
        ref_v = np.asarray(Image.fromarray(np.ascontiguousarray(strip.T), "L").resize((3, n_out), pil_f))
        assert np.array_equal(mine.T, ref_v), (n_in, n_out, name, "vertical")


def test_resize_is_pillow_bit_for_bit(emu_lib):
    dc.check_resize(emu_lib, [((64, 48, 3), (37, 29), "lanczos"), ((37, 29, 3), (64, 48), "lanczos"), ((50, 40, 3), (50, 23), "lanczos"),
                              ((50, 40, 3), (81, 40), "lanczos"), ((33, 21, 1), (16, 16), "lanczos"), ((45, 30, 3), (20, 41), "bilinear"),
                              ((45, 30, 3), (70, 11), "bicubic"), ((16, 16, 3), (16, 16), "lanczos")])


def test_composite_is_numpy_bit_for_bit(emu_lib):
    dc.check_composite(emu_lib)


def test_feather_weight_is_scipy_edt_bit_for_bit(emu_lib):
    assert dc.check_feather(emu_lib, sizes=((61, 83), (40, 33)), radii=(1, 3, 10)) == 2 * 5 * 3 * 4


def test_luminance_match(emu_lib):
    psnr, frac = dc.check_luminance(emu_lib)
    assert psnr >= 60.0


def test_klein_operator_on_the_device_tail(emu_lib):
    """the whole Klein inpainting operator, image arithmetic on the device vs on the host (RGB and RGBA pages)"""
    frac, inf = dc.check_klein_operator(emu_lib)
    assert frac < 0.05 and inf[0] * inf[1] > 900_000            # the crop went up to ~1 MP and back
    dc.check_klein_operator(emu_lib, page_hw=(200, 260), mask_box=(60, 70, 110, 150), page_mode="RGBA")
    dc.check_klein_operator(emu_lib, page_hw=(200, 260), mask_box=(60, 70, 110, 150), page_mode="RGBA", translucent=True)


def test_unload_drops_the_device_tail():
    """`unload_models()` left `_tail` behind, and the next call used a pipeline that was None"""
    import types
    from mangatranslator_amd.core.image import inpainting as ip
    for cls, unload in ((ip.FluxKontextInpainter, "unload_flux_kontext_sdnq_models"), (ip.FluxKleinInpainter, "unload_flux_klein_models")):
        inp = cls.__new__(cls)
        inp.manager = types.SimpleNamespace(**{unload: lambda *a, **k: None})
        inp.pipeline, inp._prompt_embeds, inp._tail = object(), object(), object()
        inp.unload_models()
        assert inp.pipeline is None and inp._tail is None


def test_kontext_operator_on_the_device_tail(emu_lib):
    dc.check_kontext_operator(emu_lib)
