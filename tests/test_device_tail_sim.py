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


def test_kontext_operator_on_the_device_tail(emu_lib):
    dc.check_kontext_operator(emu_lib)
