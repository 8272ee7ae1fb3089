"""GPU tier: the page-tail kernels (csrc/pagetail.hip) through the C ABI on gfx950 against Pillow / numpy at the sizes of BASELINE config 5."""
import pytest

import device_tail_checks as dc
from parity_log import record

pytestmark = pytest.mark.gpu


def test_resize_is_pillow_bit_for_bit(hip_lib):
    dc.check_resize(hip_lib, [((688, 528, 3), (1168, 896), "lanczos"), ((1168, 896, 3), (688, 528), "lanczos"), ((496, 416, 3), (1104, 928), "lanczos"),
                              ((1500, 1100, 3), (700, 520), "lanczos"), ((333, 257, 1), (129, 511), "lanczos"), ((640, 480, 3), (321, 239), "bilinear"),
                              ((200, 300, 3), (413, 177), "bicubic")])


def test_composite_is_numpy_bit_for_bit(hip_lib):
    dc.check_composite(hip_lib)


def test_feather_weight_is_scipy_edt_bit_for_bit(hip_lib):
    dc.check_feather(hip_lib, sizes=((611, 833), (97, 30), (1280, 700)), radii=(1, 2, 5, 10))


def test_luminance_match(hip_lib):
    psnr, frac = dc.check_luminance(hip_lib, h=528, w=688)
    record("device_tail.luminance_match.688x528", psnr_db_vs_host_path=psnr, bytes_differing_frac=frac)


def test_klein_operator_on_the_device_tail(hip_lib):
    frac, inf = dc.check_klein_operator(hip_lib, page_hw=(3072, 2048), mask_box=(1300, 700, 1700, 1250))
    record("device_tail.klein_operator.2048x3072", bytes_differing_frac=frac, inference_size=list(inf[:2]))
    dc.check_klein_operator(hip_lib, page_hw=(700, 500), mask_box=(200, 100, 380, 330), page_mode="RGBA")


def test_kontext_operator_on_the_device_tail(hip_lib):
    dc.check_kontext_operator(hip_lib, page_hw=(1536, 1024), mask_box=(600, 300, 900, 700))
