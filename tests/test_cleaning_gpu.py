"""GPU tier: bubble cleaning through the C ABI on gfx950 vs the numpy oracle (bit-exact masks)."""
import numpy as np
import pytest

import cleaning_checks as cc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kw", [dict(), dict(dark=True), dict(otsu=True), dict(scale=1.6, seed=2), dict(colored=True, seed=4),
                                dict(neighbors=True, seed=1), dict(touch_border=True, seed=6), dict(shrink=0)])
def test_process_bubbles_vs_oracle(hip_lib, kw):
    assert cc.compare(hip_lib, "cuda:0", **kw) >= 1


@pytest.mark.parametrize("size", [(1024, 1536), (2048, 3072)])
def test_page_scale_device_masks(hip_lib, size):
    """1024x1536 and 2048x3072 (BASELINE config 5) pages, masks already resident on the device (as SAM leaves them)"""
    import torch
    from mangatranslator_amd.core.image import cleaning as cl
    from mangatranslator_amd.utils.synthetic_pages import make_page
    from oracle import cleaning_ref as cr
    pg, boxes, _ = make_page(3, size[0], size[1], bubbles=4)
    page = np.ascontiguousarray(pg[..., ::-1])
    H, W = page.shape[:2]
    yy, xx = np.mgrid[0:H, 0:W]
    masks, bbs = [], []
    for x0, y0, x1, y1 in boxes:
        cx, cy, a, b = (x0 + x1) / 2, (y0 + y1) / 2, (x1 - x0) / 2 - 5, (y1 - y0) / 2 - 5
        masks.append(np.where(((xx - cx) / a) ** 2 + ((yy - cy) / b) ** 2 <= 1.0, 255, 0).astype(np.uint8))
        bbs.append((int(x0), int(y0), int(x1), int(y1)))
    scale = (W * H / 1e6) ** 0.5
    dk = cl.structuring_element(cl.scale_kernel(cl.DILATION_KERNEL_SIZE, scale)); ek = cl.structuring_element(cl.scale_kernel(cl.EROSION_KERNEL_SIZE, scale))
    s_px = float(cl.scale_scalar(5, scale, minimum=0.0, maximum=64.0)); min_area = cl.scale_area(50, scale, minimum=50, maximum=5000)
    dm = torch.from_numpy(np.stack(masks)).to("cuda:0")
    got = cl.process_bubbles(page, dm, bbs, 200, False, s_px, dk, ek, min_area, False, None, scale, device="cuda:0", lib=hip_lib)
    gray = cr.bgr_to_gray(page)
    for i in range(2):       # the oracle's python chamfer loops are slow: check two bubbles on crops around them
        x0, y0, x1, y1 = bbs[i]
        ys, xs = slice(max(0, y0 - 40), min(H, y1 + 40)), slice(max(0, x0 - 40), min(W, x1 + 40))
        want = cr.process_single_bubble(masks[i][ys, xs], gray[ys, xs], 200, False, s_px, None, dk, ek, min_area, False, None, scale, page[ys, xs])
        assert (want is None) == (got[i] is None)
        if want is not None:
            assert np.array_equal(got[i][0][ys, xs], want[0]) and got[i][1] == want[1]
