"""Bubble-cleaning parity: the HIP pixel kernels + native contour code vs the numpy oracle (oracle/cleaning_ref.py)
on synthetic bubbles.  Integer / mask work: bit-exact."""
import numpy as np

from mangatranslator_amd.core.image import cleaning as cl
from oracle import cleaning_ref as cr


def make_page(seed=0, H=220, W=260, dark=False, touch_border=False, gradient=False):
    """BGR page with elliptical bubbles that carry strokes, ring letters ('o' shapes), dots and 1-px hairlines."""
    rng = np.random.default_rng(seed)
    page = np.full((H, W, 3), 90 if not dark else 200, np.uint8)
    page += rng.integers(0, 12, (H, W, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:H, 0:W]
    masks, bboxes = [], []
    specs = [(70, 70, 55, 45), (185, 150, 60, 50)] if not touch_border else [(40, 50, 60, 48), (210, 160, 62, 52)]
    for k, (cx, cy, a, b) in enumerate(specs):
        inside = ((xx - cx) / a) ** 2 + ((yy - cy) / b) ** 2 <= 1.0
        ring = inside & (((xx - cx) / (a - 3)) ** 2 + ((yy - cy) / (b - 3)) ** 2 > 1.0)
        bg = 20 if dark else 245
        fg = 235 if dark else 25
        page[inside] = (np.clip(bg + rng.integers(-6, 7, (H, W, 1)), 0, 255).astype(np.uint8) * np.ones((1, 1, 3), np.uint8))[inside]
        if gradient:                                            # a non-flat interior (coloured / screentoned bubble): light ramp left to right
            ramp = np.clip(140 + 110 * (xx - (cx - a)) / (2.0 * a), 0, 255).astype(np.uint8)
            page[inside] = np.stack([ramp, np.clip(ramp.astype(int) - 6, 0, 255).astype(np.uint8), ramp], -1)[inside]
        page[ring] = fg
        for _ in range(6):                                      # strokes
            sx, sy = int(cx + rng.uniform(-0.5, 0.5) * a), int(cy + rng.uniform(-0.5, 0.5) * b)
            ln, th = int(rng.integers(8, 22)), int(rng.integers(2, 5))
            if rng.random() < 0.5:
                page[max(sy - ln // 2, 0):sy + ln // 2, sx:sx + th] = fg
            else:
                page[sy:sy + th, max(sx - ln // 2, 0):sx + ln // 2] = fg
        ox, oy = cx - 14, cy + 8                                # a ring letter with a dot inside its hole
        o = (np.abs(xx - ox) <= 9) & (np.abs(yy - oy) <= 9) & ~((np.abs(xx - ox) <= 5) & (np.abs(yy - oy) <= 5))
        page[o] = fg
        page[oy - 1:oy + 2, ox - 1:ox + 2] = fg
        page[cy - 20, cx - 15:cx + 16] = fg                     # 1-px hairline (polygon area 0)
        m = inside & ~ring
        masks.append(np.where(m, 255, 0).astype(np.uint8))
        ys, xs = np.nonzero(m)
        bboxes.append((int(xs.min()), int(ys.min()), int(xs.max()), int(ys.max())))
    return page, np.stack(masks), bboxes


def compare(lib, device, seed=0, dark=False, otsu=False, scale=1.0, colored=False, neighbors=False, touch_border=False, shrink=5):
    page, masks, bboxes = make_page(seed, dark=dark, touch_border=touch_border)
    H, W = page.shape[:2]
    dk = cl.structuring_element(cl.scale_kernel(cl.DILATION_KERNEL_SIZE, scale))
    ek = cl.structuring_element(cl.scale_kernel(cl.EROSION_KERNEL_SIZE, scale))
    assert np.array_equal(dk, cr.ellipse_kernel(cl.scale_kernel(cl.DILATION_KERNEL_SIZE, scale)))
    s_px = float(cl.scale_scalar(shrink, scale, minimum=0.0, maximum=64.0))
    min_area = cl.scale_area(cl.MIN_CONTOUR_AREA, scale, minimum=cl.MIN_CONTOUR_AREA, maximum=5000)
    nb = [[bboxes[1]], [bboxes[0]]] if neighbors else None
    got = cl.process_bubbles(page, masks, bboxes, 200, otsu, s_px, dk, ek, min_area, colored, nb, scale, device=device, lib=lib)
    gray = cr.bgr_to_gray(page)
    n_ok = 0
    for i in range(len(bboxes)):
        want = cr.process_single_bubble(masks[i], gray, 200, otsu, s_px, bboxes[i], dk, ek, min_area, colored, nb[i] if nb else None, scale, page)
        if want is None:
            assert got[i] is None
            continue
        assert got[i] is not None, "product found nothing where the oracle found text"
        n_ok += 1
        assert np.array_equal(got[i][0], want[0]), f"final mask differs on {(got[i][0] != want[0]).sum()} px"
        assert tuple(got[i][1]) == tuple(want[1]) and got[i][2] == want[2] and tuple(got[i][3]) == tuple(want[3])
        assert tuple(got[i][4]) == tuple(want[4]), (got[i][4], want[4])
        assert got[i][5] == want[5], (got[i][5], want[5])
    return n_ok
