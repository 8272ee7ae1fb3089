"""CPU tier: bubble cleaning (pixel kernels on the simulator + native contour code) vs the numpy oracle, and the
clean_speech_bubbles operator surface."""
import numpy as np
import pytest
from PIL import Image

import cleaning_checks as cc
from mangatranslator_amd.core.image import cleaning as cl
from oracle import cleaning_ref as cr


def test_structuring_elements():
    assert cl.ellipse_rows((7, 7)) == (3, [0, 2, 3, 3, 3, 2, 0])
    assert cl.ellipse_rows((5, 5)) == (2, [0, 2, 2, 2, 0])
    for k in ((3, 3), (9, 9), (11, 7), (19, 19)):
        assert np.array_equal(cl.structuring_element(k), cr.ellipse_kernel(k))
        assert cl._rows_of_kernel(cl.structuring_element(k)) == cl.ellipse_rows(k)


def test_contours_native_vs_oracle(emu_lib):
    rng = np.random.default_rng(5)
    for trial in range(6):
        img = (rng.random((40, 48)) < (0.25 + 0.08 * trial)).astype(np.uint8) * 255
        img[10:30, 12:14] = 255; img[10:12, 12:36] = 255; img[28:30, 12:36] = 255; img[10:30, 34:36] = 255; img[19:21, 22:26] = 255
        ero = np.full_like(img, 255)
        want_valid = []
        for c in cr.find_external_contours(img):
            if cr.contour_area(c) > 6 and cr.contour_centroid(c) is not None:
                want_valid.append(c)
        out = np.zeros_like(img)
        import ctypes as C
        bb = (C.c_int * 4)()
        n = emu_lib.mtx_host_text_mask(np.ascontiguousarray(img).ctypes.data, ero.ctypes.data, 48, 40, 3, 4, 100, 100, 6.0, out.ctypes.data, bb)
        assert n == len(want_valid)
        if n:
            val = cr.draw_filled(want_valid, img.shape)
            largest = max(cr.find_external_contours(val), key=cr.contour_area)
            assert np.array_equal(out, cr.draw_filled([largest], img.shape))
            x, y, w, h = cr.bounding_rect(largest)
            assert tuple(bb) == (x + 3, y + 4, w, h)


def test_distance_shrink_matches_two_pass(emu_lib):
    """the relaxed chamfer distance thresholds exactly like the two-pass fixed-point transform"""
    page, masks, bboxes = cc.make_page(3)
    for shrink in (0.0, 1.0, 5.0, 6.25, 9.0):
        crops, _ = cl._run_pixel_half(emu_lib, "cpu", page, masks, cl.ellipse_rows((7, 7)), cl.ellipse_rows((5, 5)), 200, False, shrink, None, 1.0)
        for i in range(len(bboxes)):
            x0, y0, w, h = crops.rois[i]
            roi = cr.dilate(masks[i], cr.ellipse_kernel((7, 7)))
            want = np.where(cr.distance_transform_l2_5x5(roi) >= np.float32(shrink), 255, 0).astype(np.uint8)
            assert np.array_equal(crops.plane("roi", i), roi[y0:y0 + h, x0:x0 + w])
            assert np.array_equal(crops.plane("shrunk", i), want[y0:y0 + h, x0:x0 + w])
            assert np.array_equal(crops.plane("eroded", i), cr.erode(masks[i], cr.ellipse_kernel((5, 5)))[y0:y0 + h, x0:x0 + w])


@pytest.mark.parametrize("kw", [dict(), dict(dark=True), dict(otsu=True), dict(scale=1.6, seed=2), dict(colored=True, seed=4),
                                dict(neighbors=True, seed=1), dict(touch_border=True, seed=6), dict(shrink=0)])
def test_process_bubbles_vs_oracle(emu_lib, kw):
    assert cc.compare(emu_lib, "cpu", **kw) >= 1


def test_clean_speech_bubbles_operator(emu_lib):
    page, masks, bboxes = cc.make_page(7)
    pil = Image.fromarray(page[..., ::-1].copy())
    dets = [{"bbox": bboxes[0], "sam_mask": masks[0]}, {"bbox": bboxes[1], "sam_mask": masks[1]},
            {"bbox": (0, 0, 5, 5), "sam_mask": np.zeros_like(masks[0])}, {"bbox": (1, 1, 2, 2)}]
    cleaned, info = cl.clean_speech_bubbles(pil, None, pre_computed_detections=dets, lib=emu_lib, device="cpu")
    assert cleaned.shape == page.shape and len(info) == 2
    for b in info:
        assert set(b) >= {"mask", "base_mask", "color", "bbox", "is_colored", "text_bbox", "text_color_bgr", "is_sam", "inpainted"}
        assert b["color"] == (255, 255, 255) and b["is_sam"]
        assert (cleaned[b["mask"] == 255] == 255).all()          # text pixels filled with the bubble colour
    untouched = np.bitwise_or.reduce([b["mask"] for b in info]) == 0
    assert np.array_equal(cleaned[untouched], page[untouched])
    with pytest.raises(Exception):
        cl.clean_speech_bubbles(pil, None, lib=emu_lib)          # image object without detections
    with pytest.raises(cl.CleaningError):
        cl.process_single_bubble(np.zeros_like(masks[0]), cr.bgr_to_gray(page), page.shape[0], page.shape[1], 200, False, 5, False,
                                 image_bgr=page, lib=emu_lib, device="cpu")
