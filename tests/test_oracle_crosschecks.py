"""The OpenCV restatements (oracle/cleaning_ref.py, oracle/cv2_color_ref.py — "parity unpinned": no cv2 in the build image) against
implementations that share nothing with them but the definition (tests/independent_cv.py): scipy.ndimage morphology with the same
structuring elements, the 5x5 chamfer distance as a shortest path, Otsu from cumulative sums, label / find_objects / fill_holes for
the contour stage, float64 CIE Lab.  Integer primitives must agree bit for bit; the colour conversions within one code of the float
formulas (OpenCV's 8-bit paths are fixed-point approximations of them)."""
import sys
from pathlib import Path

import numpy as np
import pytest

sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tools"))
import independent_cv as ic
import pin_oracles as po
from oracle import cleaning_ref as cr
from oracle import cv2_color_ref as cc

CASES = po.cv2_cases(0)
CASES1 = po.cv2_cases(1)


@pytest.mark.parametrize("ksize", [(3, 3), (5, 5), (7, 7), (9, 9), (11, 11), (9, 5), (5, 9), (15, 15)])
def test_elliptical_structuring_elements(ksize):
    """the hand-written 3 / 5 / 7 patterns, and for other sizes the definition's row spans"""
    assert np.array_equal(cr.ellipse_kernel(ksize), ic.ellipse(ksize))


@pytest.mark.parametrize("name", ["blobs", "speck"])
@pytest.mark.parametrize("ksize", [(7, 7), (5, 5), (3, 3), (11, 11)])
def test_dilate_erode_against_scipy(name, ksize):
    """same structuring element, borders that never win (cv2's default morphology border), 1 and 2 iterations"""
    k = cr.ellipse_kernel(ksize)
    for cases in (CASES, CASES1):
        m = cases[name]
        assert np.array_equal(cr.dilate(m, k), ic.dilate(m, k))
        assert np.array_equal(cr.erode(m, k), ic.erode(m, k))
        assert np.array_equal(cr.dilate(m, k, iterations=2), ic.dilate(m, k, 2))
    g = CASES["grey"]                                               # grey-level morphology (max / min under the footprint), not only 0 / 255
    assert np.array_equal(cr.dilate(g, k), ic.dilate(g, k))
    assert np.array_equal(cr.erode(g, k), ic.erode(g, k))


@pytest.mark.parametrize("name", ["blobs", "speck"])
def test_chamfer_distance_is_the_shortest_path(name):
    """the two-pass raster scan equals Dijkstra over the same move set and weights, bit for bit; and stays within the 5x5 mask's
    published 2 % of the Euclidean distance"""
    m = CASES[name]
    d = cr.distance_transform_l2_5x5(m)
    assert np.array_equal(d, ic.chamfer_l2_5x5(m))
    from scipy import ndimage
    e = ndimage.distance_transform_edt(m > 0)
    assert np.all(np.abs(d - e) <= 0.021 * e + 1e-3)


def test_otsu_against_the_variance_definition():
    rng = np.random.default_rng(5)
    imgs = [CASES["grey"], CASES1["grey"], rng.integers(0, 256, (40, 40)).astype(np.uint8),
            np.concatenate([np.full(300, 20), np.full(100, 230)]).astype(np.uint8),
            np.clip(rng.normal(128, 40, 4000), 0, 255).astype(np.uint8)]
    for im in imgs:
        assert cr.otsu_threshold(im) == ic.otsu(im)


@pytest.mark.parametrize("name", ["blobs", "speck"])
def test_external_contours_against_labelling(name):
    """one external contour per 8-connected component, its bounding rectangle that of the component, and all of them drawn FILLED
    give the mask with its holes closed"""
    for cases in (CASES, CASES1):
        m = cases[name]
        cs = cr.find_external_contours(m)
        lab, n, rects = ic.components8(m)
        assert len(cs) == n
        assert sorted(cr.bounding_rect(c) for c in cs) == sorted(rects)
        assert np.array_equal(cr.draw_filled(cs, m.shape), ic.filled_external(m))
        for c in cs:                                                # every contour point is a mask pixel with a background 8-neighbour or on the border
            for x, y in c:
                assert m[y, x] > 0
        # Green's area of a lattice polygon through pixel centres never exceeds the filled pixel count and is within the boundary length of it
        filled = ic.filled_external(m)
        flab, fn, _ = ic.components8(filled)
        counts = sorted(np.bincount(flab.ravel())[1:].tolist())
        areas = sorted(cr.contour_area(c) for c in cs)
        assert all(a <= n_px for a, n_px in zip(areas, counts))


def test_grey_and_saturation_against_float_formulas():
    for im in (CASES["bgr"], CASES["ramp"], CASES1["bgr"]):
        d = np.abs(cr.bgr_to_gray(im).astype(int) - ic.bgr_to_gray_float(im).astype(int))
        assert d.max() <= 1 and (d > 0).mean() < 0.02
        s = np.array([[cr.bgr_pixel_saturation(int(p[0]), int(p[1]), int(p[2])) for p in row] for row in im[:6]])
        assert np.abs(s - ic.bgr_to_hsv_s_float(im[:6]).astype(int)).max() <= 1


def test_lab_against_float64_cie():
    """8-bit RGB -> Lab within one code of the float64 CIE formulas on random colours, the grey ramp and the cube's corners — two codes
    for the darkest colours only (a channel below 48: the forward path's gamma table keeps three fractional bits, so linear values
    under 1 / 2040 are quantised; fewer than 0.1 % of random colours) — and the way back within two codes of the float inverse"""
    corners = np.array([[[r, g, b] for r in (0, 255) for g in (0, 255) for b in (0, 255)]], np.uint8)
    for im in (CASES["bgr"], CASES["ramp"], CASES1["bgr"], corners):
        lab = cc.rgb_to_lab_u8(im)
        d = np.abs(lab.astype(int) - ic.rgb_to_lab_float(im).astype(int))
        assert d.max() <= 2 and (d > 1).mean() < 1e-3, (d.max(), (d > 1).mean())
        assert d[im.min(-1) >= 48].max(initial=0) <= 1
        back = cc.lab_to_rgb_u8(lab)
        assert np.abs(back.astype(int) - ic.lab_to_rgb_float(lab).astype(int)).max() <= 2
    ramp = CASES["ramp"]
    assert np.abs(cc.lab_to_rgb_u8(cc.rgb_to_lab_u8(ramp)).astype(int) - ramp.astype(int)).max() <= 2        # greys survive the 8-bit round trip
