"""Hand-derived known-answer vectors for the OpenCV primitives the cleaning / conjoined / luminance paths rely on (SURVEY.md §8c asked
for them: OpenCV itself is absent here, so the restatements in oracle/cleaning_ref.py and the product's own implementations are pinned
to answers worked out from OpenCV's published definitions, not to each other).

  * getStructuringElement(MORPH_ELLIPSE): the 3x3, 5x5 and 7x7 elements printed in OpenCV's morphology tutorial
  * distanceTransform(DIST_L2, 5): chamfer weights a = 1, b = 1.4, c = 2.1969 — distances around a single zero pixel are sums of those
  * threshold(THRESH_OTSU): two equal spikes -> the first grey level that maximises the between-class variance = the lower spike
  * cvtColor(BGR2GRAY): (R 4899 + G 9617 + B 1868 + 8192) >> 14 -> pure red 76, green 150, blue 29, white 255
  * findContours / contourArea / moments / drawContours(FILLED) on an axis-aligned rectangle of pixels: Green's theorem over the pixel-centre
    polygon -> area (w-1)(h-1), centroid = centre, the filled polygon = the rectangle's pixels
  * dilate / erode: a single set pixel dilates to the structuring element itself; eroding that gives the pixel back
"""
import numpy as np

from mangatranslator_amd.core.image import cleaning as cl
from oracle import cleaning_ref as cr

E3 = [[0, 1, 0], [1, 1, 1], [0, 1, 0]]
E5 = [[0, 0, 1, 0, 0], [1, 1, 1, 1, 1], [1, 1, 1, 1, 1], [1, 1, 1, 1, 1], [0, 0, 1, 0, 0]]
E7 = [[0, 0, 0, 1, 0, 0, 0], [0, 1, 1, 1, 1, 1, 0], [1, 1, 1, 1, 1, 1, 1], [1, 1, 1, 1, 1, 1, 1], [1, 1, 1, 1, 1, 1, 1], [0, 1, 1, 1, 1, 1, 0],
      [0, 0, 0, 1, 0, 0, 0]]


def test_elliptical_structuring_elements():
    for k, want in ((3, E3), (5, E5), (7, E7)):
        assert cr.ellipse_kernel((k, k)).tolist() == want
        assert np.asarray(cl.structuring_element((k, k))).tolist() == want          # the product's own table builder
    assert cr.ellipse_kernel((1, 1)).tolist() == [[1]]
    wide = cr.ellipse_kernel((7, 3))                                                # 7 wide, 3 high
    assert wide.shape == (3, 7) and wide[1].tolist() == [1] * 7 and wide[0].tolist() == [0, 0, 0, 1, 0, 0, 0]
    assert np.array_equal(np.asarray(cl.structuring_element((7, 3))), wide)


def _chamfer_answers():
    a, b, c = 1.0, 1.4, 2.1969
    return {(0, 0): 0.0, (1, 0): a, (1, 1): b, (2, 0): 2 * a, (2, 1): c, (2, 2): 2 * b, (3, 0): 3 * a, (3, 1): c + a, (3, 2): c + b, (3, 3): 3 * b,
            (4, 1): c + 2 * a, (4, 2): 2 * c, (4, 3): c + 2 * b}


def test_chamfer_distance_around_a_single_zero(emu_lib):
    src = np.full((11, 11), 255, np.uint8)
    src[5, 5] = 0
    got = cr.distance_transform_l2_5x5(src)
    import ctypes as C
    native = np.zeros(src.shape, np.float32)
    assert emu_lib.mtx_host_chamfer_l2_5x5(src.ctypes.data_as(C.c_void_p), 11, 11, native.ctypes.data_as(C.c_void_p)) == 0      # the product's native transform
    for (dx, dy), want in _chamfer_answers().items():
        for sx, sy in ((1, 1), (-1, 1), (1, -1), (-1, -1)):
            for (ux, uy) in ((dx, dy), (dy, dx)):                                    # 8-fold symmetry
                y, x = 5 + sy * uy, 5 + sx * ux
                if 0 <= y < 11 and 0 <= x < 11:
                    assert abs(float(got[y, x]) - want) < 2e-4, (dx, dy, float(got[y, x]), want)
                    assert abs(float(native[y, x]) - want) < 2e-4
    assert np.array_equal(got, native)
    allz = np.zeros((4, 6), np.uint8)
    assert not cr.distance_transform_l2_5x5(allz).any()


def test_otsu_two_spikes_and_gray_weights():
    v = np.array([50] * 100 + [200] * 100, np.uint8)
    assert cr.otsu_threshold(v) == 50.0                       # every threshold in [50, 199] separates the spikes; the first maximum wins
    v2 = np.array([10] * 30 + [20] * 30 + [240] * 40, np.uint8)
    assert cr.otsu_threshold(v2) == 20.0                      # {10, 20} vs {240}: variance is maximal from 20 up to 239, first = 20
    px = np.array([[[0, 0, 255], [0, 255, 0], [255, 0, 0], [255, 255, 255], [0, 0, 0], [128, 128, 128]]], np.uint8)      # BGR
    assert cr.bgr_to_gray(px).tolist() == [[76, 150, 29, 255, 0, 128]]


def test_rectangle_contour_area_centroid_fill():
    img = np.zeros((9, 10), np.uint8)
    img[3:6, 2:7] = 255                                       # 5 x 3 pixels: x 2..6, y 3..5
    cnts = cr.find_external_contours(img)
    assert len(cnts) == 1
    pts = {tuple(int(v) for v in p) for p in np.asarray(cnts[0]).reshape(-1, 2)}
    assert {(2, 3), (2, 5), (6, 5), (6, 3)} <= pts            # the corners are on the traced border (pixel centres)
    assert abs(cr.contour_area(np.asarray(cnts[0]).reshape(-1, 2))) == 8.0      # (6 - 2) * (5 - 3)
    cx, cy = cr.contour_centroid(np.asarray(cnts[0]).reshape(-1, 2))
    assert (round(cx, 6), round(cy, 6)) == (4.0, 4.0)
    assert cr.bounding_rect(np.asarray(cnts[0]).reshape(-1, 2)) == (2, 3, 5, 3)
    filled = cr.draw_filled([np.asarray(cnts[0]).reshape(-1, 2)], img.shape)
    assert np.array_equal(filled > 0, img > 0)                # boundary pixels included
    two = img.copy(); two[0:2, 8:10] = 255                    # a second blob: two external contours, holes are not reported
    two[4, 4] = 0
    assert len(cr.find_external_contours(two)) == 2


def test_dilate_and_erode_of_a_single_pixel():
    img = np.zeros((9, 9), np.uint8)
    img[4, 4] = 255
    k = cr.ellipse_kernel((5, 5))
    d = cr.dilate(img, k)
    assert (d[2:7, 2:7] > 0).astype(int).tolist() == E5 and d.sum() == 255 * 17
    e = cr.erode(d, k)
    assert np.array_equal(e, img)                             # opening by the same element gives the seed back
    edge = np.zeros((5, 5), np.uint8); edge[0, 0] = 255       # at the image border: outside pixels never win the max
    assert (cr.dilate(edge, k)[:3, :3] > 0).astype(int).tolist() == [[1, 1, 1], [1, 1, 1], [1, 0, 0]]      # row 2 reaches the seed only through the tip of the ellipse (dx = 0)
