"""A stand-in `cv2` namespace for running the REFERENCE's cleaning code in this container (OpenCV is absent): every primitive
is served by the restatement in oracle/cleaning_ref.py.  Used only by tests/golden/make_goldens.py to pin the reference's control
flow AROUND the primitives (what is thresholded, which contours are kept, which colour fills which pixels); the primitives
themselves stay "parity unpinned"."""
import types

import numpy as np

from oracle import cleaning_ref as cr

MORPH_ELLIPSE, THRESH_BINARY, THRESH_OTSU, RETR_EXTERNAL, CHAIN_APPROX_SIMPLE, FILLED, DIST_L2 = 2, 0, 8, 0, 2, -1, 2
COLOR_BGR2GRAY, COLOR_BGRA2GRAY, COLOR_RGB2BGR, COLOR_BGR2RGB, COLOR_BGR2HSV, COLOR_RGBA2BGRA, COLOR_BGRA2RGBA = 6, 10, 4, 4, 40, 5, 5
COLOR_RGB2LAB, COLOR_LAB2RGB = 45, 57


def getStructuringElement(shape, ksize):
    assert shape == MORPH_ELLIPSE
    return cr.ellipse_kernel(ksize)


def dilate(src, kernel, iterations=1):
    return cr.dilate(np.asarray(src), kernel, iterations)


def erode(src, kernel, iterations=1):
    return cr.erode(np.asarray(src), kernel, iterations)


def threshold(src, thresh, maxval, ttype):
    src = np.asarray(src)
    if ttype & THRESH_OTSU:
        thresh = cr.otsu_threshold(src)
    return float(thresh), np.where(src > thresh, maxval, 0).astype(np.uint8)


def bitwise_and(a, b):
    return np.bitwise_and(a, b)


def bitwise_not(a):
    return np.bitwise_not(a)


def distanceTransform(src, dist_type, mask_size):
    assert dist_type == DIST_L2 and mask_size == 5
    return cr.distance_transform_l2_5x5(np.asarray(src))


def findContours(img, mode, method):
    assert mode == RETR_EXTERNAL
    return tuple(c.reshape(-1, 1, 2) for c in cr.find_external_contours(np.asarray(img))), None


def _pts(cnt):
    return np.asarray(cnt).reshape(-1, 2)


def contourArea(cnt):
    return cr.contour_area(_pts(cnt))


def moments(cnt):
    a00, a10, a01 = cr.contour_sums(_pts(cnt))
    if abs(a00) <= 1.1920929e-07:
        return {"m00": 0.0, "m10": 0.0, "m01": 0.0}
    s2, s6 = (0.5, 1.0 / 6) if a00 > 0 else (-0.5, -1.0 / 6)
    return {"m00": a00 * s2, "m10": a10 * s6, "m01": a01 * s6}


def drawContours(img, contours, idx, color, thickness=1):
    assert idx == -1 and thickness == FILLED
    filled = cr.draw_filled([_pts(c) for c in contours], img.shape[:2])
    img[filled > 0] = color
    return img


def boundingRect(cnt):
    return cr.bounding_rect(_pts(cnt))


def cvtColor(a, code):
    a = np.asarray(a)
    if code == COLOR_RGB2LAB:                       # FluxKleinInpainter luminance match (reference inpainting.py:1183, 1228, 1236)
        from oracle import cv2_color_ref
        return cv2_color_ref.rgb_to_lab_u8(a)
    if code == COLOR_LAB2RGB:
        from oracle import cv2_color_ref
        return cv2_color_ref.lab_to_rgb_u8(a)
    if code in (COLOR_BGR2GRAY, COLOR_BGRA2GRAY):
        return cr.bgr_to_gray(a[..., :3])
    if code == COLOR_BGR2HSV:                       # only ever asked for one sampled pixel: S is what the caller reads
        b, g, r = (int(v) for v in a.reshape(-1, a.shape[-1])[0][:3])
        return np.array([[[0, cr.bgr_pixel_saturation(b, g, r), max(b, g, r)]]], np.uint8)
    if code == COLOR_RGBA2BGRA:                     # (== COLOR_BGRA2RGBA) four channels in, four out
        return np.ascontiguousarray(a[..., [2, 1, 0, 3]])
    return np.ascontiguousarray(a[..., 2::-1])      # COLOR_RGB2BGR / COLOR_BGR2RGB: three channels out, whatever comes in (alpha dropped)


namespace = types.SimpleNamespace(**{k: v for k, v in globals().items() if not k.startswith("_") and k not in ("types", "np", "cr")})
