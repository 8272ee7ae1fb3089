"""Independent implementations of the OpenCV primitives the cleaning / inpainting path uses — TEST INFRASTRUCTURE.

`oracle/cleaning_ref.py` and `oracle/cv2_color_ref.py` restate OpenCV's own algorithms (two-pass fixed-point chamfer, incremental
Otsu recurrence, fixed-point colour tables).  OpenCV is not installable in the build image, so what can be done here is to check
those restatements against implementations that share NOTHING with them but the mathematical definition:

    morphology          scipy.ndimage grey_dilation / grey_erosion with a footprint (constant border that never wins)
    chamfer distance    multi-source Dijkstra over the 5x5 move set with the 16.16 weights (shortest path, not a raster scan)
    Otsu                argmax of the between-class variance from cumulative sums in float64 (not the running-mean recurrence)
    external contours   scipy.ndimage.label / find_objects / binary_fill_holes (counts, rectangles, filled regions)
    BGR -> grey, S      the float formulas, rounded
    RGB <-> Lab         float64 CIE 1976 from the sRGB transfer function and the D65 white point

The same functions, bundled as a `cv2`-shaped namespace (`stub_cv2()`), stand in for the wheel when `tools/pin_oracles.py` is
exercised on the CPU tier (tests/test_pinned_oracles.py).
"""
import heapq
import types

import numpy as np
from scipy import ndimage

# ---- morphology ----------------------------------------------------------------------------------------------
# bit patterns written down by hand from OpenCV's documentation of MORPH_ELLIPSE (also in tests/test_cv2_known_answers.py)
ELLIPSES = {
    (3, 3): ["010", "111", "010"],
    (5, 5): ["00100", "11111", "11111", "11111", "00100"],
    (7, 7): ["0001000", "0111110", "1111111", "1111111", "1111111", "0111110", "0001000"],
}


def ellipse(ksize):
    ksize = (int(ksize[0]), int(ksize[1]))
    if ksize in ELLIPSES:
        return np.array([[int(c) for c in row] for row in ELLIPSES[ksize]], np.uint8)
    # other sizes: the inscribed-ellipse row spans, written from the definition (row i spans |x - c| <= c * sqrt(1 - ((i - r) / r)^2))
    kw, kh = ksize
    r, c = kh // 2, kw // 2
    k = np.zeros((kh, kw), np.uint8)
    for i in range(kh):
        dy = (i - r) / r if r else 0.0
        if abs(dy) <= 1.0:
            dx = int(round(c * (1.0 - dy * dy) ** 0.5))
            k[i, max(c - dx, 0):min(c + dx + 1, kw)] = 1
    return k


def dilate(src, kernel, iterations=1):
    out = np.asarray(src)
    for _ in range(int(iterations)):
        out = ndimage.grey_dilation(out, footprint=np.asarray(kernel, bool)[::-1, ::-1], mode="constant", cval=0)
    return out.astype(np.uint8)


def erode(src, kernel, iterations=1):
    out = np.asarray(src)
    for _ in range(int(iterations)):
        out = ndimage.grey_erosion(out, footprint=np.asarray(kernel, bool), mode="constant", cval=255)
    return out.astype(np.uint8)


# ---- chamfer distance as a shortest path ----------------------------------------------------------------------
_MOVES = [(0, 1, 65536), (1, 0, 65536), (1, 1, 91750), (1, 2, 143976), (2, 1, 143976)]      # round(w * 2^16), w = 1, 1.4, 2.1969


def chamfer_l2_5x5(src):
    """float32 distance of every pixel to the nearest zero pixel under the 5x5 chamfer metric, by Dijkstra"""
    src = np.asarray(src)
    h, w = src.shape
    moves = set()
    for dy, dx, wt in _MOVES:
        for sy in (1, -1):
            for sx in (1, -1):
                moves.add((dy * sy, dx * sx, wt))
    inf = (2 ** 31 - 1) >> 2
    dist = np.full((h, w), inf, np.int64)
    heap = []
    for y, x in zip(*np.nonzero(src == 0)):
        dist[y, x] = 0
        heap.append((0, int(y), int(x)))
    heapq.heapify(heap)
    while heap:
        d, y, x = heapq.heappop(heap)
        if d > dist[y, x]:
            continue
        for dy, dx, wt in moves:
            yy, xx = y + dy, x + dx
            if 0 <= yy < h and 0 <= xx < w and d + wt < dist[yy, xx]:
                dist[yy, xx] = d + wt
                heapq.heappush(heap, (d + wt, yy, xx))
    return (dist.astype(np.float32) * np.float32(1.0 / 65536.0)).astype(np.float32)


# ---- thresholds ----------------------------------------------------------------------------------------------
def otsu(values):
    """threshold t maximising w0 w1 (mu0 - mu1)^2 for the split {<= t} / {> t}; the first maximum wins"""
    hist = np.bincount(np.asarray(values, np.uint8).ravel(), minlength=256).astype(np.float64)
    p = hist / hist.sum()
    w0 = np.cumsum(p)
    m = np.cumsum(p * np.arange(256))
    mt = m[-1]
    w1 = 1.0 - w0
    with np.errstate(divide="ignore", invalid="ignore"):
        sigma = np.where((w0 > 1.2e-7) & (w1 > 1.2e-7), (mt * w0 - m) ** 2 / (w0 * w1), 0.0)
    return float(np.argmax(sigma))


def threshold_binary(src, t, maxval=255):
    return np.where(np.asarray(src) > t, maxval, 0).astype(np.uint8)


# ---- connected regions -----------------------------------------------------------------------------------------
def components8(binary):
    lab, n = ndimage.label(np.asarray(binary) > 0, structure=np.ones((3, 3), int))
    rects = []
    for sl in ndimage.find_objects(lab):
        rects.append((sl[1].start, sl[0].start, sl[1].stop - sl[1].start, sl[0].stop - sl[0].start))      # x, y, w, h
    return lab, n, rects


def filled_external(binary):
    """what drawing every external contour FILLED produces: each 8-connected blob with its holes closed (background 4-connected)"""
    return np.where(ndimage.binary_fill_holes(np.asarray(binary) > 0), 255, 0).astype(np.uint8)


# ---- colour -----------------------------------------------------------------------------------------------------
def bgr_to_gray_float(bgr):
    b, g, r = (np.asarray(bgr)[..., i].astype(np.float64) for i in range(3))
    return np.floor(0.114 * b + 0.587 * g + 0.299 * r + 0.5).astype(np.uint8)


def bgr_to_hsv_s_float(bgr):
    x = np.asarray(bgr).astype(np.float64)
    v, mn = x.max(-1), x.min(-1)
    with np.errstate(divide="ignore", invalid="ignore"):
        s = np.where(v > 0, 255.0 * (v - mn) / v, 0.0)
    return np.floor(s + 0.5).astype(np.uint8)


_M = np.array([[0.412453, 0.357580, 0.180423], [0.212671, 0.715160, 0.072169], [0.019334, 0.119193, 0.950227]])     # sRGB -> XYZ, D65
_WHITE = np.array([0.950456, 1.0, 1.088754])


def rgb_to_lab_float(rgb_u8):
    """8-bit CIE L*a*b* (L * 255 / 100, a + 128, b + 128), float64 all the way, rounded at the end"""
    c = np.asarray(rgb_u8).astype(np.float64) / 255.0
    lin = np.where(c <= 0.04045, c / 12.92, ((c + 0.055) / 1.055) ** 2.4)
    xyz = lin @ _M.T / _WHITE
    f = np.where(xyz > 0.008856, np.cbrt(xyz), 7.787 * xyz + 16.0 / 116.0)
    L = np.where(xyz[..., 1] > 0.008856, 116.0 * f[..., 1] - 16.0, 903.3 * xyz[..., 1])
    a = 500.0 * (f[..., 0] - f[..., 1])
    b = 200.0 * (f[..., 1] - f[..., 2])
    out = np.stack([L * 255.0 / 100.0, a + 128.0, b + 128.0], -1)
    return np.clip(np.floor(out + 0.5), 0, 255).astype(np.uint8)


def lab_to_rgb_float(lab_u8):
    q = np.asarray(lab_u8).astype(np.float64)
    L, a, b = q[..., 0] * 100.0 / 255.0, q[..., 1] - 128.0, q[..., 2] - 128.0
    fy = (L + 16.0) / 116.0
    y = np.where(L <= 903.3 * 0.008856, L / 903.3, fy ** 3)
    fy = np.where(L <= 903.3 * 0.008856, 7.787 * y + 16.0 / 116.0, fy)
    fx, fz = fy + a / 500.0, fy - b / 200.0
    inv = lambda f: np.where(f <= 6.0 / 29.0, (f - 16.0 / 116.0) / 7.787, f ** 3)
    xyz = np.stack([inv(fx), y, inv(fz)], -1) * _WHITE
    lin = np.clip(xyz @ np.linalg.inv(_M).T, 0.0, 1.0)
    c = np.where(lin <= 0.0031308, 12.92 * lin, 1.055 * lin ** (1 / 2.4) - 0.055)
    return np.clip(np.floor(c * 255.0 + 0.5), 0, 255).astype(np.uint8)


# ---- a cv2-shaped namespace over the above (stands in for the wheel in the pinning kit's own test) -------------
def stub_cv2(broken: bool = False):
    """`broken`: dilate ignores its kernel's shape (a 3x3 square) — the kit must report that as a mismatch"""
    from oracle import cleaning_ref as cr                   # contour TRACING has no independent implementation here: polygons come from the oracle

    ns = types.SimpleNamespace(__version__="stub-0 (tests/independent_cv.py)", __name__="cv2_stub")
    ns.MORPH_ELLIPSE, ns.DIST_L2, ns.RETR_EXTERNAL, ns.CHAIN_APPROX_SIMPLE, ns.FILLED = 2, 2, 0, 2, -1
    ns.THRESH_BINARY, ns.THRESH_OTSU = 0, 8
    ns.COLOR_BGR2GRAY, ns.COLOR_BGR2HSV, ns.COLOR_RGB2LAB, ns.COLOR_LAB2RGB = 6, 40, 45, 57
    ns.getStructuringElement = lambda shape, ksize: ellipse(ksize)
    ns.dilate = (lambda src, k, iterations=1: dilate(src, np.ones((3, 3), np.uint8), iterations)) if broken else dilate
    ns.erode = erode
    ns.distanceTransform = lambda src, kind, mask: chamfer_l2_5x5(src)

    def threshold(src, t, maxval, kind):
        if kind & ns.THRESH_OTSU:
            t = otsu(src)
        return float(t), threshold_binary(src, t, maxval)

    ns.threshold = threshold
    ns.findContours = lambda m, mode, method: (tuple(c.reshape(-1, 1, 2) for c in cr.find_external_contours(m)), None)
    ns.contourArea = lambda c: cr.contour_area(np.asarray(c).reshape(-1, 2))
    ns.boundingRect = lambda c: cr.bounding_rect(np.asarray(c).reshape(-1, 2))

    def moments(c):
        a00, a10, a01 = cr.contour_sums(np.asarray(c).reshape(-1, 2))
        s2, s6 = (0.5, 1.0 / 6) if a00 > 0 else (-0.5, -1.0 / 6)
        return {"m00": a00 * s2, "m10": a10 * s6, "m01": a01 * s6}

    ns.moments = moments

    def draw_contours(img, contours, idx, color, thickness=1):
        src = np.zeros(img.shape, np.uint8)
        for c in contours:                                   # the filled external contours of a mask are the mask with its holes closed
            pts = np.asarray(c).reshape(-1, 2)
            src[pts[:, 1], pts[:, 0]] = 255
        lab, n, _ = components8(draw_contours.mask)
        img[filled_external(draw_contours.mask) > 0] = color
        return img

    ns.drawContours = draw_contours

    def find_and_remember(m, mode, method):
        draw_contours.mask = np.asarray(m).copy()
        return ns._find(m, mode, method)

    ns._find = ns.findContours
    ns.findContours = find_and_remember

    def cvt(img, code):
        if code == ns.COLOR_BGR2GRAY:
            return bgr_to_gray_float(img)
        if code == ns.COLOR_BGR2HSV:
            out = np.zeros(np.asarray(img).shape, np.uint8)
            out[..., 1] = bgr_to_hsv_s_float(img)
            out[..., 2] = np.asarray(img).max(-1)
            return out
        if code == ns.COLOR_RGB2LAB:
            return rgb_to_lab_float(img)
        if code == ns.COLOR_LAB2RGB:
            return lab_to_rgb_float(img)
        raise ValueError(code)

    ns.cvtColor = cvt
    return ns
