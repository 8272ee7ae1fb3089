"""CPU oracle of the OpenCV (non-diffusion) bubble-cleaning chain — SURVEY.md §8 row a5.
TEST INFRASTRUCTURE ONLY (imported by tests/ and bench cpu_baseline; never by the product path).

Follows reference core/image/cleaning.py:210-521 (`process_single_bubble`), :155-207
(`_build_adaptive_shrink_mask`) and the fill at :1020-1039.  The reference calls OpenCV (cv2 4.x, absent
from this image), so every cv2 primitive it uses is RESTATED here from OpenCV's published algorithms:
    getStructuringElement(MORPH_ELLIPSE), dilate / erode (default borders), threshold(BINARY [+OTSU]),
    distanceTransform(DIST_L2, 5) (two-pass 16.16 fixed-point chamfer, weights 1 / 1.4 / 2.1969),
    findContours(RETR_EXTERNAL) (outer borders of 8-connected components), contourArea / moments (Green),
    drawContours(FILLED) (even-odd over all polygons + outlines), boundingRect, cvtColor BGR2GRAY / BGR2HSV.
PARITY UNPINNED: no cv2 here to check the restatement against; anchored on the reference's call sites.
"""
import math

import numpy as np

GRAYSCALE_MIDPOINT = 128
MIN_CONTOUR_AREA = 50
DILATION_KERNEL_SIZE = (7, 7)
EROSION_KERNEL_SIZE = (5, 5)
SOLID_RATIO_THRESHOLD = 0.65
JUNCTION_ADJACENCY_MARGIN = 10
JUNCTION_MIN_SHRINK = 1.0

HV, DIAG, LONG = 65536, 91750, 143976        # round(w * 2**16) for w = 1, 1.4, 2.1969
INIT_DIST = (2 ** 31 - 1) >> 2


def bgr_to_gray(bgr: np.ndarray) -> np.ndarray:
    b, g, r = (bgr[..., i].astype(np.int64) for i in range(3))
    return ((b * 1868 + g * 9617 + r * 4899 + (1 << 13)) >> 14).astype(np.uint8)


def bgr_pixel_saturation(b: int, g: int, r: int) -> int:
    v, mn = max(b, g, r), min(b, g, r)
    if v == 0:
        return 0
    sdiv = int(round((255 << 12) / float(v)))
    return ((v - mn) * sdiv + (1 << 11)) >> 12


def ellipse_kernel(ksize) -> np.ndarray:
    kw, kh = int(ksize[0]), int(ksize[1])
    r, c = kh // 2, kw // 2
    inv_r2 = 1.0 / (r * r) if r else 0.0
    k = np.zeros((kh, kw), np.uint8)
    for i in range(kh):
        dy = i - r
        if abs(dy) <= r:
            dx = int(np.rint(c * math.sqrt((r * r - dy * dy) * inv_r2)))
            k[i, max(c - dx, 0):min(c + dx + 1, kw)] = 1
    return k


def _morph(src: np.ndarray, kernel: np.ndarray, dilate: bool) -> np.ndarray:
    h, w = src.shape
    kh, kw = kernel.shape
    ay, ax = kh // 2, kw // 2
    fill = 0 if dilate else 255           # pixels outside the image never win the max / min
    pad = np.full((h + kh - 1, w + kw - 1), fill, np.uint8)
    pad[ay:ay + h, ax:ax + w] = src
    out = np.full((h, w), fill, np.uint8)
    for i in range(kh):
        for j in range(kw):
            if kernel[i, j]:
                win = pad[i:i + h, j:j + w]
                out = np.maximum(out, win) if dilate else np.minimum(out, win)
    return out


def dilate(src, kernel, iterations=1):
    for _ in range(iterations):
        src = _morph(src, kernel, True)
    return src


def erode(src, kernel, iterations=1):
    for _ in range(iterations):
        src = _morph(src, kernel, False)
    return src


def distance_transform_l2_5x5(src: np.ndarray) -> np.ndarray:
    """float32 chamfer distance to the nearest zero pixel."""
    h, w = src.shape
    t = np.full((h + 4, w + 4), INIT_DIST, np.int64)
    fwd = ((-2, -1, LONG), (-2, 1, LONG), (-1, -2, LONG), (-1, -1, DIAG), (-1, 0, HV), (-1, 1, DIAG), (-1, 2, LONG), (0, -1, HV))
    for y in range(h):
        row = t[y + 2]
        for x in range(w):
            if src[y, x] == 0:
                row[x + 2] = 0
            else:
                row[x + 2] = min(t[y + 2 + dy, x + 2 + dx] + wt for dy, dx, wt in fwd)
    bwd = tuple((-dy, -dx, wt) for dy, dx, wt in fwd)
    for y in range(h - 1, -1, -1):
        row = t[y + 2]
        for x in range(w - 1, -1, -1):
            d = row[x + 2]
            if d > HV:
                row[x + 2] = min(d, min(t[y + 2 + dy, x + 2 + dx] + wt for dy, dx, wt in bwd))
    return (t[2:2 + h, 2:2 + w].astype(np.float32) * np.float32(1.0 / 65536.0)).astype(np.float32)


def otsu_threshold(values: np.ndarray) -> float:
    hist = np.bincount(values.astype(np.uint8).ravel(), minlength=256).astype(np.float64)
    n = hist.sum()
    scale = 1.0 / n
    mu = float((np.arange(256) * hist).sum()) * scale
    q1 = mu1 = 0.0
    best, best_val = 0.0, 0
    for i in range(256):
        p_i = hist[i] * scale
        mu1 *= q1
        q1 += p_i
        q2 = 1.0 - q1
        if min(q1, q2) < 1.1920929e-07 or max(q1, q2) > 1.0 - 1.1920929e-07:
            continue
        mu1 = (mu1 + i * p_i) / q1
        mu2 = (mu - q1 * mu1) / q2
        sigma = q1 * q2 * (mu1 - mu2) * (mu1 - mu2)
        if sigma > best:
            best, best_val = sigma, i
    return float(best_val)


# ---- contours -------------------------------------------------------------------------------------------
_N8 = ((1, 0), (1, -1), (0, -1), (-1, -1), (-1, 0), (-1, 1), (0, 1), (1, 1))      # (dx, dy), counter-clockwise from east


def _label8(binary: np.ndarray) -> np.ndarray:
    from scipy import ndimage
    lab, _ = ndimage.label(binary, structure=np.ones((3, 3), np.int32))
    return lab


def _trace_outer(comp: np.ndarray, sx: int, sy: int):
    """Outer border of the component containing (sx, sy) — its top-most, then left-most pixel — as the closed
    sequence of pixel centres visited by 8-connected border following (Suzuki & Abe outer border)."""
    h, w = comp.shape
    inside = lambda x, y: 0 <= x < w and 0 <= y < h and comp[y, x]
    # first neighbour search starts from the west pixel (background by construction) going clockwise
    start_dir = None
    d = 4                                            # direction of the west neighbour in _N8
    for k in range(8):
        dd = (d - k) % 8                             # clockwise
        nx, ny = sx + _N8[dd][0], sy + _N8[dd][1]
        if inside(nx, ny):
            start_dir = dd
            break
    if start_dir is None:
        return [(sx, sy)]
    pts = [(sx, sy)]
    i1x, i1y = sx + _N8[start_dir][0], sy + _N8[start_dir][1]     # the pixel found clockwise = "i1"
    px, py, cx, cy = i1x, i1y, sx, sy                # previous (i2) and current (i3) of Suzuki's step 3.3
    while True:
        # search counter-clockwise around (cx, cy), starting after the direction pointing at (px, py)
        dprev = _N8.index((px - cx, py - cy))
        for k in range(1, 9):
            dd = (dprev + k) % 8
            nx, ny = cx + _N8[dd][0], cy + _N8[dd][1]
            if inside(nx, ny):
                break
        if (nx, ny) == (sx, sy) and (cx, cy) == (i1x, i1y):
            break
        px, py, cx, cy = cx, cy, nx, ny
        pts.append((cx, cy))
    return pts


def find_external_contours(binary: np.ndarray):
    """list of int32 [n,2] (x,y) polygons, in the order cv2.findContours returns them (last found first)."""
    lab = _label8(binary > 0)
    out = []
    seen = set()
    ys, xs = np.nonzero(lab)
    for y, x in zip(ys, xs):                         # raster order: first pixel of a label is its top-left start
        l = lab[y, x]
        if l in seen:
            continue
        seen.add(l)
        out.append(np.asarray(_trace_outer(lab == l, int(x), int(y)), np.int32))
    return out[::-1]


def contour_sums(cnt: np.ndarray):
    x, y = cnt[:, 0].astype(np.float64), cnt[:, 1].astype(np.float64)
    xp, yp = np.roll(x, 1), np.roll(y, 1)
    dxy = xp * y - x * yp
    return float(dxy.sum()), float((dxy * (xp + x)).sum()), float((dxy * (yp + y)).sum())


def contour_area(cnt: np.ndarray) -> float:
    return abs(contour_sums(cnt)[0]) * 0.5


def contour_centroid(cnt: np.ndarray):
    a00, a10, a01 = contour_sums(cnt)
    if abs(a00) <= 1.1920929e-07:
        return None
    s2, s6 = (0.5, 1.0 / 6) if a00 > 0 else (-0.5, -1.0 / 6)
    m00, m10, m01 = a00 * s2, a10 * s6, a01 * s6
    if m00 == 0:
        return None
    return int(m10 / m00), int(m01 / m00)


def _fill_polygon_interior(cnt: np.ndarray, shape) -> np.ndarray:
    """pixels strictly inside or on the lattice polygon (outer border of one 8-connected blob): the blob plus
    everything not 4-connected to the outside of its boundary ring."""
    from scipy import ndimage
    h, w = shape
    ring = np.zeros((h, w), bool)
    n = len(cnt)
    for i in range(n):
        (x0, y0), (x1, y1) = cnt[i], cnt[(i + 1) % n]
        ring[y0, x0] = True
        ring[y1, x1] = True
    pad = np.zeros((h + 2, w + 2), bool)
    pad[1:-1, 1:-1] = ring
    outside, _ = ndimage.label(~pad, structure=[[0, 1, 0], [1, 1, 1], [0, 1, 0]])
    return (outside != outside[0, 0])[1:-1, 1:-1]


def draw_filled(contours, shape) -> np.ndarray:
    acc = np.zeros(shape, bool)
    ring = np.zeros(shape, bool)
    for c in contours:
        acc ^= _fill_polygon_interior(c, shape)
        ring[c[:, 1], c[:, 0]] = True
    return np.where(acc | ring, 255, 0).astype(np.uint8)


def bounding_rect(cnt):
    x0, y0, x1, y1 = cnt[:, 0].min(), cnt[:, 1].min(), cnt[:, 0].max(), cnt[:, 1].max()
    return int(x0), int(y0), int(x1 - x0 + 1), int(y1 - y0 + 1)


# ---- the operator ---------------------------------------------------------------------------------------
def adaptive_shrink_mask(roi_mask, roi_shrink_px, detection_bbox, neighbor_bboxes, processing_scale=1.0):
    margin = max(1, int(round(JUNCTION_ADJACENCY_MARGIN * processing_scale)))
    jmin = max(1.0, JUNCTION_MIN_SHRINK * processing_scale)
    dist = distance_transform_l2_5x5(roi_mask)
    shrunk = np.where(dist >= np.float32(roi_shrink_px), 255, 0).astype(np.uint8)
    x1, y1, x2, y2 = detection_bbox
    h, w = roi_mask.shape
    for ox1, oy1, ox2, oy2 in neighbor_bboxes:
        if x1 - margin > ox2 or ox1 - margin > x2 or y1 - margin > oy2 or oy1 - margin > y2:
            continue
        zx1, zy1 = max(0, max(x1, ox1) - margin), max(0, max(y1, oy1) - margin)
        zx2, zy2 = min(w, min(x2, ox2) + margin), min(h, min(y2, oy2) + margin)
        if zx2 <= zx1 or zy2 <= zy1:
            continue
        shrunk[zy1:zy2, zx1:zx2] |= np.where(dist[zy1:zy2, zx1:zx2] >= np.float32(jmin), 255, 0).astype(np.uint8)
    return shrunk


def process_single_bubble(base_mask, img_gray, thresholding_value, use_otsu_threshold, roi_shrink_px, detection_bbox=None,
                          dilation_kernel=None, constraint_erosion_kernel=None, min_contour_area=MIN_CONTOUR_AREA,
                          classify_colored=False, neighbor_bboxes=None, processing_scale=1.0, image_bgr=None):
    """-> (final_mask, fill_color_bgr, is_colored, sample_color_bgr, text_bbox, text_color_bgr) or None on failure"""
    h, w = img_gray.shape
    base_mask = np.where(base_mask > 0, 255, 0).astype(np.uint8)
    if dilation_kernel is None:
        dilation_kernel = ellipse_kernel(DILATION_KERNEL_SIZE)
    if constraint_erosion_kernel is None:
        constraint_erosion_kernel = ellipse_kernel(EROSION_KERNEL_SIZE)
    masked = img_gray[base_mask == 255]
    if masked.size == 0:
        return None
    mean_val = float(np.mean(masked))
    black = mean_val < GRAYSCALE_MIDPOINT
    fill = (0, 0, 0) if black else (255, 255, 255)
    is_colored, sample_color = False, fill
    roi_mask = dilate(base_mask, dilation_kernel)
    roi = roi_mask == 255
    roi_gray = np.where(roi, img_gray, 0).astype(np.uint8)
    v = (255 - roi_gray) if black else roi_gray
    thr = otsu_threshold(v[roi]) if use_otsu_threshold else thresholding_value
    thresholded = np.where(v > thr, 255, 0).astype(np.uint8) & roi_mask
    if neighbor_bboxes and detection_bbox is not None:
        shrunk = adaptive_shrink_mask(roi_mask, float(roi_shrink_px), detection_bbox, neighbor_bboxes, processing_scale)
    else:
        shrunk = np.where(distance_transform_l2_5x5(roi_mask) >= np.float32(roi_shrink_px), 255, 0).astype(np.uint8)
    thresholded &= shrunk
    eroded = erode(base_mask, constraint_erosion_kernel)
    valid = []
    for cnt in find_external_contours(thresholded):
        if contour_area(cnt) <= min_contour_area:
            continue
        cen = contour_centroid(cnt)
        if cen is None:
            continue
        cx, cy = cen
        if 0 <= cx < w and 0 <= cy < h and eroded[cy, cx] == 255:
            valid.append(cnt)
    if not valid:
        return None
    validated = draw_filled(valid, (h, w))
    boundary = find_external_contours(validated)
    if not boundary:
        return None
    largest = max(boundary, key=contour_area)
    final_mask = draw_filled([largest], (h, w))
    x, y, bw, bh = bounding_rect(largest)
    text_bbox = (x, y, x + bw, y + bh)
    text_mask = (255 - thresholded) & shrunk
    k3 = np.ones((3, 3), np.uint8)
    if classify_colored:
        sampling = erode(base_mask, constraint_erosion_kernel, iterations=2)
        sampling[dilate(text_mask, k3) == 255] = 0
        if image_bgr is not None:
            px = image_bgr[sampling == 255]
            if px.size == 0:
                px = image_bgr[base_mask == 255]
            if px.size > 0:
                med = np.median(px, axis=0).astype(int)
                diffs = np.max(np.abs(px.astype(int) - med), axis=1)
                solid_ratio = float(np.count_nonzero(diffs <= 15)) / float(len(px))
                if (med >= 245).all():
                    fill = (255, 255, 255)
                elif (med <= 10).all():
                    fill = (0, 0, 0)
                else:
                    fill = (int(med[0]), int(med[1]), int(med[2]))
            else:
                fill, solid_ratio = (255, 255, 255), 0.0
        else:
            sp = img_gray[sampling == 255]
            if sp.size == 0:
                sp = masked
            sv = sp.astype(np.uint8).flatten()
            med_val = int(np.median(sv)) if sv.size > 0 else int(mean_val)
            solid_ratio = float(np.count_nonzero(np.abs(sv.astype(int) - med_val) <= 15)) / float(max(len(sv), 1))
            fill = (255, 255, 255) if med_val >= 245 else (0, 0, 0) if med_val <= 10 else (med_val, med_val, med_val)
        is_colored = not (solid_ratio >= SOLID_RATIO_THRESHOLD)
        sample_color = fill
    text_color = None
    if image_bgr is not None:
        sm = erode(text_mask, k3)
        tp = image_bgr[sm == 255]
        if tp.size == 0:
            tp = image_bgr[text_mask == 255]
        if tp.size > 0:
            sb = tuple(int(t) for t in np.median(tp, axis=0).astype(int))
            if bgr_pixel_saturation(*sb) < 25:
                lum = 0.114 * fill[0] + 0.587 * fill[1] + 0.299 * fill[2]
                text_color = (0, 0, 0) if lum >= 128 else (255, 255, 255)
            else:
                text_color = sb
    return final_mask, fill, is_colored, sample_color, text_bbox, text_color
