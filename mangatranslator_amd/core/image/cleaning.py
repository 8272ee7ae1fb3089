"""OpenCV-style (non-diffusion) speech-bubble cleaning on MI355X — SURVEY.md §8 row a5.

Mirrors reference core/image/cleaning.py: `process_single_bubble` (:210-521), `_build_adaptive_shrink_mask`
(:155-207), `retry_cleaning_with_otsu` (:1051-1140) and `clean_speech_bubbles` (:524-1048, the flat-fill
path).  The reference runs ~10 full-page cv2 passes per bubble on the CPU; here

  * the pixel half (grey conversion, elliptical dilate / erode, fixed or Otsu threshold, 5x5 chamfer distance
    shrink with junction zones) runs for ALL bubbles of a page in one `mtx_bubble_clean` call, each bubble on
    its own crop, reading the page and the (already device-resident) SAM masks in place;
  * the contour half (external contours -> area / centroid filter -> filled union -> largest blob) runs in
    native host code on the small crops (`mtx_host_text_mask`);
  * colour statistics (medians) stay in numpy on the crops.

No cv2 anywhere; there is no CPU fallback for the pixel half (the library raises when the HIP build is absent).
"""
import ctypes as C
import math
from pathlib import Path
from typing import Any, List, Optional, Sequence, Union

import numpy as np
import torch
from PIL import Image, ImageDraw

from ...hip import abi
from ...hip.lib import get_library
from ...utils.exceptions import CleaningError, ImageProcessingError, ValidationError
from ...utils.logging import log_message
from ..scaling import scale_area, scale_kernel, scale_scalar

GRAYSCALE_MIDPOINT = 128
MIN_CONTOUR_AREA = 50
DILATION_KERNEL_SIZE = (7, 7)
EROSION_KERNEL_SIZE = (5, 5)
DISTANCE_TRANSFORM_MASK_SIZE = 5
SOLID_RATIO_THRESHOLD = 0.65
JUNCTION_ADJACENCY_MARGIN = 10
JUNCTION_MIN_SHRINK = 1.0
MAX_ZONES = 8


def ellipse_rows(ksize):
    """(half height r, [half width of row dy = -r..r]) of cv2.getStructuringElement(MORPH_ELLIPSE, ksize)."""
    kw, kh = int(ksize[0]), int(ksize[1])
    r, c = kh // 2, kw // 2
    inv_r2 = 1.0 / (r * r) if r else 0.0
    rows = []
    for i in range(kh):
        dy = i - r
        dx = int(np.rint(c * math.sqrt(max(r * r - dy * dy, 0) * inv_r2)))
        rows.append(min(dx, c))
    return r, rows


def structuring_element(ksize) -> np.ndarray:
    """uint8 kernel array shaped like cv2.getStructuringElement's (what callers pass to process_single_bubble)."""
    r, rows = ellipse_rows(ksize)
    c = int(ksize[0]) // 2
    k = np.zeros((int(ksize[1]), int(ksize[0])), np.uint8)
    for i, dx in enumerate(rows):
        k[i, c - dx:c + dx + 1] = 1
    return k


def _rows_of_kernel(kernel: np.ndarray):
    """inverse of structuring_element for symmetric row-convex kernels"""
    kh, kw = kernel.shape
    c = kw // 2
    rows = []
    for i in range(kh):
        nz = np.nonzero(kernel[i])[0]
        rows.append(int(nz.max() - c) if nz.size else -1)
    return kh // 2, rows


def _normalize_mask(mask: np.ndarray) -> np.ndarray:
    if mask.dtype != np.uint8:
        mask = mask.astype(np.uint8)
    return np.where(mask > 0, 255, 0).astype(np.uint8)


def _bgr_to_gray(bgr: np.ndarray) -> np.ndarray:
    b, g, r = (bgr[..., i].astype(np.int32) for i in range(3))
    return ((b * 1868 + g * 9617 + r * 4899 + (1 << 13)) >> 14).astype(np.uint8)


def _saturation(b: int, g: int, r: int) -> int:
    v, mn = max(b, g, r), min(b, g, r)
    if v == 0:
        return 0
    return ((v - mn) * int(round((255 << 12) / float(v))) + (1 << 11)) >> 12


def _morph_np(src: np.ndarray, rows: Sequence[int], r: int, dilate: bool) -> np.ndarray:
    """small-crop erode / dilate for the colour sampling masks (host side, crops only)"""
    h, w = src.shape
    c = max(rows)
    fill = 0 if dilate else 255
    pad = np.full((h + 2 * r, w + 2 * c), fill, np.uint8)
    pad[r:r + h, c:c + w] = src
    out = np.full((h, w), fill, np.uint8)
    for i, dx in enumerate(rows):
        for j in range(-dx, dx + 1):
            win = pad[i:i + h, c + j:c + j + w]
            out = np.maximum(out, win) if dilate else np.minimum(out, win)
    return out


class _Crops:
    """result planes of one mtx_bubble_clean call, on the host"""

    def __init__(self, rois, offsets, planes, stats):
        self.rois, self.offsets, self.planes, self.stats = rois, offsets, planes, stats

    def plane(self, name, i):
        x0, y0, w, h = self.rois[i]
        o = int(self.offsets[i])
        return self.planes[name][o:o + w * h].reshape(h, w)


def _run_pixel_half(lib, device, page_bgr: np.ndarray, masks, dil, ero, threshold, use_otsu, shrink_px, zones_per_bubble, junction_min):
    H, W = page_bgr.shape[:2]
    dev = torch.device(device)
    m = masks if torch.is_tensor(masks) else torch.from_numpy(np.ascontiguousarray(masks))
    m = m.to(dev)
    if m.dtype != torch.uint8:
        m = m.to(torch.uint8)
    m = m.contiguous()
    n = m.shape[0]
    nz = m != 0
    rows, cols = nz.any(dim=2), nz.any(dim=1)
    y0 = rows.float().argmax(1); y1 = H - 1 - rows.flip(1).float().argmax(1)
    x0 = cols.float().argmax(1); x1 = W - 1 - cols.flip(1).float().argmax(1)
    empty = ~rows.any(dim=1)
    box = torch.stack([x0, y0, x1, y1, empty.long()], 1).cpu().numpy()
    (dil_r, dil_rows), (ero_r, ero_rows) = dil, ero
    mx, my = max(dil_rows) + 2, dil_r + 2
    rois = np.zeros((n, 4), np.int32)
    offsets = np.zeros(n, np.int64)
    total = 0
    for i in range(n):
        if box[i, 4]:
            rois[i] = (0, 0, 1, 1)
        else:
            cx0, cy0 = max(0, int(box[i, 0]) - mx), max(0, int(box[i, 1]) - my)
            cx1, cy1 = min(W, int(box[i, 2]) + mx + 1), min(H, int(box[i, 3]) + my + 1)
            rois[i] = (cx0, cy0, cx1 - cx0, cy1 - cy0)
        offsets[i] = total
        total += int(rois[i, 2]) * int(rois[i, 3])
    page = torch.from_numpy(np.ascontiguousarray(page_bgr[..., :3])).to(dev)
    u8 = lambda: torch.empty(total, dtype=torch.uint8, device=dev)
    planes = {k: u8() for k in ("base", "roi", "eroded", "thresholded", "shrunk")}
    da, db = torch.empty(total, dtype=torch.int32, device=dev), torch.empty(total, dtype=torch.int32, device=dev)
    stats = torch.zeros((n, 260), dtype=torch.int32, device=dev)
    rois_d, off_d = torch.from_numpy(rois).to(dev), torch.from_numpy(offsets).to(dev)
    zones_d = None
    if zones_per_bubble is not None and any(zones_per_bubble):
        z = np.zeros((n, MAX_ZONES, 4), np.int32)
        for i, zs in enumerate(zones_per_bubble):
            for j, q in enumerate(zs[:MAX_ZONES]):
                z[i, j] = q
        zones_d = torch.from_numpy(z).to(dev)
    a = abi.CleanArgs()
    a.page_bgr, a.masks, a.rois, a.offsets = page.data_ptr(), m.data_ptr(), rois_d.data_ptr(), off_d.data_ptr()
    a.base, a.roi, a.eroded, a.thresholded, a.shrunk = (planes[k].data_ptr() for k in ("base", "roi", "eroded", "thresholded", "shrunk"))
    a.dist_a, a.dist_b, a.stats = da.data_ptr(), db.data_ptr(), stats.data_ptr()
    a.zones = zones_d.data_ptr() if zones_d is not None else None
    a.n, a.page_h, a.page_w, a.max_zones = n, H, W, MAX_ZONES
    a.dil_r, a.ero_r = dil_r, ero_r
    for i, v in enumerate(dil_rows):
        a.dil_dx[i] = v
    for i, v in enumerate(ero_rows):
        a.ero_dx[i] = v
    a.threshold, a.use_otsu = int(threshold), int(bool(use_otsu))
    s32, j32 = float(np.float32(shrink_px)), float(np.float32(junction_min))
    a.shrink_fixed, a.junction_fixed = int(math.ceil(s32 * 65536.0)), int(math.ceil(j32 * 65536.0))
    a.sweeps = int(math.ceil(s32)) + 1
    a.max_pixels = int((rois[:, 2].astype(np.int64) * rois[:, 3]).max())
    stream = torch.cuda.current_stream().cuda_stream if (dev.type == "cuda" and not lib.is_simulator) else 0
    lib.check(lib.mtx_bubble_clean(C.byref(a), C.c_void_p(stream)), "mtx_bubble_clean")
    host = {k: v.cpu().numpy() for k, v in planes.items()}
    return _Crops(rois, offsets, host, stats.cpu().numpy()), box


def _junction_zones(bbox, neighbors, scale, W, H):
    margin = max(1, int(round(JUNCTION_ADJACENCY_MARGIN * scale)))
    x1, y1, x2, y2 = bbox
    out = []
    for ox1, oy1, ox2, oy2 in neighbors:
        if x1 - margin > ox2 or ox1 - margin > x2 or y1 - margin > oy2 or oy1 - margin > y2:
            continue
        zx1, zy1 = max(0, max(x1, ox1) - margin), max(0, max(y1, oy1) - margin)
        zx2, zy2 = min(W, min(x2, ox2) + margin), min(H, min(y2, oy2) + margin)
        if zx2 > zx1 and zy2 > zy1:
            out.append((int(zx1), int(zy1), int(zx2), int(zy2)))
    return out


def process_bubbles(page_bgr: np.ndarray, masks, bboxes: Sequence, thresholding_value: int, use_otsu_threshold: bool, roi_shrink_px: float,
                    dilation_kernel=None, constraint_erosion_kernel=None, min_contour_area: float = MIN_CONTOUR_AREA,
                    classify_colored: bool = False, neighbor_bboxes: Optional[Sequence] = None, processing_scale: float = 1.0,
                    device=None, lib=None, verbose: bool = False) -> List[Optional[tuple]]:
    """All bubbles of one page.  masks: [N, H, W] (numpy or device tensor, nonzero = bubble).  Returns per bubble
    `(final_mask, fill_color_bgr, is_colored, sample_color_bgr, text_bbox, text_color_bgr)` like the reference's
    process_single_bubble, or None where the reference would raise CleaningError."""
    lib = lib if lib is not None else get_library()
    if device is None:
        device = "cpu" if lib.is_simulator else "cuda"
    H, W = page_bgr.shape[:2]
    dil = _rows_of_kernel(dilation_kernel) if dilation_kernel is not None else ellipse_rows(DILATION_KERNEL_SIZE)
    ero = _rows_of_kernel(constraint_erosion_kernel) if constraint_erosion_kernel is not None else ellipse_rows(EROSION_KERNEL_SIZE)
    n = len(bboxes)
    zones = None
    if neighbor_bboxes is not None:
        zones = [(_junction_zones(bboxes[i], neighbor_bboxes[i], processing_scale, W, H) if (neighbor_bboxes[i] and bboxes[i] is not None) else [])
                 for i in range(n)]
    jmin = max(1.0, JUNCTION_MIN_SHRINK * processing_scale)
    crops, box = _run_pixel_half(lib, device, page_bgr, masks, dil, ero, thresholding_value, use_otsu_threshold, float(roi_shrink_px), zones, jmin)
    img = np.ascontiguousarray(page_bgr[..., :3])
    out: List[Optional[tuple]] = []
    for i in range(n):
        tag = f"Detection {bboxes[i]}"
        if box[i, 4]:
            log_message(f"Skipping {tag}: empty mask", verbose=verbose)
            out.append(None)
            continue
        x0, y0, w, h = (int(v) for v in crops.rois[i])
        st = crops.stats[i]
        black = bool(st[258])
        fill = (0, 0, 0) if black else (255, 255, 255)
        mean_val = st[256] / max(int(st[257]), 1)
        log_message(f"{tag}: {'Black' if black else 'White'} bubble (mean={mean_val:.1f})", verbose=verbose)
        if use_otsu_threshold:
            log_message(f"  Otsu threshold: {float(st[259])}", verbose=verbose)
        thr, ero_c, shrunk = (np.ascontiguousarray(crops.plane(k, i)) for k in ("thresholded", "eroded", "shrunk"))
        final_c = np.zeros((h, w), np.uint8)
        bb = (C.c_int * 4)()
        nfrag = lib.mtx_host_text_mask(thr.ctypes.data, ero_c.ctypes.data, w, h, x0, y0, W, H, float(min_contour_area), final_c.ctypes.data, bb)
        if nfrag < 0:
            raise CleaningError(f"mtx_host_text_mask failed ({nfrag})")
        log_message(f"{tag}: {max(nfrag, 0)} text fragments found", verbose=verbose)
        if nfrag == 0:
            out.append(None)
            continue
        final_mask = np.zeros((H, W), np.uint8)
        final_mask[y0:y0 + h, x0:x0 + w] = final_c
        text_bbox = (int(bb[0]), int(bb[1]), int(bb[0] + bb[2]), int(bb[1] + bb[3]))
        base_c = crops.plane("base", i)
        crop_bgr = img[y0:y0 + h, x0:x0 + w]
        text_mask = (255 - thr) & shrunk
        is_colored, sample_color = False, fill
        if classify_colored:
            sampling = _morph_np(_morph_np(base_c, ero[1], ero[0], False), ero[1], ero[0], False)
            sampling = sampling.copy()
            sampling[_morph_np(text_mask, [1, 1, 1], 1, True) == 255] = 0
            px = crop_bgr[sampling == 255]
            if px.size == 0:
                px = crop_bgr[base_c == 255]
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
            is_colored = not (solid_ratio >= SOLID_RATIO_THRESHOLD)
            sample_color = fill
            log_message(f"{tag}: {'non-solid/gradient' if is_colored else f'solid color {fill}'} (solid_ratio={solid_ratio:.2f})", verbose=verbose)
        text_color = None
        tp = crop_bgr[_morph_np(text_mask, [1, 1, 1], 1, False) == 255]
        if tp.size == 0:
            tp = crop_bgr[text_mask == 255]
        if tp.size > 0:
            sb = tuple(int(t) for t in np.median(tp, axis=0).astype(int))
            if _saturation(*sb) < 25:
                lum = 0.114 * fill[0] + 0.587 * fill[1] + 0.299 * fill[2]
                text_color = (0, 0, 0) if lum >= 128 else (255, 255, 255)
            else:
                text_color = sb
        out.append((final_mask, fill, is_colored, sample_color, text_bbox, text_color))
    return out


def process_single_bubble(base_mask, img_gray, img_height, img_width, thresholding_value, use_otsu_threshold, roi_shrink_px, verbose,
                          detection_bbox=None, is_sam=False, dilation_kernel=None, constraint_erosion_kernel=None,
                          min_contour_area: float = MIN_CONTOUR_AREA, classify_colored: bool = False, neighbor_bboxes: Optional[list] = None,
                          processing_scale: float = 1.0, image_bgr: Optional[np.ndarray] = None, device=None, lib=None):
    """Reference signature (cleaning.py:210-227).  `image_bgr` is what the kernels read; when only the grey page is
    given it is replicated into three channels (BGR2GRAY of equal channels is the identity)."""
    try:
        page = image_bgr if image_bgr is not None else np.repeat(np.asarray(img_gray)[..., None], 3, axis=2)
        res = process_bubbles(page, _normalize_mask(np.asarray(base_mask))[None], [detection_bbox], thresholding_value, use_otsu_threshold,
                              roi_shrink_px, dilation_kernel, constraint_erosion_kernel, min_contour_area, classify_colored,
                              [neighbor_bboxes] if neighbor_bboxes else None, processing_scale, device=device, lib=lib, verbose=verbose)[0]
    except CleaningError:
        raise
    except Exception as e:
        log_message(f"Failed to process {'SAM' if is_sam else 'YOLO'} mask for {detection_bbox}", always_print=True)
        raise CleaningError("Failed to process bubble mask") from e
    if res is None:
        log_message(f"Failed to process {'SAM' if is_sam else 'YOLO'} mask for {detection_bbox}", always_print=True)
        raise CleaningError("Failed to process bubble mask")
    return res


def _polygon_mask(points, H, W) -> np.ndarray:
    pts = np.round(np.asarray(points, np.float32).reshape(-1, 2)).astype(int)
    im = Image.new("L", (W, H), 0)
    ImageDraw.Draw(im).polygon([tuple(int(v) for v in p) for p in pts], fill=255, outline=255)
    return np.asarray(im, np.uint8)


def clean_speech_bubbles(image_input: Union[str, Path, Image.Image], model_path, confidence=0.6, pre_computed_detections=None, device=None,
                         thresholding_value: int = 200, use_otsu_threshold: bool = False, roi_shrink_px: int = 5, verbose: bool = False,
                         processing_scale: float = 1.0, conjoined_confidence=0.35, inpaint_colored_bubbles: bool = False,
                         bubble_detector_model: str = "yolo_2", request_coordinator: Optional[Any] = None, lib=None, **flux_options):
    """-> (cleaned BGR[A] ndarray, list of per-bubble dicts) like the reference (:524-1048).  With `inpaint_colored_bubbles` the bubbles
    whose interior is not a flat colour are repainted by the configured FLUX inpainter on their text mask only (`flux_options`: the
    reference's `inpaint_method`, `flux_*` keyword arguments, :856-1015) — through the request coordinator in waves of non-overlapping
    context boxes when there are several; every other processed bubble takes the grouped flat fill."""
    try:
        if isinstance(image_input, (str, Path)):
            pil_image, image_path = Image.open(image_input), image_input
        else:
            pil_image, image_path = image_input, None
        arr = np.asarray(pil_image if pil_image.mode in ("RGB", "RGBA") else pil_image.convert("RGB"))
        image = np.ascontiguousarray(arr[..., [2, 1, 0] + ([3] if arr.shape[2] == 4 else [])])      # pil_to_cv2: RGB[A] -> BGR[A]
        H, W = image.shape[:2]
        cleaned = image.copy()
        if pre_computed_detections is not None:
            detections = pre_computed_detections
        elif image_path is not None:
            from .detection import detect_speech_bubbles
            res = detect_speech_bubbles(image_path, model_path, confidence, device=device, conjoined_confidence=conjoined_confidence,
                                        bubble_detector_model=bubble_detector_model)
            detections = res[0] if isinstance(res, tuple) else res
        else:
            raise ValidationError("Bubble detection requires an image path, but an image object was provided without pre-computed detections.")
        shrink = float(scale_scalar(roi_shrink_px, processing_scale, minimum=0.0, maximum=64.0))
        dil_k = structuring_element(scale_kernel(DILATION_KERNEL_SIZE, processing_scale))
        ero_k = structuring_element(scale_kernel(EROSION_KERNEL_SIZE, processing_scale))
        min_area = scale_area(MIN_CONTOUR_AREA, processing_scale, minimum=MIN_CONTOUR_AREA, maximum=5000)
        cand = []
        for det in detections:
            sam = det.get("sam_mask")
            if sam is not None:
                cand.append((det, _normalize_mask(np.asarray(sam)), True))
            elif det.get("mask_points"):
                pts = np.asarray(det["mask_points"], np.float32)
                if not ((pts.ndim == 3 and pts.shape[1] == 1) or (pts.ndim == 2 and pts.shape[1] == 2)):
                    log_message(f"Skipping detection {det.get('bbox')}: invalid mask format", verbose=verbose)
                    continue
                cand.append((det, _polygon_mask(pts, H, W), False))
            else:
                log_message(f"Skipping detection {det.get('bbox')}: no mask points", verbose=verbose)
        processed = []
        if cand:
            def run(items, otsu, shrink_px):
                return process_bubbles(image, np.stack([c[1] for c in items]), [c[0].get("bbox") for c in items], thresholding_value, otsu, shrink_px,
                                       dil_k, ero_k, min_area, inpaint_colored_bubbles, [c[0].get("conjoined_neighbor_bboxes") for c in items],
                                       processing_scale, device=device, lib=lib, verbose=verbose)
            results = run(cand, use_otsu_threshold, shrink)
            retry = [i for i, r in enumerate(results) if r is None] if not use_otsu_threshold else []
            if retry:           # reference: retry_cleaning_with_otsu per failed bubble (:1051-1140)
                for i in retry:
                    log_message(f"Standard cleaning failed for {cand[i][0].get('bbox')}, retrying with Otsu...", verbose=verbose)
                again = run([cand[i] for i in retry], True, shrink)
                for i, r in zip(retry, again):
                    results[i] = r
                    log_message(f"Otsu retry {'successful' if r is not None else 'failed'} for {cand[i][0].get('bbox')}", verbose=verbose)
            for (det, base, is_sam), r in zip(cand, results):
                if r is None:
                    log_message(f"Error processing {'SAM' if is_sam else 'YOLO'} mask for detection {det.get('bbox')}", always_print=True)
                    continue
                final_mask, fill, is_colored, sample, text_bbox, text_color = r
                processed.append({"mask": final_mask, "base_mask": base, "color": sample if sample else fill, "bbox": det.get("bbox"),
                                  "is_colored": is_colored, "text_bbox": text_bbox, "text_color_bgr": text_color, "is_sam": is_sam, "inpainted": False})
                log_message(f"Detection {det.get('bbox')}: processed successfully", verbose=verbose)
        method = flux_options.get("inpaint_method", "flux_kontext")
        colored = [b for b in processed if b.get("is_colored", False)]
        if inpaint_colored_bubbles and method not in ("opencv", "none") and colored:
            cleaned = _repaint_colored_bubbles(cleaned, colored, method, device, request_coordinator, flux_options, verbose)
        groups = {}
        for b in processed:
            if not b.get("inpainted", False):
                groups.setdefault(b["color"], []).append(b["mask"])
        for color, ms in groups.items():
            combined = np.bitwise_or.reduce(ms)
            if cleaned.shape[2] == 4:
                cleaned[combined == 255, :3] = color
            else:
                cleaned[combined == 255] = color
        log_message(f"Cleaned {len(processed)} speech bubbles", always_print=True)
        return cleaned, processed
    except IOError as e:
        raise ImageProcessingError(f"Error loading image {image_input}: {str(e)}")
    except (ValidationError, ImageProcessingError):
        raise
    except Exception as e:
        raise CleaningError(f"Error cleaning speech bubbles: {str(e)}")


def _repaint_colored_bubbles(cleaned: np.ndarray, colored: list, method: str, device, coordinator, opt: dict, verbose: bool) -> np.ndarray:
    """FLUX repaint of the text of non-flat bubbles (reference :63-152, :856-1015).  A failure of one bubble leaves it to the flat fill;
    a failure of the whole step leaves every bubble to it.  The reference's round trip of intermediate pages through temporary PNG files
    is a memory measure with no effect on the pixels and has no counterpart here."""
    import random
    from ..batch_coordinator import expanded_mask_bbox, partition_non_overlapping_waves, paste_image_region
    from .inpainting import FluxKleinInpainter, FluxKontextInpainter
    log_message(f"Inpainting {len(colored)} colored bubbles with Flux", always_print=True)
    working = Image.fromarray(np.ascontiguousarray(cleaned[..., 2::-1]))          # BGR[A] -> RGB
    seed = opt.get("flux_seed", 1)
    base_seed = random.randint(1, 999999) if seed == -1 else max(0, int(seed))
    try:
        backend = opt.get("flux_backend", "sdnq")
        if method in ("flux_klein_9b", "flux_klein_4b"):
            inpainter = FluxKleinInpainter(variant=method[-2:], device=device, huggingface_token=opt.get("flux_hf_token", ""),
                                           num_inference_steps=int(opt.get("flux_num_inference_steps", 8)), low_vram=opt.get("flux_low_vram", False),
                                           luminance_correction=opt.get("flux_luminance_correction", True),
                                           upscale_small_crops=opt.get("flux_upscale_small_crops", True), backend=backend,
                                           sdcpp_cache_mode=opt.get("flux_sdcpp_cache_mode", "none"),
                                           sdcpp_diffusion_quant=opt.get("flux_sdcpp_diffusion_quant", "Q4_K_M"),
                                           sdcpp_text_encoder_quant=opt.get("flux_sdcpp_text_encoder_quant", ""), verbose=verbose)
        else:
            inpainter = FluxKontextInpainter(device=device, huggingface_token=opt.get("flux_hf_token", ""),
                                             num_inference_steps=int(opt.get("flux_num_inference_steps", 8)),
                                             residual_diff_threshold=float(opt.get("flux_residual_diff_threshold", 0.15)), backend=backend,
                                             low_vram=opt.get("flux_low_vram", False) if backend == "sdnq" else False,
                                             sdcpp_cache_mode=opt.get("flux_sdcpp_cache_mode", "none"),
                                             sdcpp_diffusion_quant=opt.get("flux_sdcpp_diffusion_quant", "Q4_K_M"),
                                             sdcpp_text_encoder_quant=opt.get("flux_sdcpp_text_encoder_quant", ""))

        def resample(page, info, mask):           # the bubble's colour for the renderer: mean grey of the repainted pixels (:50-60)
            px = np.asarray(page.convert("RGB"))[..., ::-1][mask]
            if px.size > 0:
                v = int(np.clip(np.mean(px), 0, 255))
                info["color"] = (v, v, v)

        if coordinator is not None and len(colored) > 1:
            cands = [dict(info=b, mask=b["mask"].astype(bool), seed=base_seed + i if base_seed > 0 else base_seed, bbox=b.get("bbox"),
                          context=expanded_mask_bbox(b["mask"].astype(bool), working.size)) for i, b in enumerate(colored)]
            waves = partition_non_overlapping_waves(cands, lambda c: c["context"])
            log_message(f"Scheduling colored-bubble Flux in {len(waves)} wave(s)", verbose=verbose)
            for wave in waves:
                base = working

                def make_job(c):
                    def job():
                        try:
                            return c, inpainter.inpaint_mask(base.copy(), c["mask"], seed=c["seed"], verbose=verbose,
                                                             ocr_params={"type": "colored_bubble", "bbox": c["bbox"]}), None
                        except Exception as e:      # noqa: BLE001
                            return c, None, e
                    return job
                for c, image, err in coordinator.map_ordered([make_job(c) for c in wave]):
                    if err is not None:
                        log_message(f"Flux inpainting failed for bubble {c['bbox']}: {err}; falling back to standard fill", always_print=True)
                        continue
                    working = image if c["context"] is None else paste_image_region(working, image, c["context"])
                    c["info"]["inpainted"] = True
                    resample(working, c["info"], c["mask"])
        else:
            for i, b in enumerate(colored):
                mask = b["mask"].astype(bool)
                kw = dict(seed=base_seed + i if base_seed > 0 else base_seed, verbose=verbose, ocr_params={"type": "colored_bubble", "bbox": b.get("bbox")})
                try:
                    working = coordinator.run(inpainter.inpaint_mask, working, mask, **kw) if coordinator is not None else inpainter.inpaint_mask(working, mask, **kw)
                    b["inpainted"] = True
                    resample(working, b, mask)
                except Exception as e:      # noqa: BLE001
                    log_message(f"Flux inpainting failed for bubble {b.get('bbox')}: {e}; falling back to standard fill", always_print=True)
        return np.ascontiguousarray(np.asarray(working.convert("RGB"))[..., ::-1])       # BGR, 3 channels from here on (as the reference, :997-999)
    except Exception as e:      # noqa: BLE001
        log_message(f"Flux inpainting aborted; falling back to standard fill: {e}", always_print=True)
        return cleaned


def retry_cleaning_with_otsu(image, bubble_info, thresholding_value, roi_shrink_px, processing_scale: float = 1.0, verbose: bool = False,
                             classify_colored: bool = False, device=None, lib=None):
    """One bubble again with Otsu's threshold (reference :1051-1140) -> dict(mask, color, is_colored, text_bbox, text_color_bgr) or None."""
    shrink = float(scale_scalar(roi_shrink_px, processing_scale, minimum=0.0, maximum=64.0))
    dil_k = structuring_element(scale_kernel(DILATION_KERNEL_SIZE, processing_scale))
    ero_k = structuring_element(scale_kernel(EROSION_KERNEL_SIZE, processing_scale))
    min_area = scale_area(MIN_CONTOUR_AREA, processing_scale, minimum=MIN_CONTOUR_AREA, maximum=5000)
    nb = bubble_info.get("neighbor_bboxes")
    r = process_bubbles(np.asarray(image), _normalize_mask(bubble_info["base_mask"])[None], [bubble_info.get("bbox")], thresholding_value, True, shrink,
                        dil_k, ero_k, min_area, classify_colored, [nb] if nb else None, processing_scale, device=device, lib=lib, verbose=verbose)[0]
    if r is None:
        return None
    return {"mask": r[0], "color": r[3] if r[3] else r[1], "is_colored": r[2], "text_bbox": r[4], "text_color_bgr": r[5]}
