"""OSB (outside-speech-bubble) text stage of a page — SURVEY.md §8 row f3, second half: the caller that turns the OSB text
regions into FLUX requests or flat fills.  Mirrors core/outside_text_processor.py of the reference:

    prepare_outside_text_work   :217-636   detect -> min-area filter -> render expansion of narrow / tiny boxes against
                                           obstacles -> union of the bubble masks, dilated 11 x 11 -> background-brightness
                                           probe per box (2-means) -> grouped region masks
    finish_outside_text_work    :638-1691  per region: mask minus bubbles; border ring around the text box solid
                                           (>= 95 % of the ring within +-15 of its median)?  -> flat fill with the ring's
                                           (white/black-snapped) median, else FLUX — queued and run in waves of regions
                                           whose context boxes do not overlap (batch_coordinator), each result pasted back
                                           through its context box; a failed FLUX call degrades to the fallback colour fill
    process_outside_text        :1694-1737

What is NOT here (outside the vision hot path, SURVEY.md §8 "out of scope"): the translation payload
(`_build_outside_text_data`: base64 crops for the LLM), the rendered-text colour extraction (LAB contrast mask -> HSV,
reference :1096-1165, which only feeds the text renderer), page-number filtering by OCR (:281-347; refused when enabled), the
Klein / sd.cpp / nunchaku backends and `flux_group_regions`.  The returned `outside_text_data` is therefore an empty list.

All decisions are integer / uint8 arithmetic on the page (bit-exact target; pinned by tests/golden/osb_stage.json, generated
by running the reference functions on the same page with a deterministic stand-in inpainter).
"""
import random
from dataclasses import dataclass, field
from pathlib import Path
from typing import Any, Dict, List, Optional, Tuple, Union

import numpy as np
from PIL import Image
from scipy import ndimage

from ..utils.exceptions import ValidationError
from ..utils.logging import log_message
from .batch_coordinator import expanded_mask_bbox, partition_non_overlapping_waves, paste_image_region
from .image.inpainting import FluxKleinInpainter, FluxKontextInpainter
from .image.ocr_detection import OutsideTextDetector

OSB_EXPANSION_PIXEL_BUFFER = 5       # kept clear around bubbles, other OSB regions and panel borders
OSB_SOLID_RATIO_THRESHOLD = 0.95
OSB_COLOR_TOLERANCE = 15
OSB_EXPANSION_PX = 2                 # width of the border ring sampled around a text box
OSB_WHITE_SNAP_THRESH = 245
OSB_BLACK_SNAP_THRESH = 10
BUBBLE_GUARD_KERNEL = 11             # the bubble mask is dilated by an 11 x 11 square before it is subtracted


@dataclass
class OutsideTextWork:
    """state between `prepare` and `finish` (reference :39-58; the translation-side fields are not carried)"""
    pil_image: Image.Image
    config: Any
    image_path: Union[str, Path]
    image_format: Optional[str]
    verbose: bool
    outside_text_results: list
    raw_outside_text_results: list
    original_text_colors: dict
    total_bubble_mask: Any
    outside_detector: Any
    mask_groups: list
    img_w: int
    img_h: int
    outside_text_data: List[Dict[str, Any]] = field(default_factory=list)


def background_is_dark(pixels: np.ndarray) -> bool:
    """2-means over the box's pixels; the larger cluster is the background; BT.601 luma < 128 (reference :558-582, same
    scikit-learn call so the clustering is the reference's own)"""
    from sklearn.cluster import KMeans
    from threadpoolctl import threadpool_limits
    kmeans = KMeans(n_clusters=2, random_state=42, n_init=10)
    with threadpool_limits(limits=1):        # ~20 k points: one thread takes ~40 ms, the default pool spins for > 1 s on a many-core host
        kmeans.fit(pixels)
    unique, counts = np.unique(kmeans.labels_, return_counts=True)
    bg = kmeans.cluster_centers_[unique[np.argmax(counts)]]
    return bool(0.299 * bg[0] + 0.587 * bg[1] + 0.114 * bg[2] < 128)


def expand_render_boxes(results, img_w: int, img_h: int, narrow_mult: float, tiny_mult: float, aspect_thr: float, area_thr: float,
                        bubble_data=None, panels=None):
    """narrow/tall or tiny text boxes are scaled about their centre, kept inside their panel and pulled back from bubbles and the
    other boxes (reference :351-505).  Only called when a multiplier > 1 is configured."""
    buffer = OSB_EXPANSION_PIXEL_BUFFER
    out = []
    for i, (bbox, conf) in enumerate(results):
        x1, y1, x2, y2 = bbox
        w, h = x2 - x1, y2 - y1
        mult = 1.0
        if float(w) / float(max(1, h)) <= aspect_thr:
            mult = max(mult, narrow_mult)
        if (w * h) / float(max(1, img_w * img_h)) < area_thr:
            mult = max(mult, tiny_mult)
        if mult <= 1.0:
            out.append(([int(x1), int(y1), int(x2), int(y2)], conf))
            continue
        cx, cy = x1 + w / 2, y1 + h / 2
        nx1, ny1 = max(0, int(cx - w * mult / 2)), max(0, int(cy - h * mult / 2))
        nx2, ny2 = min(img_w, int(cx + w * mult / 2)), min(img_h, int(cy + h * mult / 2))
        if panels:
            panel = next((p for p in panels if p[0] <= cx <= p[2] and p[1] <= cy <= p[3]), None)
            if panel:
                px1, py1, px2, py2 = panel
                nx1, ny1 = max(min(int(px1) + buffer, int(px2)), nx1), max(min(int(py1) + buffer, int(py2)), ny1)
                nx2, ny2 = min(max(int(px2) - buffer, int(px1)), nx2), min(max(int(py2) - buffer, int(py1)), ny2)
        obstacles = []
        for b in bubble_data or []:
            bb = b.get("bbox")
            if bb and len(bb) == 4:
                bx1, by1, bx2, by2 = [int(c) for c in bb]
                obstacles.append((max(0, bx1 - buffer), max(0, by1 - buffer), min(img_w, bx2 + buffer), min(img_h, by2 + buffer)))
        for j, other in enumerate(results):
            if i == j:
                continue
            ob = out[j][0] if j < i else other[0]
            ox1, oy1, ox2, oy2 = [int(c) for c in ob]
            obstacles.append((max(0, ox1 - buffer), max(0, oy1 - buffer), min(img_w, ox2 + buffer), min(img_h, oy2 + buffer)))
        inf = float("inf")
        for ox1, oy1, ox2, oy2 in obstacles:
            if nx2 <= ox1 or nx1 >= ox2 or ny2 <= oy1 or ny1 >= oy2:
                continue
            r_x2 = (nx2 - ox1) if ox1 >= x2 else inf
            r_x1 = (ox2 - nx1) if ox2 <= x1 else inf
            r_y2 = (ny2 - oy1) if oy1 >= y2 else inf
            r_y1 = (oy2 - ny1) if oy2 <= y1 else inf
            m = min(r_x2, r_x1, r_y2, r_y1)
            if m == inf:
                continue
            if m == r_x2:
                nx2 = ox1
            elif m == r_x1:
                nx1 = ox2
            elif m == r_y2:
                ny2 = oy1
            else:
                ny1 = oy2
        out.append(([min(nx1, int(x1)), min(ny1, int(y1)), max(nx2, int(x2)), max(ny2, int(y2))], conf))
    return out


def build_bubble_guard_mask(bubble_data, img_w: int, img_h: int, verbose: bool = False) -> np.ndarray:
    """union of the bubbles' masks (their boxes where a mask is missing), dilated by an 11 x 11 square (reference :507-543;
    cv2.dilate with its default constant border == a maximum filter with zero padding)"""
    total = np.zeros((img_h, img_w), dtype=bool)
    r = BUBBLE_GUARD_KERNEL // 2

    def grow(window_mask, x0, y0):
        """OR the dilation of one bubble into `total`: dilation distributes over the union, and a bubble's dilation is confined to its
        bounding window grown by the kernel radius — same pixels as one maximum filter over the whole page (134 ms at 6 MP), at the
        cost of the bubbles' own areas"""
        ys, xs = np.flatnonzero(window_mask.any(axis=1)), np.flatnonzero(window_mask.any(axis=0))
        if ys.size == 0:
            return
        wy0, wy1, wx0, wx1 = int(ys[0]), int(ys[-1]) + 1, int(xs[0]), int(xs[-1]) + 1
        gy0, gy1 = max(0, y0 + wy0 - r), min(img_h, y0 + wy1 + r)
        gx0, gx1 = max(0, x0 + wx0 - r), min(img_w, x0 + wx1 + r)
        win = np.zeros((gy1 - gy0, gx1 - gx0), np.uint8)
        win[y0 + wy0 - gy0: y0 + wy1 - gy0, x0 + wx0 - gx0: x0 + wx1 - gx0] = window_mask[wy0:wy1, wx0:wx1]
        total[gy0:gy1, gx0:gx1] |= ndimage.maximum_filter(win, size=BUBBLE_GUARD_KERNEL, mode="constant", cval=0).astype(bool)

    for bubble in bubble_data or []:
        try:
            mask = bubble.get("sam_mask") if isinstance(bubble, dict) else None
            if mask is not None:
                m = np.asarray(mask)
                if m.ndim == 3:
                    m = m[..., 0]
                if m.shape[0] == img_h and m.shape[1] == img_w:
                    grow(m > 0, 0, 0)
                    continue
            bbox = bubble.get("bbox") if isinstance(bubble, dict) else None
            if bbox and len(bbox) == 4:
                x0, y0, x1, y1 = [int(c) for c in bbox]
                x0, x1 = max(0, min(img_w, x0)), max(0, min(img_w, x1))
                y0, y1 = max(0, min(img_h, y0)), max(0, min(img_h, y1))
                if x1 > x0 and y1 > y0:                      # a rectangle dilated by a square is the rectangle grown by its radius
                    total[max(0, y0 - r): min(img_h, y1 + r), max(0, x0 - r): min(img_w, x1 + r)] = True
        except Exception as e:
            log_message(f"Warning: Failed to apply bubble mask for OSB exclusion: {e}", verbose=verbose)
    return total


def prepare_outside_text_work(pil_image: Image.Image, config, image_path, image_format: Optional[str], verbose: bool = False,
                              bubble_data: Optional[List[Dict[str, Any]]] = None, text_free_boxes: Optional[List[List[float]]] = None,
                              panels: Optional[List[Tuple[int, int, int, int]]] = None) -> Optional[OutsideTextWork]:
    ot = config.outside_text
    if not ot.enabled:
        return None
    if getattr(ot, "enable_page_number_filtering", False):
        raise ValidationError("enable_page_number_filtering needs the OCR recogniser, which is outside this build's hot path")
    log_message("Detecting text outside speech bubbles...", verbose=verbose)
    try:
        detector = OutsideTextDetector(device=config.device, hf_token=ot.huggingface_token)
        results = detector.detect_outside_text(str(image_path), yolo_model_path=getattr(config, "yolo_model_path", None), confidence=ot.osb_confidence,
                                               conjoined_confidence=config.detection.conjoined_confidence, verbose=verbose, image_override=pil_image,
                                               existing_bubbles=bubble_data, text_free_boxes=text_free_boxes,
                                               bubble_detector_model=config.detection.bubble_detector_model,
                                               min_area_ignore_ratio=ot.min_area_ignore_ratio, text_free_only=ot.osb_text_free_only)
        if not results:
            log_message("No outside text regions found", verbose=verbose)
            return None
        img_w, img_h = pil_image.size
        min_ignore = max(0.0, min(0.05, ot.min_area_ignore_ratio))
        if min_ignore > 0.0:
            image_area = float(img_w * img_h)
            results = [(b, c) for b, c in results if ((b[2] - b[0]) * (b[3] - b[1])) / max(1.0, image_area) >= min_ignore]
            if not results:
                log_message("No outside text regions remaining after min area filter", verbose=verbose)
                return None
        raw_results = results.copy()

        narrow = getattr(ot, "osb_render_expansion_narrow_multiplier", 1.0)
        tiny = getattr(ot, "osb_render_expansion_tiny_multiplier", 1.0)
        if max(narrow, tiny) > 1.0:
            results = expand_render_boxes(results, img_w, img_h, narrow, tiny, getattr(ot, "osb_render_expansion_aspect_ratio_threshold", 0.4),
                                          getattr(ot, "osb_render_expansion_area_ratio_threshold", 0.005), bubble_data, panels)

        total_bubble_mask = build_bubble_guard_mask(bubble_data, img_w, img_h, verbose)

        original_text_colors = {}
        for bbox, _ in raw_results:
            x1, y1, x2, y2 = [int(c) for c in bbox]
            arr = np.array(pil_image.crop((x1, y1, x2, y2)))
            if arr.shape[-1] == 4:
                arr = arr[..., :3]
            original_text_colors[(x1, y1, x2, y2)] = background_is_dark(arr.reshape(-1, 3))

        mask_groups, _ = detector.get_text_masks(str(image_path), bbox_expansion_percent_width=ot.bbox_expansion_percent_width,
                                                 bbox_expansion_percent_height=ot.bbox_expansion_percent_height,
                                                 text_box_proximity_ratio=ot.text_box_proximity_ratio, verbose=verbose, image_override=pil_image,
                                                 existing_results=raw_results)
        return OutsideTextWork(pil_image=pil_image, config=config, image_path=image_path, image_format=image_format, verbose=verbose,
                               outside_text_results=results, raw_outside_text_results=raw_results, original_text_colors=original_text_colors,
                               total_bubble_mask=total_bubble_mask, outside_detector=detector, mask_groups=mask_groups, img_w=img_w, img_h=img_h)
    except ValidationError:
        raise
    except Exception as e:
        log_message(f"Error during outside text detection: {e}", always_print=True)
        return None


def _snap(med) -> Tuple[int, int, int]:
    if med[0] >= OSB_WHITE_SNAP_THRESH and med[1] >= OSB_WHITE_SNAP_THRESH and med[2] >= OSB_WHITE_SNAP_THRESH:
        return (255, 255, 255)
    if med[0] <= OSB_BLACK_SNAP_THRESH and med[1] <= OSB_BLACK_SNAP_THRESH and med[2] <= OSB_BLACK_SNAP_THRESH:
        return (0, 0, 0)
    return (int(med[0]), int(med[1]), int(med[2]))


def border_ring_pixels(image: Image.Image, x0: int, y0: int, x1: int, y1: int, img_w: int, img_h: int) -> Optional[np.ndarray]:
    """RGB pixels of the OSB_EXPANSION_PX-wide ring around box (x0, y0, x1, y1), clipped to the page; None if the window is empty"""
    sx1, sy1 = max(0, x0 - OSB_EXPANSION_PX), max(0, y0 - OSB_EXPANSION_PX)
    sx2, sy2 = min(img_w, x1 + OSB_EXPANSION_PX), min(img_h, y1 + OSB_EXPANSION_PX)
    if sx2 <= sx1 or sy2 <= sy1:
        return None
    ring = np.ones((sy2 - sy1, sx2 - sx1), dtype=bool)
    lx0, ly0 = max(0, x0 - sx1), max(0, y0 - sy1)
    lx1, ly1 = min(sx2 - sx1, x1 - sx1), min(sy2 - sy1, y1 - sy1)
    if lx1 > lx0 and ly1 > ly0:
        ring[ly0:ly1, lx0:lx1] = False
    crop = np.array(image.crop((sx1, sy1, sx2, sy2)).convert("RGB"))
    return crop[ring]


def ring_statistics(pixels: np.ndarray) -> Tuple[bool, Tuple[int, int, int]]:
    """(solid?, snapped median colour) of a border ring (reference :1167-1199)"""
    med = np.median(pixels, axis=0).astype(int)
    diffs = np.max(np.abs(pixels.astype(int) - med), axis=1)
    return float(np.mean(diffs <= OSB_COLOR_TOLERANCE)) >= OSB_SOLID_RATIO_THRESHOLD, _snap(med)


def _bounds_from(results, indices, img_w=None, img_h=None):
    x0 = int(min(results[i][0][0] for i in indices)); y0 = int(min(results[i][0][1] for i in indices))
    x1 = int(max(results[i][0][2] for i in indices)); y1 = int(max(results[i][0][3] for i in indices))
    if img_w is not None:
        x0, y0, x1, y1 = max(0, x0), max(0, y0), min(img_w, x1), min(img_h, y1)
    return x0, y0, x1, y1


def finish_outside_text_work(work: OutsideTextWork) -> Tuple[Image.Image, List[Dict[str, Any]]]:
    pil_image, config, verbose = work.pil_image, work.config, work.verbose
    results, raw_results = work.outside_text_results, work.raw_outside_text_results
    total_bubble_mask, img_w, img_h = work.total_bubble_mask, work.img_w, work.img_h
    ot = config.outside_text
    current_image = pil_image
    try:
        method = ot.inpainting_method
        inpainter = None
        if method == "flux_kontext":
            try:
                inpainter = FluxKontextInpainter(device=config.device, huggingface_token=ot.huggingface_token,
                                                 num_inference_steps=ot.flux_num_inference_steps,
                                                 residual_diff_threshold=ot.flux_residual_diff_threshold, backend=ot.flux_backend,
                                                 low_vram=ot.flux_low_vram if ot.flux_backend == "sdnq" else False)
            except Exception as e:
                log_message(f"Flux Kontext unavailable ({e}), falling back to OpenCV", verbose=verbose)
        elif method in ("flux_klein_9b", "flux_klein_4b"):          # the reference's default (core/config.py:136-144; :664-718)
            try:
                g = lambda name, default: getattr(ot, name, default)
                inpainter = FluxKleinInpainter(variant=method[-2:], device=config.device, huggingface_token=ot.huggingface_token,
                                               num_inference_steps=ot.flux_num_inference_steps, low_vram=g("flux_low_vram", False),
                                               luminance_correction=g("flux_luminance_correction", True),
                                               upscale_small_crops=g("flux_upscale_small_crops", True), backend=g("flux_backend", "sdnq"),
                                               sdcpp_cache_mode=g("flux_sdcpp_cache_mode", "none"),
                                               sdcpp_diffusion_quant=g("flux_sdcpp_diffusion_quant", ""),
                                               sdcpp_text_encoder_quant=g("flux_sdcpp_text_encoder_quant", ""), verbose=verbose)
                log_message(f"Using Flux.2 Klein {method[-2:].upper()} for inpainting", verbose=verbose)
            except Exception as e:
                log_message(f"Flux.2 Klein unavailable ({e}), falling back to OpenCV", verbose=verbose)
        if method == "none" or method == "opencv" or inpainter is None:
            inpainter = None
        if not work.mask_groups:
            return current_image, work.outside_text_data

        base_seed = random.randint(1, 999999) if ot.seed == -1 else ot.seed
        coordinator = getattr(config, "request_coordinator", None)
        if getattr(ot, "flux_group_regions", False):
            log_message("flux_group_regions is not built; regions are inpainted one by one", always_print=True)
        flux_inpaints = cv2_inpaints = none_skips = 0
        pending: List[dict] = []

        def fill_bounds(group, original_bounds):
            indices = group.get("mask_indices", [])
            if indices and results:
                return _bounds_from(results, indices, img_w, img_h)
            return original_bounds

        def simple_fill(image, group, combined_mask, original_bounds, color):
            """flat fill of the (render-expanded) text rectangle minus the bubbles (reference :1349-1437 / :777-861)"""
            new_img = image.copy()
            b = fill_bounds(group, original_bounds)
            if b is None or any(v is None for v in b):
                new_img.paste(Image.new("RGB", new_img.size, color), (0, 0), mask=Image.fromarray((combined_mask * 255).astype(np.uint8), mode="L"))
                return new_img
            x0, y0, x1, y1 = b
            if x1 > x0 and y1 > y0:
                region = np.logical_not(total_bubble_mask[y0:y1, x0:x1])
                if np.any(region):
                    new_img.paste(Image.new("RGB", (x1 - x0, y1 - y0), color), (x0, y0), mask=Image.fromarray((region * 255).astype(np.uint8), mode="L"))
            return new_img

        def flush():
            """run the queued FLUX regions in waves of non-overlapping context boxes (reference :863-949)"""
            nonlocal current_image, flux_inpaints, cv2_inpaints
            if not pending:
                return
            candidates = list(pending)
            pending.clear()
            waves = partition_non_overlapping_waves(candidates, lambda c: c["context_bbox"])
            log_message(f"Scheduling OSB Flux in {len(waves)} wave(s)", verbose=verbose)
            for wave in waves:
                base_image = current_image

                def make_job(c):
                    def job():
                        page = base_image.copy()
                        try:
                            out = inpainter.inpaint_mask(page, c["mask"], seed=c["seed"], verbose=verbose, strict_mask_clipping=True,
                                                         composite_clip_bbox=c["composite_clip_bbox"])
                            if out is page:
                                raise RuntimeError("Flux returned original image (no inpaint)")
                            return {"candidate": c, "image": out, "error": None}
                        except Exception as e:
                            return {"candidate": c, "image": None, "error": e}
                    return job

                for res in coordinator.map_ordered([make_job(c) for c in wave]):
                    c = res["candidate"]
                    if res["error"] is not None:
                        log_message(f"Flux failed for OSB region {c['index']} (Flux inpainting error: {res['error']}); "
                                    f"falling back to CV2 fill ({c['fallback_color']})", always_print=True)
                        current_image = simple_fill(current_image, c["group"], c["mask"], c["original_bounds"], c["fallback_color"])
                        cv2_inpaints += 1
                        continue
                    current_image = res["image"] if c["context_bbox"] is None else paste_image_region(current_image, res["image"], c["context_bbox"])
                    flux_inpaints += 1

        for i, group in enumerate(work.mask_groups):
            combined_mask = np.logical_and(group["combined_mask"], np.logical_not(total_bubble_mask))
            if not np.any(combined_mask):
                log_message("Skipping outside text region after bubble masking (no remaining area)", verbose=verbose)
                continue
            region_seed = base_seed + i if base_seed > 0 else base_seed
            ob = group.get("original_bbox")
            clip_bbox = fill_color = fallback_color = None
            original_bounds = None
            if ob:
                ox, oy, ow, oh = int(ob.get("x", 0)), int(ob.get("y", 0)), int(ob.get("width", 0)), int(ob.get("height", 0))
                if ow > 0 and oh > 0:
                    original_bounds = (max(0, min(img_w, ox)), max(0, min(img_h, oy)), max(0, min(img_w, ox + ow)), max(0, min(img_h, oy + oh)))
                    clip_bbox = (ox, oy, ox + ow, oy + oh)
                    # background brightness votes of the text boxes whose centre lies in the group (reference :985-1030)
                    dark = light = 0
                    for (bx1, by1, bx2, by2), is_dark in (work.original_text_colors or {}).items():
                        if ox <= (bx1 + bx2) / 2 <= ox + ow and oy <= (by1 + by2) / 2 <= oy + oh:
                            dark, light = dark + bool(is_dark), light + (not is_dark)
                    if dark or light:
                        fallback_color = (0, 0, 0) if dark >= light else (255, 255, 255)
                    indices = group.get("mask_indices", [])
                    rx0, ry0, rx1, ry1 = _bounds_from(raw_results, indices) if indices and raw_results else (ox, oy, ox + ow, oy + oh)
                    ring = None
                    sx1, sy1 = max(0, rx0 - OSB_EXPANSION_PX), max(0, ry0 - OSB_EXPANSION_PX)
                    sx2, sy2 = min(img_w, rx1 + OSB_EXPANSION_PX), min(img_h, ry1 + OSB_EXPANSION_PX)
                    if sx2 > sx1 and sy2 > sy1:
                        ring = border_ring_pixels(current_image, rx0, ry0, rx1, ry1, img_w, img_h)
                        if ring is not None and len(ring) < 20:                    # fewer than 20 ring pixels: no decision
                            ring = None
                    if ring is not None and ring.size > 0:
                        solid, detected = ring_statistics(ring)
                        if solid or fallback_color is None:
                            fallback_color = detected
                        force_fill = method == "opencv"
                        expanded_solid = solid
                        if not force_fill:                                          # the ring around the render-expanded box decides
                            px0, py0, px1, py1 = fill_bounds(group, (ox, oy, ox + ow, oy + oh))
                            ering = border_ring_pixels(current_image, px0, py0, px1, py1, img_w, img_h)
                            if ering is not None and ering.size > 0:
                                esolid, edetected = ring_statistics(ering)
                                if esolid:
                                    expanded_solid, fallback_color = True, edetected
                        if expanded_solid or force_fill:
                            fill_color = fallback_color if fallback_color is not None else detected
                            log_message((f"Using OpenCV simple fill for OSB region: {fill_color} background" if force_fill else
                                         f"Skipping Flux for OSB region: detected solid {fill_color} background"), verbose=verbose)
            if fill_color is not None:
                flush()
                current_image = simple_fill(current_image, group, combined_mask, original_bounds, fill_color)
                cv2_inpaints += 1
                continue
            if method == "none":
                none_skips += 1
                log_message(f"Skipping inpaint for non-solid OSB region {i + 1} (none mode)", verbose=verbose)
                continue
            fb = fallback_color if fallback_color else (255, 255, 255)
            if coordinator is not None and inpainter is not None:
                pending.append({"index": i + 1, "mask": combined_mask.copy(), "seed": region_seed, "composite_clip_bbox": clip_bbox, "fallback_color": fb,
                                "group": group, "original_bounds": original_bounds, "context_bbox": expanded_mask_bbox(combined_mask, current_image.size)})
                log_message(f"Queued OSB region {i + 1} for intra-page Flux scheduling", verbose=verbose)
                continue
            failed, reason, out = False, None, None
            if inpainter is None:
                failed, reason = True, "Flux inpainter unavailable"
            else:
                try:
                    out = inpainter.inpaint_mask(current_image, combined_mask, seed=region_seed, verbose=verbose, strict_mask_clipping=True,
                                                 composite_clip_bbox=clip_bbox)
                    if out is current_image:
                        failed, reason = True, "Flux returned original image (no inpaint)"
                except Exception as e:
                    failed, reason = True, f"Flux inpainting error: {e}"
            if failed:
                log_message(f"Flux failed for OSB region {i + 1} ({reason}); falling back to CV2 fill ({fb})", always_print=True)
                current_image = simple_fill(current_image, group, combined_mask, original_bounds, fb)
                cv2_inpaints += 1
                continue
            flux_inpaints += 1
            current_image = out                     # the reference's PNG round trip between regions only frees host memory
        flush()
        parts = [f"Flux: {flux_inpaints}", f"CV2: {cv2_inpaints}"] + ([f"Skipped (none): {none_skips}"] if none_skips else [])
        log_message(f"Inpainted {len(work.mask_groups)} outside text regions ({', '.join(parts)})", always_print=True)
        return current_image, work.outside_text_data
    except Exception as e:
        log_message(f"Error during outside text inpainting: {e}", always_print=True)
        return pil_image, work.outside_text_data


def process_outside_text(pil_image: Image.Image, config, image_path, image_format: Optional[str], verbose: bool = False,
                         bubble_data=None, text_free_boxes=None, panels=None) -> Tuple[Image.Image, List[Dict[str, Any]]]:
    work = prepare_outside_text_work(pil_image, config, image_path, image_format, verbose=verbose, bubble_data=bubble_data,
                                     text_free_boxes=text_free_boxes, panels=panels)
    if work is None:
        return pil_image, []
    return finish_outside_text_work(work)
