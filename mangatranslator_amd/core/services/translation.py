"""The vision half of the reference's translation service: `prepare_bubble_images_for_translation`
(core/services/translation.py:2097-2258) — every bubble cropped to the union of its box and its mask's extent, conjoined neighbours
whited out, the crop brought to the minimum side by the RCAN upscaler (`process_bubble_image_cached`) or LANCZOS, and encoded for the
request.  The request itself (`call_translation_api_batch`, prompts, providers) is the LLM side and is not in this build.

Encoding: the reference uses `cv2.imencode`; cv2 is not a dependency here, so Pillow writes the PNG / JPEG (quality 95, cv2's
default).  PNG payloads decode to the same pixels; JPEG payloads are not byte-identical to cv2's.  Parity of the crops before
encoding is pinned by tests/test_bubble_crops.py against the reference function."""
import base64
import io
from typing import Any, Dict, List

import numpy as np
from PIL import Image

from ...utils.logging import log_message
from ..image.image_utils import cv2_to_pil, process_bubble_image_cached

_METHOD_NAMES = {"model": "with 2x-AnimeSharpV4_RCAN", "model_lite": "with 2x-AnimeSharpV4_Fast_RCAN_PU (Lite)", "lanczos": "with LANCZOS"}


def _mask_plane(mask):
    """2-D view of a detection's mask (first channel of an HxWxC array), or None"""
    if mask is None:
        return None
    m = np.asarray(mask)
    if m.ndim == 3:
        m = m[..., 0]
    return m if m.ndim == 2 else None


def bubble_crop(bubble: Dict[str, Any], page_bgr: np.ndarray, masks_by_bbox: Dict[tuple, Any], whiteout_conjoined_bubbles: bool = True):
    """(crop BGR(A) ndarray, (x1, y1)) for one detection — reference :2160-2207."""
    x1, y1, x2, y2 = bubble["bbox"]
    own = _mask_plane(bubble.get("sam_mask"))
    if own is not None:
        rows, cols = np.where(own > 0)
        if rows.size and cols.size:                 # masks may reach past the detector's box: crop their whole extent
            x1, y1 = min(x1, int(cols.min())), min(y1, int(rows.min()))
            x2, y2 = max(x2, int(cols.max()) + 1), max(y2, int(rows.max()) + 1)
    crop = page_bgr[y1:y2, x1:x2].copy()
    neighbours = bubble.get("conjoined_neighbor_bboxes")
    if whiteout_conjoined_bubbles and neighbours:
        mine = own[y1:y2, x1:x2] > 0 if own is not None else None
        for nb in neighbours:
            other = _mask_plane(masks_by_bbox.get(tuple(int(round(v)) for v in nb)))
            if other is None:
                continue
            region = other[y1:y2, x1:x2] > 0
            if mine is not None:
                region &= ~mine
            crop[region] = 255                        # the neighbour's text must not be read as part of this bubble
    return crop, (x1, y1)


def encode_crop(image: Image.Image, mime_type: str) -> str:
    buf = io.BytesIO()
    if mime_type == "image/png":
        image.save(buf, format="PNG", compress_level=1)       # cv2's default PNG effort
    else:
        (image if image.mode in ("RGB", "L") else image.convert("RGB")).save(buf, format="JPEG", quality=95)
    return base64.b64encode(buf.getvalue()).decode("utf-8")


def prepare_bubble_images_for_translation(bubble_data: List[Dict[str, Any]], original_cv_image: np.ndarray, upscale_model: Any, device: Any,
                                          mime_type: str, bubble_min_side_pixels: int, upscale_method: str = "model_lite",
                                          whiteout_conjoined_bubbles: bool = True, verbose: bool = False) -> List[Dict[str, Any]]:
    """New list of detection dicts with `image_b64` / `mime_type` added (the input dicts are not touched) plus, beyond the reference,
    `image_pil`: the finished crop before encoding."""
    masks_by_bbox = {tuple(int(round(v)) for v in b["bbox"]): b.get("sam_mask") for b in bubble_data}
    how = _METHOD_NAMES.get(upscale_method)
    log_message(f"Upscaling {len(bubble_data)} bubble images {how}" if how else f"Processing {len(bubble_data)} bubble images without upscaling",
                always_print=True)
    prepared = []
    for bubble in bubble_data:
        out = bubble.copy()
        crop_bgr, (x1, y1) = bubble_crop(bubble, original_cv_image, masks_by_bbox, whiteout_conjoined_bubbles)
        crop = cv2_to_pil(crop_bgr)
        if upscale_method in ("model", "model_lite"):
            crop = process_bubble_image_cached(crop, upscale_model, device, bubble_min_side_pixels, "min", upscale_method, verbose)
        elif upscale_method == "lanczos":
            w, h = crop.size
            if min(w, h) < bubble_min_side_pixels:
                f = bubble_min_side_pixels / min(w, h)
                crop = crop.resize((int(w * f), int(h * f)), Image.LANCZOS)
        try:
            out["image_b64"], out["mime_type"], out["image_pil"] = encode_crop(crop, mime_type), mime_type, crop
            log_message(f"Bubble {x1},{y1} ({crop.size[0]}x{crop.size[1]})", verbose=verbose)
        except Exception as e:      # noqa: BLE001 — the reference keeps the bubble with no image
            log_message(f"Error encoding bubble {bubble['bbox']}: {e}", verbose=verbose)
            out["image_b64"] = None
        prepared.append(out)
    return prepared
