"""`detect_speech_bubbles` — the detect + segment operator of the hot path.

Signature and result shape of the reference operator (core/image/detection.py:1263-1277): returns
`(detections, primary_boxes)` where each detection is
`{"bbox": (x0, y0, x1, y1) ints via round(), "confidence", "class", "sam_mask": uint8 0/255 [H, W]}`.

Built so far (the simple-bubble path the reference takes when no conjoined groups are found):
    primary YOLO-seg @ imgsz (1600 for yolo_2, 640 for yolo_1)        detection.py:1337-1351
 -> IoU-0.7 confidence-ordered dedup, IoA-0.9 contained-box removal     :1358-1378
 -> seg_model "sam2": all boxes of the page through SAM-2.1 in one call  :1641-1750 (`_process_simple_bubbles`)
    with each mask ANDed with its floor/ceil-clipped prompt box          :1732-1750
    seg_model "yolo": the detector's own retina masks                    :514-565
 -> any SAM failure falls back to the YOLO masks, as the reference does  :1783-1813
Secondary RT-DETR conjoined grouping / mask splitting (rows a2 tail, a4) are not built yet: with
`conjoined_detection=True` the synthetic-overlap grouping (`_detect_overlapping_primaries`) is computed
and reported, but group members are still emitted as simple bubbles.
"""
from typing import List, Optional, Tuple

import numpy as np
import torch
from PIL import Image

from ...utils.exceptions import ImageProcessingError, ModelError
from ...utils.logging import log_message
from ..ml.model_manager import get_model_manager
from . import box_ops

IOU_DUPLICATE_THRESHOLD = 0.7
SAM_MASK_THRESHOLD = 0.5


def clip_mask_to_box(mask: np.ndarray, box, img_h: int, img_w: int) -> np.ndarray:
    """bool mask AND the floor/ceil-clipped box -> uint8 0/255 (reference :1732-1750)."""
    x0f, y0f, x1f, y1f = [float(v) for v in box]
    x0, y0 = int(np.floor(max(0, min(x0f, img_w)))), int(np.floor(max(0, min(y0f, img_h))))
    x1, y1 = int(np.ceil(max(0, min(x1f, img_w)))), int(np.ceil(max(0, min(y1f, img_h))))
    m = np.asarray(mask).astype(bool)
    if x1 > x0 and y1 > y0:
        out = np.zeros((img_h, img_w), bool)
        out[y0:y1, x0:x1] = m[y0:y1, x0:x1]
        m = out
    return m.astype(np.uint8) * 255


def rect_mask_from_box(box, img_h: int, img_w: int) -> np.ndarray:
    x0, y0, x1, y1 = [int(round(float(v))) for v in box]
    m = np.zeros((img_h, img_w), np.uint8)
    m[max(0, y0):max(0, min(img_h, y1)), max(0, x0):max(0, min(img_w, x1))] = 255
    return m


def detect_speech_bubbles(image_path, model_path=None, confidence: float = 0.6, verbose: bool = False, device=None,
                          seg_model: str = "sam2", conjoined_detection: bool = True, conjoined_confidence: float = 0.35,
                          image_override: Optional[Image.Image] = None, osb_enabled: bool = False,
                          osb_text_verification: bool = False, osb_text_hf_token: str = "",
                          bubble_detector_model: str = "yolo_2") -> Tuple[List[dict], List[List[float]]]:
    try:
        image_pil = image_override if image_override is not None else Image.open(image_path)
        rgb = np.asarray(image_pil.convert("RGB"))
    except Exception as e:
        raise ImageProcessingError(f"Failed to load image: {e}") from e
    img_h, img_w = rgb.shape[:2]
    bgr = np.ascontiguousarray(rgb[..., ::-1])
    manager = get_model_manager()
    model = manager.load_yolo_speech_bubble(bubble_detector_model)
    imgsz = 1600 if bubble_detector_model == "yolo_2" else 640
    res = model(bgr, conf=confidence, device=device, verbose=False, imgsz=imgsz, retina_masks=True)[0]
    if res.boxes is None or len(res.boxes.xyxy) == 0:
        return [], []
    boxes, confs, classes = res.boxes.xyxy, res.boxes.conf, res.boxes.cls
    keep = box_ops.deduplicate_primary_boxes(boxes, confs, IOU_DUPLICATE_THRESHOLD)
    boxes_k = boxes[keep]
    keep2 = box_ops.remove_contained_boxes(boxes_k)
    order = [keep[i] for i in keep2]
    boxes_f = boxes[order]
    if conjoined_detection:
        groups, _ = box_ops.detect_overlapping_primaries(boxes_f, list(range(len(order))))
        if groups:
            log_message(f"Detected {len(groups)} synthetic conjoined group(s)", verbose=verbose)
    masks: List[Optional[np.ndarray]] = [None] * len(order)
    if seg_model == "sam2":
        try:
            processor, sam = manager.load_sam2()
            inputs = processor(Image.fromarray(rgb), input_boxes=boxes_f.unsqueeze(0).cpu(), return_tensors="pt")
            out = sam(multimask_output=False, **inputs)
            m = processor.post_process_masks(out.pred_masks, inputs["original_sizes"])[0][:, 0]
            m = (m > SAM_MASK_THRESHOLD).cpu().numpy()
            for i in range(len(order)):
                masks[i] = clip_mask_to_box(m[i], boxes_f[i].tolist(), img_h, img_w)
        except (ModelError, RuntimeError) as e:
            log_message(f"SAM segmentation failed ({e}); falling back to YOLO masks", always_print=True)
    yolo_masks = res.masks.data.cpu().numpy() if res.masks is not None else None
    detections = []
    for i, src in enumerate(order):
        m = masks[i]
        if m is None and yolo_masks is not None:
            m = (yolo_masks[src] > 0).astype(np.uint8) * 255
        if m is None:
            m = rect_mask_from_box(boxes_f[i].tolist(), img_h, img_w)
        x0, y0, x1, y1 = boxes_f[i].tolist()
        detections.append({"bbox": (int(round(x0)), int(round(y0)), int(round(x1)), int(round(y1))),
                           "confidence": float(confs[src]), "class": model.names[int(classes[src])], "sam_mask": m})
    return detections, boxes_f.tolist()
