"""Outside-speech-bubble (OSB) text region producer — SURVEY.md §8 row f3, first half: it decides how many
FLUX regions a page has.  Mirrors `OutsideTextDetector` of the reference (core/image/ocr_detection.py:24-808):

    detect_outside_text   :189-539   OSB text boxes (YOLO OSB-text model, or the secondary detector's text_free
                                     class as fallback / in text_free_only mode), nested boxes removed, boxes that
                                     meaningfully belong to a speech bubble dropped
    get_text_masks        :541-726   boxes -> expanded int boxes -> spatial groups -> per-group bool page masks
    _group_text_boxes_spatially :728-784, _boxes_are_nearby :786-808     union-find on centre distance

All of it is integer / float64 host arithmetic on a handful of boxes (outputs are indices and rectangles ⇒ bit-exact
target, pinned by tests/golden/osb_regions.json); the detectors it calls are the libmtx_hip graphs returned by the
model manager.  Detector outputs go through the stage memo (core/caching.py) under the reference's keys: the bubble
detection of the page is the entry `detect_speech_bubbles` stored, the OSB text model's result has its own.
"""
import os
from typing import List, Optional, Tuple

import numpy as np
import torch
from PIL import Image

from ...utils.exceptions import ImageProcessingError
from ...utils.logging import log_message
from ..caching import detector_memo_path, get_cache, osb_text_memo_path
from ..device import get_best_device
from ..ml.model_manager import get_model_manager

OSB_BUBBLE_MATCH_IOA_THRESHOLD = 0.2       # minimum text-box overlap ratio for assigning a text box to a bubble
TEXT_FREE_BUBBLE_IOA_THRESHOLD = 0.5       # a bubble counts as a text_free region only with substantial overlap
MAX_GROUP_DIMENSION = 1568                 # larger groups are split back into single boxes (FLUX preferred sizes)


class OutsideTextDetector:
    def __init__(self, device: Optional[torch.device] = None, hf_token: Optional[str] = None):
        self.device = device if device is not None else get_best_device()
        self.hf_token = hf_token
        self.manager = get_model_manager()
        self.cache = get_cache()

    # ---- box predicates (reference :43-147) ---------------------------------------------------------------------
    def boxes_overlap(self, box1, box2) -> bool:
        return not (box1[2] <= box2[0] or box2[2] <= box1[0] or box1[3] <= box2[1] or box2[3] <= box1[1])

    def box_intersection_area(self, box1, box2) -> float:
        return max(0.0, min(box1[2], box2[2]) - max(box1[0], box2[0])) * max(0.0, min(box1[3], box2[3]) - max(box1[1], box2[1]))

    def box_area(self, box) -> float:
        return max(0.0, box[2] - box[0]) * max(0.0, box[3] - box[1])

    def point_in_box(self, point_x: float, point_y: float, box) -> bool:
        return box[0] <= point_x <= box[2] and box[1] <= point_y <= box[3]

    def text_box_meaningfully_matches_bubble(self, text_box, bubble_box) -> bool:
        intersection = self.box_intersection_area(text_box, bubble_box)
        if intersection <= 0.0:
            return False
        text_area = self.box_area(text_box)
        if text_area <= 0.0:
            return False
        cx, cy = (text_box[0] + text_box[2]) / 2.0, (text_box[1] + text_box[3]) / 2.0
        return intersection / text_area >= OSB_BUBBLE_MATCH_IOA_THRESHOLD or self.point_in_box(cx, cy, bubble_box)

    def box_is_inside(self, box1, box2, threshold=0.9) -> bool:
        inter_w = max(0, min(box1[2], box2[2]) - max(box1[0], box2[0]))
        inter_h = max(0, min(box1[3], box2[3]) - max(box1[1], box2[1]))
        area1 = (box1[2] - box1[0]) * (box1[3] - box1[1])
        if area1 <= 0:
            return False
        return inter_w * inter_h / area1 > threshold

    def box_ioa(self, box_inner, box_outer) -> float:
        area_inner = self.box_area(box_inner)
        if area_inner <= 0.0:
            return 0.0
        return self.box_intersection_area(box_inner, box_outer) / area_inner

    def bubble_is_text_free_region(self, bubble_box, text_free_boxes) -> bool:
        if not text_free_boxes:
            return False
        bubble = list(map(float, bubble_box[:4]))
        for tf_box in text_free_boxes:
            tf = list(map(float, tf_box[:4]))
            if self.box_ioa(bubble, tf) > TEXT_FREE_BUBBLE_IOA_THRESHOLD or self.box_ioa(tf, bubble) > TEXT_FREE_BUBBLE_IOA_THRESHOLD:
                return True
        return False

    def filter_nested_detections(self, results):
        """larger boxes first; a box more than 90 % inside an already kept one is dropped (reference :149-183)"""
        if len(results) <= 1:
            return results
        ordered = sorted(results, key=lambda r: (r[0][2] - r[0][0]) * (r[0][3] - r[0][1]), reverse=True)
        kept = []
        for cur in ordered:
            if not any(self.box_is_inside(cur[0], k[0]) for k in kept):
                kept.append(cur)
        return kept

    def unload_models(self):
        self.manager.unload_ocr_models()

    # ---- detection (reference :189-539) ---------------------------------------------------------------------------
    def detect_outside_text(self, image_path: str, yolo_model_path: Optional[str] = None, confidence: float = 0.6,
                            conjoined_confidence: float = 0.35, verbose: bool = False, image_override: Optional[Image.Image] = None,
                            existing_bubbles: Optional[List] = None, text_free_boxes: Optional[List] = None,
                            bubble_detector_model: str = "yolo_2", min_area_ignore_ratio: float = 0.0, text_free_only: bool = False):
        """-> [(bbox float32[4], confidence)] of text regions outside speech bubbles."""
        with self.cache.pixels_scope():       # the bubble key and the OSB text key digest the same page: once per call
            return self._detect_outside_text(image_path, yolo_model_path, confidence, conjoined_confidence, verbose, image_override,
                                             existing_bubbles, text_free_boxes, bubble_detector_model, min_area_ignore_ratio, text_free_only)

    def _detect_outside_text(self, image_path, yolo_model_path, confidence, conjoined_confidence, verbose, image_override, existing_bubbles,
                             text_free_boxes, bubble_detector_model, min_area_ignore_ratio, text_free_only):
        if image_override is None and not os.path.exists(image_path):
            raise FileNotFoundError(f"Error: The file '{image_path}' was not found.")
        try:
            image_pil = image_override if image_override is not None else Image.open(image_path)
            image_pil = image_pil if image_pil.mode == "RGB" else image_pil.convert("RGB")
            _bgr = []

            def image_cv_():          # BGR, what the detectors take — made on first use (a 6 MP page costs 40 ms of host copy, and
                if not _bgr:          # with provided bubbles + text_free boxes no detector runs at all)
                    _bgr.append(np.ascontiguousarray(np.asarray(image_pil)[..., ::-1]))
                return _bgr[0]
        except Exception as e:
            raise ImageProcessingError(f"Error loading image: {e}")

        provided = None
        if existing_bubbles is not None:
            try:
                provided = []
                for b in existing_bubbles:
                    bbox = b.get("bbox") if isinstance(b, dict) else b
                    if bbox is None or len(bbox) != 4:
                        continue
                    provided.append([float(v) for v in bbox])
            except Exception as e:
                log_message(f"Warning: Failed to parse provided bubbles: {e}. Falling back to YOLO.", always_print=True)
                provided = None
        text_free_boxes = list(text_free_boxes) if text_free_boxes else []

        if provided:
            yolo_boxes = torch.tensor(provided, device=self.device, dtype=torch.float32)
            log_message(f"Skipping YOLO; using provided bubbles ({len(yolo_boxes)})", verbose=verbose)
        else:
            key = self.cache.get_yolo_cache_key(image_pil, detector_memo_path(self.manager, yolo_model_path, bubble_detector_model), confidence)
            remembered = self.cache.get_yolo_detection(key)          # the page's detection from `detect_speech_bubbles`, same key (:285-313)
            if remembered is not None:
                log_message("Using cached Speech Bubble detections", verbose=verbose)
                res, yolo_boxes = remembered
            else:
                model = self.manager.load_yolo_speech_bubble(yolo_model_path if yolo_model_path is not None else bubble_detector_model)
                res = model(image_cv_(), conf=confidence, device=self.device, verbose=False,
                            imgsz=1600 if bubble_detector_model == "yolo_2" else 640, retina_masks=True)[0]
                yolo_boxes = res.boxes.xyxy if res.boxes is not None else torch.tensor([])
                self.cache.set_yolo_detection(key, (res, yolo_boxes))
            log_message(f"YOLO detected {len(yolo_boxes) if yolo_boxes.nelement() > 0 else 0} speech bubbles", verbose=verbose)

        # secondary detector: when text_free_only still lacks text_free boxes, or when the bubbles were detected here
        if (text_free_only and not text_free_boxes) or not provided:
            try:
                sec_model = self.manager.load_rtdetr_conjoined_bubble()
                sec = sec_model(image_cv_(), conf=conjoined_confidence, device=self.device, verbose=False, imgsz=640)[0]
                sec_boxes = sec.boxes.xyxy if sec.boxes is not None else torch.tensor([])
                sec_cls = sec.boxes.cls if sec.boxes is not None else torch.tensor([])
                bubble_id = tf_id = None
                if hasattr(sec_model, "names"):
                    for cid, cname in sec_model.names.items():
                        if cname == "bubble":
                            bubble_id = cid
                        elif cname == "text_free":
                            tf_id = cid
                if tf_id is not None and len(sec_boxes) > 0:
                    for i, cls_id in enumerate(sec_cls):
                        if int(cls_id) == tf_id:
                            text_free_boxes.append(sec_boxes[i].detach().cpu().numpy())
                if bubble_id is not None and len(sec_boxes) > 0:
                    extra = [sec_boxes[i] for i, cls_id in enumerate(sec_cls) if int(cls_id) == bubble_id]
                    if extra:
                        log_message(f"Secondary RT-DETR found {len(extra)} potential bubbles", verbose=verbose)
                        stacked = torch.stack(extra)
                        yolo_boxes = torch.cat((yolo_boxes, stacked.to(yolo_boxes)), dim=0) if yolo_boxes.nelement() > 0 else stacked
            except Exception as e:
                log_message(f"Secondary RT-DETR failed: {e}", verbose=verbose)

        def from_text_free():
            return (torch.tensor(np.asarray(text_free_boxes), device=self.device, dtype=torch.float32),
                    torch.ones(len(text_free_boxes), device=self.device, dtype=torch.float32))

        osb_boxes = osb_confs = None
        if text_free_only:
            log_message("Using RT-DETR text_free detections as OSB regions (skipping YOLO OSB model)", always_print=True)
            if text_free_boxes:
                osb_boxes, osb_confs = from_text_free()
            else:
                log_message("No text_free detections available; skipping OSB text detections", always_print=True)
        else:
            try:
                def osb_key():
                    return self.cache.get_yolo_cache_key(image_pil, osb_text_memo_path(self.manager), confidence)
                # :414-446; the key (a digest of the page) is only built when the one-entry detector slot holds something to compare
                # it with, or when there is a result to store — not for a model that turns out to be unavailable
                key = osb_key() if self.cache.holds_any("yolo") else None
                remembered = self.cache.get_yolo_detection(key) if key is not None else None
                if remembered is not None:
                    log_message("Using cached OSBText detections", verbose=verbose)
                    res, osb_boxes, osb_confs = remembered
                else:
                    osb_model = self.manager.load_yolo_osbtext(token=self.hf_token)
                    res = osb_model(image_cv_(), conf=confidence, device=self.device, verbose=False, imgsz=640)[0]
                    osb_boxes = res.boxes.xyxy if res.boxes is not None else None
                    osb_confs = res.boxes.conf if res.boxes is not None else None
                    self.cache.set_yolo_detection(key if key is not None else osb_key(), (res, osb_boxes, osb_confs))
            except Exception as e:
                log_message(f"OSB text model unavailable: {e}. Using text_free fallback if available.", always_print=True)
                if text_free_boxes:
                    osb_boxes, osb_confs = from_text_free()
                else:
                    log_message("No text_free fallback available; skipping OSB text detections", always_print=True)

        results = []
        if osb_boxes is not None:
            boxes_np, confs_np = osb_boxes.detach().cpu().numpy(), osb_confs.detach().cpu().numpy()
            results = [(box, float(confs_np[i])) for i, box in enumerate(boxes_np)]
        results = self.filter_nested_detections(results)

        if yolo_boxes is not None and yolo_boxes.nelement() > 0:
            bubbles_np = yolo_boxes.detach().cpu().numpy()
            outside = []
            for item in results:
                bbox = item[0]
                best_bubble, best_inter = None, 0.0
                for bubble in bubbles_np:
                    if not self.boxes_overlap(bbox, bubble) or self.bubble_is_text_free_region(bubble, text_free_boxes):
                        continue
                    inter = self.box_intersection_area(bbox, bubble)
                    if inter > best_inter:
                        best_inter, best_bubble = inter, bubble
                if best_bubble is None or not self.text_box_meaningfully_matches_bubble(bbox, best_bubble):
                    outside.append(item)
            log_message(f"Filtered out {len(results) - len(outside)} OCR results that meaningfully overlapped speech bubbles", verbose=verbose)
            results = outside

        found = len(results)
        min_ignore = max(0.0, min(0.05, min_area_ignore_ratio))
        if min_ignore > 0.0:
            image_area = float(image_pil.width * image_pil.height)
            found -= sum(1 for bbox, _ in results if ((bbox[2] - bbox[0]) * (bbox[3] - bbox[1])) / max(1.0, image_area) < min_ignore)
        log_message(f"Found {found} outside text regions", always_print=True)
        return results

    # ---- masks (reference :541-726) -------------------------------------------------------------------------------
    def get_text_masks(self, image_path: str, bbox_expansion_percent_width: float = 0.0, bbox_expansion_percent_height: float = 0.0,
                       text_box_proximity_ratio: float = 0.02, verbose: bool = False, image_override: Optional[Image.Image] = None,
                       existing_results: Optional[List] = None) -> Tuple[Optional[List], Optional[Image.Image]]:
        results = existing_results if existing_results is not None else self.detect_outside_text(image_path, verbose=verbose, image_override=image_override)
        if not results:
            return None, None
        if image_override is not None:
            image_pil = image_override.convert("RGB") if image_override.mode != "RGB" else image_override
        else:
            image_pil = Image.open(image_path).convert("RGB")
        img_w, img_h = image_pil.size

        boxes = [[int(c) for c in r[0]] for r in results]
        expanded = []
        for x0, y0, x1, y1 in boxes:
            ex, ey = (x1 - x0) * bbox_expansion_percent_width, (y1 - y0) * bbox_expansion_percent_height
            x0e, y0e = int(np.floor(max(0, x0 - ex))), int(np.floor(max(0, y0 - ey)))
            x1e, y1e = int(np.ceil(min(img_w, x1 + ex))), int(np.ceil(min(img_h, y1 + ey)))
            if x1e > x0e and y1e > y0e:
                expanded.append([x0e, y0e, x1e, y1e])
        # NOTE (kept from the reference): a box dropped above shifts `expanded` against `results`; grouping pairs them by position

        def rect(box):
            m = np.zeros((img_h, img_w), dtype=bool)
            m[box[1]:box[3], box[0]:box[2]] = True
            return m

        def xywh(x0, y0, x1, y1):
            return {"x": int(x0), "y": int(y0), "width": int(x1 - x0), "height": int(y1 - y0)}

        groups = []
        for g_boxes, g_results, g_indices in self._group_text_boxes_spatially(expanded, results, img_w, img_h, text_box_proximity_ratio, verbose):
            min_x, min_y = min(b[0] for b in g_boxes), min(b[1] for b in g_boxes)
            max_x, max_y = max(b[2] for b in g_boxes), max(b[3] for b in g_boxes)
            if max_x - min_x > MAX_GROUP_DIMENSION or max_y - min_y > MAX_GROUP_DIMENSION:
                log_message(f"  - Group too large ({max_x - min_x}x{max_y - min_y}), splitting...", verbose=verbose)
                for box, result, g_idx in zip(g_boxes, g_results, g_indices):
                    mask = rect(box)
                    raw = [int(c) for c in result[0]]
                    groups.append({"combined_mask": mask, "bbox": xywh(*box), "original_bbox": xywh(*raw), "individual_masks": [mask],
                                   "mask_indices": [g_idx], "confidence": result[1]})
                continue
            raw_boxes = [[int(c) for c in r[0]] for r in g_results]
            combined = np.zeros((img_h, img_w), dtype=bool)
            individual, total_conf = [], 0.0
            for box, result in zip(g_boxes, g_results):
                mask = rect(box)
                combined |= mask
                individual.append(mask)
                total_conf += result[1]
            groups.append({"combined_mask": combined, "bbox": xywh(min_x, min_y, max_x, max_y),
                           "original_bbox": xywh(min(b[0] for b in raw_boxes), min(b[1] for b in raw_boxes),
                                                 max(b[2] for b in raw_boxes), max(b[3] for b in raw_boxes)),
                           "individual_masks": individual, "mask_indices": list(g_indices), "confidence": total_conf / len(g_results)})
        log_message(f"Created {len(groups)} grouped text regions for inpainting", verbose=verbose)
        return groups, image_pil

    def _group_text_boxes_spatially(self, boxes, results, img_w, img_h, text_box_proximity_ratio=0.02, verbose=False):
        """union-find over box pairs whose centres are within min(W, H) * ratio; groups come out in order of their root's first
        appearance, members in index order (reference :728-784)"""
        if not boxes:
            return []
        threshold = min(img_w, img_h) * text_box_proximity_ratio
        parent = list(range(len(boxes)))

        def find(x):
            while parent[x] != x:
                parent[x] = parent[parent[x]]
                x = parent[x]
            return x

        for i in range(len(boxes)):
            for j in range(i + 1, len(boxes)):
                if self._boxes_are_nearby(boxes[i], boxes[j], threshold):
                    pi, pj = find(i), find(j)
                    if pi != pj:
                        parent[pi] = pj
        groups = {}
        for i in range(len(boxes)):
            g = groups.setdefault(find(i), ([], [], []))
            g[0].append(boxes[i]); g[1].append(results[i]); g[2].append(i)
        log_message(f"  - Grouped {len(boxes)} boxes into {len(groups)} spatial groups", verbose=verbose)
        return list(groups.values())

    def _boxes_are_nearby(self, box1, box2, threshold) -> bool:
        cx1, cy1 = (box1[0] + box1[2]) / 2, (box1[1] + box1[3]) / 2
        cx2, cy2 = (box2[0] + box2[2]) / 2, (box2[1] + box2[3]) / 2
        return bool(np.sqrt((cx1 - cx2) ** 2 + (cy1 - cy2) ** 2) <= threshold)


# ---- OCR recogniser drivers (reference :811-990) ------------------------------------------------------------------------------
# The recognisers themselves (manga-ocr, PaddleOCR-VL) are the LLM / OCR side of the reference, outside the MI355X hot path (SURVEY.md §8
# out-of-scope list).  The two drivers keep the reference's contract — one string per image, "[OCR FAILED]" for an image that could not be
# read, the same marker for every image when the recogniser cannot be had — so `core.outside_text_processor` / `core.services.translation`
# import and degrade exactly as they do when the reference fails to load the model.
OCR_FAILED = "[OCR FAILED]"


def _run_recogniser(images, what: str, get_recogniser, read_one, verbose: bool) -> List[str]:
    if not images:
        return []
    try:
        recogniser = get_recogniser()
        texts = []
        for i, img in enumerate(images):
            if img is None:
                log_message(f"Image {i + 1} is None (decode failure), skipping", always_print=True)
                texts.append(OCR_FAILED)
                continue
            try:
                log_message(f"Processing image {i + 1}/{len(images)} with {what}", verbose=verbose)
                text = read_one(recogniser, img)
                texts.append(text.strip() if text else "")
            except Exception as e:
                log_message(f"{what} failed for image {i + 1}: {e}", always_print=True)
                texts.append(OCR_FAILED)
        return texts
    except Exception as e:
        log_message(f"Error with {what}: {e}", always_print=True)
        return [OCR_FAILED] * len(images)


def extract_text_with_manga_ocr(images: List[Image.Image], verbose: bool = False) -> List[str]:
    return _run_recogniser(images, "manga-ocr", lambda: get_model_manager().get_manga_ocr(verbose=verbose), lambda ocr, img: ocr(img), verbose)


def extract_text_with_paddle_ocr_vl(images: List[Image.Image], verbose: bool = False) -> List[str]:
    def read_one(pair, img):
        processor, model = pair
        messages = [{"role": "user", "content": [{"type": "image", "image": img}, {"type": "text", "text": "OCR:"}]}]
        ip = processor.image_processor
        size = getattr(ip, "size", {}) or {}
        lo = getattr(ip, "min_pixels", None)
        for key in ("shortest_edge", "min_pixels"):
            if lo is None:
                lo = size.get(key) if isinstance(size, dict) else getattr(size, key, None)
        inputs = processor.apply_chat_template(messages, add_generation_prompt=True, tokenize=True, return_dict=True, return_tensors="pt",
                                               processor_kwargs={"images_kwargs": {"size": {"shortest_edge": lo if lo is not None else 28 * 28 * 130,
                                                                                             "longest_edge": 1280 * 28 * 28}}}).to(model.device)
        outputs = model.generate(**inputs, max_new_tokens=1024)
        return processor.decode(outputs[0][inputs["input_ids"].shape[-1]: -1])
    return _run_recogniser(images, "PaddleOCR-VL-1.6", lambda: get_model_manager().get_paddle_ocr_vl(verbose=verbose), read_one, verbose)
