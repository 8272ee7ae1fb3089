"""`detect_speech_bubbles` — the detect + segment operator of the hot path.

Signature and result shape of the reference operator (core/image/detection.py:1263-1277): returns
`(detections, primary_boxes)` where each detection is
`{"bbox": (x0, y0, x1, y1) ints via round(), "confidence", "class", "sam_mask": uint8 0/255 [H, W]}`.

Flow (reference line numbers):
    primary YOLO-seg @ imgsz (1600 for yolo_2, 640 for yolo_1)                        detection.py:1337-1351
 -> IoU-0.7 confidence-ordered dedup, IoA-0.9 contained-box removal                     :1358-1378
 -> secondary RT-DETR-v2 @640 (conjoined_detection): contained removal, class routing (bubble / text_free /
    text_bubble), bubbles the primary missed are appended, primaries marked text_free are dropped      :1392-1546
 -> simple vs conjoined (>= 2 secondaries inside a primary), synthetic groups of mutually overlapping
    primaries                                                                                             :1571-1620
 -> seg_model "sam2": simple boxes + conjoined parents + synthetic parents through SAM-2.1 in ONE call, each
    mask ANDed with its floor/ceil-clipped prompt box                                                     :1660-1750
 -> detection dicts; conjoined parents partitioned per child (core/image/conjoined.py)                    :1075-1260
 -> any SAM failure falls back to the detector's own masks                                                :1783-1813
 -> osb_text_verification: primary boxes grown to cover the OSB text boxes that belong to them, the same text boxes
    steer the conjoined partition (text-safe cuts)                                                      :120-201, 1555-1571
Stage memo (core/caching.py; reference :1330-1351, 1646-1656, 1781): the primary detector's result is remembered under (pixels,
model path, confidence), the finished SAM detections under (pixels, prompt boxes, seg model, conjoined settings); a SAM hit returns
the remembered list itself, as the reference does.  The OSB text model runs once per call.  Returns `(detections, text_free_boxes)`
like the reference.
"""
from typing import List, Optional, Tuple

import numpy as np
import torch
from PIL import Image

from ...utils.exceptions import ImageProcessingError, ModelError
from ...utils.logging import log_message
from ..caching import detector_memo_path, get_cache, osb_text_memo_path
from ..ml.model_manager import get_model_manager
from . import box_ops, conjoined

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


def _ioa(a, b) -> float:
    area = max(0.0, a[2] - a[0]) * max(0.0, a[3] - a[1])
    if area <= 0:
        return 0.0
    return max(0.0, min(a[2], b[2]) - max(a[0], b[0])) * max(0.0, min(a[3], b[3]) - max(a[1], b[1])) / area


IOA_THRESHOLD = 0.50
IOA_OVERLAP_THRESHOLD = 0.5


def expand_boxes_with_osb_text(image_cv, primary_boxes: torch.Tensor, model_manager, device, confidence: float, hf_token: str,
                               verbose: bool, image_pil=None, cache=None):
    """Grow each speech-bubble box to fully contain the OSB text boxes that meaningfully belong to it (reference
    `_expand_boxes_with_osb_text`, :120-198).  Returns `(boxes, osb_text_boxes_np or None)`; any failure (model unavailable)
    leaves the boxes untouched, as in the reference.  With `image_pil` + `cache` the OSB text model's result is read from / stored in
    the stage memo under the key the OSB stage uses (:135-161), so the page runs that model once."""
    if primary_boxes is None or len(primary_boxes) == 0:
        return primary_boxes, None
    try:
        key = cache.get_yolo_cache_key(image_pil, osb_text_memo_path(model_manager), confidence) if cache is not None and image_pil is not None else None
        remembered = cache.get_yolo_detection(key) if key is not None else None
        if remembered is not None:
            _, osb_boxes, _ = remembered
        else:
            osb_model = model_manager.load_yolo_osbtext(token=hf_token)
            res = osb_model(image_cv, conf=confidence, device=device, verbose=False, imgsz=640)[0]
            osb_boxes = res.boxes.xyxy if res.boxes is not None else torch.tensor([])
            if key is not None:
                cache.set_yolo_detection(key, (res, osb_boxes, res.boxes.conf if res.boxes is not None else torch.tensor([])))
        if osb_boxes is None or len(osb_boxes) == 0:
            return primary_boxes, None
        pb_np, osb_np = primary_boxes.detach().cpu().numpy(), osb_boxes.detach().cpu().numpy()
        for t_box in osb_np:
            best_idx, best_inter = None, 0.0
            for i, b_box in enumerate(pb_np):
                inter = conjoined.box_intersection_area(t_box, b_box)
                if inter > best_inter:
                    best_inter, best_idx = inter, i
            if best_idx is None or best_inter <= 0.0:
                continue
            b = pb_np[best_idx]
            if not conjoined.text_box_belongs_to(t_box, b) or (t_box[0] >= b[0] and t_box[1] >= b[1] and t_box[2] <= b[2] and t_box[3] <= b[3]):
                continue
            pb_np[best_idx] = [min(b[0], t_box[0]), min(b[1], t_box[1]), max(b[2], t_box[2]), max(b[3], t_box[3])]
        return torch.tensor(pb_np, device=primary_boxes.device, dtype=primary_boxes.dtype), osb_np
    except Exception as e:
        log_message(f"OSB text verification skipped: {e}", verbose=verbose)
        return primary_boxes, None


def _open_like_imread(image_path) -> Image.Image:
    """a page opened from its path the way the reference's `cv2.imread(str(image_path))` sees it (reference detection.py:1296-1310):
    EXIF orientation applied, 8-bit 3-channel (palette / CMYK / 16-bit sources reduced by Pillow's converter)"""
    from PIL import ImageOps
    img = Image.open(image_path)
    try:
        img = ImageOps.exif_transpose(img)
    except Exception:
        pass
    return img


def detect_speech_bubbles(image_path, model_path=None, confidence: float = 0.6, verbose: bool = False, device=None,
                          seg_model: str = "yolo", conjoined_detection: bool = True, conjoined_confidence: float = 0.35,
                          image_override: Optional[Image.Image] = None, osb_enabled: bool = False,
                          osb_text_verification: bool = False, osb_text_hf_token: str = "",
                          bubble_detector_model: str = "yolo_2") -> Tuple[List[dict], List[List[float]]]:
    with get_cache().pixels_scope():          # the detector key and the SAM key digest the same page: once per call
        return _detect_speech_bubbles(image_path, model_path, confidence, verbose, device, seg_model, conjoined_detection, conjoined_confidence,
                                      image_override, osb_enabled, osb_text_verification, osb_text_hf_token, bubble_detector_model)


def _detect_speech_bubbles(image_path, model_path, confidence, verbose, device, seg_model, conjoined_detection, conjoined_confidence,
                           image_override, osb_enabled, osb_text_verification, osb_text_hf_token, bubble_detector_model):
    detections: List[dict] = []
    text_free_boxes: List[List[float]] = []
    try:
        image_pil = image_override if image_override is not None else _open_like_imread(image_path)
        if image_pil.mode != "RGB":
            image_pil = image_pil.convert("RGB")
        rgb = np.asarray(image_pil)
    except Exception as e:
        raise ImageProcessingError(f"Error loading image: {e}") from e
    img_h, img_w = rgb.shape[:2]
    bgr = np.ascontiguousarray(rgb[..., ::-1])
    manager = get_model_manager()
    try:
        # the reference names the detector by its checkpoint path (:1325); with no path the detector's short name stands in
        primary_model = manager.load_yolo_speech_bubble(model_path if model_path is not None else bubble_detector_model)
    except Exception as e:
        raise ModelError(f"Error loading primary model: {e}") from e
    cache = get_cache()
    yolo_key = cache.get_yolo_cache_key(image_pil, detector_memo_path(manager, model_path, bubble_detector_model), confidence)
    remembered = cache.get_yolo_detection(yolo_key)
    early_secondary, early_secondary_error = None, None
    if remembered is not None:
        log_message("Using cached YOLO detections", verbose=verbose)
        primary_results, primary_boxes = remembered
    else:
        imgsz = 1600 if bubble_detector_model == "yolo_2" else 640
        if hasattr(primary_model, "submit"):
            # both detectors of the page are queued before either is waited for (hip/plan.py AsyncLane): the secondary network's graph
            # runs beside the primary's and the primary's NMS / mask assembly beside the secondary's kernels.  The reference calls them
            # one after the other (:1337-1351, 1401-1407) and skips the secondary on a page without bubbles; here its result is simply
            # dropped in that case.  A secondary that cannot be loaded or queued fails where the reference would have met it, below.
            ticket = primary_model.submit(bgr, conf=confidence, imgsz=imgsz)
            if conjoined_detection:
                try:
                    sm = manager.load_rtdetr_conjoined_bubble()
                    early_secondary = (sm, sm.submit(bgr, conf=conjoined_confidence, imgsz=640)) if hasattr(sm, "submit") else None
                except Exception as e:      # noqa: BLE001
                    early_secondary_error = e
            try:
                primary_results = primary_model.collect(ticket)[0]
            except BaseException:
                if early_secondary is not None:
                    early_secondary[0].collect(early_secondary[1])      # never leave a model busy behind a failed page
                raise
        else:
            primary_results = primary_model(bgr, conf=confidence, device=device, verbose=False, imgsz=imgsz, retina_masks=True)[0]
        primary_boxes = primary_results.boxes.xyxy if primary_results.boxes is not None else torch.zeros((0, 4))
        cache.set_yolo_detection(yolo_key, (primary_results, primary_boxes))
    try:
        primary_sources = [("primary", i) for i in range(len(primary_boxes))]
        if len(primary_boxes) > 1:
            keep = box_ops.deduplicate_primary_boxes(primary_boxes, primary_results.boxes.conf, IOU_DUPLICATE_THRESHOLD)
            primary_boxes, primary_sources = primary_boxes[keep], [primary_sources[i] for i in keep]
        if len(primary_boxes) > 1:
            keep = box_ops.remove_contained_boxes(primary_boxes)
            primary_boxes, primary_sources = primary_boxes[keep], [primary_sources[i] for i in keep]
    except BaseException:
        if early_secondary is not None:
            early_secondary[0].collect(early_secondary[1])              # a queued model is always collected: its lane stays busy until then
        raise
    if len(primary_boxes) == 0:
        log_message("No detections found", verbose=verbose)
        if early_secondary is not None:
            early_secondary[0].collect(early_secondary[1])          # frees the model; the reference never ran it on such a page
        return detections, text_free_boxes
    log_message(f"Detected {len(primary_boxes)} speech bubbles with YOLO", always_print=True)

    secondary_boxes, secondary_sources, secondary_results = torch.zeros((0, 4)), [], None
    if conjoined_detection:
        try:
            if early_secondary_error is not None:
                raise early_secondary_error
            if early_secondary is not None:
                secondary_model = early_secondary[0]
                secondary_results, early_secondary = secondary_model.collect(early_secondary[1])[0], None
            else:
                secondary_model = manager.load_rtdetr_conjoined_bubble()
                secondary_results = secondary_model(bgr, conf=conjoined_confidence, device=device, verbose=False, imgsz=640)[0]
            secondary_boxes = secondary_results.boxes.xyxy if secondary_results.boxes is not None else torch.zeros((0, 4))
            secondary_sources = [("secondary", i) for i in range(len(secondary_boxes))]
            if len(secondary_boxes) > 1:
                keep = box_ops.remove_contained_boxes(secondary_boxes)
                secondary_boxes, secondary_sources = secondary_boxes[keep], [secondary_sources[i] for i in keep]
            if len(secondary_boxes) > 0 and hasattr(secondary_model, "names"):
                ids = {name: cid for cid, name in secondary_model.names.items()}
                bubble_id, text_free_id = ids.get("bubble"), ids.get("text_free")
                kept_b, kept_s = [], []
                for i, sb in enumerate(secondary_boxes):
                    cid = int(secondary_results.boxes.cls[secondary_sources[i][1]])
                    if text_free_id is not None and cid == text_free_id:
                        text_free_boxes.append(sb.tolist())
                    elif bubble_id is None or cid == bubble_id:
                        kept_b.append(sb); kept_s.append(secondary_sources[i])
                secondary_boxes = torch.stack(kept_b) if kept_b else secondary_boxes[:0]
                secondary_sources = kept_s
                if len(secondary_boxes) > 0:           # bubbles the primary model missed
                    plist = primary_boxes.tolist()
                    new_b, new_s = [], []
                    for i, sb in enumerate(secondary_boxes):
                        sl = sb.tolist()
                        if not any(_ioa(sl, pl) > IOA_OVERLAP_THRESHOLD or _ioa(pl, sl) > IOA_OVERLAP_THRESHOLD for pl in plist):
                            new_b.append(sb); new_s.append(secondary_sources[i])
                    if new_b:
                        log_message(f"Found {len(new_b)} missed bubbles from secondary model", always_print=True)
                        primary_boxes = torch.cat((primary_boxes, torch.stack(new_b).to(primary_boxes)), dim=0)
                        primary_sources.extend(new_s)
            if text_free_boxes and len(primary_boxes) > 0:
                drop = [i for i, pb_ in enumerate(primary_boxes.tolist())
                        if any(_ioa(pb_, tf) > IOA_OVERLAP_THRESHOLD or _ioa(tf, pb_) > IOA_OVERLAP_THRESHOLD for tf in text_free_boxes)]
                if drop:
                    action = "routing to OSB pipeline" if osb_enabled else "discarding (OSB disabled)"
                    log_message(f"Removing {len(drop)} bubbles marked text_free ({action})", always_print=True)
                    keep = [i for i in range(len(primary_boxes)) if i not in drop]
                    primary_boxes = primary_boxes[keep] if keep else primary_boxes[:0]
                    primary_sources = [primary_sources[i] for i in keep]
        except Exception as e:
            log_message(f"Warning: Could not load/run secondary RT-DETR model: {e}. Proceeding without conjoined/fallback detection.", verbose=verbose)
            secondary_boxes, secondary_sources = torch.zeros((0, 4)), []
    if len(primary_boxes) == 0:
        return detections, text_free_boxes

    grouping_primary_boxes = primary_boxes.clone()
    osb_text_boxes_np = None
    if osb_text_verification and len(primary_boxes) > 0:
        primary_boxes, osb_text_boxes_np = expand_boxes_with_osb_text(bgr, primary_boxes, manager, device, confidence, osb_text_hf_token, verbose,
                                                                          image_pil=image_pil, cache=cache)
    conjoined_indices, simple_indices = [], list(range(len(primary_boxes)))
    if len(secondary_boxes) > 0 and conjoined_detection:
        conjoined_indices, simple_indices = box_ops.categorize_detections(grouping_primary_boxes, secondary_boxes, ioa_threshold=IOA_THRESHOLD)
        if conjoined_indices:
            log_message(f"Detected {len(conjoined_indices)} conjoined speech bubbles with RT-DETR", always_print=True)
    synthetic_groups: List[dict] = []
    if len(simple_indices) > 1:
        groups, simple_indices = box_ops.detect_overlapping_primaries(grouping_primary_boxes, simple_indices)
        for members in groups:
            st = grouping_primary_boxes[members]
            synthetic_groups.append({"member_indices": members, "parent_mask": None,
                                     "parent_box": torch.cat([st[:, :2].min(dim=0).values, st[:, 2:].max(dim=0).values])})

    def assemble(sam_masks):
        return conjoined.build_segmentation_detections(primary_boxes, grouping_primary_boxes, primary_sources, primary_results, primary_model,
                                                       secondary_boxes, secondary_sources, secondary_results, simple_indices, conjoined_indices,
                                                       img_h, img_w, conjoined_confidence, osb_text_boxes_np=osb_text_boxes_np, verbose=verbose,
                                                       sam_masks=sam_masks, synthetic_conjoined_groups=synthetic_groups)

    if seg_model not in ("sam2", "sam3"):
        log_message("SAM disabled, using YOLO segmentation masks", verbose=verbose)
        return assemble(None), text_free_boxes
    try:
        sam_key = cache.get_sam_cache_key(image_pil, primary_boxes, seg_model, conjoined_detection, conjoined_confidence,
                                          arithmetic=f"{getattr(manager, 'sam_precision', 'high')}-{getattr(manager, 'sam_storage', 'auto')}")
        remembered = cache.get_sam_masks(sam_key)
        if remembered is not None:
            log_message("Using cached SAM masks", verbose=verbose)
            return remembered, text_free_boxes
        if seg_model == "sam3":                 # reference :1661-1666; the loader raises here (SAM 3 is not built) and the page keeps its YOLO masks
            processor, sam = manager.load_sam3(token=osb_text_hf_token, verbose=verbose)
        else:
            processor, sam = manager.load_sam2(verbose=verbose)
        prompts, owners = [], []
        for idx in simple_indices:
            prompts.append(primary_boxes[idx]); owners.append(idx)
        for p_idx, s_indices in conjoined_indices:           # the prompt must cover every child
            st = torch.cat([primary_boxes[p_idx].unsqueeze(0)] + [secondary_boxes[s].unsqueeze(0).to(primary_boxes) for s in s_indices], dim=0)
            prompts.append(torch.cat([st[:, :2].min(dim=0).values, st[:, 2:].max(dim=0).values])); owners.append(p_idx)
        synth_start = len(prompts)
        prompts += [sg["parent_box"] for sg in synthetic_groups]
        sam_masks = [None] * len(primary_boxes)
        if prompts:
            all_boxes = torch.stack(prompts)
            inputs = processor(image_pil, input_boxes=all_boxes.unsqueeze(0).cpu(), return_tensors="pt")
            out = sam(multimask_output=False, **inputs)
            m = processor.post_process_masks(out.pred_masks, inputs["original_sizes"])[0][:, 0]
            m = (m > SAM_MASK_THRESHOLD).cpu().numpy()
            for i, box in enumerate(prompts):
                clipped = clip_mask_to_box(m[i], box.tolist(), img_h, img_w)
                if i < synth_start:
                    sam_masks[owners[i]] = clipped
                else:
                    synthetic_groups[i - synth_start]["parent_mask"] = clipped
            log_message(f"Generated {len(prompts)} primary masks with SAM 2.1", always_print=True)
        detections = assemble(sam_masks)
        cache.set_sam_masks(sam_key, detections)
        return detections, text_free_boxes
    except Exception as e:
        log_message(f"{'SAM 3' if seg_model == 'sam3' else 'SAM 2.1'} segmentation failed: {e}. Falling back to YOLO segmentation masks.", always_print=True)
        for sg in synthetic_groups:
            sg["parent_mask"] = None
        return assemble(None), text_free_boxes


def submit_panels(image_pil: Image.Image, confidence: float = 0.25):
    """First half of `detect_panels` for a page flow that knows early that it will want panels: the panel network is queued on its own stream
    NOW and runs beside the page's other detectors; `detect_panels(..., ticket=...)` later only collects.  With front halves running side by
    side (`batch_vision_images(front_workers=N)`) this is also the window in which their panel calls meet in one graph replay
    (core/ml/detector_batch.py).  Returns None — and `detect_panels` then does everything itself, as the reference does — when the loaded
    model has no submit half or anything at all goes wrong here."""
    try:
        model = get_model_manager().load_yolo_panel()
        if not hasattr(model, "submit"):
            return None
        if image_pil.mode != "RGB":
            image_pil = image_pil.convert("RGB")
        bgr = np.ascontiguousarray(np.asarray(image_pil)[..., ::-1])
        return (model, model.submit(bgr, conf=confidence, imgsz=640))
    except Exception:      # noqa: BLE001 — the collect side reports failures the way the reference does
        return None


def detect_panels(image_path, confidence: float = 0.25, device=None, verbose: bool = False, image_override: Optional[Image.Image] = None, ticket=None):
    """Panel rectangles `(x1, y1, x2, y2)` (ints via round()) of the detections whose class is "frame" — every detection when the
    model names no such class (reference `detect_panels`, :1817-1915).  Image and loader failures raise ImageProcessingError / ModelError;
    a failure while running the model degrades to `[]`, as there.  `ticket` = what `submit_panels` returned for this page (same confidence)."""
    if ticket is not None:
        model, tk = ticket
        try:
            res = model.collect(tk)[0]
            return _panels_of(model, res, verbose)
        except Exception as e:
            log_message(f"Panel detection failed: {e}. Proceeding without panel information.", always_print=True)
            return []
    try:
        image_pil = image_override if image_override is not None else _open_like_imread(image_path)
        if image_pil.mode != "RGB":
            image_pil = image_pil.convert("RGB")
        bgr = np.ascontiguousarray(np.asarray(image_pil)[..., ::-1])
        log_message(f"Processing image for panel detection: {getattr(image_path, 'name', image_path) if image_path else 'override'} "
                    f"({bgr.shape[1]}x{bgr.shape[0]})", verbose=verbose)
    except Exception as e:
        raise ImageProcessingError(f"Error loading image: {e}") from e
    try:
        model = get_model_manager().load_yolo_panel(verbose=verbose)
    except Exception as e:
        raise ModelError(f"Error loading panel model: {e}") from e
    try:
        res = model(bgr, conf=confidence, device=device, verbose=False, imgsz=640)[0]
        return _panels_of(model, res, verbose)
    except Exception as e:
        log_message(f"Panel detection failed: {e}. Proceeding without panel information.", always_print=True)
        return []


def _panels_of(model, res, verbose):
    boxes = res.boxes.xyxy if res.boxes is not None else torch.zeros((0, 4))
    classes = res.boxes.cls if res.boxes is not None else torch.zeros((0,))
    if len(boxes) == 0:
        log_message("No panels detected", verbose=verbose)
        return []
    frame_id = next((cid for cid, name in getattr(model, "names", {}).items() if name.lower() == "frame"), None)
    panels = []
    for box, cid in zip(boxes.tolist(), classes.tolist()):
        if frame_id is None or int(cid) == frame_id:
            panels.append(tuple(int(round(v)) for v in box))
    return panels
