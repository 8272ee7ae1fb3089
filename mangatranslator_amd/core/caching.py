"""Stage memo of the vision path (reference core/caching.py:12-658): detector results, segmentation
detections, upscaled pages and inpainted patches remembered under keys derived from a SHA-256 of the pixels plus the stage's
parameters, so a page processed twice with the same settings does not run its models again.  Keys are byte-identical to the
reference's (tests/golden/cache_keys.json is produced by the reference class), which keeps mixed deployments — reference operators
over these loaders, or these operators beside the reference's translation stage — on one memo.

The LLM-side stores of the reference (translation, manga-ocr) exist here as empty slots only: their key builders belong to
the translation service, which is out of scope."""
import contextlib
import hashlib
import pickle
import threading
import time
from collections import OrderedDict

import numpy as np
from PIL import Image

from ..utils.logging import log_message

_SEG_MODEL_IDS = {"sam2": "facebook/sam2.1-hiera-large", "sam3": "facebook/sam3", "yolo": "yolo"}


def _sha(data: bytes, n: int = 64) -> str:
    return hashlib.sha256(data).hexdigest()[:n]


class _Slot:
    """bounded most-recently-used store (reference core/text/font_manager.py:13-40)"""

    def __init__(self, capacity: int):
        self.capacity, self.cache = capacity, OrderedDict()

    def get(self, key):
        if key not in self.cache:
            return None
        self.cache.move_to_end(key)
        return self.cache[key]

    def put(self, key, value):
        if key in self.cache:
            del self.cache[key]
        elif len(self.cache) >= self.capacity:
            self.cache.popitem(last=False)
        self.cache[key] = value


class UnifiedCache:
    _SLOTS = (("yolo", 1), ("sam", 1), ("translation", 1), ("manga_ocr", 20), ("upscale", 20), ("inpaint", 20))

    def __init__(self):
        self._lock = threading.Lock()
        self._slots = {name: _Slot(cap) for name, cap in self._SLOTS}
        self._current_image_hash = None
        self._tls = threading.local()

    # ---- digests -------------------------------------------------------------------------------
    @contextlib.contextmanager
    def pixels_scope(self):
        """Within the scope (one operator call on this thread, during which the operator does not write to its input) the digest of an
        image OBJECT is computed once: `detect_speech_bubbles` keys two stores with the same page, `upscale_image` two more — each a
        SHA-256 over 4.7 MB at 1024x1536 (10 ms on a host core without SHA extensions).  Not in the reference; keys are unchanged."""
        outer = getattr(self._tls, "digests", None)
        self._tls.digests = {} if outer is None else outer
        try:
            yield self
        finally:
            self._tls.digests = outer

    def _hash_image(self, image: Image.Image) -> str:
        memo = getattr(self._tls, "digests", None)
        if memo is not None:
            hit = memo.get(id(image))
            if hit is not None and hit[0] is image:
                return hit[1]
            digest = self._digest_image(image)
            memo[id(image)] = (image, digest)          # the entry keeps the object alive, so its id cannot be reused inside the scope
            return digest
        return self._digest_image(image)

    hash_seconds = 0.0          # time spent digesting page pixels, whole process (the batch bench reports it per page: SURVEY.md §8d "report separately")
    hash_calls = 0

    @staticmethod
    def _digest_image(image: Image.Image) -> str:
        t0 = time.perf_counter()
        try:
            return UnifiedCache._digest_image_untimed(image)
        finally:
            UnifiedCache.hash_seconds += time.perf_counter() - t0
            UnifiedCache.hash_calls += 1

    @staticmethod
    def _digest_image_untimed(image: Image.Image) -> str:
        if image.mode == "RGBA":                     # alpha flattened on white, as the page would print
            flat = Image.new("RGB", image.size, (255, 255, 255))
            flat.paste(image, mask=image.getchannel("A"))
            image = flat
        h = hashlib.sha256(f"{image.mode}_{image.width}_{image.height}".encode())
        h.update(image.tobytes())
        return h.hexdigest()[:16]

    def _hash_numpy(self, array: np.ndarray) -> str:
        if array.size == 0:
            return _sha(b"empty_array", 16)
        h = hashlib.sha256(f"{array.shape}_{array.dtype}".encode())
        h.update(array.tobytes())
        return h.hexdigest()[:16]

    def _hash_dict(self, data: dict) -> str:
        return _sha(pickle.dumps(data, protocol=pickle.HIGHEST_PROTOCOL), 16)

    # ---- keys ----------------------------------------------------------------------------------
    def get_yolo_cache_key(self, image, model_path: str, confidence: float) -> str:
        return _sha(f"yolo_{self._hash_image(image)}_{_sha(model_path.encode(), 16)}_conf{confidence:.3f}".encode())

    def get_sam_cache_key(self, image, yolo_boxes, seg_model: str = "yolo", conjoined_detection: bool = True,
                          conjoined_confidence: float = 0.35, arithmetic: str = "") -> str:
        """`arithmetic` (this package only; "" = the reference's key): the precision / storage type the SAM graphs were built with — masks
        remembered under one arithmetic are not served to a batch that asked for another"""
        boxes = yolo_boxes.cpu().numpy() if hasattr(yolo_boxes, "cpu") else np.array(yolo_boxes)
        model = _sha(_SEG_MODEL_IDS.get(seg_model, "yolo").encode(), 8)
        return _sha((f"sam_{self._hash_image(image)}_{self._hash_numpy(boxes)}_{model}_seg{seg_model}"
                     f"_conjoined{int(conjoined_detection)}_conf{conjoined_confidence:.3f}" + (f"_arith{arithmetic}" if arithmetic else "")).encode())

    def get_upscale_cache_key(self, image, factor: float, model_type: str = "model") -> str:
        return _sha(f"upscale_{self._hash_image(image)}_factor{factor:.3f}_model{model_type}".encode())

    def get_upscale_dimension_cache_key(self, image, target: int, mode: str, model_type: str = "model") -> str:
        return _sha(f"upscale_dim_{self._hash_image(image)}_target{target}_mode{mode}_model{model_type}".encode())

    def get_bubble_processing_cache_key(self, image, target: int, mode: str, model_type: str = "model") -> str:
        return _sha(f"bubble_proc_{self._hash_image(image)}_target{target}_mode{mode}_model{model_type}".encode())

    def get_inpaint_cache_key(self, image, mask: np.ndarray, seed: int, num_inference_steps: int, residual_diff_threshold: float,
                              guidance_scale: float, prompt: str, ocr_params=None) -> str:
        extra = "_" + "_".join(f"{k}{v}" for k, v in sorted(ocr_params.items())) if ocr_params else ""
        return _sha((f"inpaint_{self._hash_image(image)}_{self._hash_numpy(mask)}_seed{seed}_steps{num_inference_steps}_"
                     f"thresh{residual_diff_threshold:.3f}_guide{guidance_scale:.2f}_{prompt}{extra}").encode())

    def should_use_inpaint_cache(self, seed: int) -> bool:
        return seed != -1                           # -1 = fresh noise every call: nothing to remember

    # ---- stores --------------------------------------------------------------------------------
    def _get(self, slot, key):
        with self._lock:
            return self._slots[slot].get(key)

    def _put(self, slot, key, value, what, verbose):
        with self._lock:
            self._slots[slot].put(key, value)
            n = len(self._slots[slot].cache)
        log_message(f"  - Cached {what} (cache size: {n})", verbose=verbose)

    def holds_any(self, slot: str) -> bool:
        """False when a lookup in `slot` cannot hit — lets a caller postpone the page digest its key needs (10 ms per 4.7 MB page on a
        host core) until there is something to look up or to store.  Not in the reference."""
        with self._lock:
            return bool(self._slots[slot].cache)

    def get_yolo_detection(self, cache_key):
        return self._get("yolo", cache_key)

    def set_yolo_detection(self, cache_key, results, verbose: bool = False):
        self._put("yolo", cache_key, results, "YOLO detection", verbose)

    def get_sam_masks(self, cache_key):
        return self._get("sam", cache_key)

    def set_sam_masks(self, cache_key, masks, verbose: bool = False):
        self._put("sam", cache_key, masks, "SAM masks", verbose)

    def get_upscaled_image(self, cache_key):
        return self._get("upscale", cache_key)

    def set_upscaled_image(self, cache_key, image, verbose: bool = False):
        self._put("upscale", cache_key, image, "upscaled image", verbose)

    def get_inpainted_image(self, cache_key):
        return self._get("inpaint", cache_key)

    def set_inpainted_image(self, cache_key, image, verbose: bool = False):
        self._put("inpaint", cache_key, image, "inpainted image", verbose)

    # ---- lifecycle -----------------------------------------------------------------------------
    def _clear(self, names):
        with self._lock:
            for name in names:
                self._slots[name].cache.clear()

    def clear_yolo_cache(self, verbose: bool = False):
        self._clear(["yolo"]); log_message("YOLO cache cleared", verbose=verbose)

    def clear_sam_cache(self, verbose: bool = False):
        self._clear(["sam"]); log_message("SAM cache cleared", verbose=verbose)

    def clear_translation_cache(self, verbose: bool = False):
        self._clear(["translation"]); log_message("Translation cache cleared", verbose=verbose)

    def clear_manga_ocr_cache(self, verbose: bool = False):
        self._clear(["manga_ocr"]); log_message("manga-ocr cache cleared", verbose=verbose)

    def clear_upscale_cache(self, verbose: bool = False):
        self._clear(["upscale"]); log_message("Upscale cache cleared", verbose=verbose)

    def clear_inpaint_cache(self, verbose: bool = False):
        self._clear(["inpaint"]); log_message("Inpaint cache cleared", verbose=verbose)

    def clear_all(self):
        self._clear(list(self._slots))
        log_message("All caches cleared", always_print=True)

    def reset(self):
        """quiet clear_all that also forgets the current page (not in the reference; bench.py calls it before every page so that no
        timed step is served from the memo, tests call it between cases)"""
        self._clear(list(self._slots))
        with self._lock:
            self._current_image_hash = None

    def set_current_image(self, image, verbose: bool = False):
        """a new page drops everything remembered for the previous one; the same page (by pixels) keeps it"""
        digest = self._hash_image(image)
        with self._lock:
            previous, self._current_image_hash = self._current_image_hash, digest
            if previous is not None and previous != digest:
                for slot in self._slots.values():
                    slot.cache.clear()
        state = "initialized for new image" if previous is None else ("Same image detected - reusing caches" if previous == digest else
                                                                     "Different image detected - clearing all caches")
        log_message(f"Cache {state}" if previous is None else state, verbose=verbose)

    def get_cache_stats(self) -> dict:
        with self._lock:
            return {name: len(slot.cache) for name, slot in self._slots.items()}


_global_cache = None
_global_cache_lock = threading.Lock()


def get_cache() -> UnifiedCache:
    global _global_cache
    with _global_cache_lock:
        if _global_cache is None:
            _global_cache = UnifiedCache()
        return _global_cache


def detector_memo_path(manager, model_path, bubble_detector_model: str) -> str:
    """What stands for "which bubble detector" in a detector memo key.  The reference keys on the checkpoint path its callers pass
    (`detect_speech_bubbles` :1330, `OutsideTextDetector` :285-292); this build's manager picks the checkpoint from
    `bubble_detector_model`, so without an explicit path the manager's path for that detector is used — the same string from both
    operators, so the OSB stage reuses the page's detection like the reference does."""
    if model_path is not None:
        return str(model_path)
    paths = getattr(manager, "model_paths", None)
    if paths:
        for model_type, path in paths.items():
            if getattr(model_type, "name", "") == ("YOLO_SPEECH_BUBBLE_2" if bubble_detector_model == "yolo_2" else "YOLO_SPEECH_BUBBLE"):
                return str(path)
    return str(bubble_detector_model)


def osb_text_memo_path(manager) -> str:
    """the OSB text model's stand-in in a detector memo key: the manager's checkpoint path (reference detection.py:135, ocr_detection.py:411)"""
    paths = getattr(manager, "model_paths", None) or {}
    return next((str(p) for t, p in paths.items() if getattr(t, "name", "") == "YOLO_OSBTEXT"), "yolo_osbtext")
