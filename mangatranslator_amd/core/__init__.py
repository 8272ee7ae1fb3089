"""Host-side mirror of the reference's `core` package for the vision hot path (reference core/__init__.py:8-41).

The names below are the subset of the reference's re-exports this build implements; `translate_and_render`, `batch_translate_images`,
`render_text_skia`, `call_translation_api_batch` and `sort_bubbles_by_reading_order` belong to the LLM / rendering half and stay the
reference's (mangatranslator_amd.integration.install() serves the hot-path MODULES under the reference's own `core` package, whose
`__init__` then re-exports all of them).  The vision half of `translate_and_render` is `pipeline.process_page_vision`, of
`batch_translate_images` `pipeline.batch_process_images`.
"""
from .._version import __version__, __version_info__  # noqa: F401
from .caching import UnifiedCache, get_cache  # noqa: F401
from .image.cleaning import clean_speech_bubbles  # noqa: F401
from .image.detection import detect_speech_bubbles  # noqa: F401
from .image.image_utils import cv2_to_pil, pil_to_cv2, save_image_with_compression  # noqa: F401
from .image.inpainting import FluxKleinInpainter, FluxKontextInpainter  # noqa: F401
from .image.ocr_detection import OutsideTextDetector  # noqa: F401
from .ml.model_manager import ModelManager, get_model_manager  # noqa: F401
from .pipeline import batch_process_images, batch_vision_images, process_page_vision  # noqa: F401

__all__ = ["__version__", "__version_info__", "get_cache", "UnifiedCache", "detect_speech_bubbles", "clean_speech_bubbles", "pil_to_cv2",
           "cv2_to_pil", "save_image_with_compression", "get_model_manager", "ModelManager", "OutsideTextDetector", "FluxKontextInpainter",
           "FluxKleinInpainter", "process_page_vision", "batch_process_images", "batch_vision_images"]
