"""Binding this package under the reference application (Option A) — one call, before `core` is imported:

    import mangatranslator_amd.integration as amd
    amd.install()            # then: import core / from core.pipeline import translate_and_render / app.py / main.py as usual

What it does, in the reference's own terms:
  * the reference's `utils.exceptions` and `utils.logging` become this package's `utils` modules, so a ModelError raised by a HIP
    loader IS the class `core/pipeline.py` catches (`except ModelError` compares classes, not names);
  * the hot-path modules of `core` (`HOT_PATH_MODULES`) are served by `mangatranslator_amd.core.*` — same module paths, names,
    arguments and error behaviour (reference core/__init__.py:8-41, core/pipeline.py:33-54 import from them);
  * everything else (`core.pipeline`, `core.config`, the LLM / OCR / rendering side) stays the reference's.
"""
import importlib
import sys

HOT_PATH_MODULES = ("device", "caching", "batch_coordinator", "ml.model_manager", "image.image_utils", "image.detection", "image.ocr_detection",
                    "image.inpainting", "image.cleaning", "outside_text_processor")


def install(include_caching: bool = True, share_utils: bool = True) -> list:
    """-> the list of `core.*` module names now served by this package.  `include_caching=False` keeps the reference's own stage memo
    (needed when its translation / manga-ocr key builders are in use: this build restates the vision-side keys only)."""
    if "core" in sys.modules and getattr(sys.modules["core"], "__file__", None):
        raise RuntimeError("mangatranslator_amd.integration.install() must run before the reference's `core` package is imported")
    if share_utils:
        for name in ("exceptions", "logging"):
            try:
                ref = importlib.import_module(f"utils.{name}")
            except ImportError:
                continue                                    # stand-alone use: this package's own copies
            if name == "exceptions" and not all(hasattr(ref, c) for c in ("ValidationError", "ModelError", "ImageProcessingError", "DetectionError",
                                                                            "CleaningError", "CancellationError")):
                continue                                    # some other top-level `utils`
            if name == "logging" and not hasattr(ref, "log_message"):
                continue
            loaded = sys.modules.get(f"mangatranslator_amd.utils.{name}")
            if loaded is not None and loaded is not ref and any(m.startswith("mangatranslator_amd.core") for m in sys.modules):
                raise RuntimeError("install() must run before any mangatranslator_amd.core module is imported (exception classes are bound at import)")
            sys.modules[f"mangatranslator_amd.utils.{name}"] = ref
            import mangatranslator_amd.utils as u
            setattr(u, name, ref)
    served = []
    for name in HOT_PATH_MODULES:
        if name == "caching" and not include_caching:
            continue
        sys.modules[f"core.{name}"] = importlib.import_module(f"mangatranslator_amd.core.{name}")
        served.append(f"core.{name}")
    return served
