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


def set_hardware_queues(n: int = 16) -> bool:
    """`GPU_MAX_HW_QUEUES=n` for this process — ROCm multiplexes a process's HIP streams onto that many hardware queues (default 4) and reads
    the variable when the HIP runtime is LOADED, i.e. at `import torch`: too late from inside a library that imports torch, in time from
    here (this module imports nothing heavy).  Sixteen is what a service that mostly detects / segments / cleans wants (every model owns a
    stream; two pages' detect stages in flight = ten streams: config 2 33 -> 37 pages/s); diffusion-bound services keep the
    default (config 5 measured slower with more).  A value already in the environment is left alone.  -> True when the setting will
    take effect, False (with a message) when torch is already loaded."""
    import os
    if os.environ.get("GPU_MAX_HW_QUEUES"):
        return True
    if "torch" in sys.modules:
        print(f"mangatranslator_amd: GPU_MAX_HW_QUEUES={n} requested after torch was imported — the HIP runtime has read its settings already; "
              "export it in the environment (or call install() / set_hardware_queues() before importing torch)", file=sys.stderr)
        return False
    os.environ["GPU_MAX_HW_QUEUES"] = str(int(n))
    return True


def pin_rank_to_gpu_cpus() -> dict:
    """For services that run one process per GPU (LOCAL_RANK set by the launcher): this process's host threads move to the CPUs of the NUMA
    node its GPU is attached to — its share of them when several GPUs hang off one node (core/device.py pin_host_threads_to_gpu).  Call
    it after install() and before the page loop starts its thread pools; a single process on a one-GPU host is left alone."""
    import os
    import torch
    from .core.device import pin_host_threads_to_gpu
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        return {"pinned": False, "reason": "one GPU"}
    return pin_host_threads_to_gpu(int(os.environ.get("LOCAL_RANK", "0")), n_devices=torch.cuda.device_count())


def install(include_caching: bool = True, share_utils: bool = True, hardware_queues: "int | None" = None) -> list:
    """-> the list of `core.*` module names now served by this package.  `include_caching=False` keeps the reference's own stage memo
    (needed when its translation / manga-ocr key builders are in use: this build restates the vision-side keys only).
    `hardware_queues=16`: see `set_hardware_queues` (detect / segment / clean services; makes `batch_vision_images` pick two front workers)."""
    if hardware_queues is not None:
        set_hardware_queues(hardware_queues)
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
