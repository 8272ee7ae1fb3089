"""Build-container check of the drop-in boundary (INTEGRATION.md §2 Option A): this package bound under the UNMODIFIED
reference, the reference's own `core` / `core.pipeline` imported on top of it, and the reference's `translate_and_render` run in
cleaning-only mode — compared with `process_page_vision` on the same page.  Skipped where /root/reference does not exist (GPU box)."""
import json
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.skipif(not Path("/root/reference/core/pipeline.py").exists(), reason="the reference checkout only exists in the build container")
def test_reference_runs_on_top_of_this_package(emu_lib):
    r = subprocess.run([sys.executable, str(ROOT / "tests" / "dropin_runner.py")], capture_output=True, text=True, timeout=900, cwd=str(ROOT))
    line = [l for l in r.stdout.splitlines() if l.startswith("DROPIN_REPORT ")]
    assert r.returncode == 0 and line, r.stdout[-2000:] + r.stderr[-4000:]
    rep = json.loads(line[0][len("DROPIN_REPORT "):])
    assert rep["reference_core_file"] == "/root/reference/core/__init__.py"                 # `core` really is the reference's package
    assert rep["same_exception_classes"]                                                     # a HIP loader's ModelError is the class the reference catches
    for name in ("detect_speech_bubbles", "clean_speech_bubbles", "FluxKleinInpainter", "FluxKontextInpainter", "OutsideTextDetector", "get_model_manager"):
        assert rep["core_reexports"][name].startswith("mangatranslator_amd.core."), name   # reference core/__init__.py:8-41 resolved through us
    assert rep["core_reexports"]["translate_and_render"] == "core.pipeline"                   # ... and the page flow is still the reference's
    assert all(m.startswith("mangatranslator_amd.core.") for m in rep["pipeline_binds"].values())      # core/pipeline.py:33-54
    assert rep["detector_calls_reference"] == 1 and rep["bubbles_cleaned"] == 2
    assert rep["ref_mode"] == "RGBA" and rep["ref_size"] == [256, 384]
    assert rep["text_removed"] == [True, True] and rep["changed_pixels"] > 500
    assert rep["same_pixels"], "reference translate_and_render(cleaning_only) and process_page_vision disagree"


def test_import_surface_of_the_served_modules():
    """every name the reference imports FROM the modules this package serves exists here (reference core/__init__.py:8-41,
    core/pipeline.py:33-54, core/outside_text_processor.py:16-25, core/services/translation.py:11-17, core/text/text_renderer.py:8)"""
    from mangatranslator_amd.core import batch_coordinator, caching, outside_text_processor
    from mangatranslator_amd.core.image import cleaning, detection, image_utils, inpainting, ocr_detection
    from mangatranslator_amd.core.ml import model_manager
    want = {caching: ["UnifiedCache", "get_cache"], cleaning: ["clean_speech_bubbles", "retry_cleaning_with_otsu"],
            detection: ["detect_speech_bubbles", "detect_panels"],
            image_utils: ["cv2_to_pil", "pil_to_cv2", "save_image_with_compression", "convert_image_to_target_mode", "resize_to_max_side",
                          "upscale_image", "upscale_image_to_dimension", "process_bubble_image_cached", "calculate_centroid_expansion_box"],
            inpainting: ["FluxKleinInpainter", "FluxKontextInpainter"],
            ocr_detection: ["OutsideTextDetector", "extract_text_with_manga_ocr", "extract_text_with_paddle_ocr_vl"],
            model_manager: ["ModelManager", "ModelType", "get_model_manager"],
            outside_text_processor: ["finish_outside_text_work", "prepare_outside_text_work", "process_outside_text"],
            batch_coordinator: ["BatchRequestCoordinator", "expanded_mask_bbox", "partition_non_overlapping_waves", "paste_image_region"]}
    for mod, names in want.items():
        for n in names:
            assert hasattr(mod, n), f"{mod.__name__}.{n}"
    import inspect
    assert inspect.signature(detection.detect_speech_bubbles).parameters["seg_model"].default == "yolo"     # reference detection.py:1269


def test_ocr_drivers_degrade_like_the_reference(monkeypatch):
    from PIL import Image
    from mangatranslator_amd.core.image import ocr_detection as od
    imgs = [Image.new("RGB", (8, 8)), None]
    assert od.extract_text_with_manga_ocr([]) == [] and od.extract_text_with_paddle_ocr_vl([]) == []
    assert od.extract_text_with_manga_ocr(imgs) == ["[OCR FAILED]", "[OCR FAILED]"]          # no recogniser in this build: the reference's marker
    assert od.extract_text_with_paddle_ocr_vl(imgs) == ["[OCR FAILED]", "[OCR FAILED]"]
    mgr = od.get_model_manager()
    monkeypatch.setattr(mgr, "get_manga_ocr", lambda verbose=False: (lambda img: "  text "))
    assert od.extract_text_with_manga_ocr(imgs) == ["text", "[OCR FAILED]"]                   # a staged recogniser is driven with the reference's call shape


def test_centroid_expansion_box():
    """reference docstring example (image_utils.py:199-203): ellipse 40 x 30 at (50, 50) in a 100 x 100 mask"""
    import numpy as np
    from mangatranslator_amd.core.image.image_utils import calculate_centroid_expansion_box
    from mangatranslator_amd.utils.exceptions import ImageProcessingError
    yy, xx = np.mgrid[0:100, 0:100]
    m = ((((xx - 50) / 40.0) ** 2 + ((yy - 50) / 30.0) ** 2) <= 1.0).astype(np.uint8) * 255
    (x, y, w, h), (cx, cy) = calculate_centroid_expansion_box(m, padding_pixels=10.0)
    assert abs(cx - 50) < 0.6 and abs(cy - 50) < 0.6 and abs(w - 58) <= 2 and abs(h - 38) <= 2 and x == int(round(cx - w / 2.0))
    with pytest.raises(ImageProcessingError):
        calculate_centroid_expansion_box(np.zeros((10, 10), np.uint8))
    with pytest.raises(ImageProcessingError):
        calculate_centroid_expansion_box(m, padding_pixels=60.0)
    two = np.zeros((60, 160), np.uint8)                            # two lobes joined by a neck: the anchor moves to the deepest point
    two[10:50, 10:70] = 255; two[10:50, 90:150] = 255; two[27:33, 70:90] = 255
    (_, _, w2, _), (cx2, _) = calculate_centroid_expansion_box(two, padding_pixels=4.0)
    assert (cx2 < 70 or cx2 > 90) and w2 > 20


def test_hardware_queue_setting_needs_to_come_before_torch():
    """`integration.set_hardware_queues` / `install(hardware_queues=)`: in time in a fresh interpreter (the module imports nothing heavy), refused
    with a message once torch is loaded, and never over an explicit environment setting"""
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = str(Path(__file__).resolve().parent.parent)
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    code = ("import sys, os; sys.path.insert(0, %r); import mangatranslator_amd.integration as a; "
            "print(a.set_hardware_queues(16), os.environ.get('GPU_MAX_HW_QUEUES'), 'torch' in sys.modules); "
            "import torch; os.environ.pop('GPU_MAX_HW_QUEUES'); print(a.set_hardware_queues(16), os.environ.get('GPU_MAX_HW_QUEUES'))") % root
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.strip().splitlines()
    assert lines[0] == "True 16 False" and lines[1] == "False None" and "after torch was imported" in r.stderr
    r = subprocess.run([sys.executable, "-c", "import sys, os; sys.path.insert(0, %r); import mangatranslator_amd.integration as a; print(a.set_hardware_queues(16), os.environ['GPU_MAX_HW_QUEUES'])" % root],
                       capture_output=True, text=True, env=dict(env, GPU_MAX_HW_QUEUES="8"), timeout=300)
    assert r.stdout.strip() == "True 8"
