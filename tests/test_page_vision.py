"""`process_page_vision` — the vision half of the reference's `translate_and_render` (core/pipeline.py:638-1000, cleaning-only flow):
stage order, argument plumbing, mode handling and the degrade paths, with the four operators replaced by recorders."""
import math
import types

import numpy as np
import pytest
from PIL import Image

from mangatranslator_amd.core import pipeline
from mangatranslator_amd.core.image import cleaning, detection, image_utils
from mangatranslator_amd.core import outside_text_processor as otp


def _config(**over):
    c = types.SimpleNamespace(
        device="cpu", yolo_model_path=None, upscaling_only=False, request_coordinator=None,
        preprocessing=types.SimpleNamespace(auto_scale=True),
        detection=types.SimpleNamespace(confidence=0.6, seg_model="sam2", conjoined_detection=True, conjoined_confidence=0.35,
                                        use_osb_text_verification=False, bubble_detector_model="yolo_2"),
        cleaning=types.SimpleNamespace(thresholding_value=200, use_otsu_threshold=False, roi_shrink_px=5, inpaint_colored_bubbles=False),
        outside_text=types.SimpleNamespace(enabled=True, huggingface_token=""),
        output=types.SimpleNamespace(upscale_final_image=True, image_upscale_factor=2.0, image_upscale_model="model_lite"))
    for k, v in over.items():
        setattr(c, k, v)
    return c


@pytest.fixture
def calls(monkeypatch):
    log = []
    bubbles = [{"bbox": (1, 2, 30, 40), "confidence": 0.9, "class": "bubble", "sam_mask": np.zeros((60, 50), np.uint8)}]

    def detect(image_path, model_path, confidence, **kw):
        log.append(("detect", kw["image_override"].mode, kw["seg_model"], kw["osb_enabled"], kw["bubble_detector_model"]))
        return bubbles, [[5.0, 5.0, 9.0, 9.0]]

    # the OSB stage in its two halves (prepare in the page's front half, finish in its back half: core/pipeline.py)
    def osb_prepare(page, config, image_path, image_format, verbose=False, bubble_data=None, text_free_boxes=None, panels=None):
        log.append(("osb", len(bubble_data), text_free_boxes, panels))
        return ("work", page)

    def osb_finish(work):
        page = work[1]
        out = page.copy(); out.putpixel((0, 0), (1, 2, 3, 255) if page.mode == "RGBA" else (1, 2, 3))
        return out, []

    def clean(page, model_path, confidence, pre_computed_detections=None, processing_scale=1.0, **kw):
        log.append(("clean", page.getpixel((0, 0))[:3], len(pre_computed_detections), round(processing_scale, 6), kw["thresholding_value"]))
        arr = np.asarray(page)
        bgr = np.ascontiguousarray(arr[..., [2, 1, 0] + ([3] if arr.shape[2] == 4 else [])]).copy()
        bgr[1, 1, :3] = (10, 20, 30)                                   # B, G, R
        return bgr, [{"bbox": (1, 2, 30, 40)}]

    def upscale(image, factor, model_type="model", verbose=False):
        log.append(("upscale", image.size, image.getpixel((1, 1))[:3], factor, model_type))
        return image.resize((int(image.width * factor), int(image.height * factor))).convert("RGB")

    monkeypatch.setattr(detection, "detect_speech_bubbles", detect)
    monkeypatch.setattr(otp, "prepare_outside_text_work", osb_prepare)
    monkeypatch.setattr(otp, "finish_outside_text_work", osb_finish)
    monkeypatch.setattr(cleaning, "clean_speech_bubbles", clean)
    monkeypatch.setattr(image_utils, "upscale_image", upscale)
    return log


def test_stage_order_and_plumbing(calls):
    page = Image.new("RGBA", (50, 60), (250, 250, 250, 255))
    out, info = pipeline.process_page_vision(page, _config())
    assert [c[0] for c in calls] == ["detect", "osb", "clean", "upscale"]
    assert calls[0] == ("detect", "RGBA", "sam2", True, "yolo_2")
    assert calls[1] == ("osb", 1, [[5.0, 5.0, 9.0, 9.0]], None)
    assert calls[2] == ("clean", (1, 2, 3), 1, round(math.sqrt(50 * 60 / 1e6), 6), 200)      # cleaning sees the OSB stage's output
    assert calls[3] == ("upscale", (50, 60), (30, 20, 10), 2.0, "model_lite")                 # BGR -> RGB on the way back
    assert out.mode == "RGBA" and out.size == (100, 120)                                      # the upscaler's RGB is brought back to the target mode
    assert info["processing_scale"] == pytest.approx(math.sqrt(0.003)) and len(info["bubbles"]) == 1 and len(info["cleaned"]) == 1


def test_degrade_paths_and_modes(calls, monkeypatch):
    def boom(*a, **k):
        raise RuntimeError("no model")
    monkeypatch.setattr(detection, "detect_speech_bubbles", boom)
    page = Image.new("RGB", (40, 40), (9, 9, 9))
    cfg = _config(); cfg.output.upscale_final_image = False; cfg.preprocessing.auto_scale = False
    out, info = pipeline.process_page_vision(page, cfg)
    assert [c[0] for c in calls] == ["osb"] and calls[0][1] == 0           # no bubbles: OSB still runs, cleaning is skipped
    assert out.mode == "RGB" and info["bubbles"] == [] and info["processing_scale"] == 1.0
    calls.clear()
    up = _config(upscaling_only=True)
    out, _ = pipeline.process_page_vision(Image.new("RGBA", (10, 12)), up)
    assert [c[0] for c in calls] == ["upscale"] and out.size == (20, 24) and out.mode == "RGBA"


def test_cleaning_failure_keeps_the_page(calls, monkeypatch):
    from mangatranslator_amd.utils.exceptions import CleaningError

    def bad_clean(*a, **k):
        raise CleaningError("contours")
    monkeypatch.setattr(cleaning, "clean_speech_bubbles", bad_clean)
    cfg = _config(); cfg.output.upscale_final_image = False
    out, info = pipeline.process_page_vision(Image.new("RGBA", (50, 60), (250, 250, 250, 255)), cfg)
    assert out.getpixel((0, 0))[:3] == (1, 2, 3) and info["cleaned"] == []


def test_panels_reach_the_osb_stage(calls, monkeypatch):
    """use_panel_sorting: panels from `detect_panels` are handed to the OSB stage (reference pipeline.py:804-831, :1733); a failing panel
    detector leaves panels = None and the page goes on."""
    cfg = _config()
    cfg.detection.use_panel_sorting, cfg.detection.panel_confidence = True, 0.25
    seen = []

    def panels(image_path, confidence=0.25, device=None, verbose=False, image_override=None):
        seen.append((confidence, image_override.size))
        return [(0, 0, 25, 60), (25, 0, 50, 60)]
    monkeypatch.setattr(detection, "detect_panels", panels)
    page = Image.new("RGB", (50, 60), (250, 250, 250))
    _, info = pipeline.process_page_vision(page, cfg)
    assert seen == [(0.25, (50, 60))] and calls[1] == ("osb", 1, [[5.0, 5.0, 9.0, 9.0]], [(0, 0, 25, 60), (25, 0, 50, 60)])
    assert info["panels"] == [(0, 0, 25, 60), (25, 0, 50, 60)]
    calls.clear()
    monkeypatch.undo()                                   # the product operator: no panel model in this build -> ModelError -> None
    monkeypatch.setattr(detection, "detect_speech_bubbles", lambda *a, **k: ([], []))
    monkeypatch.setattr(otp, "prepare_outside_text_work", lambda page, *a, panels="unset", **k: (calls.append(("osb", panels)), ("work", page))[1])
    monkeypatch.setattr(otp, "finish_outside_text_work", lambda work: (work[1], []))
    monkeypatch.setattr(image_utils, "upscale_image", lambda image, *a, **k: image)
    _, info = pipeline.process_page_vision(page, cfg)
    assert calls == [("osb", None)] and info["panels"] is None


def test_initial_upscale_rule(calls, monkeypatch):
    """`preprocessing.enabled` + `factor`: the page goes through the upscaler before detection (reference pipeline.py:602-635, :718-720) —
    factor rule and model choice against the reference's two functions (tests/golden/make_pre_upscale_golden.py)"""
    import json
    from pathlib import Path
    gold = json.loads((Path(__file__).resolve().parent / "golden" / "pre_upscale.json").read_text())
    T = types.SimpleNamespace
    for row in gold["resolve"]:
        cfg = None if row["cfg"] is None else T(**row["cfg"])
        assert pipeline.resolve_pre_upscale_factor(cfg) == row["factor"], row
    seen = []
    monkeypatch.setattr(image_utils, "upscale_image", lambda image, factor, model_type="model", verbose=False: (seen.append([factor, model_type]), "up")[1])
    r = [pipeline.apply_pre_upscale_if_needed("img", T(preprocessing=T(enabled=True, factor=2.0), output=T(image_upscale_model="model"))),
         pipeline.apply_pre_upscale_if_needed("img", T(preprocessing=T(enabled=True, factor=3.0))),
         pipeline.apply_pre_upscale_if_needed("img", T(preprocessing=T(enabled=False, factor=3.0), output=T(image_upscale_model="model")))]
    assert [list(x) for x in r] == gold["apply"] and seen == gold["calls"]
    # in the page flow: detection sees the enlarged page, the processing scale follows it
    monkeypatch.undo()
    log = []
    monkeypatch.setattr(image_utils, "upscale_image", lambda image, factor, model_type="model", verbose=False: (log.append(("up", factor, model_type)), image.resize((int(image.width * factor), int(image.height * factor))))[1])
    monkeypatch.setattr(detection, "detect_speech_bubbles", lambda *a, **k: (log.append(("detect", k["image_override"].size)), ([], []))[1])
    monkeypatch.setattr(otp, "prepare_outside_text_work", lambda page, *a, **k: None)
    cfg = _config()
    cfg.preprocessing = T(auto_scale=True, enabled=True, factor=2.0)
    cfg.output.upscale_final_image = False
    out, info = pipeline.process_page_vision(Image.new("RGB", (50, 60)), cfg)
    assert log == [("up", 2.0, "model_lite"), ("detect", (100, 120))] and out.size == (100, 120)
    assert info["pre_upscale_factor"] == 2.0 and info["processing_scale"] == pytest.approx(math.sqrt(100 * 120 / 1e6))


def test_batch_vision_images_equals_page_by_page(calls, tmp_path, monkeypatch):
    """`batch_vision_images`: the folder through the front / back halves inside the harness — files, order and pixels of
    `process_page_vision` page by page; front_workers chosen from the configuration (short back half + sixteen hardware queues -> 2)"""
    root = tmp_path / "in"
    root.mkdir()
    for i in range(5):
        Image.new("RGB", (50, 60), (200 + i, 250, 250)).save(root / f"p{i}.png")
    cfg = _config(verbose=False, output=types.SimpleNamespace(upscale_final_image=True, image_upscale_factor=2.0, image_upscale_model="model_lite",
                                                             output_format="png", jpeg_quality=95, png_compression=2))
    cfg.detection.use_panel_sorting = False
    monkeypatch.delenv("GPU_MAX_HW_QUEUES", raising=False)
    assert pipeline.default_front_workers(cfg) == 1                      # FLUX / upscale in the back half
    light = _config(outside_text=types.SimpleNamespace(enabled=False, huggingface_token=""),
                    output=types.SimpleNamespace(upscale_final_image=False))
    assert pipeline.default_front_workers(light) == 1                    # runtime default of four hardware queues
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "16")
    assert pipeline.default_front_workers(light) == 2 and pipeline.default_front_workers(cfg) == 1
    res = pipeline.batch_vision_images(root, cfg, tmp_path / "out", io_threads=2, front_workers=2)
    assert res["success_count"] == 5 and res["error_count"] == 0 and res["io"]["pages_in_flight"] == 3
    assert [c[0] for c in calls].count("detect") == 5 and [c[0] for c in calls].count("upscale") == 5
    formats = {c[1] for c in calls if c[0] == "detect"}
    assert formats == {"RGBA"}                                           # png output: pages are converted up front (load_page)
    for i in range(5):
        want, _ = pipeline.process_page_vision(Image.open(root / f"p{i}.png").convert("RGBA"), cfg, root / f"p{i}.png")
        got = Image.open(tmp_path / "out" / f"p{i}_translated.png").convert("RGBA")
        assert got.size == (100, 120) and np.array_equal(np.asarray(got), np.asarray(want))
