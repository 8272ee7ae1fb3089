"""`detect_panels` (host half) vs the reference operator on canned detector outputs (tests/golden/make_panel_goldens.py):
class filter ("frame", case-insensitive; every box when the model names no such class), round-half-even corners, empty / missing boxes,
a failing model -> [], a failing loader -> ModelError; and the page flow's use of it."""
import json
import types
from pathlib import Path

import numpy as np
import pytest
import torch
from PIL import Image

from mangatranslator_amd.core.image import detection
from mangatranslator_amd.utils.exceptions import ModelError

GOLD = json.loads((Path(__file__).resolve().parent / "golden" / "panels.json").read_text())


class _Model:
    def __init__(self, boxes, classes, names):
        self.b = None if boxes is None else types.SimpleNamespace(xyxy=torch.tensor(boxes, dtype=torch.float32).reshape(-1, 4),
                                                                  cls=torch.tensor(classes, dtype=torch.float32))
        if names is not None:
            self.names = {int(k): v for k, v in names.items()}

    def __call__(self, image, conf=None, device=None, verbose=False, imgsz=None):
        assert image.dtype == np.uint8 and image.shape == (300, 400, 3) and imgsz == 640 and conf == 0.25
        return [types.SimpleNamespace(boxes=self.b)]


def test_detect_panels_matches_reference(monkeypatch):
    img = Image.fromarray(np.zeros((300, 400, 3), np.uint8)).convert("RGBA")
    for name, c in GOLD["cases"].items():
        m = _Model(c["boxes"], c["classes"], c["names"])
        monkeypatch.setattr(detection, "get_model_manager", lambda m=m: types.SimpleNamespace(load_yolo_panel=lambda verbose=False: m))
        got = detection.detect_panels(Path("page.png"), confidence=0.25, device="cpu", image_override=img)
        assert [list(p) for p in got] == GOLD["results"][name], name
        assert all(isinstance(v, int) for p in got for v in p)

    class Boom:
        names = {0: "frame"}

        def __call__(self, *a, **k):
            raise RuntimeError("kernel fault")
    monkeypatch.setattr(detection, "get_model_manager", lambda: types.SimpleNamespace(load_yolo_panel=lambda verbose=False: Boom()))
    assert detection.detect_panels(Path("page.png"), image_override=img) == GOLD["results"]["model_raises"] == []

    def no_model(verbose=False):
        raise RuntimeError("download failed")
    monkeypatch.setattr(detection, "get_model_manager", lambda: types.SimpleNamespace(load_yolo_panel=no_model))
    with pytest.raises(ModelError) as e:
        detection.detect_panels(Path("page.png"), image_override=img)
    assert [type(e.value).__name__, str(e.value)] == GOLD["results"]["loader_raises"]


def test_product_loader_raises_until_the_graph_exists():
    from mangatranslator_amd.core.ml.model_manager import ModelType, get_model_manager
    mgr = get_model_manager()
    with pytest.raises(ModelError):
        mgr.load_yolo_panel()
    sentinel = object()
    mgr.models[ModelType.YOLO_PANEL] = sentinel          # a deployment that brings its own panel model puts it in the slot
    try:
        assert mgr.load_yolo_panel() is sentinel
    finally:
        mgr.models.pop(ModelType.YOLO_PANEL, None)


def test_speech_bubble_loader_path_rule():
    """`load_yolo_speech_bubble(model_path)`: the reference's rule (model_manager.py:702-709) plus this build's short names"""
    from mangatranslator_amd.core.ml.model_manager import ModelType, get_model_manager
    mgr = get_model_manager()
    first, second = mgr.model_paths[ModelType.YOLO_SPEECH_BUBBLE], mgr.model_paths[ModelType.YOLO_SPEECH_BUBBLE_2]
    assert mgr._resolve_speech_bubble_model(None) == (ModelType.YOLO_SPEECH_BUBBLE, first)
    assert mgr._resolve_speech_bubble_model(str(second)) == (ModelType.YOLO_SPEECH_BUBBLE_2, second)
    assert mgr._resolve_speech_bubble_model(str(first)) == (ModelType.YOLO_SPEECH_BUBBLE, first)
    assert mgr._resolve_speech_bubble_model("/data/custom.safetensors") == (ModelType.YOLO_SPEECH_BUBBLE, Path("/data/custom.safetensors"))
    assert mgr._resolve_speech_bubble_model("yolo_2") == (ModelType.YOLO_SPEECH_BUBBLE_2, second)
    assert mgr._resolve_speech_bubble_model("yolo_1") == (ModelType.YOLO_SPEECH_BUBBLE, first)
    sentinel = object()
    mgr.models[ModelType.YOLO_SPEECH_BUBBLE_2] = sentinel
    try:
        assert mgr.load_yolo_speech_bubble(str(second)) is sentinel and mgr.load_yolo_speech_bubble("yolo_2") is sentinel
        mgr.unload_all()
        assert not mgr.is_loaded(ModelType.YOLO_SPEECH_BUBBLE_2)
    finally:
        mgr.models.pop(ModelType.YOLO_SPEECH_BUBBLE_2, None)
