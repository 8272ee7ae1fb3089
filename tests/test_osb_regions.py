"""OSB (outside-speech-bubble) text regions — SURVEY.md §8 row f3, first half — vs goldens produced by running the REFERENCE
`OutsideTextDetector` (core/image/ocr_detection.py:189-808) on the same canned detector outputs
(tests/golden/make_goldens.py gen_osb): which text boxes survive the nested / in-bubble filters in each mode (all models,
provided bubbles, text_free_only, OSB model unavailable), their order, and the grouped page masks — bit-exact, including the
reference's positional pairing after a degenerate box is dropped."""
import json
import types
from pathlib import Path

import numpy as np
import pytest
import torch
from PIL import Image

from mangatranslator_amd.core.image import ocr_detection

G = Path(__file__).resolve().parent / "golden"
GOLD = json.loads((G / "osb_regions.json").read_text())
MASKS = np.load(G / "osb_regions_masks.npz")


class _Boxes:
    def __init__(self, xyxy, conf, cls):
        self.xyxy = torch.tensor(xyxy, dtype=torch.float32).reshape(-1, 4)
        self.conf, self.cls = torch.tensor(conf, dtype=torch.float32), torch.tensor(cls, dtype=torch.float32)

    def __len__(self):
        return len(self.xyxy)


class _Model:
    def __init__(self, boxes, names):
        self.boxes, self.names, self.calls = boxes, names, 0

    def __call__(self, *a, **k):
        self.calls += 1
        return [types.SimpleNamespace(boxes=self.boxes)]


@pytest.fixture
def rig(monkeypatch):
    inp = GOLD["inputs"]
    names = {int(k): v for k, v in inp["names"].items()}
    bub = _Model(_Boxes(inp["bubbles"], inp["bconf"], [0, 0]), {0: "speech_bubble"})
    sec = _Model(_Boxes(inp["secondary"], inp["sconf"], inp["scls"]), names)
    osb = _Model(_Boxes(inp["osb"], inp["oconf"], [0] * len(inp["osb"])), {0: "text"})
    state = dict(osb_ok=True)

    def load_osb(token=None):
        if not state["osb_ok"]:
            raise RuntimeError("gated repo")
        return osb

    mgr = types.SimpleNamespace(load_yolo_speech_bubble=lambda *a, **k: bub, load_rtdetr_conjoined_bubble=lambda *a, **k: sec,
                                load_yolo_osbtext=load_osb, device="cpu")
    monkeypatch.setattr(ocr_detection, "get_model_manager", lambda: mgr)
    det = ocr_detection.OutsideTextDetector(device="cpu")
    img = Image.fromarray((np.random.default_rng(5).random((inp["H"], inp["W"], 3)) * 255).astype(np.uint8))
    return types.SimpleNamespace(inp=inp, det=det, img=img, state=state, bub=bub, sec=sec, osb=osb)


def _check(results, gold):
    assert len(results) == len(gold)
    for (bbox, conf), g in zip(results, gold):
        assert [float(v) for v in bbox] == g["bbox"]              # same float32 values, same order
        assert float(conf) == g["conf"]


def test_detect_outside_text_modes(rig):
    gold, inp, det, img = GOLD["detect"], rig.inp, rig.det, rig.img
    _check(det.detect_outside_text("page.png", image_override=img), gold["detect_all_models"])
    assert (rig.bub.calls, rig.sec.calls, rig.osb.calls) == (1, 1, 1)
    provided = [dict(bbox=inp["bubbles"][0]), inp["bubbles"][1], dict(bbox=None), [1, 2, 3]]
    _check(det.detect_outside_text("page.png", image_override=img, existing_bubbles=provided, text_free_boxes=[inp["secondary"][1]]),
           gold["detect_provided_bubbles"])
    assert (rig.bub.calls, rig.sec.calls) == (1, 1)              # provided bubbles: neither bubble detector runs again
    _check(det.detect_outside_text("page.png", image_override=img, existing_bubbles=provided, text_free_only=True), gold["detect_text_free_only"])
    assert (rig.bub.calls, rig.sec.calls) == (1, 2)              # text_free_only: secondary only, OSB model skipped
    rig.state["osb_ok"] = False
    _check(det.detect_outside_text("page.png", image_override=img, min_area_ignore_ratio=0.01), gold["detect_osb_model_unavailable"])
    rig.state["osb_ok"] = True
    _check(det.detect_outside_text("page.png", image_override=img, existing_bubbles=[]), gold["detect_no_bubbles_given_empty_list"])


def test_missing_file_and_product_loader():
    det = ocr_detection.OutsideTextDetector(device="cpu")
    with pytest.raises(FileNotFoundError):
        det.detect_outside_text("/nonexistent/page.png")
    from mangatranslator_amd.core.ml.model_manager import get_model_manager
    from mangatranslator_amd.utils.exceptions import ModelError
    with pytest.raises(ModelError):                               # not built this round: callers fall back like the reference does
        get_model_manager().load_yolo_osbtext()


def _check_groups(tag, groups, gold, H, W):
    assert len(groups) == len(gold)
    for gi, (g, e) in enumerate(zip(groups, gold)):
        assert g["bbox"] == e["bbox"] and g["original_bbox"] == e["original_bbox"]
        assert [int(i) for i in g["mask_indices"]] == e["mask_indices"]
        assert float(g["confidence"]) == e["confidence"]
        assert len(g["individual_masks"]) == e["n_individual"]
        assert g["combined_mask"].dtype == bool and g["combined_mask"].shape == (H, W)
        assert np.array_equal(np.packbits(g["combined_mask"]), MASKS[f"{tag}_{gi}_combined"])
        assert np.array_equal(np.packbits(np.stack(g["individual_masks"])), MASKS[f"{tag}_{gi}_individual"])


def test_get_text_masks(rig):
    inp, det, img = rig.inp, rig.det, rig.img
    res_a = [(np.asarray(r["bbox"], np.float32), r["conf"]) for r in GOLD["detect"]["detect_all_models"]]
    for tag, m in GOLD["masks"].items():
        if tag == "too_large":
            continue
        if m["source"] == "raw":
            results = [(np.asarray(inp["osb"][i], np.float32), float(np.float32(inp["oconf"][i]))) for i in m["raw_order"]]
            results = [(b, float(inp["oconf"][i])) for (b, _), i in zip(results, m["raw_order"])]
        else:
            results = res_a
        ew, eh, ratio = m["args"]
        groups, page = det.get_text_masks("page.png", ew, eh, ratio, image_override=img, existing_results=results)
        assert page is img
        _check_groups(tag, groups, m["groups"], inp["H"], inp["W"])
    m = GOLD["masks"]["too_large"]
    big = Image.new("RGB", tuple(m["size"]), "white")
    far = [(np.asarray(r["bbox"], np.float32), r["conf"]) for r in m["results"]]
    groups, _ = det.get_text_masks("page.png", *m["args"], image_override=big, existing_results=far)
    _check_groups("too_large", groups, m["groups"], m["size"][1], m["size"][0])
    assert det.get_text_masks("page.png", image_override=img, existing_results=[]) == (None, None)


def test_grouping_order_and_threshold():
    det = ocr_detection.OutsideTextDetector(device="cpu")
    boxes = [[0, 0, 10, 10], [100, 100, 110, 110], [4, 4, 14, 14], [104, 100, 114, 110], [300, 300, 310, 310]]
    res = [(b, 0.5) for b in boxes]
    groups = det._group_text_boxes_spatially(boxes, res, 500, 400, 0.02)          # threshold 8 px
    assert [g[2] for g in groups] == [[0, 2], [1, 3], [4]]
    assert det._boxes_are_nearby([0, 0, 10, 10], [8, 0, 18, 10], 8.0) and not det._boxes_are_nearby([0, 0, 10, 10], [8, 1, 18, 11], 8.0)
    assert det._group_text_boxes_spatially([], [], 10, 10) == []


def test_detector_memo_counts(rig):
    """how often each detector runs over a script of calls on one page — the one-entry detector slot is shared by the bubble detector
    and the OSB text model, which evict each other — equal to the reference with its real memo (tests/golden/make_cache_goldens.py osb_memo)"""
    memo = json.loads((G / "cache_keys.json").read_text())["osb_memo"]
    inp, det, img = rig.inp, rig.det, rig.img
    provided = [dict(bbox=inp["bubbles"][0]), inp["bubbles"][1], dict(bbox=None), [1, 2, 3]]
    script = dict(all_models={}, all_models_again={}, provided=dict(existing_bubbles=provided, text_free_boxes=[inp["secondary"][1]]),
                  text_free_only=dict(existing_bubbles=provided, text_free_only=True), osb_unavailable=dict(min_area_ignore_ratio=0.01),
                  empty_list=dict(existing_bubbles=[]), other_confidence=dict(confidence=0.5))
    from mangatranslator_amd.core.caching import get_cache
    for row in memo:
        rig.state["osb_ok"] = row["tag"] != "osb_unavailable"
        res = det.detect_outside_text("page.png", image_override=img, **script[row["tag"]])
        assert [rig.bub.calls, rig.sec.calls, rig.osb.calls] == row["calls"], row["tag"]
        assert len(res) == row["n"] and get_cache().get_cache_stats()["yolo"] == row["yolo_slot"]
