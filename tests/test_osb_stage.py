"""The OSB stage of a page (second half): `prepare_outside_text_work` + `finish_outside_text_work` vs goldens
produced by running the REFERENCE functions (core/outside_text_processor.py:217-1691) on the synthetic page of
tests/golden/osb_page.py with the same deterministic stand-in inpainter — render-expanded boxes, background-brightness probes,
region groups, the dilated bubble guard mask, which regions go to FLUX (with which seed / clip box / mask) and which are
flat-filled with which colour, wave scheduling, failure fallbacks, and the final page — bit-exact."""
import json
import sys
import types
from pathlib import Path

import numpy as np
import pytest
import torch

G = Path(__file__).resolve().parent / "golden"
sys.path.insert(0, str(G))
import osb_page  # noqa: E402

from mangatranslator_amd.core import outside_text_processor as otp  # noqa: E402
from mangatranslator_amd.core.batch_coordinator import BatchRequestCoordinator  # noqa: E402
from mangatranslator_amd.core.image import ocr_detection  # noqa: E402
from mangatranslator_amd.utils.exceptions import ValidationError  # noqa: E402

GOLD = json.loads((G / "osb_stage.json").read_text())
ARR = np.load(G / "osb_stage_arrays.npz")


class _Boxes:
    def __init__(self, xyxy, conf):
        self.xyxy, self.conf, self.cls = torch.tensor(xyxy, dtype=torch.float32).reshape(-1, 4), torch.tensor(conf, dtype=torch.float32), torch.zeros(len(conf))


@pytest.fixture
def rig(monkeypatch):
    def boom(*a, **k):
        raise RuntimeError("bubbles are provided: no bubble detector may run")
    osb_model = lambda *a, **k: [types.SimpleNamespace(boxes=_Boxes(osb_page.OSB, osb_page.OSB_CONF))]
    mgr = types.SimpleNamespace(load_yolo_speech_bubble=boom, load_rtdetr_conjoined_bubble=boom, load_yolo_osbtext=lambda token=None: osb_model, device="cpu")
    monkeypatch.setattr(ocr_detection, "get_model_manager", lambda: mgr)
    monkeypatch.setattr(otp, "FluxKontextInpainter", osb_page.StandInInpainter)
    return osb_page.make_page()


@pytest.mark.parametrize("tag", list(GOLD))
def test_stage_matches_reference(rig, tag):
    gold = GOLD[tag]
    over = dict(gold["over"])
    coord = None if tag == "flux_no_coordinator" else BatchRequestCoordinator(2)
    cfg = osb_page.make_config(coord, gold["method"], **over)
    osb_page.StandInInpainter.calls = []
    work = otp.prepare_outside_text_work(rig, cfg, "page.png", "PNG", bubble_data=osb_page.bubble_data(), text_free_boxes=osb_page.TEXT_FREE,
                                         panels=osb_page.PANELS)
    p = gold["prepare"]
    assert [[[int(v) for v in b], float(c)] for b, c in work.outside_text_results] == p["results"]
    assert [[[float(v) for v in b], float(c)] for b, c in work.raw_outside_text_results] == p["raw"]
    assert {",".join(map(str, k)): bool(v) for k, v in work.original_text_colors.items()} == p["colors"]
    assert [dict(bbox=g["bbox"], original_bbox=g["original_bbox"], mask_indices=[int(i) for i in g["mask_indices"]]) for g in work.mask_groups] == p["groups"]
    assert np.array_equal(np.packbits(work.total_bubble_mask), ARR[f"{tag}_bubble_mask"])
    final, data = otp.finish_outside_text_work(work)
    assert data == []
    assert sorted(osb_page.StandInInpainter.calls, key=lambda c: c["seed"]) == gold["calls"]
    assert np.array_equal(np.asarray(final.convert("RGB")), ARR[f"{tag}_final"])


def test_disabled_and_refused_options(rig):
    cfg = osb_page.make_config(None, enabled=False)
    assert otp.prepare_outside_text_work(rig, cfg, "p.png", "PNG") is None
    assert otp.process_outside_text(rig, cfg, "p.png", "PNG") == (rig, [])
    with pytest.raises(ValidationError):
        otp.prepare_outside_text_work(rig, osb_page.make_config(None, enable_page_number_filtering=True), "p.png", "PNG")


def test_ring_statistics():
    ring = np.full((100, 3), 250, np.uint8)
    ring[:4] = 0                                         # 4 % outliers: still solid, snapped to white
    assert otp.ring_statistics(ring) == (True, (255, 255, 255))
    ring[:6] = 0                                         # 6 %: not solid
    assert otp.ring_statistics(ring)[0] is False
    assert otp.ring_statistics(np.full((30, 3), 9, np.uint8)) == (True, (0, 0, 0))
    assert otp.ring_statistics(np.tile(np.array([[120, 130, 140]], np.uint8), (30, 1))) == (True, (120, 130, 140))


def test_synthetic_bench_page_sends_its_block_to_flux(monkeypatch):
    """the benchmark's page generator: the text box inside the gradient block fails the solid-border test and is
    inpainted through the manager's Kontext pipeline; pixels outside the text box's clip rectangle are untouched"""
    from PIL import Image
    from mangatranslator_amd.core.ml.model_manager import ModelType, get_model_manager
    from mangatranslator_amd.utils.synthetic_pages import make_page
    W_, H_ = 512, 768
    pg, boxes, regions = make_page(3, W_, H_, bubbles=8, osb_regions=1)
    x0, y0, x1, y1 = regions[0]
    assert not any(x0 < b[2] and b[0] < x1 and y0 < b[3] and b[1] < y1 for b in boxes)        # blocks are placed clear of the bubbles
    text_boxes = [[x0 + 5.0, y0 + 5.0, x1 - 5.0, y1 - 5.0]]

    class Pipe:
        calls = 0

        def encode_prompt(self, **kw):
            return torch.zeros(512, 8), torch.zeros(8), None

        def __call__(self, image=None, width=None, height=None, **kw):
            Pipe.calls += 1
            return types.SimpleNamespace(images=[torch.full((3, height, width), 0.5)])

    mgr = get_model_manager()
    monkeypatch.setitem(mgr.models, ModelType.FLUX_KONTEXT_SDNQ_PIPELINE, Pipe())
    cfg = osb_page.make_config(BatchRequestCoordinator(1), osb_render_expansion_narrow_multiplier=1.0, text_box_proximity_ratio=0.02, seed=1)
    cfg.device = torch.device("cpu")
    page = Image.fromarray(pg)
    out, _ = otp.process_outside_text(page, cfg, "page.png", "PNG", bubble_data=[{"bbox": tuple(float(v) for v in b)} for b in boxes],
                                      text_free_boxes=text_boxes)
    assert Pipe.calls == 1
    a, b = np.asarray(page), np.asarray(out)
    changed = np.argwhere((a != b).any(-1))
    assert len(changed) > 0
    cx0, cy0, cx1, cy1 = [int(v) for v in text_boxes[0]]
    assert changed[:, 1].min() >= cx0 and changed[:, 1].max() < cx1 and changed[:, 0].min() >= cy0 and changed[:, 0].max() < cy1


def test_bubble_guard_mask_equals_full_page_dilation():
    """the windowed per-bubble dilation is the same set as one 11 x 11 maximum filter over the whole page (what cv2.dilate computes)"""
    import numpy as np
    from scipy import ndimage
    from mangatranslator_amd.core import outside_text_processor as otp
    rng = np.random.default_rng(4)
    H, W = 140, 190
    yy, xx = np.mgrid[0:H, 0:W]
    bubbles = []
    for k in range(6):
        cx, cy, a, b = rng.uniform(0, W), rng.uniform(0, H), rng.uniform(3, 40), rng.uniform(3, 30)
        m = (((xx - cx) / a) ** 2 + ((yy - cy) / b) ** 2 <= 1.0)
        if k % 3 == 0:
            bubbles.append({"bbox": (cx - a, cy - b, cx + a, cy + b)})                       # no mask: the box stands in
        elif k % 3 == 1:
            bubbles.append({"bbox": (0, 0, 1, 1), "sam_mask": m.astype(np.uint8) * 255})
        else:
            bubbles.append({"bbox": (0, 0, 1, 1), "sam_mask": np.repeat(m[..., None], 3, -1)})  # HxWx3 mask
    bubbles.append({"bbox": (0, 0, 1, 1), "sam_mask": np.zeros((H, W), bool)})                # empty mask
    total = np.zeros((H, W), bool)
    for bub in bubbles:
        m = bub.get("sam_mask")
        if m is not None:
            m = np.asarray(m)
            total |= (m[..., 0] if m.ndim == 3 else m) > 0
        else:
            x0, y0, x1, y1 = [int(c) for c in bub["bbox"]]
            total[max(0, y0):max(0, min(H, y1)), max(0, x0):max(0, min(W, x1))] = True
    want = ndimage.maximum_filter(total.astype(np.uint8), size=11, mode="constant", cval=0).astype(bool)
    assert np.array_equal(otp.build_bubble_guard_mask(bubbles, W, H), want)
