"""Bubble crops for the OCR / translation request vs the reference `prepare_bubble_images_for_translation`
(core/services/translation.py:2097-2258) run on the same synthetic page with the same stand-in 2x model
(tests/golden/make_bubble_crop_goldens.py): crop extent (box ∪ mask), conjoined white-out, model passes, exact min-side fit, memo hits —
the BGR(A) array each crop has when it reaches the encoder, bit-exact (sha-256)."""
import base64
import hashlib
import io
import json
from pathlib import Path

import numpy as np
import torch
from PIL import Image

from mangatranslator_amd.core.image import image_utils as iu
from mangatranslator_amd.core.services import translation as tr
from test_image_utils import _fake_upscaler

GOLD = json.loads((Path(__file__).resolve().parent / "golden" / "bubble_crops.json").read_text())


def _scene(channels):
    H, W = 160, 200
    yy, xx = np.mgrid[0:H, 0:W]
    page = np.stack([(xx * 5 + yy * 3) % 256, (xx * 2 + yy * 7) % 256, (xx * yy // 5) % 256] + ([np.full((H, W), 255)] if channels == 4 else []), -1).astype(np.uint8)

    def ellipse(cx, cy, a, b):
        return (((xx - cx) / a) ** 2 + ((yy - cy) / b) ** 2 <= 1).astype(np.uint8) * 255
    m0 = ellipse(40, 40, 30, 22)
    m1, m2 = ellipse(110, 50, 28, 30), ellipse(150, 60, 26, 32)
    m2 = np.where(m1 > 0, 0, m2).astype(np.uint8)
    dets = [dict(bbox=(14, 20, 70, 60), sam_mask=m0, confidence=0.9),
            dict(bbox=(82, 20, 138, 80), sam_mask=m1, conjoined_neighbor_bboxes=[(124, 28, 176, 92)], confidence=0.8),
            dict(bbox=(124, 28, 176, 92), sam_mask=np.repeat(m2[..., None], 3, axis=2), conjoined_neighbor_bboxes=[(82, 20, 138, 80), (1, 2, 3, 4)], confidence=0.7),
            dict(bbox=(20, 100, 90, 150), confidence=0.6),
            dict(bbox=(120, 110, 150, 140), sam_mask=np.zeros((H, W), np.uint8), confidence=0.5)]
    return page, dets


def test_crops_match_reference():
    passes = [0]

    def model(t):
        passes[0] += 1
        return _fake_upscaler(t)
    for name, g in GOLD.items():                 # in the generator's order: "model_again" must be served from the memo
        page, dets = _scene(g["channels"])
        before = [dict(d) for d in dets]
        n0 = passes[0]
        res = tr.prepare_bubble_images_for_translation(dets, page, model, torch.device("cpu"), g["mime"], g["min_side"], g["method"], g["whiteout"])
        assert passes[0] - n0 == g["passes"], name
        assert len(res) == len(dets) and all(set(b) == set(d) for b, d in zip(before, dets)), "the input dicts are not touched"
        for i, r in enumerate(res):
            arr = iu.pil_to_cv2(r["image_pil"])
            assert list(arr.shape) == g["shapes"][i], (name, i)
            assert hashlib.sha256(arr.tobytes()).hexdigest() == g["sha256"][i], (name, i)
            assert sorted(k for k in r if k not in ("sam_mask", "image_pil")) == g["keys"][i] and r["mime_type"] == g["mime"]
            decoded = Image.open(io.BytesIO(base64.b64decode(r["image_b64"])))
            assert decoded.format == ("PNG" if g["mime"] == "image/png" else "JPEG") and decoded.size == r["image_pil"].size
            if g["mime"] == "image/png":
                assert np.array_equal(np.asarray(decoded), np.asarray(r["image_pil"]))
        from mangatranslator_amd.core.caching import get_cache
        assert get_cache().get_cache_stats()["upscale"] == g["memo"], name


def test_channel_helpers_and_min_side():
    rgb = Image.fromarray(np.arange(2 * 3 * 3, dtype=np.uint8).reshape(2, 3, 3))
    assert np.array_equal(iu.pil_to_cv2(rgb), np.asarray(rgb)[..., ::-1]) and iu.cv2_to_pil(iu.pil_to_cv2(rgb)).tobytes() == rgb.tobytes()
    rgba = rgb.convert("RGBA")
    assert np.array_equal(iu.pil_to_cv2(rgba)[..., 3], np.asarray(rgba)[..., 3]) and iu.cv2_to_pil(iu.pil_to_cv2(rgba)).tobytes() == rgba.tobytes()
    gray = rgb.convert("L")
    assert iu.pil_to_cv2(gray).shape == (2, 3) and iu.cv2_to_pil(iu.pil_to_cv2(gray)).mode == "L"
    im = Image.new("RGB", (30, 45))
    assert iu.resize_to_min_side(im, 30) is im and iu.resize_to_min_side(im, 100).size == (100, 150) and iu.resize_to_min_side(im, 7).size == (7, 10)
