"""Page mode rule: `convert_image_to_target_mode` / `resize_to_max_side` vs the reference functions
(tests/golden/make_mode_goldens.py) for every mode a page can arrive in — transparency is flattened onto white on the way to RGB, not
dropped — and `load_page` applying it."""
import hashlib
import json
import sys
from pathlib import Path

import numpy as np
import pytest
from PIL import Image

from mangatranslator_amd.core import pipeline
from mangatranslator_amd.core.image import image_utils as iu

G = Path(__file__).resolve().parent / "golden"
GOLD = json.loads((G / "page_modes.json").read_text())
sys.path.insert(0, str(G))


def _sources():
    rng = np.random.default_rng(77)
    rgba = rng.integers(0, 256, (9, 7, 4), dtype=np.uint8)
    rgba[..., 3] = np.where(rng.random((9, 7)) < 0.3, 0, np.where(rng.random((9, 7)) < 0.5, 255, rgba[..., 3]))
    out = dict(rgba=Image.fromarray(rgba, "RGBA"), rgb=Image.fromarray(rgba[..., :3].copy(), "RGB"),
               la=Image.fromarray(rgba[..., [0, 3]].copy(), "LA"), l=Image.fromarray(rgba[..., 1].copy(), "L"))
    pal = out["rgb"].convert("P", palette=Image.ADAPTIVE, colors=16)
    out["p"] = pal
    pt = pal.copy(); pt.info["transparency"] = 3
    out["p_transparent"] = pt
    out["cmyk"] = out["rgb"].convert("CMYK")
    out["one_bit"] = out["l"].convert("1")
    out["i16"] = Image.fromarray((rgba[..., 0].astype(np.uint16) * 200), "I;16")
    return out


def test_mode_conversion_matches_reference():
    src = _sources()
    for key, want in GOLD.items():
        if "->" not in key:
            continue
        name, target = key.split("->")
        if "error" in want:
            with pytest.raises(Exception) as e:
                iu.convert_image_to_target_mode(src[name], target)
            assert type(e.value).__name__ == want["error"], key
            continue
        o = iu.convert_image_to_target_mode(src[name], target)
        assert (o.mode, list(o.size), o is src[name]) == (want["mode"], want["size"], want["same_object"]), key
        assert hashlib.sha256(o.tobytes()).hexdigest() == want["sha256"], key
    for t, want in GOLD["max_side"].items():
        o = iu.resize_to_max_side(src["rgb"], int(t))
        assert (list(o.size), o is src["rgb"]) == (want["size"], want["same_object"]) and hashlib.sha256(o.tobytes()).hexdigest() == want["sha256"]


def test_load_page_flattens_transparency_for_jpeg_output(tmp_path):
    src = _sources()
    src["rgba"].save(tmp_path / "a.png")
    page = pipeline.load_page(tmp_path / "a.png", "jpeg")
    assert page.mode == "RGB" and hashlib.sha256(page.tobytes()).hexdigest() == GOLD["rgba->RGB"]["sha256"]
    assert page.tobytes() != src["rgba"].convert("RGB").tobytes()                # dropping alpha would show what hides under it
    assert pipeline.load_page(tmp_path / "a.png", "png").mode == "RGBA"
    src["rgb"].save(tmp_path / "b.jpg", quality=95)
    assert pipeline.load_page(tmp_path / "b.jpg", "auto").mode == "RGB" and pipeline.load_page(tmp_path / "b.jpg", "webp").mode == "RGBA"
