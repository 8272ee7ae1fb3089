"""FluxKleinInpainter host math vs golden vectors produced by running the reference class here (tests/golden/make_klein_goldens.py):
geometry, luminance match, and the whole operator around a deterministic stand-in pipeline — bit-exact pages."""
import json
import threading
import types
from pathlib import Path

import numpy as np
import pytest
from PIL import Image

from mangatranslator_amd.core.image import color, inpainting

G = Path(__file__).resolve().parent / "golden"
GEO = json.load(open(G / "klein_geometry.json"))
ARR = np.load(G / "klein_arrays.npz")
CALLS = []


def fake_pipeline(**kw):
    """the generator's stand-in (same arithmetic)"""
    img = kw["image"].convert("RGB").resize((kw["width"], kw["height"]), Image.BILINEAR)
    a = np.asarray(img, dtype=np.float32)
    out = np.clip(a * np.array([0.62, 0.70, 0.78], np.float32) + np.array([30.0, 18.0, 4.0], np.float32), 0, 255).astype(np.uint8)
    CALLS.append(dict(width=int(kw["width"]), height=int(kw["height"]), steps=int(kw["num_inference_steps"]), guidance=float(kw["guidance_scale"]),
                      mode=kw["image"].mode, size=list(kw["image"].size), keys=sorted(k for k in kw if k not in ("image", "generator"))))
    return types.SimpleNamespace(images=[Image.fromarray(out)])


def make(**kw):
    inp = inpainting.FluxKleinInpainter.__new__(inpainting.FluxKleinInpainter)
    inp.variant, inp.backend = "4b", "sdnq"
    inp.num_inference_steps, inp.low_vram, inp.verbose = 4, False, False
    inp.luminance_correction = kw.get("luminance_correction", True)
    inp.upscale_small_crops = kw.get("upscale_small_crops", True)
    inp.sdcpp_cache_mode, inp.sdcpp_diffusion_quant, inp.sdcpp_text_encoder_quant = "none", "", ""
    inp.pipeline = fake_pipeline
    inp.load_models = lambda *a, **k: None
    inp._prompt_kwargs = lambda: {"prompt_embeds": "EMBEDS"}
    inp.manager = types.SimpleNamespace(flux_inference_lock=threading.Lock())
    inp.cache = types.SimpleNamespace(should_use_inpaint_cache=lambda seed: False)
    return inp


def _mask(name, h, w):
    return np.unpackbits(ARR[name])[: h * w].reshape(h, w).astype(bool)


def test_klein_geometry_matches_reference():
    inp = make()
    for d, q in GEO["quantize"]:
        assert inp._quantize_dimension(d) == q
    assert len(GEO["expand"]) == 40
    for args, want in GEO["expand"]:
        assert list(inp._expand_bounds_to_min_size(*args)) == want
    for c in GEO["prepare"]:
        inp.upscale_small_crops = c["upscale"]
        out, ow, oh = inp._prepare_image_for_inference(Image.new("RGB", tuple(c["size"])))
        assert list(out.size) == c["out"] and [ow, oh] == c["orig"]


def test_klein_luminance_match_matches_reference():
    inp = make()
    assert list(inp._compute_luminance_stats(np.zeros((4, 4, 3), np.uint8), np.zeros((4, 4), bool))) == GEO["luminance_empty"]
    for ci, c in enumerate(GEO["luminance"]):
        orig, gen, m = ARR[f"lum_orig{ci}"], ARR[f"lum_gen{ci}"], _mask(f"lum_mask{ci}", c["h"], c["w"])
        assert list(inp._compute_luminance_stats(orig, ~m)) == c["stats_orig"]
        assert list(inp._compute_luminance_stats(gen, ~m)) == c["stats_gen"]
        g = Image.fromarray(gen)
        out = inp._match_luminance(g, Image.fromarray(orig), m)
        assert np.array_equal(np.asarray(out), ARR[f"lum_out{ci}"])
        if c["same_object"]:
            assert out is g
    assert sum(not c["same_object"] for c in GEO["luminance"]) >= 4        # the correction branch really ran


def test_klein_inpaint_mask_matches_reference():
    assert len(GEO["pages"]) == 10
    for ci, c in enumerate(GEO["pages"]):
        inp = make(**c["cfg"])
        pg, m = ARR[f"page{ci}"], _mask(f"mask{ci}", c["h"], c["w"])
        img = Image.fromarray(pg)
        assert img.mode == c["mode"]
        CALLS.clear()
        out = inp.inpaint_mask(img, m, seed=7, strict_mask_clipping=c["strict"], composite_clip_bbox=c["clip"])
        assert CALLS == c["calls"], f"page {ci}: pipeline called differently"
        assert out.mode == c["mode"] and np.array_equal(np.asarray(out), ARR[f"out{ci}"]), f"page {ci}: composited page differs"
        assert not np.array_equal(np.asarray(out), pg)
    img = Image.new("RGB", (64, 64))
    assert make().inpaint_mask(img, np.zeros((64, 64), bool)) is img
    assert make().inpaint_mask(img, np.zeros((64, 64), np.uint8)) is img


def test_klein_surface():
    """constructor contract of the reference class (:1020-1068)"""
    with pytest.raises(ValueError):
        inpainting.FluxKleinInpainter(variant="2b")
    with pytest.raises(ValueError):
        inpainting.FluxKleinInpainter(backend="nunchaku")
    k = inpainting.FluxKleinInpainter
    assert (k.KLEIN_MAX_STEPS, k.KLEIN_DEFAULT_STEPS, k.KLEIN_GUIDANCE_SCALE, k.MIN_RESOLUTION, k.MAX_RESOLUTION, k.RESOLUTION_MULTIPLE,
            k.MAX_INFERENCE_PIXELS, k.KLEIN_PADDING_MULTIPLIER) == (12, 4, 1.0, 64, 2048, 16, 4_000_000, 2.0)


def test_lab_known_values():
    """hand-derived known answers of the 8-bit Lab convention (L * 255 / 100, a + 128, b + 128): black, white, mid grey (L* = 53.6 -> 137),
    sRGB red (L* 53.2, a* 80.1, b* 67.2 -> 136, 208, 195)"""
    px = np.array([[[0, 0, 0], [255, 255, 255], [128, 128, 128], [255, 0, 0]]], np.uint8)
    assert color.rgb_to_lab_u8(px).tolist() == [[[0, 128, 128], [255, 128, 128], [137, 128, 128], [136, 208, 195]]]
    grey = np.repeat(np.arange(256, dtype=np.uint8)[None, :, None], 3, -1)
    back = color.lab_to_rgb_u8(color.rgb_to_lab_u8(grey))
    assert np.abs(back.astype(int) - grey.astype(int)).max() <= 1
