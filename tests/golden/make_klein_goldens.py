"""Golden vectors for the host half of the reference's DEFAULT inpainter, FluxKleinInpainter (core/image/inpainting.py:980-1665),
produced by IMPORTING THE REFERENCE in this container:

    python tests/golden/make_klein_goldens.py        # rewrites tests/golden/klein_geometry.json, klein_arrays.npz

  geometry      :1126-1163 (_quantize_dimension, _expand_bounds_to_min_size), :1258-1313 (_prepare_image_for_inference: ~1 MP / 4 MP
                cap / multiples of 16 in [64, 2048])
  luminance     :1165-1256 (_compute_luminance_stats, _match_luminance) — the two cv2.cvtColor calls (COLOR_RGB2LAB / COLOR_LAB2RGB;
                OpenCV is absent here) are served by the restatement in oracle/cv2_color_ref.py: the FLOW is pinned, the primitive is not
  operator      :1350-1665 (inpaint_mask) with a deterministic stand-in pipeline: crop geometry with doubled padding, minimum size,
                /16 quantisation and slide-back, EDT feather inside the crop, strict / clip-bbox masking, LANCZOS round trip,
                luminance match, fp32 alpha composite with uint8 truncation, and what the pipeline was called with
"""
import importlib.machinery
import importlib.util
import json
import sys
import threading
import types
from pathlib import Path
from unittest.mock import MagicMock

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
REF = "/root/reference"
sys.path.insert(0, str(HERE.parent.parent))
from tests.golden import cv2_shim  # noqa: E402

for name in ["spandrel", "ultralytics", "oxipng", "diffusers", "sdnq", "skia", "uharfbuzz", "manga_ocr", "pythainlp", "pythainlp.tokenize",
             "gradio", "torchvision"]:
    try:
        if importlib.util.find_spec(name) is not None:
            continue
    except (ImportError, ValueError):
        pass
    stub = MagicMock(name=name)
    stub.__spec__ = importlib.machinery.ModuleSpec(name, None)
    sys.modules.setdefault(name, stub)
cv2 = types.ModuleType("cv2")
cv2.__dict__.update(vars(cv2_shim.namespace))
sys.modules["cv2"] = cv2
for pkg in ["core", "core.image", "core.ml", "core.text", "core.services"]:
    m = types.ModuleType(pkg)
    m.__path__ = [REF + "/" + pkg.replace(".", "/")]
    sys.modules[pkg] = m
sys.path.insert(0, REF)
for modname in ("core.ml.model_manager", "core.ml.sdcpp_server"):
    _mm = MagicMock(name=modname)
    _mm.__spec__ = importlib.machinery.ModuleSpec(modname, None)
    sys.modules[modname] = _mm

from core.image import inpainting  # noqa: E402
from PIL import Image  # noqa: E402

CALLS = []


def fake_pipeline(**kw):
    """stand-in for Flux2KleinPipeline: the input at the requested size, darker, lower contrast and with a colour cast"""
    img = kw["image"].convert("RGB").resize((kw["width"], kw["height"]), Image.BILINEAR)
    a = np.asarray(img, dtype=np.float32)
    out = np.clip(a * np.array([0.62, 0.70, 0.78], np.float32) + np.array([30.0, 18.0, 4.0], np.float32), 0, 255).astype(np.uint8)
    CALLS.append(dict(width=int(kw["width"]), height=int(kw["height"]), steps=int(kw["num_inference_steps"]), guidance=float(kw["guidance_scale"]),
                      mode=kw["image"].mode, size=list(kw["image"].size), keys=sorted(k for k in kw if k not in ("image", "generator"))))
    return types.SimpleNamespace(images=[Image.fromarray(out)])


def make(variant="4b", **kw):
    inp = inpainting.FluxKleinInpainter.__new__(inpainting.FluxKleinInpainter)
    inp.variant, inp.backend = variant, "sdnq"
    inp.num_inference_steps = 4
    inp.low_vram = False
    inp.luminance_correction = kw.get("luminance_correction", True)
    inp.upscale_small_crops = kw.get("upscale_small_crops", True)
    inp.sdcpp_cache_mode, inp.sdcpp_diffusion_quant, inp.sdcpp_text_encoder_quant = "none", "", ""
    inp.verbose = False
    inp.DEVICE = torch.device("cpu")
    inp.pipeline = fake_pipeline
    inp.sdcpp_assets = None
    inp.load_models = lambda *a, **k: None
    inp._get_prompt_embeddings = lambda *a, **k: ("EMBEDS", None)
    inp.manager = types.SimpleNamespace(flux_inference_lock=threading.Lock())
    inp.cache = types.SimpleNamespace(should_use_inpaint_cache=lambda seed: False)
    return inp


def page(rng, h, w, kind):
    yy, xx = np.mgrid[0:h, 0:w]
    if kind == 0:      # screentone-like grey page
        g = (128 + 90 * np.sin(xx / 3.0) * np.sin(yy / 4.0)).astype(np.uint8)
        return np.stack([g, g, g], -1)
    if kind == 1:      # colour gradient
        return np.stack([(xx * 3 + yy) % 256, (xx + yy * 2) % 256, (xx * yy // 7) % 256], -1).astype(np.uint8)
    return rng.integers(0, 256, (h, w, 3), dtype=np.uint8)


def rand_mask(rng, h, w, kind):
    m = np.zeros((h, w), bool)
    if kind == 0:
        x0, y0 = int(rng.integers(0, w - 40)), int(rng.integers(0, h - 40))
        m[y0:y0 + int(rng.integers(8, 70)), x0:x0 + int(rng.integers(8, 90))] = True
    elif kind == 1:
        yy, xx = np.mgrid[0:h, 0:w]
        cx, cy = rng.uniform(0.3, 0.7) * w, rng.uniform(0.3, 0.7) * h
        m = ((xx - cx) / rng.uniform(10, 45)) ** 2 + ((yy - cy) / rng.uniform(8, 35)) ** 2 <= 1
    elif kind == 2:
        m[0:int(rng.integers(3, 20)), 0:int(rng.integers(3, 30))] = True            # tiny, flush top-left: minimum-size growth
    elif kind == 3:
        m[h - int(rng.integers(5, 40)):, w - int(rng.integers(5, 50)):] = True      # flush bottom-right
    else:
        m[int(h * 0.1):int(h * 0.9), int(w * 0.1):int(w * 0.9)] = True              # almost the whole page
    return m


def main():
    rng = np.random.default_rng(21)
    geo = dict(quantize=[], expand=[], prepare=[], luminance=[], pages=[])
    arrays = {}
    inp = make()
    for d in [0, 1, 15, 16, 63, 64, 65, 79, 80, 1000, 1023, 2047, 2048, 2049, 5000]:
        geo["quantize"].append([d, inp._quantize_dimension(d)])
    for _ in range(40):
        iw, ih = int(rng.integers(20, 400)), int(rng.integers(20, 400))
        x1, y1 = int(rng.integers(0, iw)), int(rng.integers(0, ih))
        x2, y2 = int(rng.integers(x1, iw + 1)), int(rng.integers(y1, ih + 1))
        geo["expand"].append([[x1, y1, x2, y2, iw, ih], list(inp._expand_bounds_to_min_size(x1, y1, x2, y2, iw, ih))])
    for up in (True, False):
        inp.upscale_small_crops = up
        for (w, h) in [(64, 64), (100, 37), (333, 517), (1024, 1024), (1500, 900), (2048, 2048), (3000, 2000), (4000, 300), (640, 64), (2500, 1700), (80, 4000)]:
            im = Image.new("RGB", (w, h))
            out, ow, oh = inp._prepare_image_for_inference(im)
            geo["prepare"].append(dict(upscale=up, size=[w, h], out=list(out.size), orig=[ow, oh]))
    inp.upscale_small_crops = True
    # luminance match on its own
    for ci in range(6):
        h, w = int(rng.integers(40, 90)), int(rng.integers(40, 90))
        orig = page(rng, h, w, ci % 3)
        m = rand_mask(rng, h, w, ci % 2)
        if ci == 4:
            gen = orig.copy()                                   # already matched: returned untouched
        elif ci == 5:
            gen = np.clip(orig.astype(np.int32) + rng.integers(-1, 2, orig.shape), 0, 255).astype(np.uint8)
        else:
            gen = np.clip(orig.astype(np.float32) * rng.uniform(0.5, 1.3) + rng.uniform(-40, 40, 3), 0, 255).astype(np.uint8)
        stats_o = inp._compute_luminance_stats(orig, ~m)
        stats_g = inp._compute_luminance_stats(gen, ~m)
        out = inp._match_luminance(Image.fromarray(gen), Image.fromarray(orig), m)
        arrays[f"lum_orig{ci}"], arrays[f"lum_gen{ci}"], arrays[f"lum_mask{ci}"] = orig, gen, np.packbits(m)
        arrays[f"lum_out{ci}"] = np.asarray(out)
        geo["luminance"].append(dict(h=h, w=w, stats_orig=list(stats_o), stats_gen=list(stats_g), same_object=bool(np.array_equal(np.asarray(out), gen))))
    empty = inp._compute_luminance_stats(np.zeros((4, 4, 3), np.uint8), np.zeros((4, 4), bool))
    geo["luminance_empty"] = list(empty)
    # the operator
    cfgs = [dict(), dict(upscale_small_crops=False), dict(luminance_correction=False), dict(), dict(upscale_small_crops=False, luminance_correction=False),
            dict(), dict(), dict(upscale_small_crops=False), dict(), dict()]
    for ci, cfg in enumerate(cfgs):
        inp = make(**cfg)
        h, w = int(rng.integers(90, 260)), int(rng.integers(90, 260))
        if ci == 8:
            h, w = 50, 70                                        # page smaller than the 64 px minimum
        kind = ci % 5
        m = rand_mask(rng, h, w, kind)
        pg = page(rng, h, w, ci % 3)
        mode = "RGBA" if ci == 6 else "RGB"
        img = Image.fromarray(pg)
        if mode == "RGBA":
            img = img.convert("RGBA")
            img.putalpha(Image.fromarray(((np.mgrid[0:h, 0:w][0] * 2) % 256).astype(np.uint8)))
        ys, xs = np.where(m)
        strict = bool(ci % 2)
        clip = None if ci % 3 else [int(xs.min()) - 3, int(ys.min()) - 3, int(xs.max()) + 9, int(ys.max()) + 9]
        CALLS.clear()
        out = inp.inpaint_mask(img, m, seed=7, strict_mask_clipping=strict, composite_clip_bbox=clip)
        arrays[f"page{ci}"], arrays[f"mask{ci}"], arrays[f"out{ci}"] = np.asarray(img), np.packbits(m), np.asarray(out)
        geo["pages"].append(dict(h=h, w=w, kind=kind, mode=mode, strict=strict, clip=clip, cfg=cfg, calls=list(CALLS)))
    # empty mask and non-bool mask
    inp = make()
    img = Image.fromarray(page(rng, 64, 64, 1))
    geo["empty_returns_same"] = bool(inp.inpaint_mask(img, np.zeros((64, 64), bool)) is img)
    json.dump(geo, open(HERE / "klein_geometry.json", "w"), indent=1)
    np.savez_compressed(HERE / "klein_arrays.npz", **arrays)
    print("wrote", len(arrays), "arrays;", len(geo["pages"]), "operator pages")


if __name__ == "__main__":
    main()
