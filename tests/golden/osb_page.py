"""The synthetic page + canned detector outputs + stand-in inpainter of the OSB-stage golden, shared by the generator
(tests/golden/make_goldens.py gen_osb_stage, which runs the REFERENCE on it) and tests/test_osb_stage.py (which runs this build)."""
import types

import numpy as np
from PIL import Image

W, H = 640, 480

BUBBLES = [dict(bbox=[40, 40, 200, 160], mask="ellipse"), dict(bbox=[420, 30, 560, 120], mask=None)]
TEXT_FREE = [[15.5, 200.5, 150.5, 290.5]]
# OSB text boxes as the (stand-in) OSB text model reports them
OSB = [[230.4, 50.2, 330.7, 100.9],        # R0 on solid white                 -> flat white fill
       [400.2, 200.5, 520.8, 260.1],       # R1 on texture                     -> FLUX  (grouped with R1b)
       [405.0, 262.0, 500.0, 290.0],       # R1b just below R1                 -> same group
       [60.3, 330.2, 180.6, 400.7],        # R2 on texture, far from R1        -> FLUX, same wave as R1
       [470.1, 295.3, 600.4, 345.9],       # R3 on texture next to R1          -> FLUX, context overlaps R1: next wave
       [250.5, 380.2, 350.1, 430.6],       # R4 on solid grey 128              -> flat (128,128,128) fill
       [30.2, 430.4, 120.8, 465.9],        # R5 on near-black 5                -> snapped black fill
       [562.0, 40.0, 566.0, 100.0],        # R6 hugging bubble B1              -> nothing left after the bubble guard
       [606.2, 300.1, 621.7, 420.9],       # R7 narrow / tall, in a panel      -> render-expanded, texture -> FLUX
       [215.3, 190.6, 300.9, 240.2],       # R8 on texture                     -> FLUX call raises -> fallback fill
       [212.3, 262.6, 290.9, 300.2]]       # R9 on texture                     -> FLUX returns the page untouched -> fallback fill
OSB_CONF = [0.91, 0.82, 0.8, 0.73, 0.64, 0.55, 0.86, 0.77, 0.68, 0.9, 0.7]
PANELS = [(590, 250, 638, 470)]
SEED = 11


def make_page() -> Image.Image:
    yy, xx = np.mgrid[0:H, 0:W]
    tex = ((xx * 7 + yy * 13) % 97 + 80).astype(np.uint8)                   # deterministic non-solid texture
    page = np.stack([tex, np.roll(tex, 5, axis=1), np.roll(tex, 9, axis=0)], axis=-1)
    page[0:180, 0:360] = 255                                                # white area (bubble B0 and R0 live here)
    page[360:450, 230:370] = 128                                            # grey block
    page[420:480, 0:140] = 5                                                # near-black block
    page[20:130, 410:570] = 250                                             # near-white area around bubble B1
    for x0, y0, x1, y1 in OSB:                                              # a few dark strokes inside every text box
        for k in range(3):
            y = int(y0 + (k + 1) * (y1 - y0) / 4)
            page[y:y + 2, int(x0) + 3:max(int(x0) + 4, int(x1) - 3)] = 20 if page[y, int(x0) + 3, 0] > 60 else 230
    return Image.fromarray(page)


def bubble_data():
    out = []
    yy, xx = np.mgrid[0:H, 0:W]
    for b in BUBBLES:
        x0, y0, x1, y1 = b["bbox"]
        d = dict(bbox=tuple(b["bbox"]), confidence=0.9)
        if b["mask"] == "ellipse":
            d["sam_mask"] = ((((xx - (x0 + x1) / 2) / ((x1 - x0) / 2)) ** 2 + ((yy - (y0 + y1) / 2) / ((y1 - y0) / 2)) ** 2 <= 1.0) * 255).astype(np.uint8)
        out.append(d)
    return out


def make_config(coordinator, method="flux_kontext", **over):
    ot = dict(enabled=True, enable_page_number_filtering=False, page_filter_margin_threshold=0.1, page_filter_min_area_ratio=0.05,
              min_area_ignore_ratio=0.0, seed=SEED, huggingface_token="", inpainting_method=method, flux_backend="sdnq", flux_low_vram=False,
              flux_sdcpp_cache_mode="none", flux_sdcpp_diffusion_quant="", flux_sdcpp_text_encoder_quant="", flux_num_inference_steps=4,
              flux_luminance_correction=True, flux_upscale_small_crops=True, flux_group_regions=False, flux_residual_diff_threshold=0.15,
              osb_confidence=0.5, osb_text_free_only=False, bbox_expansion_percent_width=0.1, bbox_expansion_percent_height=0.1,
              osb_render_expansion_narrow_multiplier=1.5, osb_render_expansion_tiny_multiplier=1.0,
              osb_render_expansion_aspect_ratio_threshold=0.4, osb_render_expansion_area_ratio_threshold=0.005, text_box_proximity_ratio=0.1)
    ot.update(over)
    return types.SimpleNamespace(device="cpu", yolo_model_path=None, outside_text=types.SimpleNamespace(**ot),
                                 detection=types.SimpleNamespace(conjoined_confidence=0.35, bubble_detector_model="yolo_2"),
                                 request_coordinator=coordinator)


class StandInInpainter:
    """deterministic stand-in for FluxKontextInpainter: paints the (clipped) mask with a function of position and seed, records its
    calls; the region whose clip box starts at x = 215 raises, the one starting at x = 212 hands the page back untouched"""
    calls = []

    def __init__(self, **kw):
        self.kw = kw

    def inpaint_mask(self, image_pil, mask_np, seed=1, verbose=False, ocr_params=None, strict_mask_clipping=False, composite_clip_bbox=None):
        m = np.asarray(mask_np).astype(bool).copy()
        ys, xs = np.nonzero(m)
        StandInInpainter.calls.append(dict(seed=int(seed), clip=[int(v) for v in composite_clip_bbox] if composite_clip_bbox else None,
                                           mask_bbox=[int(xs.min()), int(ys.min()), int(xs.max()) + 1, int(ys.max()) + 1], area=int(m.sum()),
                                           strict=bool(strict_mask_clipping)))
        if composite_clip_bbox and composite_clip_bbox[0] == 215:
            raise RuntimeError("stand-in failure")
        if composite_clip_bbox and composite_clip_bbox[0] == 212:
            return image_pil
        if strict_mask_clipping and composite_clip_bbox:
            x0, y0, x1, y1 = composite_clip_bbox
            clip = np.zeros_like(m)
            clip[max(0, y0):max(0, y1), max(0, x0):max(0, x1)] = True
            m &= clip
        arr = np.array(image_pil.convert("RGB"))
        ys, xs = np.nonzero(m)
        for c in range(3):
            arr[ys, xs, c] = (xs * 3 + ys * 5 + seed * 7 + c * 40) % 256
        return Image.fromarray(arr)
