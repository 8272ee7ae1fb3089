"""Wall time of the host-side pieces of FluxKleinInpainter.inpaint_mask on a 2048x3072 page (BASELINE config 5) with a stand-in pipeline
(no GPU needed): which of them are worth moving to the device."""
import sys, os, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from PIL import Image
from mangatranslator_amd.core.image import inpainting as ip
from mangatranslator_amd.utils.synthetic_pages import make_page

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2048, 3072)
pg, boxes, regions = make_page(0, W, H, bubbles=8, osb_regions=1)
x0, y0, x1, y1 = regions[0]
mask = np.zeros((H, W), bool); mask[y0 + 10:y1 - 10, x0 + 10:x1 - 10] = True
img = Image.fromarray(pg)
inp = ip.FluxKleinInpainter.__new__(ip.FluxKleinInpainter)
inp.variant, inp.luminance_correction, inp.upscale_small_crops, inp.verbose = "4b", True, True, False
T = {}


def lap(name, t0):
    T[name] = T.get(name, 0.0) + 1e3 * (time.perf_counter() - t0)
    return time.perf_counter()


for rep in range(3):
    T.clear()
    t = time.perf_counter()
    x, y, w, h, padding, blur = inp.region_for_mask(mask); t = lap("region_for_mask", t)
    crop = img.crop((x, y, x + w, y + h)); mask_crop = mask[y:y + h, x:x + w]; t = lap("crop", t)
    alpha = inp._crop_alpha(mask_crop, blur); t = lap("_crop_alpha (EDT feather)", t)
    scaled, _, _ = inp._prepare_image_for_inference(crop); t = lap("_prepare_image_for_inference (LANCZOS to ~1 MP)", t)
    patch = Image.fromarray(np.asarray(scaled)[:, ::-1].copy()); t = lap("(stand-in pipeline)", t)
    if patch.size != (w, h):
        patch = patch.resize((w, h), Image.Resampling.LANCZOS); t = lap("resize back (LANCZOS)", t)
    patch = inp._match_luminance(patch, crop, mask_crop); t = lap("_match_luminance (Lab)", t)
    out = Image.fromarray(ip.composite_u8(np.asarray(img), np.asarray(patch), alpha, x, y)); t = lap("composite_u8 + fromarray", t)
print(f"page {W}x{H}, crop {w}x{h} -> inference {scaled.size}")
for k, v in T.items():
    print(f"  {k:50s} {v:8.1f} ms")
print(f"  {'total':50s} {sum(v for k, v in T.items() if 'stand-in' not in k):8.1f} ms")
