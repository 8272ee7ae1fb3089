"""wall time of the host-side pieces of FluxKontextInpainter.inpaint_mask on a 1024x1536 page (no GPU needed)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from PIL import Image
from mangatranslator_amd.core.image.inpainting import FluxKontextInpainter, composite_u8
from mangatranslator_amd.utils.synthetic_pages import make_page
pg, boxes, regions = make_page(0, 1024, 1536, bubbles=8, osb_regions=1)
x0, y0, x1, y1 = regions[0]
mask = np.zeros((1536, 1024), bool); mask[y0:y1, x0:x1] = True
inp = FluxKontextInpainter.__new__(FluxKontextInpainter)
inp.PREFERED_KONTEXT_RESOLUTIONS = None
from mangatranslator_amd.core.image import inpainting as ip
inp.PREFERED_KONTEXT_RESOLUTIONS = list(ip.PREFERRED_KONTEXT_RESOLUTIONS); inp.context_padding_ratio = ip.CONTEXT_PADDING_RATIO; inp.max_context_padding = ip.MAX_CONTEXT_PADDING
img = Image.fromarray(pg)
for _ in range(2):
    t0 = time.perf_counter(); alpha, x, y, w, h, pad, blur = inp.region_for_mask(mask, False, None); t1 = time.perf_counter()
    crop = img.crop((x, y, x + w, y + h)); scaled = inp.flux_kontext_image_scale(crop); t2 = time.perf_counter()
    patch = scaled.resize((w, h), Image.Resampling.LANCZOS); t3 = time.perf_counter()
    out = Image.fromarray(composite_u8(np.asarray(img), np.asarray(patch), alpha, x, y)); t4 = time.perf_counter()
print(f"region_for_mask {1e3*(t1-t0):.1f} ms, crop+scale-up {1e3*(t2-t1):.1f} ms, resize back {1e3*(t3-t2):.1f} ms, composite {1e3*(t4-t3):.1f} ms; crop {w}x{h} -> {scaled.size}")
