"""Deterministic stand-in for FluxKleinInpainter / FluxKontextInpainter, shared by the golden generator (plugged into the REFERENCE's
cleaning module) and the tests (plugged into this package's): accepts either class's constructor arguments, repaints the masked pixels
with a fixed function of the page and the seed, and records how it was called."""
import numpy as np
from PIL import Image


class StandInInpainter:
    calls = []
    fail = False

    def __init__(self, *a, **kw):
        self.kind = "klein" if "variant" in kw else "kontext"
        self.variant = kw.get("variant")
        self.steps = kw.get("num_inference_steps")

    def inpaint_mask(self, image_pil, mask_np, seed=1, verbose=False, ocr_params=None, **kw):
        mask = np.asarray(mask_np).astype(bool)
        StandInInpainter.calls.append(dict(kind=self.kind, variant=self.variant, steps=int(self.steps), seed=int(seed), mode=image_pil.mode,
                                           pixels=int(mask.sum()), ocr_type=(ocr_params or {}).get("type"),
                                           ocr_bbox=[int(v) for v in (ocr_params or {}).get("bbox", ())]))
        if StandInInpainter.fail and seed % 2 == 1:
            raise RuntimeError("stand-in failure")
        arr = np.asarray(image_pil).copy()
        yy, xx = np.nonzero(mask)
        arr[yy, xx, :3] = np.stack([(xx * 3 + seed) % 200 + 30, (yy * 5 + 2 * seed) % 180 + 40, (xx + yy) % 160 + 50], -1).astype(np.uint8)
        return Image.fromarray(arr, image_pil.mode)
