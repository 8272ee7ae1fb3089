"""GPU tier: `prepare_bubble_images_for_translation` with the RCAN upscaler on the HIP path — 40 bubble crops of 40
different small sizes through the shared masked bucket plans, every finished crop against the same function driven by the fp32 CPU oracle
model (PSNR >= 40 dB), plus a crops/s figure."""
import time

import numpy as np
import pytest
import torch

from mangatranslator_amd.core.services.translation import prepare_bubble_images_for_translation
from parity_log import record

pytestmark = pytest.mark.gpu


def _page_and_bubbles(n=40, seed=0):
    rng = np.random.default_rng(seed)
    H, W = 1536, 1024
    yy, xx = np.mgrid[0:H, 0:W]
    page = np.stack([(xx * 3 + yy) % 256, (xx + yy * 2) % 256, (xx * yy // 7) % 256], -1).astype(np.uint8)
    bubbles = []
    for i in range(n):
        w, h = int(rng.integers(21, 120)), int(rng.integers(21, 120))            # min side < 128: at least one model pass each
        x0, y0 = int(rng.integers(0, W - w)), int(rng.integers(0, H - h))
        page[y0:y0 + h, x0:x0 + w] = 255 - page[y0:y0 + h, x0:x0 + w] // 3
        b = {"bbox": (x0, y0, x0 + w, y0 + h), "confidence": 0.9, "class": "bubble"}
        if i % 3 == 0:
            m = np.zeros((H, W), np.uint8)
            m[max(0, y0 - 2):y0 + h + 1, x0:x0 + w + 3] = 255                       # a mask reaching past the box: the crop follows it
            b["sam_mask"] = m
        bubbles.append(b)
    return page, bubbles


def test_bubble_crops_through_bucket_plans(hip_lib):
    from mangatranslator_amd.core.ml.rcan import RCANUpscaler
    from oracle.rcan_ref import load_ref, make_state_dict
    sd = make_state_dict(n_feats=64, n_resgroups=4, n_resblocks=8, unshuffle=2, seed=5)           # the lite (pixel-unshuffle) shape
    model = RCANUpscaler(sd, device="cuda:0", lib=hip_lib)
    oracle = load_ref(sd)
    page, bubbles = _page_and_bubbles()
    bgr = np.ascontiguousarray(page[..., ::-1])
    from mangatranslator_amd.core import caching
    caching.get_cache().reset()
    got = prepare_bubble_images_for_translation(bubbles, bgr, model, "cuda:0", "image/png", 128, "model_lite")
    caching.get_cache().reset()
    want = prepare_bubble_images_for_translation(bubbles, bgr, oracle, "cpu", "image/png", 128, "model_lite")
    worst = 99.0
    for g, w in zip(got, want):
        a, b = np.asarray(g["image_pil"], np.float64), np.asarray(w["image_pil"], np.float64)
        assert a.shape == b.shape and min(a.shape[:2]) >= 128
        mse = ((a - b) ** 2).mean()
        worst = min(worst, 99.0 if mse == 0 else 10 * np.log10(255.0 ** 2 / mse))
    assert worst >= 40.0, f"worst crop PSNR {worst:.1f} dB"
    # 72 model passes over 60-odd distinct sizes (second passes run on up to 2 x 123 px): 15 canvases in 64 px steps, no per-size plan
    assert len(model._buckets) <= 16 and len(model._plans) == 0, (len(model._buckets), len(model._plans))
    caching.get_cache().reset()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    prepare_bubble_images_for_translation(bubbles, bgr, model, "cuda:0", "image/png", 128, "model_lite")
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"bubble crops: {len(bubbles)} crops in {dt * 1e3:.1f} ms = {len(bubbles) / dt:.0f} crops/s (crop -> RCAN passes -> LANCZOS fit -> PNG base64), "
          f"worst PSNR vs the oracle-model path {worst:.1f} dB, {len(model._buckets)} bucket plans")
    record("bubble_crops.40_sizes.lite", crops_per_s=len(bubbles) / dt, worst_psnr_db=float(worst), bucket_plans=len(model._buckets))
