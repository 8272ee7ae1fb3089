"""SAM-2.1 parity: libmtx_hip graph vs HF transformers Sam2Model on CPU fp32 (oracle/sam2_ref.py)."""
import numpy as np
import torch

from mangatranslator_amd.core.ml.sam2 import Sam2Hip
from oracle import sam2_ref as sr


def make_page(h, w, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    page = (128 + 60 * np.sin(xx / 17.0)[..., None] + 50 * np.cos(yy / 23.0)[..., None] + rng.normal(0, 12, (h, w, 3)))
    for _ in range(4):
        cx, cy, r = rng.uniform(0.2, 0.8) * w, rng.uniform(0.2, 0.8) * h, rng.uniform(0.08, 0.2) * min(h, w)
        page[((xx - cx) ** 2 + (yy - cy) ** 2) < r * r] = rng.uniform(200, 255)
    return np.clip(page, 0, 255).astype(np.uint8)


stats = {}          # figures of the last check (recorded by the GPU tests)


def check_sam2(lib, device, size="tiny_test", h=300, w=200, n_boxes=3, seed=0, logit_tol=0.08, mask_tol=0.01):
    model, cfg = sr.make_model(size, seed)
    page = make_page(h, w, seed)
    rng = np.random.default_rng(seed + 5)
    x0 = rng.uniform(0, 0.5 * w, n_boxes); y0 = rng.uniform(0, 0.5 * h, n_boxes)
    boxes = np.stack([x0, y0, x0 + rng.uniform(0.2, 0.45, n_boxes) * w, y0 + rng.uniform(0.2, 0.45, n_boxes) * h], 1).astype(np.float32)
    ref = sr.run(model, page, boxes)
    hipm = Sam2Hip(model.state_dict(), cfg, device=device, lib=lib)
    masks, low, iou, sel = hipm.segment(page, boxes, return_logits=True)
    low_ref = ref["pred_masks"][0, :, 0]
    scale = low_ref.abs().max().item()
    err = (low.float().cpu() - low_ref).abs().max().item() / scale
    iou_ref = ref["iou_scores"][0, :, 0]
    iou_sel = iou.cpu()[torch.arange(n_boxes), sel.cpu().long()]
    iou_err = (iou_sel - iou_ref).abs().max().item()
    diff = masks.cpu().bool() != ref["masks"]
    mism = diff.float().mean().item()
    assert err < logit_tol, f"low-res mask logits differ: rel err {err:.4f}"
    assert iou_err < 0.03, f"iou scores differ by {iou_err:.4f}"
    assert mism < mask_tol, f"{mism:.4%} of page-resolution mask pixels differ"
    # the sharp statement (BASELINE.json: bit-exact masks after the threshold): the page-resolution logit is a convex combination of
    # low-resolution logits, so it is off by at most the largest low-resolution error `delta`; every pixel whose fp32 logit is farther
    # than delta from the threshold MUST come out the same, and only pixels inside that band may flip
    import torch.nn.functional as F
    delta = (low.float().cpu() - low_ref).abs().max().item()
    up_ref = F.interpolate(ref["pred_masks"][0].float(), (h, w), mode="bilinear", align_corners=False)[:, 0]
    decided = up_ref.abs() > delta
    wrong_decided = int((diff & decided).sum().item())
    band = float((~decided).float().mean().item())
    stats.update(logit_rel_err=err, logit_abs_err=delta, mask_mismatch_frac=mism, undecidable_band_frac=band, decided_pixels_wrong=wrong_decided)
    assert wrong_decided == 0, f"{wrong_decided} pixels outside the +-{delta:.3f} logit band differ"
    return err, mism
