"""SAM-2.1 parity: libmtx_hip graph vs HF transformers Sam2Model on CPU fp32 (oracle/sam2_ref.py)."""
import numpy as np
import torch

from mangatranslator_amd.core.ml.sam2 import Sam2Hip
from oracle import sam2_ref as sr


last = {}


def make_page(h, w, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    page = (128 + 60 * np.sin(xx / 17.0)[..., None] + 50 * np.cos(yy / 23.0)[..., None] + rng.normal(0, 12, (h, w, 3)))
    for _ in range(4):
        cx, cy, r = rng.uniform(0.2, 0.8) * w, rng.uniform(0.2, 0.8) * h, rng.uniform(0.08, 0.2) * min(h, w)
        page[((xx - cx) ** 2 + (yy - cy) ** 2) < r * r] = rng.uniform(200, 255)
    return np.clip(page, 0, 255).astype(np.uint8)


stats = {}          # figures of the last check (recorded by the GPU tests)


def calibrate_logits(model, ref, target_std=5.0):
    """Scale the mask tokens' hypernetwork output layers so that the low-resolution logits have the spread a trained SAM-2.1 shows
    (std of a few units: |logit| of 5-10 inside / outside a mask) instead of a seeded network's 0.1.  Errors are then quoted in LOGIT
    UNITS: what matters for a mask is how far a pixel's logit is from 0 compared with the error, and that ratio is what a checkpoint
    would show.  (The caller runs the oracle again: with `multimask_output=False` the mask token is picked by a stability score over
    logit thresholds of +-0.05, so the selection itself depends on the scale.)"""
    gain = target_std / ref["pred_masks"].float().std().item()
    with torch.no_grad():
        for name, p in model.named_parameters():
            if "output_hypernetworks_mlps" in name and ".proj_out." in name:
                p.mul_(gain)
    return gain


def check_sam2(lib, device, size="tiny_test", h=300, w=200, n_boxes=3, seed=0, logit_tol=0.08, mask_tol=0.01, calibrated=False, abs_tol=1.0, rms_tol=0.2, **hip_kw):
    model, cfg = sr.make_model(size, seed)
    page = make_page(h, w, seed)
    rng = np.random.default_rng(seed + 5)
    x0 = rng.uniform(0, 0.5 * w, n_boxes); y0 = rng.uniform(0, 0.5 * h, n_boxes)
    boxes = np.stack([x0, y0, x0 + rng.uniform(0.2, 0.45, n_boxes) * w, y0 + rng.uniform(0.2, 0.45, n_boxes) * h], 1).astype(np.float32)
    ref = sr.run(model, page, boxes)
    gain = 1.0
    keep = torch.arange(n_boxes)
    if calibrated:
        gain = calibrate_logits(model, ref)
        # with trained-model logit spread the single-mask token's stability score (IoU of the masks at logit thresholds +-0.05) sits near
        # its 0.98 threshold for some boxes, and the three multimask IoU predictions can be near ties: capture what the oracle's selection
        # was based on, demand the same selection wherever it is decisive, and leave a box out of the pixel comparison when an
        # arbitrarily small error may legitimately flip it (the tokens' logits themselves are compared regardless)
        cap = {}
        md = model.mask_decoder
        orig = md._dynamic_multimask_via_stability

        def spy(all_logits, all_iou):
            cap["logits"], cap["iou"], cap["stab"] = all_logits.clone(), all_iou.clone(), md._get_stability_scores(all_logits[:, :, 0:1]).flatten()
            return orig(all_logits, all_iou)

        md._dynamic_multimask_via_stability = spy
        try:
            ref = sr.run(model, page, boxes)
        finally:
            md._dynamic_multimask_via_stability = orig
    hipm = Sam2Hip(model.state_dict(), cfg, device=device, lib=lib, **hip_kw)
    last["hip"] = hipm                      # (tools/sam_frontier.py times the plans of the model it has just checked)
    masks, low, iou, sel = hipm.segment(page, boxes, return_logits=True)
    if calibrated:
        thresh = md.dynamic_multimask_stability_thresh
        multi = cap["iou"][0][:, 1:]
        sel_o = torch.where(cap["stab"] >= thresh, torch.zeros(n_boxes, dtype=torch.long), 1 + multi.argmax(-1))
        top2 = multi.topk(2, -1).values
        decisive = ((cap["stab"] - thresh).abs() > 0.01) & ((cap["stab"] >= thresh) | ((top2[:, 0] - top2[:, 1]) > 0.005))
        same = sel.cpu().long() == sel_o
        assert bool(same[decisive].all()), f"mask token selection differs on a decisive box: {sel.cpu().tolist()} vs {sel_o.tolist()}, stability {cap['stab'].tolist()}"
        keep = torch.nonzero(same).flatten()
        assert len(keep) * 2 >= n_boxes, "too few boxes left to compare"
        # every token's logits, selected or not
        dec = hipm._decoder(n_boxes)
        lg = dec.logits.view(n_boxes, -1, 4).permute(0, 2, 1).float().cpu()
        ol = cap["logits"][0].flatten(-2)
        tok_err = ((lg - ol).abs().amax(-1) / ol.abs().amax(-1)).max().item()
        assert tok_err < logit_tol, f"all-token logits rel err {tok_err:.4f}"
    masks, low, iou, sel = masks[keep.to(masks.device)], low[keep.to(low.device)], iou[keep.to(iou.device)], sel[keep.to(sel.device)]
    n_cmp = len(keep)
    low_ref = ref["pred_masks"][0, keep, 0]
    scale = low_ref.abs().max().item()
    err = (low.float().cpu() - low_ref).abs().max().item() / scale
    iou_ref = ref["iou_scores"][0, keep, 0]
    iou_sel = iou.cpu()[torch.arange(n_cmp), sel.cpu().long()]
    iou_err = (iou_sel - iou_ref).abs().max().item()
    diff = masks.cpu().bool() != ref["masks"][keep]
    mism = diff.float().mean().item()
    assert err < logit_tol, f"low-res mask logits differ: rel err {err:.4f}"
    assert iou_err < 0.03, f"iou scores differ by {iou_err:.4f}"
    assert mism < mask_tol, f"{mism:.4%} of page-resolution mask pixels differ"
    # the sharp statement (BASELINE.json: bit-exact masks after the threshold): the page-resolution logit is a convex combination of
    # low-resolution logits, so it is off by at most the largest low-resolution error `delta`; every pixel whose fp32 logit is farther
    # than delta from the threshold MUST come out the same, and only pixels inside that band may flip
    import torch.nn.functional as F
    delta = (low.float().cpu() - low_ref).abs().max().item()
    up_ref = F.interpolate(ref["pred_masks"][0, keep].float(), (h, w), mode="bilinear", align_corners=False)[:, 0]
    decided = up_ref.abs() > delta
    wrong_decided = int((diff & decided).sum().item())
    band = float((~decided).float().mean().item())
    stats.clear()
    stats.update(logit_rel_err=err, logit_abs_err=delta, mask_mismatch_frac=mism, undecidable_band_frac=band, decided_pixels_wrong=wrong_decided)
    assert wrong_decided == 0, f"{wrong_decided} pixels outside the +-{delta:.3f} logit band differ"
    if calibrated:
        # a-priori statement, not derived from the measured error: with logits of std 5 the low-resolution error stays under `abs_tol`
        # logit units (rms far below), hence no page pixel whose fp32 logit is farther than abs_tol from 0 may differ — and the pixels a
        # trained model leaves that close to the threshold are the one-pixel rims of its masks
        d = (low.float().cpu() - low_ref).abs()
        rim = up_ref.abs() <= abs_tol
        stats.update(logit_gain=gain, logit_std=low_ref.std().item(), boxes_compared=n_cmp, stability_scores=[round(v, 4) for v in cap["stab"].tolist()], logit_abs_err_rms=d.pow(2).mean().sqrt().item(),
                     logit_abs_err_p999=d.flatten().kthvalue(max(1, int(0.999 * d.numel()))).values.item(),
                     pixels_within_1_logit_frac=float(rim.float().mean().item()), wrong_beyond_1_logit=int((diff & ~rim).sum().item()))
        assert delta < abs_tol, f"low-resolution logit error {delta:.3f} logit units at std 5"
        assert stats["logit_abs_err_rms"] < rms_tol, stats["logit_abs_err_rms"]      # measured at Hiera-L, logit std 11.2: 0.138 with bf16 storage, 0.017 with f16
        assert stats["wrong_beyond_1_logit"] == 0
    return err, mism
