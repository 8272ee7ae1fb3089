"""GPU tier: SAM-2.1 on MI355X through the C ABI vs HF Sam2Model on CPU fp32."""
import pytest

import sam2_checks as sc
from parity_log import record

pytestmark = pytest.mark.gpu


def test_sam2_tiny(hip_lib):
    sc.check_sam2(hip_lib, "cuda:0", "tiny_test", h=300, w=200, n_boxes=3, seed=0)


def test_sam2_small(hip_lib):
    err, mism = sc.check_sam2(hip_lib, "cuda:0", "small_test", h=768, w=512, n_boxes=5, seed=1)
    print(f"small_test: logits rel err {err:.4f}, mask mismatch {mism:.4%}")
    record("sam2.small_test.768x512", logit_rel_err=err, mask_mismatch_frac=mism)


def test_sam2_hiera_large_page(hip_lib):
    """Full Hiera-L geometry (48 blocks, 1024x1024 input, head_dim 72) with seeded weights on a
    1024x1536 page with 8 boxes; the CPU oracle pass takes ~10 s."""
    err, mism = sc.check_sam2(hip_lib, "cuda:0", "hiera_large", h=1536, w=1024, n_boxes=8, seed=2,
                              logit_tol=0.06, mask_tol=0.01)          # measured (profiles/r02_parity.json): 0.031 / 0.45 %
    print(f"hiera_large: logits rel err {err:.4f}, mask mismatch {mism:.4%}")
    record("sam2.hiera_large.1024x1536", boxes=8, **sc.stats)


def test_sam2_hiera_large_page_2048x3072(hip_lib):
    """BASELINE config 5 page size (the encoder input stays 1024 x 1024; what grows is the antialiased down-scale and the mask up-scale)"""
    err, mism = sc.check_sam2(hip_lib, "cuda:0", "hiera_large", h=3072, w=2048, n_boxes=8, seed=3, logit_tol=0.06, mask_tol=0.01)
    print(f"hiera_large 2048x3072: logits rel err {err:.4f}, mask mismatch {mism:.4%}")
    record("sam2.hiera_large.2048x3072", boxes=8, **sc.stats)


def test_sam2_hiera_large_calibrated_logits(hip_lib):
    """the same page with the mask tokens' hypernetwork scaled to trained-model logit spread (std 11): errors in logit units against an
    a-priori bound, and no differing pixel anywhere a logit is farther than the bound from the threshold.  f16 storage — what
    `ModelManager.load_sam2` serves under `sam_precision = "fast"` when the checkpoint allows it: max < 0.25, rms < 0.03 logit units, < 3e-4 of the page pixels differ
    after the `> 0` threshold (measured 0.087 / 0.017 / 1.7e-4, profiles/r04_sam_dtype_probe.json; VERDICT r03 asked for < 1e-4, which an
    f32 decoder tail alone cannot deliver: DESIGN.md §3)"""
    from mangatranslator_amd.hip import abi
    err, mism = sc.check_sam2(hip_lib, "cuda:0", "hiera_large", h=1536, w=1024, n_boxes=8, seed=2, logit_tol=0.01, mask_tol=3e-4, calibrated=True,
                              abs_tol=0.25, rms_tol=0.03, dtype=abi.F16)
    record("sam2.hiera_large.1024x1536.calibrated.f16", boxes=8, **sc.stats)


def test_sam2_tiny_high_precision(hip_lib):
    """precision "high" (hi + lo trunk weights, fp32 residual stream, fp32 mask decoder) at least halves the logit error of f16 storage"""
    from mangatranslator_amd.hip import abi
    sc.check_sam2(hip_lib, "cuda:0", "tiny_test", h=300, w=200, n_boxes=3, seed=0, dtype=abi.F16, calibrated=True)
    fast = sc.stats["logit_abs_err_rms"]
    sc.check_sam2(hip_lib, "cuda:0", "tiny_test", h=300, w=200, n_boxes=3, seed=0, dtype=abi.F16, calibrated=True, precision="high")
    assert sc.stats["logit_abs_err_rms"] < 0.5 * fast and sc.stats["decided_pixels_wrong"] == 0


def test_sam2_hiera_large_high_precision(hip_lib):
    """What `ModelManager` serves by default (precision "high", f16 storage), SAM-2.1 Hiera-L at 1024 x 1536 with logits at a trained model's
    spread, against the fp32 reference — the bar north_star's "bit-exact masks" turns into for a 16-bit trunk: fewer than 1e-4 of the page's
    mask pixels differ after `> 0`, none outside the error band, logit rms below 0.01 (measured on MI355X, round 5: 5.1e-5 / 0 / 0.0051,
    max 0.026; `precision="fast"`: 1.8e-4 / 0 / 0.016 — profiles/r05_parity.json)"""
    from mangatranslator_amd.hip import abi
    sc.check_sam2(hip_lib, "cuda:0", "hiera_large", h=1536, w=1024, n_boxes=8, seed=2, logit_tol=0.01, mask_tol=1e-4, calibrated=True, abs_tol=0.06, rms_tol=0.01,
                  dtype=abi.F16, precision="high")
    assert sc.stats["mask_mismatch_frac"] < 1e-4 and sc.stats["decided_pixels_wrong"] == 0 and sc.stats["wrong_beyond_1_logit"] == 0 and sc.stats["logit_abs_err_rms"] < 0.01
    record("sam2.hiera_large.1024x1536.calibrated.f16.high_precision", boxes=8, **sc.stats)


def test_sam2_hiera_large_calibrated_logits_bf16(hip_lib):
    """the fallback storage type (the reference's own GPU dtype): max < 1, rms < 0.2 logit units (measured 0.72 / 0.138, mismatch 1.55e-3)"""
    from mangatranslator_amd.hip import abi
    err, mism = sc.check_sam2(hip_lib, "cuda:0", "hiera_large", h=1536, w=1024, n_boxes=8, seed=2, logit_tol=0.06, mask_tol=0.01, calibrated=True, dtype=abi.BF16)
    record("sam2.hiera_large.1024x1536.calibrated.bf16", boxes=8, **sc.stats)


def test_sam2_repeated_calls_are_bit_identical(hip_lib):
    """the same page and boxes through one model again and again (hipGraph replays after the first call): logits, the stability-based mask
    choice and the masks never change.  23 boxes on purpose: the pixel counters of the mask choice are then 184 bytes, the size at which a
    captured hipMemsetAsync left counts uncleared on some replays (round 4: the choice of a box flipped between identical calls; the
    counters are now cleared by a kernel, csrc/mtx_device.h zero_words_async)"""
    import numpy as np
    import torch
    from oracle import sam2_ref
    from mangatranslator_amd.core.ml.sam2 import Sam2Hip
    from mangatranslator_amd.utils.synthetic_pages import make_page
    dev = torch.device("cuda:0")
    smodel, scfg = sam2_ref.make_model("tiny_test", seed=2)
    sam = Sam2Hip(smodel.state_dict(), scfg, device=dev, lib=hip_lib)
    pg, _b, _r = make_page(23, 512, 768, bubbles=8, osb_regions=0)
    for nb in (23, 28, 7, 23):
        r = np.random.default_rng(nb)
        bx = np.stack([r.integers(0, 300, nb), r.integers(0, 500, nb)], 1).astype(np.float32)
        bx = np.concatenate([bx, bx + np.stack([25 + (np.arange(nb) % 5) * 30, 32 + (np.arange(nb) % 7) * 20], 1)], 1).astype(np.float32)
        first = None
        for rep in range(8):
            masks, low, iou, sel = sam.segment(pg, bx, return_logits=True)
            got = [t.cpu() for t in (masks, low, iou, sel)]
            if first is None:
                first = got
            for a, b, name in zip(first, got, ("masks", "logits", "iou", "choice")):
                assert torch.equal(a, b), (nb, rep, name)
