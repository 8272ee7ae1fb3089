"""RCAN upscaler parity: libmtx_hip graph vs the fp32 CPU oracle (oracle/rcan_ref.py)."""
import math

import torch

from mangatranslator_amd.core.ml.rcan import RCANUpscaler
from oracle.rcan_ref import load_ref, make_state_dict

PSNR_MIN_DB = 40.0   # BASELINE.json north_star: PSNR >= 40 dB on upscaled pixels


def psnr_unit(y, ref):
    """PSNR on the reference's own [0,1] image scale after tensor_to_image's clamp
    (reference core/image/image_utils.py:361-366)."""
    a, b = y.float().clamp(0, 1), ref.float().clamp(0, 1)
    mse = ((a - b) ** 2).mean().item()
    return 99.0 if mse == 0 else 10 * math.log10(1.0 / mse)


def check_rcan(lib, device, h, w, n_resgroups, n_resblocks, n_feats=64, unshuffle=1, mean_shift=False, seed=0, n=1):
    sd = make_state_dict(n_feats=n_feats, n_resgroups=n_resgroups, n_resblocks=n_resblocks,
                         unshuffle=unshuffle, mean_shift=mean_shift, seed=seed)
    ref = load_ref(sd)
    model = RCANUpscaler(sd, device=device, lib=lib)
    g = torch.Generator().manual_seed(seed + 100)
    x = torch.rand(n, 3, h, w, generator=g)
    y = model(x).cpu()
    yr = ref(x)
    assert y.shape == yr.shape == (n, 3, 2 * h, 2 * w)
    p = psnr_unit(y, yr)
    # uint8 pages as tensor_to_image produces them
    u8 = (y.clamp(0, 1) * 255).to(torch.uint8)
    u8r = (yr.clamp(0, 1) * 255).to(torch.uint8)
    frac_off = ((u8.int() - u8r.int()).abs() > 1).float().mean().item()
    assert p >= PSNR_MIN_DB, f"PSNR {p:.1f} dB < {PSNR_MIN_DB}"
    max_off = (u8.int() - u8r.int()).abs().max().item()
    print(f"RCAN {n_resgroups}x{n_resblocks}: PSNR {p:.2f} dB, uint8 |diff|>1 on {frac_off:.2%} of pixels, max {max_off} levels")
    assert max_off <= 16, f"uint8 pages differ by up to {max_off} levels"
    return p
