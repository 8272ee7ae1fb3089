"""GPU tier: FLUX.1-Kontext graphs through the C ABI on gfx950 vs the fp32 CPU oracle."""
import pytest

import flux_checks as fc
from parity_log import record

pytestmark = pytest.mark.gpu

MID = dict(d=256, heads=2, layers=2, single_layers=3, joint_dim=128, pooled_dim=64, axes_dim=(16, 56, 56), vae_ch=(32, 64, 128, 128), groups=8)


def test_dit_step_hd64(hip_lib):
    fc.check_dit_step(hip_lib, "cuda:0")


def test_dit_step_hd128(hip_lib):
    record("flux1.dit_step.mid.bf16", velocity_rel_err=fc.check_dit_step(hip_lib, "cuda:0", h2=8, w2=12, t_txt=32, **MID))


def test_vae(hip_lib):
    fc.check_vae(hip_lib, "cuda:0", h=128, w=192, **MID)


def test_kontext_loop(hip_lib):
    e, p = fc.check_kontext(hip_lib, "cuda:0", h=128, w=192, t_txt=32, steps=4, **MID)
    record("flux1.kontext.4steps.mid.bf16", latent_rel_err=e, image_psnr_db=p)
    assert p >= fc.PSNR_MIN_DB


def test_full_depth_kontext_step(hip_lib):
    """57 blocks at d = 3072, T = 2 048: one step of the real geometry against a single fp32 pass of the oracle (streamed block by block)"""
    e, c = fc.check_full_depth_step(hip_lib, "cuda:0")
    record("flux1.full_depth_step.19+38.d3072.T2048", velocity_rel_err=e, cosine=c)


def test_first_block_cache(hip_lib):
    """`residual_diff_threshold` (reference core/ml/model_manager.py:1159-1162): off and never-passing thresholds give the one-plan step's bytes
    (hipGraph replays of head + body), the skip path reproduces a computed step on equal inputs, an always-passing threshold skips every
    step after the first.  Parity with nunchaku's implementation itself is unpinned (the wheel cannot be installed here)."""
    record("flux1.first_block_cache.mid.bf16", skip_vs_computed_velocity_rel_err=fc.check_first_block_cache(hip_lib, "cuda:0", h=128, w=192, t_txt=32, steps=4, **MID))
