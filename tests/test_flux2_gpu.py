"""GPU tier: FLUX.2-Klein graphs (bf16 and MX-fp8 block linears) through the C ABI on gfx950 vs the fp32 CPU oracle, plus the full-width
MMDiT blocks of both FLUX families (d = 3072 was never compared with anything)."""
import pytest

import flux2_checks as f2c
import flux_checks as fc
from parity_log import record

pytestmark = pytest.mark.gpu

MID = dict(d=256, heads=2, layers=2, single_layers=3, joint_dim=128, axes_dim=(32, 32, 32, 32), vae_ch=(32, 64, 128, 128), groups=8)
DEEP = dict(d=256, heads=2, layers=5, single_layers=20, joint_dim=128, axes_dim=(32, 32, 32, 32), vae_ch=(32, 64, 128, 128), groups=8)    # Klein-4B's depth


def test_dit_step_hd64(hip_lib):
    record("flux2.dit_step.tiny.bf16", velocity_rel_err=f2c.check_dit_step(hip_lib, "cuda:0"))


def test_dit_step_hd128(hip_lib):
    record("flux2.dit_step.mid.bf16", velocity_rel_err=f2c.check_dit_step(hip_lib, "cuda:0", h2=8, w2=12, t_txt=32, **MID))


def test_dit_step_reference_grid(hip_lib):
    f2c.check_dit_step(hip_lib, "cuda:0", h2=8, w2=12, rh2=6, rw2=10, t_txt=32, **MID)


def test_dit_step_fp8(hip_lib):
    record("flux2.dit_step.mid.fp8", velocity_rel_err=f2c.check_dit_step(hip_lib, "cuda:0", h2=8, w2=12, t_txt=32, fp8=True, **MID))
    record("flux2.dit_step.klein_depth.fp8", velocity_rel_err=f2c.check_dit_step(hip_lib, "cuda:0", h2=8, w2=12, t_txt=32, fp8=True, fp8_tol=0.2, **DEEP))


def test_vae(hip_lib):
    e1, e2 = f2c.check_vae(hip_lib, "cuda:0", h=128, w=192, **MID)
    record("flux2.vae.mid", encoder_rel_err=e1, decoder_rel_err=e2)


def test_klein_loop(hip_lib):
    e, p, _ = f2c.check_klein(hip_lib, "cuda:0", h=128, w=192, t_txt=32, steps=4, **MID)
    record("flux2.klein.4steps.mid.bf16", latent_rel_err=e, image_psnr_db=p)
    assert p >= f2c.PSNR_MIN_DB


def test_klein_fp8_vs_bf16_psnr(hip_lib):
    """BASELINE config 5: the fp8 pipeline against the bf16 pipeline of the same weights (all block linears in fp8, Klein-4B's depth)"""
    p_all = f2c.check_klein_fp8_vs_bf16(hip_lib, "cuda:0", h=128, w=192, t_txt=32, steps=4, **DEEP)
    p_mlp = f2c.check_klein_fp8_vs_bf16(hip_lib, "cuda:0", h=128, w=192, t_txt=32, steps=4, fp8=("ff_in", "ff_out", "single_in", "single_out"), **DEEP)
    record("flux2.klein.4steps.klein_depth.fp8_vs_bf16", psnr_all_linears_db=p_all, psnr_mlp_and_single_only_db=p_mlp)
    assert p_all >= f2c.PSNR_MIN_DB and p_mlp >= f2c.PSNR_MIN_DB        # BASELINE.json's 40 dB bar, held by the fp8 pipeline against the bf16 pipeline (r02 measured 41.9)


def test_klein_fp8_attention_scores_psnr(hip_lib):
    """the fp8-score experiment (Flux2DiTHip(attn_qk_f8=True)): 4 Klein steps at Klein-4B's depth and a token count that takes the
    long-sequence kernel (T = 1568), image PSNR against the bf16 pipeline — recorded beside the fp8-linears-only pipeline's.  The option is the fp8 path's
    default, so BASELINE.json's 40 dB bar is asserted on it (first measured: 41.9 dB; fp8 linears alone 41.8)."""
    p_lin, p_sc, p_between, p_pv = f2c.check_klein_fp8_scores(hip_lib, "cuda:0", h=384, w=512, t_txt=32, steps=4, **DEEP)
    record("flux2.klein.4steps.klein_depth.T1568.fp8_scores", psnr_fp8_linears_vs_bf16_db=p_lin, psnr_fp8_linears_and_scores_vs_bf16_db=p_sc,
           psnr_fp8_scores_vs_fp8_linears_db=p_between, psnr_fp8_linears_scores_and_p_v_vs_bf16_db=p_pv)
    assert p_sc >= f2c.PSNR_MIN_DB and p_lin >= f2c.PSNR_MIN_DB and p_pv >= f2c.PSNR_MIN_DB      # fp8 P V is the fp8 path's default too (first measured: 41.8 dB)


def test_full_width_blocks_flux1(hip_lib):
    """one double + one single MMDiT block at FLUX.1 width and the bench's token count (d = 3072, 24 heads, T = 512 + 2 x 4070 = 8652)"""
    e = fc.check_dit_step(hip_lib, "cuda:0", h2=55, w2=74, t_txt=512, d=3072, heads=24, layers=1, single_layers=1, joint_dim=4096, pooled_dim=768,
                          axes_dim=(16, 56, 56))
    record("flux1.full_width_blocks.T8652", velocity_rel_err=e)


def test_full_width_blocks_flux2(hip_lib):
    """one double + one single Flux2 block at Klein-4B width, T = 512 + 2 x 4096 = 8704: bf16, then all linears in MX fp8"""
    kw = dict(h2=64, w2=64, t_txt=512, d=3072, heads=24, layers=1, single_layers=1, joint_dim=7680, axes_dim=(32, 32, 32, 32))
    e = f2c.check_dit_step(hip_lib, "cuda:0", **kw)
    e8 = f2c.check_dit_step(hip_lib, "cuda:0", fp8=True, **kw)
    record("flux2.full_width_blocks.T8704", velocity_rel_err_bf16=e, velocity_rel_err_fp8=e8)


def test_klein_step_without_quantiser_launch(hip_lib):
    """d = 3072, 1 + 2 blocks, T = 2064: gated MLP-in GEMMs + attention with MX fp8 output give the velocity bits of the step with separate
    quantiser launches, and no mtx_quantize_mx launch is left in the plan."""
    f2c.check_no_quantiser_step(hip_lib, "cuda:0", h2=32, w2=32, t_txt=16, d=3072, heads=24, axes_dim=(32, 32, 32, 32), layers=1, single_layers=2,
                                joint_dim=7680)
    f2c.check_glu_epilogue_step(hip_lib, "cuda:0", d=256, heads=2, axes_dim=(32, 32, 32, 32), layers=1, single_layers=2)
