"""CPU tier: the FLUX.2-Klein graphs (core/ml/flux2.py) executed by the kernel simulator on tiny geometry against the fp32 oracle."""
import flux2_checks as fc


def test_flux2_dit_step(emu_lib):
    fc.check_dit_step(emu_lib, "cpu")


def test_flux2_dit_step_reference_of_other_size(emu_lib):
    """reference-image tokens on their own grid (the pipeline caps the conditioning image at 1 MP)"""
    fc.check_dit_step(emu_lib, "cpu", h2=4, w2=5, rh2=3, rw2=4, t_txt=8)


def test_flux2_gated_epilogue_step_is_identical(emu_lib):
    """SwiGLU + MX quantisation inside the MLP-in GEMMs: same velocity bits, no SwiGLU launch left in the step"""
    fc.check_glu_epilogue_step(emu_lib, "cpu", d=256, heads=2, axes_dim=(32, 32, 32, 32), layers=1, single_layers=2)


def test_flux2_step_without_a_quantiser_launch(emu_lib):
    """both epilogue fusions at a sequence length that takes the long-sequence attention kernel (T = 1072): same velocity bits, zero
    mtx_quantize_mx launches in the step"""
    fc.check_no_quantiser_step(emu_lib, "cpu", d=256, heads=2, axes_dim=(32, 32, 32, 32), layers=1, single_layers=1)


def test_flux2_dit_step_fp8(emu_lib):
    """every block linear on the MX fp8 kernel: bounded distance to the fp32 oracle"""
    fc.check_dit_step(emu_lib, "cpu", fp8=True)


def test_flux2_vae(emu_lib):
    fc.check_vae(emu_lib, "cpu", h=32, w=48)


def test_flux2_vae_token_count_not_multiple_of_8(emu_lib):
    """a 48 x 80 crop has 6 x 10 = 60 latent positions in the mid-block attention: the key axis is padded to 64 and masked in the softmax
    (Klein crops are any multiple of 16, e.g. 1296 x 784 -> 15876 positions; the first config-5 bench run raised here)"""
    fc.check_vae(emu_lib, "cpu", h=48, w=80)


def test_flux2_klein_pipeline(emu_lib):
    fc.check_klein(emu_lib, "cpu", h=32, w=48, t_txt=8, steps=2)


def test_flux2_step_with_fp8_attention_scores(emu_lib):
    """Flux2DiTHip(attn_qk_f8=True) at T >= 1024: the rotary launches carry the e4m3 twin, every attention launch the fp8 operands, and the step
    stays close to the step with 16-bit scores (the op-level test holds the exact comparison)"""
    import torch
    from mangatranslator_amd.hip import abi
    kw = dict(d=128, heads=1, layers=1, single_layers=1, joint_dim=64, axes_dim=(32, 32, 32, 32))
    t, v = fc.models(**kw)
    lat, pe = fc.step_inputs(t, 22, 24, 22, 24, 16)
    vels = []
    for scores in (False, True):
        dit, _ = fc.hip_models(t, v, emu_lib, "cpu", fp8=True, attn_qk_f8=scores)
        vel, plan = fc.run_step(dit, lat, pe, 22, 24, 22, 24, 0.7, "cpu")
        assert plan.T >= 1024
        attn = [o for o in plan.ops if o.kind == abi.OP_ATTN]
        rope = [o for o in plan.ops if o.kind == abi.OP_EW and o.u.ew.kind == abi.EW_QK_NORM_ROPE]
        assert attn and rope and all(bool(o.u.attn.k_f8) == scores for o in attn) and all(bool(o.u.ew.y8) == scores for o in rope)
        vels.append(vel)
    e = fc.rel(vels[1], vels[0])
    print(f"fp8 attention scores vs 16-bit scores, one step: velocity rel diff {e:.4f}")
    assert torch.isfinite(vels[1]).all() and 0 < e < 0.1
