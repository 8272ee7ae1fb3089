"""CPU tier: FLUX.1-Kontext graphs (DiT step, VAE, full Euler loop) on the kernel simulator vs the fp32 oracle."""
import flux_checks as fc


def test_dit_step(emu_lib):
    fc.check_dit_step(emu_lib, "cpu")


def test_vae(emu_lib):
    fc.check_vae(emu_lib, "cpu", h=32, w=48)


def test_kontext_loop(emu_lib):
    fc.check_kontext(emu_lib, "cpu", h=32, w=48, t_txt=8, steps=2)


def test_dit_step_with_the_text_stream_inside_the_image_launches(emu_lib):
    """t_txt a multiple of the 256-row tile (the real prompt length is 512): the double-stream blocks' text linears ride in the image
    stream's launches as a row-split second operand set (mtx_gemm_args.alt_*); the side-lane form of the same step stays available"""
    e_merged = fc.check_dit_step(emu_lib, "cpu", h2=4, w2=6, t_txt=256)
    e_lanes = fc.check_dit_step(emu_lib, "cpu", h2=4, w2=6, t_txt=256, dit_kw=dict(merge_text_stream=False))
    assert abs(e_merged - e_lanes) < 5e-3
