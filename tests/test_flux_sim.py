"""CPU tier: FLUX.1-Kontext graphs (DiT step, VAE, full Euler loop) on the kernel simulator vs the fp32 oracle."""
import flux_checks as fc


def test_dit_step(emu_lib):
    fc.check_dit_step(emu_lib, "cpu")


def test_vae(emu_lib):
    fc.check_vae(emu_lib, "cpu", h=32, w=48)


def test_kontext_loop(emu_lib):
    fc.check_kontext(emu_lib, "cpu", h=32, w=48, t_txt=8, steps=2)


def test_first_block_cache(emu_lib):
    """`residual_diff_threshold` (the reference's nunchaku first-block cache): off == the one-plan step, never-passing == the same bytes through
    head + body, the skip path reproduces a computed step on equal inputs, always-passing skips every step after the first"""
    fc.check_first_block_cache(emu_lib, "cpu", h=32, w=48, t_txt=8, steps=3)
