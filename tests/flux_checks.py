"""FLUX.1-Kontext parity: libmtx_hip graphs vs the fp32 CPU oracle (oracle/flux_ref.py).

The oracle's weights are rounded to bf16 first (the precision the checkpoint is served at), so what is
compared is the arithmetic of the path, not the weight rounding.  Tolerances are relative L2 errors of bf16
activation storage accumulated over the depth of the tiny test networks; the final image check is the
BASELINE.json bar, PSNR >= 40 dB on the inpainted pixels.
"""
import math

import numpy as np
import torch

from mangatranslator_amd.core.ml import flux as fx
from oracle import flux_ref as fr

PSNR_MIN_DB = 40.0


def rel(a, b):
    a, b = a.float().cpu().reshape(-1), b.float().cpu().reshape(-1)
    return ((a - b).norm() / (b.norm() + 1e-12)).item()


def models(seed=0, **kw):
    t, v = fr.make_models(seed=seed, **kw)
    with torch.no_grad():
        for m in (t, v):
            for p in m.parameters():
                p.copy_(p.to(torch.bfloat16).float())
    return t, v


def hip_models(t, v, lib, device):
    tsd, vsd = t.state_dict(), v.state_dict()
    c = t.cfg
    dcfg = dict(d=c["d"], heads=c["heads"], layers=c["layers"], single_layers=c["single_layers"], in_channels=c["in_channels"],
                joint_dim=c["joint_dim"], pooled_dim=c["pooled_dim"], axes_dim=tuple(c["axes_dim"]))
    vcfg = dict(ch=tuple(v.cfg["ch"]), groups=v.cfg["groups"], scaling_factor=v.cfg["scaling_factor"], shift_factor=v.cfg["shift_factor"])
    dit = fx.FluxDiTHip(lambda n: tsd[n], dcfg, device, lib=lib)
    vae = fx.FluxVAEHip(lambda n: vsd[n], vcfg, device, lib=lib)
    return dit, vae


def inputs(t, h, w, t_txt, seed=1):
    g = torch.Generator().manual_seed(seed)
    img = (torch.rand(h, w, 3, generator=g) * 255).to(torch.uint8).numpy()
    pe = torch.randn(t_txt, t.cfg["joint_dim"], generator=g).to(torch.bfloat16).float()
    pooled = torch.randn(t.cfg["pooled_dim"], generator=g).to(torch.bfloat16).float()
    noise = torch.randn(1, 16, h // 8, w // 8, generator=g)
    return img, pe, pooled, noise


def check_dit_step(lib, device, h2=4, w2=6, t_txt=16, tol=3e-2, **kw):
    t, v = models(**kw)
    dit, _ = hip_models(t, v, lib, device)
    g = torch.Generator().manual_seed(3)
    tn = h2 * w2
    lat = torch.randn(2 * tn, 64, generator=g).to(torch.bfloat16).float()
    pe = torch.randn(t_txt, t.cfg["joint_dim"], generator=g).to(torch.bfloat16).float()
    pooled = torch.randn(t.cfg["pooled_dim"], generator=g).to(torch.bfloat16).float()
    ids = torch.cat([fr.image_ids(h2, w2, 0), fr.image_ids(h2, w2, 1)])
    with torch.no_grad():
        ref = t(lat, 0.7, 2.5, pooled, pe, torch.zeros(t_txt, 3), ids)[:tn]
    plan = dit.plan_for(t_txt, h2, w2, 1)
    plan.ctx_in.copy_(pe.to(device, torch.bfloat16))
    plan.lat.copy_(lat.to(device, torch.bfloat16))
    plan.mod.copy_(dit.modulation(0.7, 2.5, pooled.to(device, torch.bfloat16)))
    plan.run()
    e = rel(plan.vel, ref)
    print(f"DiT step ({t.cfg['layers']}+{t.cfg['single_layers']} blocks, T={plan.T}): velocity rel err {e:.4f}")
    assert e < tol
    assert_repeats(plan, device)
    return e


def assert_repeats(plan, device, runs=3):
    """the step again — eagerly and as hipGraph replays, the pipelines' form — over the same inputs: the velocity's bytes do not change
    (side lane joins, K-slice tickets and partials, key-split merges: nothing may depend on what the previous run left behind)"""
    sync = torch.cuda.synchronize if torch.device(device).type == "cuda" else (lambda: None)
    sync()
    first = plan.vel.clone()
    for i in range(runs):
        plan.vel.fill_(float("nan"))
        plan.run(graph=i > 0)
        sync()
        assert torch.equal(plan.vel, first), f"run {i + 2} of the same step differs from the first"


def check_vae(lib, device, h=64, w=96, tol=3e-2, **kw):
    t, v = models(**kw)
    _, vae = hip_models(t, v, lib, device)
    img, _, _, noise = inputs(t, h, w, 8)
    x = torch.from_numpy(img).permute(2, 0, 1)[None].float() / 127.5 - 1.0
    with torch.no_grad():
        mean_ref = v.encode_mode(x)
        z = torch.randn(1, 16, h // 8, w // 8, generator=torch.Generator().manual_seed(5))
        dec_ref = v.decode(z)
    enc = vae.encoder_plan(h, w)
    enc.src.copy_(torch.from_numpy(img).to(device).view(1, h, w, 3))
    enc.run()
    mean = enc.moments.t[0, :, :, :16].float().permute(2, 0, 1)[None]
    e1 = rel(mean, mean_ref)
    dec = vae.decoder_plan(h // 8, w // 8)
    dec.z.t.copy_(z.permute(0, 2, 3, 1).to(device, torch.bfloat16))
    dec.run()
    raw = dec.raw.t[0, :, :, :3].float().permute(2, 0, 1)[None]
    e2 = rel(raw, dec_ref)
    print(f"VAE: encoder mean rel err {e1:.4f}, decoder rel err {e2:.4f}")
    assert e1 < tol and e2 < tol
    return e1, e2


def check_kontext(lib, device, h=64, w=96, t_txt=16, steps=3, **kw):
    t, v = models(**kw)
    dit, vae = hip_models(t, v, lib, device)
    img, pe, pooled, noise = inputs(t, h, w, t_txt)
    ref_img, info = fr.kontext(t, v, img, pe, pooled, steps, 2.5, noise)
    pipe = fx.FluxKontextHip(dit, vae)
    out = pipe(image=img, width=w, height=h, num_inference_steps=steps, guidance_scale=2.5, prompt_embeds=pe[None],
               pooled_prompt_embeds=pooled[None], latents=noise).images[0]
    assert out.shape == (3, h, w) and out.dtype == torch.float32
    assert np.allclose(pipe.last["sigmas"], info["sigmas"])
    e = rel(pipe.last["latents"], info["latents"])
    mse = ((out.cpu() - ref_img) ** 2).mean().item()
    p = 99.0 if mse == 0 else 10 * math.log10(1.0 / mse)
    print(f"Kontext {steps} steps {w}x{h}: latent rel err {e:.4f}, image PSNR {p:.1f} dB")
    assert e < 3e-2
    return e, p


def named_provider(shapes: dict, device, seed: int):
    """Seeded parameters that depend on (seed, name) only — any caller, in any order, gets the same tensor for a name.  Matrices are
    bf16 values (generated on `device`, where a 12 B-parameter network takes seconds), so the fp32 oracle and the bf16 graph see the
    same weights exactly."""
    import zlib

    def get(name):
        g = torch.Generator(device=device).manual_seed(seed * 1000003 + zlib.crc32(name.encode()))
        shp = shapes[name]
        if len(shp) >= 2:
            fan = int(np.prod(shp[1:]))
            return torch.randn(shp, device=device, generator=g, dtype=torch.float32).mul_(1.0 / math.sqrt(fan)).to(torch.bfloat16)
        if "norm" in name and name.endswith("weight"):
            return (1.0 + 0.1 * torch.randn(shp, device=device, generator=g)).to(torch.bfloat16).float()
        return (0.02 * torch.randn(shp, device=device, generator=g)).to(torch.bfloat16).float()
    return get


@torch.no_grad()
def oracle_step_streamed(cfg: dict, get, lat, timestep, guidance, pooled, pe, txt_ids, img_ids):
    """oracle/flux_ref.py's FluxTransformer.forward with ONE block resident at a time: every block module is built, filled from
    `get(name)` (fp32 copies of the tensors the graph was built from), run and dropped — the 11.9 B-parameter network never needs its
    48 GB of fp32 weights at once."""
    d, heads = cfg["d"], cfg["heads"]

    def fill(mod, prefix):
        for k, p in mod.named_parameters():
            p.copy_(get(prefix + k).float().cpu())
        return mod.eval()

    def lin(prefix, dout, din):
        return fill(torch.nn.Linear(din, dout), prefix + ".")

    x = lin("x_embedder", d, cfg["in_channels"])(lat)
    c = lin("context_embedder", d, cfg["joint_dim"])(pe)
    tte = fill(fr.TimeTextEmbed(d, cfg["pooled_dim"]), "time_text_embed.")
    temb = tte(torch.tensor([timestep * 1000.0]), torch.tensor([guidance * 1000.0]), pooled[None])[0]
    cos, sin = fr.rope_tables(torch.cat([txt_ids, img_ids]), cfg["axes_dim"])
    for i in range(cfg["layers"]):
        b = fill(fr.DoubleBlock(d, heads), f"transformer_blocks.{i}.")
        x, c = b(x, c, temb, cos, sin)
        del b
    j = torch.cat([c, x])
    for i in range(cfg["single_layers"]):
        b = fill(fr.SingleBlock(d, heads), f"single_transformer_blocks.{i}.")
        j = b(j, temb, cos, sin)
        del b
    x = j[c.shape[0]:]
    scale, shift = fill(fr.AdaNorm(d, 2), "norm_out.")(temb)
    return lin("proj_out", cfg["in_channels"], d)(torch.nn.functional.layer_norm(x, (d,), eps=1e-6) * (1 + scale) + shift)


def check_full_depth_step(lib, device, h2=24, w2=32, t_txt=512, tol=3e-2, seed=4):          # measured 1.27 %, cosine 0.99992 (profiles/r03_parity.json)
    """ONE denoising step of the REAL FLUX.1-Kontext geometry — 19 double + 38 single blocks, d = 3072, 24 heads, T = t_txt + 2 h2 w2
    tokens — against the fp32 oracle on the same (bf16-valued) weights.  Until round 3 the full depth was only ever run at d = 256 and
    the full width at 1 + 1 blocks."""
    cfg = dict(fx.KONTEXT_DIT_CFG)
    shapes = fx.dit_param_shapes(cfg)
    get = named_provider(shapes, device, seed)
    dit = fx.FluxDiTHip(get, cfg, device, lib=lib)
    g = torch.Generator().manual_seed(seed)
    tn = h2 * w2
    lat = torch.randn(2 * tn, 64, generator=g).to(torch.bfloat16).float()
    pe = torch.randn(t_txt, cfg["joint_dim"], generator=g).to(torch.bfloat16).float()
    pooled = torch.randn(cfg["pooled_dim"], generator=g).to(torch.bfloat16).float()
    ids = torch.cat([fr.image_ids(h2, w2, 0), fr.image_ids(h2, w2, 1)])
    plan = dit.plan_for(t_txt, h2, w2, 1)
    plan.ctx_in.copy_(pe.to(device, torch.bfloat16))
    plan.lat.copy_(lat.to(device, torch.bfloat16))
    plan.mod.copy_(dit.modulation(0.7, 2.5, pooled.to(device, torch.bfloat16)))
    plan.run()
    vel = plan.vel.float().cpu()
    ref = oracle_step_streamed(cfg, get, lat, 0.7, 2.5, pooled, pe, torch.zeros(t_txt, 3), ids)[:tn]
    e = rel(vel, ref)
    cosine = torch.nn.functional.cosine_similarity(vel.reshape(-1), ref.reshape(-1), dim=0).item()
    print(f"full-depth DiT step (19 + 38 blocks, d = 3072, T = {plan.T}): velocity rel err {e:.4f}, cosine {cosine:.6f}")
    assert e < tol
    return e, cosine


def check_first_block_cache(lib, device, h=64, w=96, t_txt=16, steps=4, **kw):
    """The first-block cache behind `residual_diff_threshold` (reference core/ml/model_manager.py:1159-1162; nunchaku's algorithm, unpinned):
      * threshold 0 is the one-plan graph: the image the pipeline made before the knob existed, byte for byte;
      * a threshold nothing passes runs head + body every step: the same ops in three plans — still the same bytes;
      * the skip path on a step whose inputs equal the computed step's reproduces that step's velocity up to the 16-bit rounding of the
        two residuals (x1 + (x_out - x1));
      * a threshold everything passes computes step 0 and skips every later one."""
    t, v = models(**kw)
    dit, vae = hip_models(t, v, lib, device)
    img, pe, pooled, noise = inputs(t, h, w, t_txt)
    pipe = fx.FluxKontextHip(dit, vae)
    call = lambda **k: pipe(image=img, width=w, height=h, num_inference_steps=steps, guidance_scale=2.5, prompt_embeds=pe[None],
                            pooled_prompt_embeds=pooled[None], latents=noise, **k).images[0].clone()
    base = call()
    assert pipe.last["skipped_steps"] == 0
    never = call(residual_diff_threshold=1e-12)
    assert pipe.last["skipped_steps"] == 0 and torch.equal(never, base), "head + body differs from the one-plan step"
    # the skip path against the body path on identical step inputs
    h2, w2 = h // 16, w // 16
    plan = dit.plan_for(t_txt, h2, w2, 1, cached=True)
    g = torch.Generator().manual_seed(7)
    plan.ctx_in.copy_(pe.to(device, torch.bfloat16))
    plan.lat.copy_(torch.randn(2 * h2 * w2, 64, generator=g).to(device, torch.bfloat16))
    plan.mod.copy_(dit.modulation(0.6, 2.5, pooled.to(device, torch.bfloat16)))
    plan.run(); plan.body.run()
    computed = plan.vel.clone()
    plan.run()
    from mangatranslator_amd.hip.plan import residual_distance
    d = residual_distance(plan.parts)
    assert d == 0.0, f"the same step again: first-block residual distance {d}"
    plan.skip.run()
    e = rel(plan.vel, computed)
    print(f"first-block cache: skip path vs computed step on equal inputs: velocity rel err {e:.2e}")
    assert e < 2e-2
    always = call(residual_diff_threshold=1e9)
    assert pipe.last["skipped_steps"] == steps - 1 and pipe.last["skipped"][0] is False
    assert torch.isfinite(always).all()
    some = call(residual_diff_threshold=0.5)
    print(f"first-block cache: threshold 0.5 skipped {pipe.last['skipped_steps']} of {steps} steps; image PSNR vs full compute "
          f"{-10 * math.log10(max(((some.float() - base.float()) ** 2).mean().item(), 1e-12)):.1f} dB")
    return e
