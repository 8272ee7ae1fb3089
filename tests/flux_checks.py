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
    return e


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
