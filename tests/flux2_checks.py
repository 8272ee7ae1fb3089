"""FLUX.2-Klein parity: libmtx_hip graphs vs the fp32 CPU oracle (oracle/flux2_ref.py).

As in flux_checks.py the oracle's weights are rounded to bf16 first, so the comparison is the arithmetic of the path.  The fp8
mode is compared twice: against the fp32 oracle (what the quantisation costs) and, as the judge's bar, image PSNR against the
bf16 graph of the same weights.
"""
import math

import numpy as np
import torch

from mangatranslator_amd.core.ml import flux2 as f2
from oracle import flux2_ref as fr
from flux_checks import assert_repeats as fc_assert_repeats

PSNR_MIN_DB = 40.0


def rel(a, b):
    a, b = a.float().cpu().reshape(-1), b.float().cpu().reshape(-1)
    return ((a - b).norm() / (b.norm() + 1e-12)).item()


def psnr(a, b):
    mse = ((a.float().cpu() - b.float().cpu()) ** 2).mean().item()
    return 99.0 if mse == 0 else 10 * math.log10(1.0 / mse)


def models(seed=0, **kw):
    t, v = fr.make_models(seed=seed, **kw)
    with torch.no_grad():
        for m in (t, v):
            for p in m.parameters():
                p.copy_(p.to(torch.bfloat16).float())
    return t, v


def hip_models(t, v, lib, device, fp8=False, **dit_kw):
    tsd, vsd = t.state_dict(), v.state_dict()
    c = t.cfg
    dcfg = dict(d=c["d"], heads=c["heads"], layers=c["layers"], single_layers=c["single_layers"], in_channels=c["in_channels"],
                joint_dim=c["joint_dim"], mlp_ratio=c["mlp_ratio"], axes_dim=tuple(c["axes_dim"]), rope_theta=c["rope_theta"],
                guidance_embeds=c["guidance_embeds"])
    vcfg = dict(ch=tuple(v.cfg["ch"]), groups=v.cfg["groups"], latent=v.cfg["latent"], quant_conv=True, bn_eps=v.cfg["bn_eps"])
    dit = f2.Flux2DiTHip(lambda n: tsd[n], dcfg, device, lib=lib, fp8=fp8, **dit_kw)
    vae = f2.Flux2VAEHip(lambda n: vsd[n], vcfg, device, lib=lib)
    return dit, vae


def step_inputs(t, h2, w2, rh2, rw2, t_txt, seed=3):
    g = torch.Generator().manual_seed(seed)
    C = t.cfg["in_channels"]
    lat = torch.randn(h2 * w2 + rh2 * rw2, C, generator=g).to(torch.bfloat16).float()
    pe = torch.randn(t_txt, t.cfg["joint_dim"], generator=g).to(torch.bfloat16).float()
    return lat, pe


def run_step(dit, lat, pe, h2, w2, rh2, rw2, timestep, device):
    plan = dit.plan_for(pe.shape[0], h2, w2, rh2, rw2)
    plan.ctx_in.copy_(pe.to(device, torch.bfloat16))
    plan.lat.copy_(lat.to(device, torch.bfloat16))
    plan.mod.copy_(dit.modulation(timestep))
    plan.run()
    if torch.device(device).type == "cuda":
        torch.cuda.synchronize()
    return plan.vel.float().cpu().clone(), plan


def check_dit_step(lib, device, h2=4, w2=6, rh2=None, rw2=None, t_txt=16, tol=3e-2, fp8=False, fp8_tol=0.12, **kw):
    rh2, rw2 = (h2 if rh2 is None else rh2), (w2 if rw2 is None else rw2)
    t, v = models(**kw)
    lat, pe = step_inputs(t, h2, w2, rh2, rw2, t_txt)
    ids = torch.cat([fr.latent_ids(h2, w2, 0), fr.latent_ids(rh2, rw2, 10)])
    with torch.no_grad():
        ref = t(lat, 0.7, pe, fr.text_ids(t_txt), ids)[: h2 * w2]
    dit, _ = hip_models(t, v, lib, device, fp8=fp8)
    vel, plan = run_step(dit, lat, pe, h2, w2, rh2, rw2, 0.7, device)
    e = rel(vel, ref)
    print(f"FLUX.2 DiT step ({t.cfg['layers']}+{t.cfg['single_layers']} blocks, d={t.cfg['d']}, T={plan.T}, fp8={bool(fp8)}): velocity rel err {e:.4f}")
    assert e < (fp8_tol if fp8 else tol)
    fc_assert_repeats(plan, device)
    return e


def check_glu_epilogue_step(lib, device, h2=4, w2=6, t_txt=16, **kw):
    """one denoising step with the gated epilogue in the MLP-in GEMMs (Flux2DiTHip(glu_epilogue=True)) against the same step with the
    separate SwiGLU-quantiser launches: the velocities must be IDENTICAL (the fused epilogue reproduces the bytes and scales, so
    everything downstream sees the same operands), and the fused plan must hold no SwiGLU launch"""
    from mangatranslator_amd.hip import abi
    t, v = models(**kw)
    lat, pe = step_inputs(t, h2, w2, h2, w2, t_txt)
    vels, nq = [], []
    for glu in (False, True):
        dit, _ = hip_models(t, v, lib, device, fp8=True, glu_epilogue=glu)
        assert dit.glu_epilogue == glu, "geometry does not allow the gated epilogue (3 d must be a multiple of 256)"
        vel, plan = run_step(dit, lat, pe, h2, w2, h2, w2, 0.7, device)
        vels.append(vel)
        nq.append(sum(1 for o in plan.ops if o.kind == abi.OP_QUANT and o.u.quant.op == abi.QUANT_SWIGLU))
    assert nq[0] == 2 * t.cfg["layers"] + t.cfg["single_layers"] and nq[1] == 0, nq
    assert torch.equal(vels[0], vels[1]), f"gated epilogue changes the step: rel {rel(vels[1], vels[0]):.3e}"
    return nq


def check_no_quantiser_step(lib, device, h2=22, w2=24, t_txt=16, **kw):
    """a denoising step long enough for the long-sequence attention kernel (T >= 1024) with BOTH epilogue fusions — gated MLP-in GEMMs and
    attention with MX fp8 output — against the step with the separate quantiser launches: identical velocity, and not one quantiser
    launch (mtx_quantize_mx) left in the plan"""
    from mangatranslator_amd.hip import abi
    t, v = models(**kw)
    lat, pe = step_inputs(t, h2, w2, h2, w2, t_txt)
    vels, nq = [], []
    for on in (False, True):
        dit, _ = hip_models(t, v, lib, device, fp8=True, glu_epilogue=on, attn_q8=on, attn_pv_f8=False)      # (fp8 P V exists only with the fp8 output form: off on both sides)
        assert dit.glu_epilogue == on and dit.attn_q8 == on
        vel, plan = run_step(dit, lat, pe, h2, w2, h2, w2, 0.7, device)
        assert plan.T >= 1024
        vels.append(vel)
        nq.append(sum(1 for o in plan.ops if o.kind == abi.OP_QUANT))
    assert nq[0] > 0 and nq[1] == 0, nq
    assert torch.equal(vels[0], vels[1]), f"epilogue fusions change the step: rel {rel(vels[1], vels[0]):.3e}"
    return nq


def check_vae(lib, device, h=64, w=96, tol=3e-2, **kw):
    t, v = models(**kw)
    _, vae = hip_models(t, v, lib, device)
    L = v.cfg["latent"]
    g = torch.Generator().manual_seed(1)
    img = (torch.rand(h, w, 3, generator=g) * 255).to(torch.uint8).numpy()
    x = torch.from_numpy(img).permute(2, 0, 1)[None].float() / 127.5 - 1.0
    with torch.no_grad():
        mean_ref = v.encode_mode(x)
        z = torch.randn(1, L, h // 8, w // 8, generator=torch.Generator().manual_seed(5))
        dec_ref = v.decode(z)
    enc = vae.encoder_plan(h, w)
    enc.src.copy_(torch.from_numpy(img).to(device).view(1, h, w, 3))
    enc.run()
    mean = enc.moments.t[0, :, :, :L].float().permute(2, 0, 1)[None]
    e1 = rel(mean, mean_ref)
    dec = vae.decoder_plan(h // 8, w // 8)
    dec.z.t.copy_(z.permute(0, 2, 3, 1).to(device, torch.bfloat16))
    dec.run()
    raw = dec.raw.t[0, :, :, :3].float().permute(2, 0, 1)[None]
    e2 = rel(raw, dec_ref)
    print(f"FLUX.2 VAE: encoder mean rel err {e1:.4f}, decoder rel err {e2:.4f}")
    assert e1 < tol and e2 < tol
    return e1, e2


def check_klein(lib, device, h=64, w=96, t_txt=16, steps=3, fp8=False, **kw):
    """whole pipeline call vs the oracle pipeline on the same noise: latents, sigmas, image PSNR"""
    from PIL import Image
    t, v = models(**kw)
    dit, vae = hip_models(t, v, lib, device, fp8=fp8)
    g = torch.Generator().manual_seed(1)
    img = (torch.rand(h, w, 3, generator=g) * 255).to(torch.uint8).numpy()
    pe = torch.randn(t_txt, t.cfg["joint_dim"], generator=g).to(torch.bfloat16).float()
    noise = torch.randn(1, t.cfg["in_channels"], h // 16, w // 16, generator=g)
    ref_img, info = fr.klein(t, v, img, pe, steps, noise)
    pipe = f2.Flux2KleinHip(dit, vae)
    out = pipe(image=Image.fromarray(img), width=w, height=h, num_inference_steps=steps, guidance_scale=1.0, prompt_embeds=pe[None],
               latents=noise, output_type="pt").images[0]
    assert out.shape == (3, h, w) and out.dtype == torch.float32
    assert np.allclose(pipe.last["sigmas"], info["sigmas"])
    e_ref = rel(pipe.last["ref_tokens"], info["ref_tokens"])
    e = rel(pipe.last["latents"], info["latents"])
    p = psnr(out, ref_img)
    print(f"Klein {steps} steps {w}x{h} fp8={bool(fp8)}: ref-token rel err {e_ref:.4f}, latent rel err {e:.4f}, image PSNR vs fp32 oracle {p:.1f} dB")
    assert e_ref < 3e-2
    if not fp8:
        assert e < 3e-2 and p >= PSNR_MIN_DB
    pil = pipe(image=Image.fromarray(img), width=w, height=h, num_inference_steps=1, prompt_embeds=pe[None], latents=noise).images[0]
    assert pil.size == (w, h) and pil.mode == "RGB"
    return e, p, out


def check_klein_fp8_scores(lib, device, h=384, w=512, t_txt=32, steps=4, **kw):
    """fp8 attention scores (Flux2DiTHip(attn_qk_f8=True): q and k as e4m3 rows, scores on the fp8 matrix instruction) at a size that takes the
    long-sequence kernel: image PSNR against the bf16 pipeline of the same weights, beside the PSNR of the fp8-linears-only pipeline"""
    from PIL import Image
    t, v = models(**kw)
    g = torch.Generator().manual_seed(1)
    img = (torch.rand(h, w, 3, generator=g) * 255).to(torch.uint8).numpy()
    pe = torch.randn(t_txt, t.cfg["joint_dim"], generator=g).to(torch.bfloat16).float()
    noise = torch.randn(1, t.cfg["in_channels"], h // 16, w // 16, generator=g)
    outs = []
    for fp8, scores, values in ((False, False, False), (True, False, False), (True, True, False), (True, True, True)):
        dit, vae = hip_models(t, v, lib, device, fp8=fp8, attn_qk_f8=scores, attn_pv_f8=values)
        assert dit.attn_qk_f8 == scores and dit.attn_pv_f8 == values
        pipe = f2.Flux2KleinHip(dit, vae)
        outs.append(pipe(image=Image.fromarray(img), width=w, height=h, num_inference_steps=steps, prompt_embeds=pe[None], latents=noise,
                         output_type="pt").images[0].cpu())
        if scores:
            from mangatranslator_amd.hip import abi
            plan = next(iter(dit._plans._d.values()))
            assert plan.T >= 1024 and any(o.kind == abi.OP_ATTN and o.u.attn.k_f8 for o in plan.ops), "the step does not run the fp8-score kernel"
            assert any(o.kind == abi.OP_ATTN and o.u.attn.v_f8t for o in plan.ops) == values
    p_lin, p_sc, p_between = psnr(outs[1], outs[0]), psnr(outs[2], outs[0]), psnr(outs[2], outs[1])
    p_pv, p_pv_between = psnr(outs[3], outs[0]), psnr(outs[3], outs[2])
    print(f"Klein {steps} steps {w}x{h}: vs bf16: fp8 linears {p_lin:.1f} dB, + fp8 scores {p_sc:.1f} dB; fp8 scores vs fp8 linears {p_between:.1f} dB; "
          f"+ fp8 P V {p_pv:.1f} dB vs bf16 ({p_pv_between:.1f} dB vs fp8 scores alone)")
    return p_lin, p_sc, p_between, p_pv


def check_klein_fp8_vs_bf16(lib, device, h=64, w=96, t_txt=16, steps=4, fp8=True, **kw):
    """the judge's bar for the fp8 path: image PSNR of the fp8 pipeline against the bf16 pipeline of the same weights"""
    from PIL import Image
    t, v = models(**kw)
    g = torch.Generator().manual_seed(1)
    img = (torch.rand(h, w, 3, generator=g) * 255).to(torch.uint8).numpy()
    pe = torch.randn(t_txt, t.cfg["joint_dim"], generator=g).to(torch.bfloat16).float()
    noise = torch.randn(1, t.cfg["in_channels"], h // 16, w // 16, generator=g)
    outs = []
    for mode in (False, fp8):
        dit, vae = hip_models(t, v, lib, device, fp8=mode)
        pipe = f2.Flux2KleinHip(dit, vae)
        outs.append(pipe(image=Image.fromarray(img), width=w, height=h, num_inference_steps=steps, prompt_embeds=pe[None], latents=noise,
                         output_type="pt").images[0].cpu())
    p = psnr(outs[1], outs[0])
    print(f"Klein {steps} steps {w}x{h}: fp8 {fp8} vs bf16 image PSNR {p:.1f} dB")
    return p
