"""GPU probe: full-size FLUX.1-Kontext (random-init bf16) — per-step time and per-op breakdown."""
import sys, time, json
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mangatranslator_amd.core.ml import flux as fx
from mangatranslator_amd.hip.lib import get_library

lib = get_library(); lib.init(0)
dev = torch.device("cuda:0")
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1248, 832)
t0 = time.time()
dit = fx.FluxDiTHip(fx.synthetic_provider(fx.dit_param_shapes(fx.KONTEXT_DIT_CFG), dev, 1), fx.KONTEXT_DIT_CFG, dev)
vae = fx.FluxVAEHip(fx.synthetic_provider(fx.vae_param_shapes(fx.KONTEXT_VAE_CFG), dev, 2), fx.KONTEXT_VAE_CFG, dev)
torch.cuda.synchronize(); print("weights", time.time() - t0, "s", torch.cuda.memory_allocated() / 2**30, "GiB")
h2, w2 = H // 16, W // 16
plan = dit.plan_for(512, h2, w2, 1)
plan.ctx_in.normal_(); plan.lat.normal_()
pooled = torch.randn(768, device=dev).bfloat16()
plan.mod.copy_(dit.modulation(0.7, 2.5, pooled))
plan.run(); torch.cuda.synchronize()
print("finite", bool(torch.isfinite(plan.vel).all()), float(plan.vel.abs().mean()))
ms = plan.time(3); msg = plan.time(3, graph=True)
fl = dit.flops_per_step(512, h2, w2)
print(f"step T={fl['tokens']}: {ms:.1f} ms eager, {msg:.1f} ms graph; {(fl['gemm'] + fl['attention']) / ms / 1e9:.0f} TFLOP/s overall")
agg = {}
for i, lab in enumerate(plan.labels):
    k = lab.split(".", 1)[1] if "." in lab and lab[:3] in ("dbl", "sgl") else lab
    k = lab[:3] + "." + k if lab[:3] in ("dbl", "sgl") else k
    agg.setdefault(k, []).append(i)
rows = []
for k, idx in agg.items():
    t = plan.time_range(idx[0], idx[0], 3)
    rows.append((t * len(idx), t, len(idx), k))
for tot, t, n, k in sorted(rows, reverse=True)[:24]:
    print(f"  {k:24s} {t:8.3f} ms x{n:3d} = {tot:8.1f} ms")
tT = fl["tokens"]; D = 3072
a_ms = [r for r in rows if r[3] == "sgl.attn"][0][1]
print(f"attention {fl['attention_per_layer'] / a_ms / 1e9:.0f} TFLOP/s; qkv GEMM {2 * tT * D * 3 * D / [r for r in rows if r[3] == 'sgl.qkv'][0][1] / 1e9:.0f} TFLOP/s")
t0 = time.time(); m = dit.modulation(0.5, 2.5, pooled); torch.cuda.synchronize(); print("modulation (uncached)", (time.time() - t0) * 1e3, "ms")
enc = vae.encoder_plan(H, W); dec = vae.decoder_plan(H // 8, W // 8)
enc.run(); dec.run(); torch.cuda.synchronize()
print("vae encoder", enc.time(3), "ms; decoder", dec.time(3), "ms")
for name, p in (("enc", enc), ("dec", dec)):
    rows = sorted(((p.time_range(i, i, 2), p.labels[i]) for i in range(len(p.labels))), reverse=True)[:8]
    print(name, [(round(t, 2), l) for t, l in rows])
