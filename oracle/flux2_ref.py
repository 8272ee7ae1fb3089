"""CPU oracle for the FLUX.2-Klein inpainting pipeline.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the reference drives diffusers' `Flux2KleinPipeline` (reference
core/image/inpainting.py:1577-1589; loaded at core/ml/model_manager.py:1254-1337 with `diffusers>=0.37.0`
+ `sdnq`, neither installed here, no checkpoints present).  This file restates the published architecture in
plain torch fp32 with diffusers' module / state-dict names:

  Flux2Transformer2DModel   x_embedder, context_embedder (no biases anywhere), time_guidance_embed (sinusoidal
      timestep MLP; Klein is guidance-distilled: guidance_embeds = False), ONE modulation per stream type shared
      by all blocks (double_stream_modulation_img / _txt: 2 x (shift, scale, gate); single_stream_modulation:
      1 x (shift, scale, gate)), N double-stream blocks (LayerNorm + modulation, joint attention with per-head
      RMSNorm on q/k and 4-axis RoPE theta = 2000, SwiGLU feed-forward of ratio 3), M single-stream blocks
      (one fused projection to q|k|v|mlp_gate|mlp_up, attention and SwiGLU in parallel, one output projection
      over their concat), AdaLayerNormContinuous, proj_out
  AutoencoderKLFlux2        the AutoencoderKL encoder / decoder with 32 latent channels, 1x1 quant_conv /
      post_quant_conv, and a BatchNorm (running statistics, no affine) over the 2x2-patchified latents in place
      of scaling / shift factors
  Flux2KleinPipeline        noise drawn in the patchified shape [1, 128, H/16, W/16]; reference-image latents
      appended as extra tokens whose position id is t = 10 on the first axis; ids are (t, h, w, l) with the
      text tokens on the l axis; FlowMatchEuler sigmas = linspace(1, 1/N, N) exponentially shifted with
      mu = compute_empirical_mu(image_seq_len, N); Euler update x += (sigma_next - sigma) * v

Klein-4B: d = 3072 (24 heads x 128), 5 double + 20 single blocks, joint_attention_dim 7680 (three Qwen3-4B
hidden states); Klein-9B: d = 4096 (32 heads), 8 + 24 blocks, joint_attention_dim 12288.
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .flux_ref import Decoder, Encoder, RMSNorm, apply_rope, rope_tables, sdpa, timestep_embedding


def lin(din, dout):
    return nn.Linear(din, dout, bias=False)


def swiglu(x):
    a, b = x.chunk(2, -1)
    return F.silu(a) * b


class TimeEmbed(nn.Module):
    def __init__(self, d, guidance):
        super().__init__()
        self.timestep_embedder = nn.Module()
        self.timestep_embedder.linear_1, self.timestep_embedder.linear_2 = lin(256, d), lin(d, d)
        if guidance:
            self.guidance_embedder = nn.Module()
            self.guidance_embedder.linear_1, self.guidance_embedder.linear_2 = lin(256, d), lin(d, d)

    def forward(self, t, g=None):
        e = self.timestep_embedder
        out = e.linear_2(F.silu(e.linear_1(timestep_embedding(t))))
        if g is not None and hasattr(self, "guidance_embedder"):
            e = self.guidance_embedder
            out = out + e.linear_2(F.silu(e.linear_1(timestep_embedding(g))))
        return out


class Modulation(nn.Module):
    def __init__(self, d, sets):
        super().__init__()
        self.linear, self.sets = lin(d, 3 * sets * d), sets

    def forward(self, temb):
        c = self.linear(F.silu(temb)).chunk(3 * self.sets, -1)
        return [c[3 * i: 3 * i + 3] for i in range(self.sets)]          # (shift, scale, gate) per set


class FeedForward(nn.Module):
    def __init__(self, d, ratio):
        super().__init__()
        hid = int(d * ratio)
        self.linear_in, self.linear_out = lin(d, 2 * hid), lin(hid, d)

    def forward(self, x):
        return self.linear_out(swiglu(self.linear_in(x)))


class JointAttn(nn.Module):
    def __init__(self, d, heads):
        super().__init__()
        hd = d // heads
        self.heads, self.hd = heads, hd
        self.to_q, self.to_k, self.to_v = lin(d, d), lin(d, d), lin(d, d)
        self.norm_q, self.norm_k = RMSNorm(hd), RMSNorm(hd)
        self.add_q_proj, self.add_k_proj, self.add_v_proj = lin(d, d), lin(d, d), lin(d, d)
        self.norm_added_q, self.norm_added_k = RMSNorm(hd), RMSNorm(hd)
        self.to_out = nn.ModuleList([lin(d, d)])
        self.to_add_out = lin(d, d)

    def qkv(self, x, ctx):
        S = x.shape[0]
        sh = lambda t: t.view(S, self.heads, self.hd)
        if ctx:
            return self.norm_added_q(sh(self.add_q_proj(x))), self.norm_added_k(sh(self.add_k_proj(x))), sh(self.add_v_proj(x))
        return self.norm_q(sh(self.to_q(x))), self.norm_k(sh(self.to_k(x))), sh(self.to_v(x))


def ln(t):
    return F.layer_norm(t, (t.shape[-1],), eps=1e-6)


class DoubleBlock(nn.Module):
    def __init__(self, d, heads, ratio):
        super().__init__()
        self.attn = JointAttn(d, heads)
        self.ff, self.ff_context = FeedForward(d, ratio), FeedForward(d, ratio)

    def forward(self, x, c, mod_img, mod_txt, cos, sin):
        (sh, sc, g), (sh2, sc2, g2) = mod_img
        (csh, csc, cg), (csh2, csc2, cg2) = mod_txt
        q, k, v = self.attn.qkv(ln(x) * (1 + sc) + sh, False)
        cq, ck, cv = self.attn.qkv(ln(c) * (1 + csc) + csh, True)
        Q, K, V = torch.cat([cq, q]), torch.cat([ck, k]), torch.cat([cv, v])
        o = sdpa(apply_rope(Q, cos, sin), apply_rope(K, cos, sin), V)
        oc, ox = o[: c.shape[0]], o[c.shape[0]:]
        x = x + g * self.attn.to_out[0](ox)
        x = x + g2 * self.ff(ln(x) * (1 + sc2) + sh2)
        c = c + cg * self.attn.to_add_out(oc)
        c = c + cg2 * self.ff_context(ln(c) * (1 + csc2) + csh2)
        return x, c


class ParallelAttn(nn.Module):
    def __init__(self, d, heads, ratio):
        super().__init__()
        self.heads, self.hd, self.d, self.hid = heads, d // heads, d, int(d * ratio)
        self.to_qkv_mlp_proj = lin(d, 3 * d + 2 * self.hid)
        self.norm_q, self.norm_k = RMSNorm(self.hd), RMSNorm(self.hd)
        self.to_out = lin(d + self.hid, d)

    def forward(self, x, cos, sin):
        S = x.shape[0]
        h = self.to_qkv_mlp_proj(x)
        q, k, v, mlp = h[:, : self.d], h[:, self.d: 2 * self.d], h[:, 2 * self.d: 3 * self.d], h[:, 3 * self.d:]
        sh = lambda t: t.reshape(S, self.heads, self.hd)
        o = sdpa(apply_rope(self.norm_q(sh(q)), cos, sin), apply_rope(self.norm_k(sh(k)), cos, sin), sh(v))
        return self.to_out(torch.cat([o, swiglu(mlp)], -1))


class SingleBlock(nn.Module):
    def __init__(self, d, heads, ratio):
        super().__init__()
        self.attn = ParallelAttn(d, heads, ratio)

    def forward(self, x, mod, cos, sin):
        sh, sc, g = mod
        return x + g * self.attn(ln(x) * (1 + sc) + sh, cos, sin)


class NormOut(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.linear = lin(d, 2 * d)

    def forward(self, x, temb):
        scale, shift = self.linear(F.silu(temb)).chunk(2, -1)
        return ln(x) * (1 + scale) + shift


class Flux2Transformer(nn.Module):
    def __init__(self, d=3072, heads=24, layers=5, single_layers=20, in_channels=128, joint_dim=7680, mlp_ratio=3.0,
                 axes_dim=(32, 32, 32, 32), rope_theta=2000.0, guidance_embeds=False):
        super().__init__()
        self.cfg = dict(d=d, heads=heads, layers=layers, single_layers=single_layers, in_channels=in_channels, joint_dim=joint_dim,
                        mlp_ratio=mlp_ratio, axes_dim=tuple(axes_dim), rope_theta=rope_theta, guidance_embeds=guidance_embeds)
        self.x_embedder, self.context_embedder = lin(in_channels, d), lin(joint_dim, d)
        self.time_guidance_embed = TimeEmbed(d, guidance_embeds)
        self.double_stream_modulation_img, self.double_stream_modulation_txt = Modulation(d, 2), Modulation(d, 2)
        self.single_stream_modulation = Modulation(d, 1)
        self.transformer_blocks = nn.ModuleList(DoubleBlock(d, heads, mlp_ratio) for _ in range(layers))
        self.single_transformer_blocks = nn.ModuleList(SingleBlock(d, heads, mlp_ratio) for _ in range(single_layers))
        self.norm_out = NormOut(d)
        self.proj_out = lin(d, in_channels)

    @torch.no_grad()
    def forward(self, hidden, timestep, enc, txt_ids, img_ids, guidance=None):
        """hidden [S_img, C]; enc [S_txt, joint]; timestep scalar in [0, 1]; ids [S, 4] = (t, h, w, l)"""
        temb = self.time_guidance_embed(torch.tensor([timestep * 1000.0]),
                                        None if guidance is None else torch.tensor([guidance * 1000.0]))[0]
        mod_img, mod_txt = self.double_stream_modulation_img(temb), self.double_stream_modulation_txt(temb)
        mod_single = self.single_stream_modulation(temb)[0]
        x, c = self.x_embedder(hidden), self.context_embedder(enc)
        cos, sin = rope_tables(torch.cat([txt_ids, img_ids]), self.cfg["axes_dim"], theta=self.cfg["rope_theta"])
        for b in self.transformer_blocks:
            x, c = b(x, c, mod_img, mod_txt, cos, sin)
        j = torch.cat([c, x])
        for b in self.single_transformer_blocks:
            j = b(j, mod_single, cos, sin)
        return self.proj_out(self.norm_out(j[c.shape[0]:], temb))


class VAE2(nn.Module):
    def __init__(self, ch=(128, 256, 512, 512), latent=32, groups=32, bn_eps=1e-4):
        super().__init__()
        self.cfg = dict(ch=tuple(ch), latent=latent, groups=groups, bn_eps=bn_eps)
        self.encoder, self.decoder = Encoder(ch, latent, groups), Decoder(ch, latent, groups)
        self.quant_conv, self.post_quant_conv = nn.Conv2d(2 * latent, 2 * latent, 1), nn.Conv2d(latent, latent, 1)
        self.bn = nn.BatchNorm2d(4 * latent, eps=bn_eps, affine=False, track_running_stats=True)

    @torch.no_grad()
    def encode_mode(self, x):
        return self.quant_conv(self.encoder(x))[:, : self.cfg["latent"]]

    @torch.no_grad()
    def decode(self, z):
        return self.decoder(self.post_quant_conv(z))

    def bn_stats(self):
        return self.bn.running_mean.view(1, -1, 1, 1), torch.sqrt(self.bn.running_var.view(1, -1, 1, 1) + self.cfg["bn_eps"])


# ---- pipeline ---------------------------------------------------------------------------------------------
def patchify(lat):           # [1, C, H, W] -> [1, 4C, H/2, W/2], channel = c*4 + dy*2 + dx
    _, c, h, w = lat.shape
    return lat.view(1, c, h // 2, 2, w // 2, 2).permute(0, 1, 3, 5, 2, 4).reshape(1, c * 4, h // 2, w // 2)


def unpatchify(lat):         # inverse
    _, c4, h2, w2 = lat.shape
    return lat.view(1, c4 // 4, 2, 2, h2, w2).permute(0, 1, 4, 2, 5, 3).reshape(1, c4 // 4, h2 * 2, w2 * 2)


def pack(lat):               # [1, C, H, W] -> [H*W, C]
    return lat[0].flatten(1).t()


def unpack(tok, h, w):       # [H*W, C] -> [1, C, H, W]
    return tok.t().reshape(1, -1, h, w)


def latent_ids(h2, w2, t=0):
    ids = torch.zeros(h2, w2, 4)
    ids[..., 0] = t
    ids[..., 1] = torch.arange(h2)[:, None]
    ids[..., 2] = torch.arange(w2)[None, :]
    return ids.view(-1, 4)


def text_ids(n):
    ids = torch.zeros(n, 4)
    ids[:, 3] = torch.arange(n)
    return ids


def compute_empirical_mu(image_seq_len: int, num_steps: int) -> float:
    a1, b1 = 8.73809524e-05, 1.89833333
    a2, b2 = 0.00016927, 0.45666666
    if image_seq_len > 4300:
        return float(a2 * image_seq_len + b2)
    m_200 = a2 * image_seq_len + b2
    m_10 = a1 * image_seq_len + b1
    a = (m_200 - m_10) / 190.0
    b = m_200 - 200.0 * a
    return float(a * num_steps + b)


def flow_sigmas(steps, image_seq_len):
    s = np.linspace(1.0, 1.0 / steps, steps)
    mu = compute_empirical_mu(image_seq_len, steps)
    s = math.exp(mu) / (math.exp(mu) + (1.0 / s - 1.0))
    return np.append(s, 0.0).astype(np.float32)


@torch.no_grad()
def klein(transformer, vae, image_u8: np.ndarray, prompt_embeds, steps, noise, ref_image_u8=None):
    """image uint8 [H,W,3] (H, W multiples of 16) is both the output geometry and (unless `ref_image_u8` gives a
    resized copy) the reference image; noise [1, 128, H/16, W/16] fp32 -> float image [3,H,W] in 0..1"""
    H, W = image_u8.shape[:2]
    h2, w2 = H // 16, W // 16
    ref_u8 = image_u8 if ref_image_u8 is None else ref_image_u8
    rh2, rw2 = ref_u8.shape[0] // 16, ref_u8.shape[1] // 16
    mean, std = vae.bn_stats()
    x = torch.from_numpy(ref_u8).permute(2, 0, 1)[None].float() / 127.5 - 1.0
    ref_tok = pack((patchify(vae.encode_mode(x)) - mean) / std)
    lat = pack(noise)
    ids = torch.cat([latent_ids(h2, w2, 0), latent_ids(rh2, rw2, 10)])
    tids = text_ids(prompt_embeds.shape[0])
    sig = flow_sigmas(steps, lat.shape[0])
    for i in range(steps):
        v = transformer(torch.cat([lat, ref_tok]), float(sig[i]), prompt_embeds, tids, ids)[: lat.shape[0]]
        lat = lat + (float(sig[i + 1]) - float(sig[i])) * v
    z = unpatchify(unpack(lat, h2, w2) * std + mean)
    return (vae.decode(z)[0] / 2 + 0.5).clamp(0, 1), dict(latents=lat, ref_tokens=ref_tok, sigmas=sig)


def make_models(seed=0, d=128, heads=2, layers=2, single_layers=2, joint_dim=96, axes_dim=(16, 16, 16, 16),
                vae_ch=(32, 64, 64, 64), groups=8, latent=32):
    torch.manual_seed(seed)
    t = Flux2Transformer(d, heads, layers, single_layers, 4 * latent, joint_dim, 3.0, axes_dim).eval()
    v = VAE2(vae_ch, latent, groups).eval()
    with torch.no_grad():
        for m in (t, v):
            for name, p in m.named_parameters():
                if p.dim() >= 2:
                    p.normal_(0, 1.0 / math.sqrt(p[0].numel()))
                elif "norm" in name and name.endswith("weight"):
                    p.normal_(1.0, 0.1)
                else:
                    p.normal_(0, 0.05)
        v.bn.running_mean.normal_(0, 0.3)
        v.bn.running_var.uniform_(0.5, 2.0)
    return t, v
