"""CPU oracle for the FLUX.1-Kontext inpainting pipeline.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the reference drives diffusers' `FluxKontextPipeline` (reference
core/image/inpainting.py:877-887, 918-928; loaded at core/ml/model_manager.py:1176-1252 with
`diffusers>=0.37.0` + `sdnq`, neither installed here, no checkpoints present).  This file restates the
published architecture in plain torch fp32, with diffusers' module / state-dict names:

  FluxTransformer2DModel  x_embedder, context_embedder, time_text_embed (sinusoidal timestep +
      guidance + pooled-text MLPs), N double-stream blocks (AdaLayerNormZero, joint attention with
      per-head RMSNorm on q/k and 3-axis RoPE, gated residuals, GELU-tanh MLP), M single-stream blocks
      (AdaLayerNormZeroSingle, fused attention + MLP, proj_out over their concat), AdaLayerNormContinuous,
      proj_out
  AutoencoderKL           ResNet/GroupNorm/SiLU encoder and decoder with a single-head attention mid block,
      scaling_factor / shift_factor
  FlowMatchEulerDiscrete  sigmas = linspace(1, 1/N, N), exponential time shift with
      mu = calculate_shift(image_seq_len), Euler update x += (sigma_next - sigma) * v
  Kontext                 reference image latents appended as extra tokens (ids[...,0] = 1), only the noise
      tokens' prediction is kept
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


# ---------------------------------------------------------------------------------------------------
def timestep_embedding(t: torch.Tensor, dim=256) -> torch.Tensor:
    half = dim // 2
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    return torch.cat([args.cos(), args.sin()], -1)          # flip_sin_to_cos=True


def rope_tables(ids: torch.Tensor, axes_dim, theta=10000.0):
    """ids [S, 3] -> cos, sin [S, sum(axes_dim)] with every frequency repeated for its (even, odd) pair."""
    cos, sin = [], []
    for i, d in enumerate(axes_dim):
        freqs = 1.0 / (theta ** (torch.arange(0, d, 2, dtype=torch.float64) / d))
        ang = ids[:, i].double()[:, None] * freqs[None]
        cos.append(ang.cos().repeat_interleave(2, 1).float())
        sin.append(ang.sin().repeat_interleave(2, 1).float())
    return torch.cat(cos, 1), torch.cat(sin, 1)


def apply_rope(x, cos, sin):
    """x [S, H, D]; pairs (2k, 2k+1): (a, b) -> (a cos - b sin, b cos + a sin)"""
    xr = x.reshape(*x.shape[:-1], -1, 2)
    rot = torch.stack([-xr[..., 1], xr[..., 0]], -1).flatten(-2)
    return x * cos[:, None] + rot * sin[:, None]


class RMSNorm(nn.Module):
    def __init__(self, d, eps=1e-6):
        super().__init__()
        self.weight, self.eps = nn.Parameter(torch.ones(d)), eps

    def forward(self, x):
        return x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + self.eps) * self.weight


class MLPEmbed(nn.Module):
    def __init__(self, din, d):
        super().__init__()
        self.linear_1, self.linear_2 = nn.Linear(din, d), nn.Linear(d, d)

    def forward(self, x):
        return self.linear_2(F.silu(self.linear_1(x)))


class TimeTextEmbed(nn.Module):
    def __init__(self, d, pooled):
        super().__init__()
        self.timestep_embedder, self.guidance_embedder, self.text_embedder = MLPEmbed(256, d), MLPEmbed(256, d), MLPEmbed(pooled, d)

    def forward(self, t, g, pooled):
        return self.timestep_embedder(timestep_embedding(t)) + self.guidance_embedder(timestep_embedding(g)) + self.text_embedder(pooled)


class AdaNorm(nn.Module):
    def __init__(self, d, n):
        super().__init__()
        self.linear, self.n = nn.Linear(d, n * d), n

    def forward(self, temb):
        return self.linear(F.silu(temb)).chunk(self.n, -1)


class Attn(nn.Module):
    def __init__(self, d, heads, context):
        super().__init__()
        hd = d // heads
        self.heads, self.hd = heads, hd
        self.to_q, self.to_k, self.to_v = nn.Linear(d, d), nn.Linear(d, d), nn.Linear(d, d)
        self.norm_q, self.norm_k = RMSNorm(hd), RMSNorm(hd)
        if context:
            self.add_q_proj, self.add_k_proj, self.add_v_proj = nn.Linear(d, d), nn.Linear(d, d), nn.Linear(d, d)
            self.norm_added_q, self.norm_added_k = RMSNorm(hd), RMSNorm(hd)
            self.to_out = nn.ModuleList([nn.Linear(d, d)])
            self.to_add_out = nn.Linear(d, d)

    def qkv(self, x, ctx=False):
        S = x.shape[0]
        if ctx:
            q, k, v = self.add_q_proj(x), self.add_k_proj(x), self.add_v_proj(x)
            nq, nk = self.norm_added_q, self.norm_added_k
        else:
            q, k, v = self.to_q(x), self.to_k(x), self.to_v(x)
            nq, nk = self.norm_q, self.norm_k
        sh = lambda t: t.view(S, self.heads, self.hd)
        return nq(sh(q)), nk(sh(k)), sh(v)


def sdpa(q, k, v):
    o = F.scaled_dot_product_attention(q.transpose(0, 1)[None], k.transpose(0, 1)[None], v.transpose(0, 1)[None])[0]
    return o.transpose(0, 1).reshape(q.shape[0], -1)


class FF(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.net = nn.ModuleList([nn.Module(), nn.Identity(), nn.Linear(4 * d, d)])
        self.net[0].proj = nn.Linear(d, 4 * d)

    def forward(self, x):
        return self.net[2](F.gelu(self.net[0].proj(x), approximate="tanh"))


class DoubleBlock(nn.Module):
    def __init__(self, d, heads):
        super().__init__()
        self.norm1, self.norm1_context = AdaNorm(d, 6), AdaNorm(d, 6)
        self.attn = Attn(d, heads, True)
        self.ff, self.ff_context = FF(d), FF(d)

    def forward(self, x, c, temb, cos, sin):
        ln = lambda t: F.layer_norm(t, (t.shape[-1],), eps=1e-6)
        sh, sc, g, sh2, sc2, g2 = self.norm1(temb)
        csh, csc, cg, csh2, csc2, cg2 = self.norm1_context(temb)
        q, k, v = self.attn.qkv(ln(x) * (1 + sc) + sh)
        cq, ck, cv = self.attn.qkv(ln(c) * (1 + csc) + csh, ctx=True)
        Q, K, V = torch.cat([cq, q]), torch.cat([ck, k]), torch.cat([cv, v])
        o = sdpa(apply_rope(Q, cos, sin), apply_rope(K, cos, sin), V)
        oc, ox = o[: c.shape[0]], o[c.shape[0]:]
        x = x + g * self.attn.to_out[0](ox)
        x = x + g2 * self.ff(ln(x) * (1 + sc2) + sh2)
        c = c + cg * self.attn.to_add_out(oc)
        c = c + cg2 * self.ff_context(ln(c) * (1 + csc2) + csh2)
        return x, c


class SingleBlock(nn.Module):
    def __init__(self, d, heads):
        super().__init__()
        self.norm = AdaNorm(d, 3)
        self.proj_mlp, self.proj_out = nn.Linear(d, 4 * d), nn.Linear(5 * d, d)
        self.attn = Attn(d, heads, False)

    def forward(self, x, temb, cos, sin):
        sh, sc, g = self.norm(temb)
        n = F.layer_norm(x, (x.shape[-1],), eps=1e-6) * (1 + sc) + sh
        q, k, v = self.attn.qkv(n)
        o = sdpa(apply_rope(q, cos, sin), apply_rope(k, cos, sin), v)
        return x + g * self.proj_out(torch.cat([o, F.gelu(self.proj_mlp(n), approximate="tanh")], -1))


class FluxTransformer(nn.Module):
    def __init__(self, d=3072, heads=24, layers=19, single_layers=38, in_channels=64, joint_dim=4096, pooled_dim=768,
                 axes_dim=(16, 56, 56)):
        super().__init__()
        self.cfg = dict(d=d, heads=heads, layers=layers, single_layers=single_layers, in_channels=in_channels,
                        joint_dim=joint_dim, pooled_dim=pooled_dim, axes_dim=tuple(axes_dim))
        self.x_embedder, self.context_embedder = nn.Linear(in_channels, d), nn.Linear(joint_dim, d)
        self.time_text_embed = TimeTextEmbed(d, pooled_dim)
        self.transformer_blocks = nn.ModuleList(DoubleBlock(d, heads) for _ in range(layers))
        self.single_transformer_blocks = nn.ModuleList(SingleBlock(d, heads) for _ in range(single_layers))
        self.norm_out = AdaNorm(d, 2)
        self.proj_out = nn.Linear(d, in_channels)

    @torch.no_grad()
    def forward(self, hidden, timestep, guidance, pooled, enc, txt_ids, img_ids):
        """hidden [S_img, C]; enc [S_txt, joint]; timestep/guidance scalars in [0,1]-scale; ids [S,3]"""
        x, c = self.x_embedder(hidden), self.context_embedder(enc)
        temb = self.time_text_embed(torch.tensor([timestep * 1000.0]), torch.tensor([guidance * 1000.0]), pooled[None])[0]
        cos, sin = rope_tables(torch.cat([txt_ids, img_ids]), self.cfg["axes_dim"])
        for b in self.transformer_blocks:
            x, c = b(x, c, temb, cos, sin)
        j = torch.cat([c, x])
        for b in self.single_transformer_blocks:
            j = b(j, temb, cos, sin)
        x = j[c.shape[0]:]
        scale, shift = self.norm_out(temb)
        return self.proj_out(F.layer_norm(x, (x.shape[-1],), eps=1e-6) * (1 + scale) + shift)


# ---- VAE ---------------------------------------------------------------------------------------------
class Res(nn.Module):
    def __init__(self, cin, cout, groups):
        super().__init__()
        self.norm1, self.conv1 = nn.GroupNorm(groups, cin, eps=1e-6), nn.Conv2d(cin, cout, 3, padding=1)
        self.norm2, self.conv2 = nn.GroupNorm(groups, cout, eps=1e-6), nn.Conv2d(cout, cout, 3, padding=1)
        if cin != cout:
            self.conv_shortcut = nn.Conv2d(cin, cout, 1)

    def forward(self, x):
        h = self.conv2(F.silu(self.norm2(self.conv1(F.silu(self.norm1(x))))))
        return (self.conv_shortcut(x) if hasattr(self, "conv_shortcut") else x) + h


class VAttn(nn.Module):
    def __init__(self, c, groups):
        super().__init__()
        self.group_norm = nn.GroupNorm(groups, c, eps=1e-6)
        self.to_q, self.to_k, self.to_v = nn.Linear(c, c), nn.Linear(c, c), nn.Linear(c, c)
        self.to_out = nn.ModuleList([nn.Linear(c, c)])

    def forward(self, x):
        b, c, h, w = x.shape
        t = self.group_norm(x).flatten(2).transpose(1, 2)
        o = F.scaled_dot_product_attention(self.to_q(t)[:, None], self.to_k(t)[:, None], self.to_v(t)[:, None])[:, 0]
        return x + self.to_out[0](o).transpose(1, 2).view(b, c, h, w)


class Mid(nn.Module):
    def __init__(self, c, groups):
        super().__init__()
        self.resnets = nn.ModuleList([Res(c, c, groups), Res(c, c, groups)])
        self.attentions = nn.ModuleList([VAttn(c, groups)])

    def forward(self, x):
        return self.resnets[1](self.attentions[0](self.resnets[0](x)))


class _Sampler(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, padding=0)


class Down(nn.Module):
    def __init__(self, cin, cout, groups, down):
        super().__init__()
        self.resnets = nn.ModuleList([Res(cin, cout, groups), Res(cout, cout, groups)])
        if down:
            self.downsamplers = nn.ModuleList([_Sampler(cout)])

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        if hasattr(self, "downsamplers"):
            x = F.conv2d(F.pad(x, (0, 1, 0, 1)), self.downsamplers[0].conv.weight, self.downsamplers[0].conv.bias, stride=2)
        return x


class Up(nn.Module):
    def __init__(self, cin, cout, groups, up):
        super().__init__()
        self.resnets = nn.ModuleList([Res(cin if i == 0 else cout, cout, groups) for i in range(3)])
        if up:
            self.upsamplers = nn.ModuleList([_Sampler(cout)])

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        if hasattr(self, "upsamplers"):
            x = F.conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest"), self.upsamplers[0].conv.weight, self.upsamplers[0].conv.bias, padding=1)
        return x


class Encoder(nn.Module):
    def __init__(self, ch, latent, groups):
        super().__init__()
        self.conv_in = nn.Conv2d(3, ch[0], 3, padding=1)
        self.down_blocks = nn.ModuleList(Down(ch[max(i - 1, 0)], ch[i], groups, i < len(ch) - 1) for i in range(len(ch)))
        self.mid_block = Mid(ch[-1], groups)
        self.conv_norm_out, self.conv_out = nn.GroupNorm(groups, ch[-1], eps=1e-6), nn.Conv2d(ch[-1], 2 * latent, 3, padding=1)

    def forward(self, x):
        x = self.conv_in(x)
        for d in self.down_blocks:
            x = d(x)
        return self.conv_out(F.silu(self.conv_norm_out(self.mid_block(x))))


class Decoder(nn.Module):
    def __init__(self, ch, latent, groups):
        super().__init__()
        rc = list(reversed(ch))
        self.conv_in = nn.Conv2d(latent, rc[0], 3, padding=1)
        self.mid_block = Mid(rc[0], groups)
        self.up_blocks = nn.ModuleList(Up(rc[max(i - 1, 0)], rc[i], groups, i < len(rc) - 1) for i in range(len(rc)))
        self.conv_norm_out, self.conv_out = nn.GroupNorm(groups, rc[-1], eps=1e-6), nn.Conv2d(rc[-1], 3, 3, padding=1)

    def forward(self, z):
        x = self.mid_block(self.conv_in(z))
        for u in self.up_blocks:
            x = u(x)
        return self.conv_out(F.silu(self.conv_norm_out(x)))


class VAE(nn.Module):
    def __init__(self, ch=(128, 256, 512, 512), latent=16, groups=32, scaling_factor=0.3611, shift_factor=0.1159):
        super().__init__()
        self.cfg = dict(ch=tuple(ch), latent=latent, groups=groups, scaling_factor=scaling_factor, shift_factor=shift_factor)
        self.encoder, self.decoder = Encoder(ch, latent, groups), Decoder(ch, latent, groups)

    @torch.no_grad()
    def encode_mode(self, x):
        return self.encoder(x)[:, : self.cfg["latent"]]

    @torch.no_grad()
    def decode(self, z):
        return self.decoder(z)


# ---- pipeline ------------------------------------------------------------------------------------------
def pack(lat):        # [1, C, H, W] -> [(H/2)(W/2), 4C]
    _, c, h, w = lat.shape
    return lat.view(1, c, h // 2, 2, w // 2, 2).permute(0, 2, 4, 1, 3, 5).reshape((h // 2) * (w // 2), c * 4)


def unpack(tok, h, w, c):   # [(H/2)(W/2), 4C] -> [1, C, H, W]
    return tok.view(1, h // 2, w // 2, c, 2, 2).permute(0, 3, 1, 4, 2, 5).reshape(1, c, h, w)


def image_ids(h2, w2, first):
    ids = torch.zeros(h2, w2, 3)
    ids[..., 0] = first
    ids[..., 1] = torch.arange(h2)[:, None]
    ids[..., 2] = torch.arange(w2)[None, :]
    return ids.view(-1, 3)


def flow_sigmas(steps, image_seq_len):
    s = np.linspace(1.0, 1.0 / steps, steps)
    m = (1.15 - 0.5) / (4096 - 256)
    mu = image_seq_len * m + (0.5 - m * 256)
    s = math.exp(mu) / (math.exp(mu) + (1.0 / s - 1.0))
    return np.append(s, 0.0).astype(np.float32)


@torch.no_grad()
def kontext(transformer, vae, image_u8: np.ndarray, prompt_embeds, pooled, steps, guidance, noise):
    """image uint8 [H,W,3] (H, W multiples of 16); noise [1,16,H/8,W/8] fp32 -> float image [3,H,W] in 0..1"""
    H, W = image_u8.shape[:2]
    cfg = vae.cfg
    x = torch.from_numpy(image_u8).permute(2, 0, 1)[None].float() / 127.5 - 1.0
    ref = (vae.encode_mode(x) - cfg["shift_factor"]) * cfg["scaling_factor"]
    ref_tok, lat = pack(ref), pack(noise)
    h2, w2 = H // 16, W // 16
    ids = torch.cat([image_ids(h2, w2, 0), image_ids(h2, w2, 1)])
    txt_ids = torch.zeros(prompt_embeds.shape[0], 3)
    sig = flow_sigmas(steps, lat.shape[0])
    for i in range(steps):
        v = transformer(torch.cat([lat, ref_tok]), float(sig[i]), guidance, pooled, prompt_embeds, txt_ids, ids)[: lat.shape[0]]
        lat = lat + (float(sig[i + 1]) - float(sig[i])) * v
    z = unpack(lat, H // 8, W // 8, cfg["latent"]) / cfg["scaling_factor"] + cfg["shift_factor"]
    return (vae.decode(z)[0] / 2 + 0.5).clamp(0, 1), dict(latents=lat, ref_tokens=ref_tok, sigmas=sig)


def make_models(seed=0, d=128, heads=2, layers=2, single_layers=2, joint_dim=64, pooled_dim=32, axes_dim=(8, 28, 28),
                vae_ch=(32, 64, 64, 64), groups=8):
    torch.manual_seed(seed)
    t = FluxTransformer(d, heads, layers, single_layers, 64, joint_dim, pooled_dim, axes_dim).eval()
    v = VAE(vae_ch, 16, groups).eval()
    with torch.no_grad():
        for m in (t, v):
            for name, p in m.named_parameters():
                if p.dim() >= 2:
                    fan = p[0].numel()
                    p.normal_(0, 1.0 / math.sqrt(fan))
                elif "norm" in name and name.endswith("weight"):
                    p.normal_(1.0, 0.1)
                else:
                    p.normal_(0, 0.05)
    return t, v
