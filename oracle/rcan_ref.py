"""CPU oracle for the 2x-AnimeSharpV4 upscaler (RCAN family).  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the reference loads `2x-AnimeSharpV4_RCAN.safetensors` /
`2x-AnimeSharpV4_Fast_RCAN_PU.safetensors` through spandrel's architecture auto-detection
(reference core/ml/model_manager.py:617-700, `ModelLoader().load_from_state_dict`) and calls the
result as `model(tensor[1,3,H,W] fp32 in 0..1)` (reference core/image/image_utils.py:369-374).
spandrel (>=0.3.0, un-pinned, requirements.txt) is NOT installed in this image and no checkpoint
is present, so this file restates the published RCAN architecture (Zhang et al., ECCV 2018,
"Image Super-Resolution Using Very Deep Residual Channel Attention Networks", the EDSR-PyTorch
`model/rcan.py` layout that spandrel's RCAN state-dict keys follow):

    x*rgb_range -> [sub_mean] -> head conv3x3
      -> n_resgroups x [ n_resblocks x RCAB(conv3x3, ReLU, conv3x3, CA(avgpool,1x1,ReLU,1x1,sigmoid)) + conv3x3 ] (+skip)
      -> conv3x3 (+ long skip) -> Upsampler(conv3x3 -> 4C, PixelShuffle(2)) x log2(scale') -> conv3x3
      -> [add_mean] -> /rgb_range
    "PU" (pixel-unshuffle) variants first fold 2x2 pixel blocks into channels (12 input channels; odd sizes are padded
    bottom / right by reflection and the output cropped — the padding mode of the absent upstream wheel is ASSUMED: parity unpinned)
    and upsample 4x internally.

State-dict keys: head.0, body.{g}.body.{b}.body.{0,2}, body.{g}.body.{b}.body.3.conv_du.{0,2},
body.{g}.body.{n_resblocks}, body.{n_resgroups}, tail.0.{0,2,..}, tail.1, optional sub_mean/add_mean.
Hyper-parameters are derived from tensor shapes exactly as an auto-detecting loader must.
"""
import math
import re

import torch
import torch.nn as nn
import torch.nn.functional as F


def rcan_hparams(sd: dict) -> dict:
    """Derive the architecture from a state dict (shapes only)."""
    n_feats, in_ch = sd["head.0.weight"].shape[:2]
    groups = set()
    blocks = {}
    for k in sd:
        m = re.match(r"body\.(\d+)\.body\.(\d+)\.", k)
        if m:
            g, b = int(m.group(1)), int(m.group(2))
            groups.add(g)
            blocks[g] = max(blocks.get(g, -1), b)
    n_resgroups = len(groups)
    n_resblocks = blocks[min(groups)]  # last index in a group is the group's tail conv
    cr = sd["body.0.body.0.body.3.conv_du.0.weight"].shape[0]
    ups = sorted({int(m.group(1)) for k in sd for m in [re.match(r"tail\.0\.(\d+)\.weight", k)] if m})
    up_factors = []
    for u in ups:
        ratio = sd[f"tail.0.{u}.weight"].shape[0] // n_feats
        r = int(round(math.sqrt(ratio)))
        assert r * r == ratio
        up_factors.append(r)
    n_colors = sd["tail.1.weight"].shape[0]
    unshuffle = int(round(math.sqrt(in_ch // n_colors)))
    total_up = 1
    for r in up_factors:
        total_up *= r
    return dict(n_feats=int(n_feats), n_resgroups=n_resgroups, n_resblocks=n_resblocks, cr=int(cr),
                up_factors=up_factors, up_keys=ups, n_colors=int(n_colors), unshuffle=unshuffle,
                scale=total_up // unshuffle, mean_shift=("sub_mean.weight" in sd))


class _CA(nn.Module):
    def __init__(self, c, cr):
        super().__init__()
        self.conv_du = nn.Sequential(nn.Conv2d(c, cr, 1), nn.ReLU(), nn.Conv2d(cr, c, 1), nn.Sigmoid())

    def forward(self, x):
        return x * self.conv_du(x.mean(dim=(2, 3), keepdim=True))


class _RCAB(nn.Module):
    def __init__(self, c, cr):
        super().__init__()
        self.body = nn.Sequential(nn.Conv2d(c, c, 3, padding=1), nn.ReLU(), nn.Conv2d(c, c, 3, padding=1), _CA(c, cr))

    def forward(self, x):
        return self.body(x) + x


class _Group(nn.Module):
    def __init__(self, c, cr, nb):
        super().__init__()
        self.body = nn.Sequential(*[_RCAB(c, cr) for _ in range(nb)], nn.Conv2d(c, c, 3, padding=1))

    def forward(self, x):
        return self.body(x) + x


class RCANRef(nn.Module):
    def __init__(self, hp: dict, rgb_range: float = 255.0):
        super().__init__()
        c, cr = hp["n_feats"], hp["cr"]
        self.hp, self.rgb_range = hp, float(rgb_range)
        in_ch = hp["n_colors"] * hp["unshuffle"] ** 2
        if hp["mean_shift"]:
            self.sub_mean = nn.Conv2d(hp["n_colors"], hp["n_colors"], 1)
            self.add_mean = nn.Conv2d(hp["n_colors"], hp["n_colors"], 1)
        self.head = nn.Sequential(nn.Conv2d(in_ch, c, 3, padding=1))
        self.body = nn.Sequential(*[_Group(c, cr, hp["n_resblocks"]) for _ in range(hp["n_resgroups"])],
                                  nn.Conv2d(c, c, 3, padding=1))
        up = nn.Sequential()
        for key, r in zip(hp["up_keys"], hp["up_factors"]):
            while len(up) < key:
                up.append(nn.Identity())
            up.append(nn.Conv2d(c, c * r * r, 3, padding=1))
            up.append(nn.PixelShuffle(r))
        self.tail = nn.Sequential(up, nn.Conv2d(c, hp["n_colors"], 3, padding=1))

    @torch.no_grad()
    def forward(self, x):
        u = self.hp["unshuffle"]
        h0, w0 = x.shape[-2:]
        ph, pw = (-h0) % u, (-w0) % u
        if ph or pw:        # the pixel-unshuffle variants take any size: pad bottom / right by reflection, crop the result
            x = F.pad(x, (0, pw, 0, ph), mode="reflect" if (h0 > ph and w0 > pw) else "replicate")
        x = x * self.rgb_range
        if self.hp["mean_shift"]:
            x = self.sub_mean(x)
        if u > 1:
            x = F.pixel_unshuffle(x, u)
        h = self.head(x)
        y = self.tail(self.body(h) + h)
        if self.hp["mean_shift"]:
            y = self.add_mean(y)
        y = y / self.rgb_range
        s = y.shape[-2] // (h0 + ph)
        return y[..., : h0 * s, : w0 * s]


def make_state_dict(n_feats=64, n_resgroups=10, n_resblocks=20, reduction=16, scale=2, unshuffle=1,
                    mean_shift=False, seed=0, gain=0.7) -> dict:
    """Seeded synthetic checkpoint with the real key layout (no real weights exist offline).
    Residual branches are damped so activations stay O(rgb_range) through 200+ layers."""
    g = torch.Generator().manual_seed(seed)
    sd = {}

    def conv(name, co, ci, k, wgain=1.0, bias_std=0.01):
        std = wgain / math.sqrt(ci * k * k)
        sd[name + ".weight"] = torch.randn(co, ci, k, k, generator=g) * std
        sd[name + ".bias"] = torch.randn(co, generator=g) * bias_std

    c, cr = n_feats, max(n_feats // reduction, 1)
    conv("head.0", c, 3 * unshuffle ** 2, 3, 1.0)
    for gi in range(n_resgroups):
        for b in range(n_resblocks):
            p = f"body.{gi}.body.{b}.body"
            conv(p + ".0", c, c, 3, 1.4)
            conv(p + ".2", c, c, 3, gain * 0.5)
            conv(p + ".3.conv_du.0", cr, c, 1, 1.0, 0.1)
            conv(p + ".3.conv_du.2", c, cr, 1, 1.0, 0.1)
        conv(f"body.{gi}.body.{n_resblocks}", c, c, 3, gain * 0.3)
    conv(f"body.{n_resgroups}", c, c, 3, gain * 0.5)
    total = scale * unshuffle
    k = 0
    while total > 1:
        conv(f"tail.0.{k}", 4 * c, c, 3, 1.0)
        k += 2
        total //= 2
    conv("tail.1", 3, c, 3, 0.6, 0.0)
    sd["tail.1.bias"] = torch.full((3,), 0.5 * 255.0)
    if mean_shift:
        mean = torch.tensor([0.4488, 0.4371, 0.4040])
        sd["sub_mean.weight"] = torch.eye(3).view(3, 3, 1, 1)
        sd["sub_mean.bias"] = -255.0 * mean
        sd["add_mean.weight"] = torch.eye(3).view(3, 3, 1, 1)
        sd["add_mean.bias"] = 255.0 * mean
    return sd


def load_ref(sd: dict, rgb_range: float = 255.0) -> RCANRef:
    hp = rcan_hparams(sd)
    m = RCANRef(hp, rgb_range)
    remap = {}
    for k, v in sd.items():
        k2 = re.sub(r"^tail\.0\.", "tail.0.", k)
        remap[k2] = v.float()
    missing, unexpected = m.load_state_dict(remap, strict=False)
    assert not unexpected, unexpected
    assert not [k for k in missing if "Identity" not in k], missing
    return m.eval()
