"""CPU oracle for the YOLO11 / YOLO12 detectors of the hot path.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the reference loads three more ultralytics checkpoints — the default bubble detector `yolo_2`
(manga109-segmentation-bubble, a YOLO11-seg; core/ml/model_manager.py:120-125, 183-190), the panel detector (YOLO11-L,
:809-838, called at core/image/detection.py:1867-1873) and the outside-text detector (AnimeText YOLO12x, :780-808, called at
detection.py:144-150 and ocr_detection.py:425-431) — through `ultralytics>=8.3.94`, which is not installed here, and no checkpoint is
present.  This file restates the published architectures (ultralytics cfg/models/11/yolo11{,-seg}.yaml, cfg/models/12/yolo12.yaml,
nn/modules/{conv,block,head}.py) with ultralytics' fused state-dict names:

  YOLO11   Conv-Conv-C3k2-Conv-C3k2-Conv-C3k2-Conv-C3k2-SPPF-C2PSA backbone, PAN neck of C3k2 blocks, Detect / Segment head whose
           class branch is DWConv3x3 + Conv1x1 twice.  C3k2 = C2f whose inner blocks are C3k (CSP with two 3x3 bottlenecks) on the
           m / l / x scales and plain bottlenecks (hidden = c / 2) otherwise; C2PSA = split, n x PSABlock (multi-head attention with
           key_dim = head_dim / 2 and a depthwise 3x3 positional conv on v, then a 2x-wide 1x1 MLP), merge.
  YOLO12   the same skeleton with A2C2f blocks at P4 / P5: area attention (the flattened H*W sequence cut into `area` contiguous
           chunks that attend only inside themselves, head_dim 32, depthwise 7x7 positional conv on v), MLP of ratio 1.2 on the
           l / x scales and a learnt residual scale `gamma`; the neck's A2C2f blocks are the C3k form (a2 = False).
Letterbox, DFL decode, NMS, box scaling and retina masks are those of oracle/yolo_ref.py.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .yolo_ref import Conv, Proto, SPPF, make_divisible

SCALES11 = {"n": (0.50, 0.25, 1024), "s": (0.50, 0.50, 1024), "m": (0.50, 1.00, 512), "l": (1.00, 1.00, 512), "x": (1.00, 1.50, 512)}


class ConvG(nn.Module):
    """Conv with groups (depthwise when g == c): ultralytics DWConv / Conv(g=...)"""

    def __init__(self, c1, c2, k=1, s=1, g=1, act=True):
        super().__init__()
        self.conv = nn.Conv2d(c1, c2, k, s, k // 2, groups=g, bias=True)
        self.act = act

    def forward(self, x):
        y = self.conv(x)
        return F.silu(y) if self.act else y


class Bottleneck(nn.Module):
    def __init__(self, c1, c2, shortcut=True, e=0.5):
        super().__init__()
        c_ = int(c2 * e)
        self.cv1, self.cv2, self.add = Conv(c1, c_, 3), Conv(c_, c2, 3), shortcut and c1 == c2

    def forward(self, x):
        y = self.cv2(self.cv1(x))
        return x + y if self.add else y


class C3k(nn.Module):
    def __init__(self, c1, c2, n=2, shortcut=True, e=0.5):
        super().__init__()
        c_ = int(c2 * e)
        self.cv1, self.cv2, self.cv3 = Conv(c1, c_, 1), Conv(c1, c_, 1), Conv(2 * c_, c2, 1)
        self.m = nn.Sequential(*(Bottleneck(c_, c_, shortcut, e=1.0) for _ in range(n)))

    def forward(self, x):
        return self.cv3(torch.cat((self.m(self.cv1(x)), self.cv2(x)), 1))


class C3k2(nn.Module):
    def __init__(self, c1, c2, n=1, c3k=False, e=0.5, shortcut=True):
        super().__init__()
        self.c = int(c2 * e)
        self.cv1, self.cv2 = Conv(c1, 2 * self.c, 1), Conv((2 + n) * self.c, c2, 1)
        self.m = nn.ModuleList(C3k(self.c, self.c, 2, shortcut) if c3k else Bottleneck(self.c, self.c, shortcut) for _ in range(n))

    def forward(self, x):
        y = list(self.cv1(x).chunk(2, 1))
        for m in self.m:
            y.append(m(y[-1]))
        return self.cv2(torch.cat(y, 1))


class Attention(nn.Module):
    def __init__(self, dim, num_heads, attn_ratio=0.5):
        super().__init__()
        self.nh, self.hd = num_heads, dim // num_heads
        self.kd = int(self.hd * attn_ratio)
        self.scale = self.kd ** -0.5
        self.qkv = Conv(dim, dim + 2 * self.kd * num_heads, 1, act=False)
        self.proj = Conv(dim, dim, 1, act=False)
        self.pe = ConvG(dim, dim, 3, 1, g=dim, act=False)

    def forward(self, x):
        B, C, H, W = x.shape
        N = H * W
        q, k, v = self.qkv(x).view(B, self.nh, 2 * self.kd + self.hd, N).split([self.kd, self.kd, self.hd], 2)
        attn = ((q.transpose(-2, -1) @ k) * self.scale).softmax(-1)
        y = (v @ attn.transpose(-2, -1)).view(B, C, H, W) + self.pe(v.reshape(B, C, H, W))
        return self.proj(y)


class PSABlock(nn.Module):
    def __init__(self, c, num_heads):
        super().__init__()
        self.attn = Attention(c, num_heads)
        self.ffn = nn.Sequential(Conv(c, 2 * c, 1), Conv(2 * c, c, 1, act=False))

    def forward(self, x):
        x = x + self.attn(x)
        return x + self.ffn(x)


class C2PSA(nn.Module):
    def __init__(self, c1, n=1, e=0.5):
        super().__init__()
        self.c = int(c1 * e)
        self.cv1, self.cv2 = Conv(c1, 2 * self.c, 1), Conv(2 * self.c, c1, 1)
        self.m = nn.Sequential(*(PSABlock(self.c, max(self.c // 64, 1)) for _ in range(n)))

    def forward(self, x):
        a, b = self.cv1(x).split((self.c, self.c), 1)
        return self.cv2(torch.cat((a, self.m(b)), 1))


class AAttn(nn.Module):
    def __init__(self, dim, num_heads, area=1):
        super().__init__()
        self.area, self.nh, self.hd = area, num_heads, dim // num_heads
        self.qkv = Conv(dim, 3 * dim, 1, act=False)
        self.proj = Conv(dim, dim, 1, act=False)
        self.pe = ConvG(dim, dim, 7, 1, g=dim, act=False)

    def forward(self, x):
        B, C, H, W = x.shape
        N = H * W
        qkv = self.qkv(x).flatten(2).transpose(1, 2)                 # [B, N, 3C], per head the channels run q | k | v
        if self.area > 1:
            qkv = qkv.reshape(B * self.area, N // self.area, 3 * C)
        Bn, Nn, _ = qkv.shape
        q, k, v = qkv.view(Bn, Nn, self.nh, 3 * self.hd).permute(0, 2, 3, 1).split([self.hd] * 3, 2)
        attn = ((q.transpose(-2, -1) @ k) * self.hd ** -0.5).softmax(-1)
        y = (v @ attn.transpose(-2, -1)).permute(0, 3, 1, 2)         # [Bn, Nn, nh, hd]
        v = v.permute(0, 3, 1, 2)
        y = y.reshape(B, H, W, C).permute(0, 3, 1, 2)
        v = v.reshape(B, H, W, C).permute(0, 3, 1, 2)
        return self.proj(y + self.pe(v))


class ABlock(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=1.2, area=1):
        super().__init__()
        self.attn = AAttn(dim, num_heads, area)
        hid = int(dim * mlp_ratio)
        self.mlp = nn.Sequential(Conv(dim, hid, 1), Conv(hid, dim, 1, act=False))

    def forward(self, x):
        x = x + self.attn(x)
        return x + self.mlp(x)


class A2C2f(nn.Module):
    def __init__(self, c1, c2, n=1, a2=True, area=1, residual=False, mlp_ratio=2.0, e=0.5, shortcut=True):
        super().__init__()
        c_ = int(c2 * e)
        self.cv1, self.cv2 = Conv(c1, c_, 1), Conv((1 + n) * c_, c2, 1)
        if a2 and residual:
            self.gamma = nn.Parameter(0.01 * torch.ones(c2))
        self.m = nn.ModuleList(nn.Sequential(*(ABlock(c_, c_ // 32, mlp_ratio, area) for _ in range(2))) if a2 else C3k(c_, c_, 2, shortcut)
                               for _ in range(n))

    def forward(self, x):
        y = [self.cv1(x)]
        for m in self.m:
            y.append(m(y[-1]))
        y = self.cv2(torch.cat(y, 1))
        return x + self.gamma.view(1, -1, 1, 1) * y if hasattr(self, "gamma") else y


class Detect(nn.Module):
    """non-legacy head (YOLO11 / 12): box branch of two 3x3 convs, class branch of two (depthwise 3x3 + 1x1) pairs"""

    def __init__(self, ch, nc, reg_max=16, nm=0, npr=0):
        super().__init__()
        self.nc, self.nm, self.reg_max = nc, nm, reg_max
        c2, c3 = max(16, ch[0] // 4, reg_max * 4), max(ch[0], min(nc, 100))
        self.cv2 = nn.ModuleList(nn.Sequential(Conv(x, c2, 3), Conv(c2, c2, 3), nn.Conv2d(c2, 4 * reg_max, 1)) for x in ch)
        self.cv3 = nn.ModuleList(nn.Sequential(nn.Sequential(ConvG(x, x, 3, g=x), Conv(x, c3, 1)), nn.Sequential(ConvG(c3, c3, 3, g=c3), Conv(c3, c3, 1)),
                                               nn.Conv2d(c3, nc, 1)) for x in ch)
        if nm:
            c4 = max(ch[0] // 4, nm)
            self.cv4 = nn.ModuleList(nn.Sequential(Conv(x, c4, 3), Conv(c4, c4, 3), nn.Conv2d(c4, nm, 1)) for x in ch)
            self.proto = Proto(ch[0], npr, nm)

    def forward(self, xs):
        outs, anchors, strides = [], [], []
        for i, x in enumerate(xs):
            _, _, h, w = x.shape
            parts = [self.cv2[i](x), self.cv3[i](x)] + ([self.cv4[i](x)] if self.nm else [])
            outs.append(torch.cat(parts, 1).flatten(2))
            sy, sx = torch.meshgrid(torch.arange(h, dtype=torch.float32) + 0.5, torch.arange(w, dtype=torch.float32) + 0.5, indexing="ij")
            anchors.append(torch.stack([sx, sy], -1).view(-1, 2))
            strides.append(torch.full((h * w,), float(8 * 2 ** i)))
        y = torch.cat(outs, 2)
        anc, st = torch.cat(anchors).t(), torch.cat(strides)
        box, cls, mc = y.split([4 * self.reg_max, self.nc, self.nm], 1)
        b, _, a = box.shape
        dist = (box.view(b, 4, self.reg_max, a).softmax(2) * torch.arange(self.reg_max, dtype=torch.float32).view(1, 1, -1, 1)).sum(2)
        lt, rb = dist.chunk(2, 1)
        xyxy = torch.cat([anc[None] - lt, anc[None] + rb], 1) * st
        return torch.cat([xyxy, cls.sigmoid(), mc], 1), (self.proto(xs[0]) if self.nm else None)


def arch(family="11", scale="l", nc=1, seg=False):
    d, w, mc = SCALES11[scale]
    ch = lambda c: make_divisible(min(c, mc) * w, 8)
    dep = lambda n: max(round(n * d), 1)
    return dict(family=family, scale=scale, nc=nc, seg=seg, c=[ch(64), ch(128), ch(256), ch(512), ch(1024)], n2=dep(2), n4=dep(4), c3k=scale in "mlx",
                residual=scale in "lx", mlp_ratio=1.2 if scale in "lx" else 2.0, nm=32 if seg else 0, npr=ch(256), reg_max=16)


class Yolo11(nn.Module):
    """`model.{i}` indices as in yolo11.yaml (23 = head) / yolo12.yaml (21 = head)"""

    def __init__(self, a):
        super().__init__()
        self.a = a
        c, n2, k3 = a["c"], a["n2"], a["c3k"]
        I = nn.Identity
        if a["family"] == "11":
            m = [Conv(3, c[0], 3, 2), Conv(c[0], c[1], 3, 2), C3k2(c[1], c[2], n2, k3, 0.25), Conv(c[2], c[2], 3, 2), C3k2(c[2], c[3], n2, k3, 0.25),
                 Conv(c[3], c[3], 3, 2), C3k2(c[3], c[3], n2, True), Conv(c[3], c[4], 3, 2), C3k2(c[4], c[4], n2, True), SPPF(c[4], c[4]), C2PSA(c[4], n2),
                 I(), I(), C3k2(c[4] + c[3], c[3], n2, k3), I(), I(), C3k2(c[3] + c[3], c[2], n2, k3), Conv(c[2], c[2], 3, 2), I(),
                 C3k2(c[2] + c[3], c[3], n2, k3), Conv(c[3], c[3], 3, 2), I(), C3k2(c[3] + c[4], c[4], n2, True),
                 Detect([c[2], c[3], c[4]], a["nc"], a["reg_max"], a["nm"], a["npr"])]
        else:
            A = lambda c1, c2, n, a2, area: A2C2f(c1, c2, n, a2, area, a["residual"], a["mlp_ratio"])
            m = [Conv(3, c[0], 3, 2), Conv(c[0], c[1], 3, 2), C3k2(c[1], c[2], n2, k3, 0.25), Conv(c[2], c[2], 3, 2), C3k2(c[2], c[3], n2, k3, 0.25),
                 Conv(c[3], c[3], 3, 2), A(c[3], c[3], a["n4"], True, 4), Conv(c[3], c[4], 3, 2), A(c[4], c[4], a["n4"], True, 1),
                 I(), I(), A(c[4] + c[3], c[3], n2, False, -1), I(), I(), A(c[3] + c[3], c[2], n2, False, -1), Conv(c[2], c[2], 3, 2), I(),
                 A(c[2] + c[3], c[3], n2, False, -1), Conv(c[3], c[3], 3, 2), I(), C3k2(c[3] + c[4], c[4], n2, True),
                 Detect([c[2], c[3], c[4]], a["nc"], a["reg_max"], a["nm"], a["npr"])]
        self.model = nn.ModuleList(m)

    @torch.no_grad()
    def forward(self, x):
        m = self.model
        up = lambda t: F.interpolate(t, scale_factor=2.0, mode="nearest")
        x = m[1](m[0](x))
        p3 = m[4](m[3](m[2](x)))
        p4 = m[6](m[5](p3))
        if self.a["family"] == "11":
            p5 = m[10](m[9](m[8](m[7](p4))))
            h4 = m[13](torch.cat([up(p5), p4], 1))
            h3 = m[16](torch.cat([up(h4), p3], 1))
            n4 = m[19](torch.cat([m[17](h3), h4], 1))
            n5 = m[22](torch.cat([m[20](n4), p5], 1))
            return m[23]([h3, n4, n5])
        p5 = m[8](m[7](p4))
        h4 = m[11](torch.cat([up(p5), p4], 1))
        h3 = m[14](torch.cat([up(h4), p3], 1))
        n4 = m[17](torch.cat([m[15](h3), h4], 1))
        n5 = m[20](torch.cat([m[18](n4), p5], 1))
        return m[21]([h3, n4, n5])


def make_model(family="11", scale="n", nc=1, seg=False, seed=0):
    torch.manual_seed(seed)
    net = Yolo11(arch(family, scale, nc, seg)).eval().float()
    with torch.no_grad():
        for name, p in net.named_parameters():
            if name.endswith("gamma"):
                p.fill_(0.3)                     # a trained layer scale; the 0.01 initial value would hide the branch from a parity check
            elif p.dim() == 4:
                fan = p.shape[1] * p.shape[2] * p.shape[3]
                # the branches that are ADDED to the stream (attention proj, MLP output, positional conv) are damped like a trained
                # network's: eight undamped residual blocks in a row push seeded activations past 1e4, where f16 storage has no precision left
                damp = 0.25 if any(t in name for t in (".attn.proj.", ".mlp.1.", ".ffn.1.", ".attn.pe.")) else 1.0
                p.normal_(0, damp * (1.2 if fan > 49 else 0.8) / math.sqrt(fan))
            else:
                p.normal_(0, 0.1)
        head = net.model[-1]
        for l in range(3):
            head.cv3[l][2].bias.fill_(-2.0)
    return net


@torch.no_grad()
def calibrate(net, x):
    """Data-dependent conditioning of a seeded network (test infrastructure): one forward pass over `x` in which every convolution's
    weights and bias are rescaled so that its output on this input has zero mean and unit spread.  A plain seeded network loses the
    input-dependent part of its features within a dozen layers (each SiLU layer shrinks it, the biases do not shrink), and a parity
    check on spatially constant features would only test how constants propagate."""
    hooks = []

    def fix(mod, inp, out):
        s, m = out.std().item(), out.mean().item()
        if not (s > 1e-12):
            return out
        mod.weight.div_(s)
        if mod.bias is not None:
            mod.bias.sub_(m).div_(s)
            return (out - m) / s
        return out / s

    for mod in net.modules():
        if isinstance(mod, nn.Conv2d) and mod.weight.requires_grad:          # the DFL projection is a constant, not a parameter
            hooks.append(mod.register_forward_hook(fix))
    net(x)
    for h in hooks:
        h.remove()
    return net


@torch.no_grad()
def predict(net, img_bgr, imgsz=640, conf=0.25):
    from .yolo_ref import letterbox, postprocess
    x, lp = letterbox(img_bgr, imgsz)
    pred, proto = net(x)
    a = net.a
    if a["seg"]:
        out = postprocess(pred[0], proto[0], lp, img_bgr.shape[:2], conf=conf, nc=a["nc"])
        return dict(boxes=out[0], conf=out[1], cls=out[2], masks=out[3], pred=pred[0], proto=proto[0], lp=lp)
    fake = torch.zeros(1, 8, 8)
    out = postprocess(torch.cat([pred[0], torch.zeros(1, pred.shape[2])], 0), fake, lp, img_bgr.shape[:2], conf=conf, nc=a["nc"])
    return dict(boxes=out[0], conf=out[1], cls=out[2], masks=None, pred=pred[0], proto=None, lp=lp)
