"""Seeded stand-in checkpoints of the hot path's small networks, by PARAMETER INVENTORY.

No checkpoint can be downloaded where this package is benchmarked, so `bench.py` (and anyone who wants to time the graphs without the
weights) needs state dicts of the right names and shapes.  Each function below lists a network's parameters the way the checkpoint
the reference loads names them — ultralytics' fused `model.{i}...conv.weight / .bias` layout for the YOLO families
(reference core/ml/model_manager.py:711-743, 780-838), the RCAN safetensors' `body.{g}.body.{b}.body.{0,2,3.conv_du.*}` layout
(:617-700) — and fills them from a seeded generator with fan-in scaling; branches that are ADDED to a residual stream are damped the
way a trained network's are, so activations stay in 16-bit range through the depth.  SAM-2.1 and RT-DETR-v2 are instantiated from
HF `transformers` (the reference's own dependency for them) on their published configurations.

The product graph builders (core/ml/yolo.py, yolo11.py, rcan.py, sam2.py, rtdetr.py) consume these dicts exactly like real ones;
`tests/test_synthetic_checkpoints.py` checks every inventory against the CPU oracles' modules (names and shapes).
"""
import math
from collections import OrderedDict
from typing import Dict, Tuple

import torch

Shapes = "OrderedDict[str, Tuple[int, ...]]"


def _divisible(x: float, d: int = 8) -> int:
    return int(math.ceil(x / d) * d)


class _Inventory:
    """ordered name -> shape, with the few layer kinds the detectors are made of"""

    def __init__(self):
        self.shapes: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()

    def conv(self, name, cin, cout, k=1, groups=1):          # ultralytics Conv with its BatchNorm folded: <name>.conv.weight / .bias
        self.shapes[f"{name}.conv.weight"] = (cout, cin // groups, k, k)
        self.shapes[f"{name}.conv.bias"] = (cout,)

    def plain(self, name, cin, cout, k=1):                   # nn.Conv2d with bias
        self.shapes[f"{name}.weight"] = (cout, cin, k, k)
        self.shapes[f"{name}.bias"] = (cout,)

    def bottleneck(self, name, c, hidden=None):
        hidden = c if hidden is None else hidden
        self.conv(f"{name}.cv1", c, hidden, 3)
        self.conv(f"{name}.cv2", hidden, c, 3)

    def sppf(self, name, c1, c2):
        self.conv(f"{name}.cv1", c1, c1 // 2, 1)
        self.conv(f"{name}.cv2", c1 // 2 * 4, c2, 1)

    def branch(self, name, cin, mid, cout):                  # head branch: two 3x3 Convs and a 1x1 Conv2d
        self.conv(f"{name}.0", cin, mid, 3)
        self.conv(f"{name}.1", mid, mid, 3)
        self.plain(f"{name}.2", mid, cout, 1)

    def proto(self, name, cin, mid, nm):
        self.conv(f"{name}.cv1", cin, mid, 3)
        self.shapes[f"{name}.upsample.weight"] = (mid, mid, 2, 2)
        self.shapes[f"{name}.upsample.bias"] = (mid,)
        self.conv(f"{name}.cv2", mid, mid, 3)
        self.conv(f"{name}.cv3", mid, nm, 1)


# ---- YOLOv8-seg (yolo_1: yolov8m_seg-speech-bubble) -------------------------------------------------------------------------------
_V8_SCALES = {"n": (0.33, 0.25, 1024), "s": (0.33, 0.50, 1024), "m": (0.67, 0.75, 768), "l": (1.00, 1.00, 512), "x": (1.00, 1.25, 512)}


def yolov8_seg_shapes(scale: str = "m", nc: int = 1):
    depth, width, cmax = _V8_SCALES[scale]
    ch = [_divisible(min(c, cmax) * width) for c in (64, 128, 256, 512, 1024)]
    rep = [max(round(n * depth), 1) for n in (3, 6, 6, 3)]
    nh, nm, npr, reg_max = max(round(3 * depth), 1), 32, _divisible(min(256, cmax) * width), 16
    inv = _Inventory()

    def c2f(i, cin, cout, n):
        c = cout // 2
        inv.conv(f"model.{i}.cv1", cin, 2 * c, 1)
        inv.conv(f"model.{i}.cv2", (2 + n) * c, cout, 1)
        for j in range(n):
            inv.bottleneck(f"model.{i}.m.{j}", c)

    inv.conv("model.0", 3, ch[0], 3); inv.conv("model.1", ch[0], ch[1], 3); c2f(2, ch[1], ch[1], rep[0])
    inv.conv("model.3", ch[1], ch[2], 3); c2f(4, ch[2], ch[2], rep[1])
    inv.conv("model.5", ch[2], ch[3], 3); c2f(6, ch[3], ch[3], rep[2])
    inv.conv("model.7", ch[3], ch[4], 3); c2f(8, ch[4], ch[4], rep[3]); inv.sppf("model.9", ch[4], ch[4])
    c2f(12, ch[4] + ch[3], ch[3], nh); c2f(15, ch[3] + ch[2], ch[2], nh)
    inv.conv("model.16", ch[2], ch[2], 3); c2f(18, ch[2] + ch[3], ch[3], nh)
    inv.conv("model.19", ch[3], ch[3], 3); c2f(21, ch[3] + ch[4], ch[4], nh)
    feats = (ch[2], ch[3], ch[4])
    c_box, c_cls, c_mask = max(16, feats[0] // 4, reg_max * 4), max(feats[0], min(nc, 100)), max(feats[0] // 4, nm)
    for kind, mid, cout in (("cv2", c_box, 4 * reg_max), ("cv3", c_cls, nc), ("cv4", c_mask, nm)):
        for lvl, cin in enumerate(feats):
            inv.branch(f"model.22.{kind}.{lvl}", cin, mid, cout)
    inv.proto("model.22.proto", feats[0], npr, nm)
    return inv.shapes, 22


# ---- YOLO11 / YOLO11-seg / YOLO12 (yolo_2, panels, outside text) ----------------------------------------------------------------------
_V11_SCALES = {"n": (0.50, 0.25, 1024), "s": (0.50, 0.50, 1024), "m": (0.50, 1.00, 512), "l": (1.00, 1.00, 512), "x": (1.00, 1.50, 512)}


def yolo11_shapes(family: str = "11", scale: str = "l", nc: int = 1, seg: bool = False):
    depth, width, cmax = _V11_SCALES[scale]
    ch = [_divisible(min(c, cmax) * width) for c in (64, 128, 256, 512, 1024)]
    n2, n4 = max(round(2 * depth), 1), max(round(4 * depth), 1)
    big = scale in "mlx"
    residual, mlp_ratio = scale in "lx", (1.2 if scale in "lx" else 2.0)
    nm, npr, reg_max = (32 if seg else 0), _divisible(min(256, cmax) * width), 16
    inv = _Inventory()

    def c3k(name, c):                      # CSP block with two full-width 3x3 bottlenecks
        h = c // 2
        inv.conv(f"{name}.cv1", c, h, 1); inv.conv(f"{name}.cv2", c, h, 1); inv.conv(f"{name}.cv3", 2 * h, c, 1)
        for j in range(2):
            inv.bottleneck(f"{name}.m.{j}", h)

    def c3k2(i, cin, cout, n, inner_c3k, e=0.5):
        c = int(cout * e)
        inv.conv(f"model.{i}.cv1", cin, 2 * c, 1)
        inv.conv(f"model.{i}.cv2", (2 + n) * c, cout, 1)
        for j in range(n):
            if inner_c3k:
                c3k(f"model.{i}.m.{j}", c)
            else:
                inv.bottleneck(f"model.{i}.m.{j}", c, c // 2)

    def c2psa(i, c1, n):
        c = c1 // 2
        heads = max(c // 64, 1)
        key = int((c // heads) * 0.5)
        inv.conv(f"model.{i}.cv1", c1, 2 * c, 1); inv.conv(f"model.{i}.cv2", 2 * c, c1, 1)
        for j in range(n):
            p = f"model.{i}.m.{j}"
            inv.conv(f"{p}.attn.qkv", c, c + 2 * key * heads, 1); inv.conv(f"{p}.attn.proj", c, c, 1); inv.conv(f"{p}.attn.pe", c, c, 3, groups=c)
            inv.conv(f"{p}.ffn.0", c, 2 * c, 1); inv.conv(f"{p}.ffn.1", 2 * c, c, 1)

    def a2c2f(i, cin, cout, n, area_attention):
        h = cout // 2
        inv.conv(f"model.{i}.cv1", cin, h, 1); inv.conv(f"model.{i}.cv2", (1 + n) * h, cout, 1)
        if area_attention and residual:
            inv.shapes[f"model.{i}.gamma"] = (cout,)
        for j in range(n):
            if not area_attention:
                c3k(f"model.{i}.m.{j}", h)
                continue
            for b in range(2):
                p = f"model.{i}.m.{j}.{b}"
                inv.conv(f"{p}.attn.qkv", h, 3 * h, 1); inv.conv(f"{p}.attn.proj", h, h, 1); inv.conv(f"{p}.attn.pe", h, h, 7, groups=h)
                hid = int(h * mlp_ratio)
                inv.conv(f"{p}.mlp.0", h, hid, 1); inv.conv(f"{p}.mlp.1", hid, h, 1)

    inv.conv("model.0", 3, ch[0], 3); inv.conv("model.1", ch[0], ch[1], 3); c3k2(2, ch[1], ch[2], n2, big, 0.25)
    inv.conv("model.3", ch[2], ch[2], 3); c3k2(4, ch[2], ch[3], n2, big, 0.25); inv.conv("model.5", ch[3], ch[3], 3)
    if family == "11":
        c3k2(6, ch[3], ch[3], n2, True); inv.conv("model.7", ch[3], ch[4], 3); c3k2(8, ch[4], ch[4], n2, True)
        inv.sppf("model.9", ch[4], ch[4]); c2psa(10, ch[4], n2)
        c3k2(13, ch[4] + ch[3], ch[3], n2, big); c3k2(16, ch[3] + ch[3], ch[2], n2, big)
        inv.conv("model.17", ch[2], ch[2], 3); c3k2(19, ch[2] + ch[3], ch[3], n2, big)
        inv.conv("model.20", ch[3], ch[3], 3); c3k2(22, ch[3] + ch[4], ch[4], n2, True)
        head = 23
    else:
        a2c2f(6, ch[3], ch[3], n4, True); inv.conv("model.7", ch[3], ch[4], 3); a2c2f(8, ch[4], ch[4], n4, True)
        a2c2f(11, ch[4] + ch[3], ch[3], n2, False); a2c2f(14, ch[3] + ch[3], ch[2], n2, False)
        inv.conv("model.15", ch[2], ch[2], 3); a2c2f(17, ch[2] + ch[3], ch[3], n2, False)
        inv.conv("model.18", ch[3], ch[3], 3); c3k2(20, ch[3] + ch[4], ch[4], n2, True)
        head = 21
    feats = (ch[2], ch[3], ch[4])
    c_box, c_cls = max(16, feats[0] // 4, reg_max * 4), max(feats[0], min(nc, 100))
    for lvl, cin in enumerate(feats):
        inv.branch(f"model.{head}.cv2.{lvl}", cin, c_box, 4 * reg_max)
    for lvl, cin in enumerate(feats):                        # class branch: (depthwise 3x3 + 1x1) twice, then the 1x1 classifier
        p = f"model.{head}.cv3.{lvl}"
        inv.conv(f"{p}.0.0", cin, cin, 3, groups=cin); inv.conv(f"{p}.0.1", cin, c_cls, 1)
        inv.conv(f"{p}.1.0", c_cls, c_cls, 3, groups=c_cls); inv.conv(f"{p}.1.1", c_cls, c_cls, 1)
        inv.plain(f"{p}.2", c_cls, nc, 1)
    if seg:
        c_mask = max(feats[0] // 4, nm)
        for lvl, cin in enumerate(feats):
            inv.branch(f"model.{head}.cv4.{lvl}", cin, c_mask, nm)
        inv.proto(f"model.{head}.proto", feats[0], npr, nm)
    return inv.shapes, head


def seeded_detector(shapes, head: int, seed: int, class_bias: float = -2.0, class_gain: float = 1.0, box_gain: float = 1.0) -> Dict[str, torch.Tensor]:
    """fan-in scaled normal weights; residual-added branches (attention projection, MLP / FFN output, positional conv) damped; layer
    scales at 0.3; the classifier's bias at `class_bias` (logit) so only a handful of anchors clear a confidence threshold"""
    g = torch.Generator().manual_seed(seed)
    sd = OrderedDict()
    for name, shp in shapes.items():
        if name.endswith("gamma"):
            sd[name] = torch.full(shp, 0.3)
        elif len(shp) == 4:
            fan = shp[1] * shp[2] * shp[3]
            damp = 0.25 if any(t in name for t in (".attn.proj.", ".mlp.1.", ".ffn.1.", ".attn.pe.")) else 1.0
            sd[name] = torch.randn(shp, generator=g) * (damp * (1.2 if fan > 49 else 0.8) / math.sqrt(fan))
        else:
            sd[name] = torch.randn(shp, generator=g) * 0.1
    for lvl in range(3):
        sd[f"model.{head}.cv3.{lvl}.2.weight"] *= class_gain
        sd[f"model.{head}.cv3.{lvl}.2.bias"].fill_(class_bias)
        sd[f"model.{head}.cv2.{lvl}.2.weight"] *= box_gain
    return sd


# ---- RCAN (2x-AnimeSharpV4) ----------------------------------------------------------------------------------------------------------
def rcan_state_dict(n_feats=64, n_resgroups=10, n_resblocks=20, reduction=16, scale=2, unshuffle=1, seed=0, gain=0.7) -> Dict[str, torch.Tensor]:
    """the safetensors' key layout; the second conv of every RCAB and the group / trunk closers are damped (they are added to the stream)"""
    g = torch.Generator().manual_seed(seed)
    sd = OrderedDict()

    def conv(name, cout, cin, k, wgain=1.0, bias_std=0.01):
        sd[name + ".weight"] = torch.randn(cout, cin, k, k, generator=g) * (wgain / math.sqrt(cin * k * k))
        sd[name + ".bias"] = torch.randn(cout, generator=g) * bias_std

    c, cr = n_feats, max(n_feats // reduction, 1)
    conv("head.0", c, 3 * unshuffle ** 2, 3)
    for gi in range(n_resgroups):
        for b in range(n_resblocks):
            p = f"body.{gi}.body.{b}.body"
            conv(p + ".0", c, c, 3, 1.4); conv(p + ".2", c, c, 3, gain * 0.5)
            conv(p + ".3.conv_du.0", cr, c, 1, 1.0, 0.1); conv(p + ".3.conv_du.2", c, cr, 1, 1.0, 0.1)
        conv(f"body.{gi}.body.{n_resblocks}", c, c, 3, gain * 0.3)
    conv(f"body.{n_resgroups}", c, c, 3, gain * 0.5)
    total, k = scale * unshuffle, 0
    while total > 1:
        conv(f"tail.0.{k}", 4 * c, c, 3)
        k += 2
        total //= 2
    conv("tail.1", 3, c, 3, 0.6, 0.0)
    sd["tail.1.bias"] = torch.full((3,), 0.5 * 255.0)
    return sd


# ---- SAM-2.1 Hiera-L and RT-DETR-v2 R50: HF transformers' own modules on the published configurations -----------------------------------
def sam2_hiera_large_config():
    """facebook/sam2.1-hiera-large's config.json values (reference model_manager.py:203)"""
    from transformers import Sam2Config
    from transformers.models.sam2.configuration_sam2 import Sam2HieraDetConfig, Sam2VisionConfig
    trunk = Sam2HieraDetConfig(hidden_size=144, num_attention_heads=2, blocks_per_stage=[2, 6, 36, 4], embed_dim_per_stage=[144, 288, 576, 1152],
                               num_attention_heads_per_stage=[2, 4, 8, 16], window_size_per_stage=[8, 4, 16, 8], global_attention_blocks=[23, 33, 43],
                               window_positional_embedding_background_size=[7, 7])
    return Sam2Config(vision_config=Sam2VisionConfig(backbone_config=trunk, backbone_channel_list=[1152, 576, 288, 144]))


def sam2_state_dict(config, seed: int) -> Dict[str, torch.Tensor]:
    """HF `Sam2Model(config)` with seeded weights (HF's initialiser; the zero-initialised position tables re-seeded; matrices widened so
    signals survive 48 blocks)"""
    from transformers import Sam2Model
    torch.manual_seed(seed)
    m = Sam2Model(config).eval().float()
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        trunk = m.vision_encoder.backbone
        trunk.pos_embed.copy_(torch.randn(trunk.pos_embed.shape, generator=g) * 0.02)
        trunk.pos_embed_window.copy_(torch.randn(trunk.pos_embed_window.shape, generator=g) * 0.02)
        m.no_memory_embedding.copy_(torch.randn(m.no_memory_embedding.shape, generator=g) * 0.02)
        for name, p in m.named_parameters():
            if p.dim() >= 2 and "embed" not in name:
                p.mul_(2.0)
    return {k: v for k, v in m.state_dict().items()}


def sam2_shapes(config) -> Dict[str, Tuple[int, ...]]:
    from transformers import Sam2Model
    with torch.device("meta"):
        return {k: tuple(v.shape) for k, v in Sam2Model(config).state_dict().items()}


def rtdetr_r50_config():
    """the RT-DETR-v2 R50vd geometry of the reference's secondary detector (ogkalu/comic-text-and-bubble-detector, three classes): HF's
    default RTDetrV2Config IS that geometry (ResNet-50-vd, 256-wide hybrid encoder, 6 decoder layers, 300 queries); inference settings:
    anchors per input size, no denoising queries"""
    from transformers import RTDetrV2Config
    return RTDetrV2Config(num_labels=3, anchor_image_size=None, num_denoising=0)


def rtdetr_state_dict(config, seed: int) -> Dict[str, torch.Tensor]:
    """HF `RTDetrV2ForObjectDetection(config)` with seeded weights; BatchNorm statistics and the constant-initialised heads re-seeded so
    every path carries signal"""
    from transformers import RTDetrV2ForObjectDetection
    torch.manual_seed(seed)
    m = RTDetrV2ForObjectDetection(config).eval().float()
    with torch.no_grad():
        for mod in m.modules():
            if getattr(mod, "running_mean", None) is not None:
                mod.running_mean.normal_(0, 0.1); mod.running_var.uniform_(0.6, 1.4)
                mod.weight.normal_(1.0, 0.1); mod.bias.normal_(0, 0.1)
        for name, p in m.named_parameters():
            if p.dim() == 1 and "bias" in name and p.abs().sum() == 0:
                p.normal_(0, 0.05)
            if "sampling_offsets.weight" in name or "attention_weights.weight" in name or ("bbox_embed" in name and name.endswith("2.weight")) or \
               "enc_bbox_head.layers.2.weight" in name:
                p.normal_(0, 0.05)
    return {k: v for k, v in m.state_dict().items() if v.is_floating_point()}
