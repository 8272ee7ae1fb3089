"""CPU oracle of the RT-DETR-v2 secondary detector — TEST INFRASTRUCTURE ONLY.
It IS the upstream implementation the reference depends on: HF transformers `RTDetrV2ForObjectDetection`
(reference core/ml/rtdetr_adapter.py:61-113), with seeded weights (no checkpoint offline); pre/post-processing restated from
`RTDetrImageProcessor` (resize BILINEAR to imgsz, 1/255; sigmoid -> top-k over queries x classes -> cxcywh to xyxy, scaled)."""
import numpy as np
import torch
from PIL import Image


def make_config(size="tiny_test"):
    from transformers import RTDetrV2Config
    from transformers.models.rt_detr.configuration_rt_detr_resnet import RTDetrResNetConfig
    if size == "r50":          # the HF default = the geometry of RT-DETR-v2 R50 checkpoints
        return RTDetrV2Config(num_labels=3, anchor_image_size=None, num_denoising=0)
    bb = RTDetrResNetConfig(embedding_size=16, hidden_sizes=[32, 64, 128, 256], depths=[1, 2, 1, 1], layer_type="bottleneck",
                            out_features=["stage2", "stage3", "stage4"])
    return RTDetrV2Config(backbone_config=bb, d_model=64, encoder_hidden_dim=64, encoder_in_channels=[64, 128, 256], encoder_ffn_dim=128,
                          encoder_attention_heads=4, decoder_ffn_dim=128, decoder_attention_heads=4, decoder_layers=2, num_queries=30,
                          decoder_n_points=4, num_labels=3, decoder_in_channels=[64, 64, 64], anchor_image_size=None, num_denoising=0)


def make_model(size="tiny_test", seed=0):
    from transformers import RTDetrV2ForObjectDetection
    torch.manual_seed(seed)
    cfg = make_config(size)
    m = RTDetrV2ForObjectDetection(cfg).eval().float()
    with torch.no_grad():       # HF initialises BN statistics to (0, 1) and several heads to constants: re-seed so every path matters
        for name, mod in m.named_modules():
            if hasattr(mod, "running_mean") and mod.running_mean is not None:
                mod.running_mean.normal_(0, 0.1); mod.running_var.uniform_(0.6, 1.4)
                mod.weight.normal_(1.0, 0.1); mod.bias.normal_(0, 0.1)
        for name, p in m.named_parameters():
            if p.dim() == 1 and ("bias" in name) and p.abs().sum() == 0:
                p.normal_(0, 0.05)
            if "sampling_offsets.weight" in name or "attention_weights.weight" in name or ("bbox_embed" in name and name.endswith("2.weight")) or \
               ("enc_bbox_head.layers.2.weight" in name):
                p.normal_(0, 0.05)
    return m, cfg


@torch.no_grad()
def spread_class_scores(model, seed=77, std=0.6):
    """seeded class heads are nearly constant over the queries (every score 0.954...): widen them so thresholds and top-k order matter"""
    g = torch.Generator().manual_seed(seed)
    for name, p in model.named_parameters():
        if "class_embed" in name or "enc_score_head" in name:
            p.copy_(torch.randn(p.shape, generator=g) * std)
    return model


@torch.no_grad()
def run(model, img_u8_resized: np.ndarray):
    """resized RGB uint8 [H, W, 3] -> (logits [Q, C], boxes cxcywh [Q, 4])"""
    x = torch.from_numpy(img_u8_resized).permute(2, 0, 1)[None].float() / 255.0
    out = model(pixel_values=x)
    return out.logits[0], out.pred_boxes[0]


def predict(model, pil: Image.Image, conf=0.35, imgsz=640):
    ow, oh = pil.size
    img = np.asarray(pil.convert("RGB").resize((imgsz, imgsz), resample=Image.Resampling.BILINEAR))
    logits, boxes = run(model, img)
    nc = logits.shape[-1]
    scores = logits.sigmoid()
    top_s, idx = scores.flatten().topk(min(logits.shape[0], scores.numel()))
    labels, qi = idx % nc, idx // nc
    cx, cy, w, h = boxes[qi].unbind(-1)
    xyxy = torch.stack([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], -1) * torch.tensor([ow, oh, ow, oh], dtype=torch.float32)
    keep = top_s > conf
    return xyxy[keep], top_s[keep], labels[keep]
