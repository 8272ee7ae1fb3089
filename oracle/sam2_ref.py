"""CPU oracle for the SAM-2.1 box-prompted mask path.  TEST INFRASTRUCTURE ONLY.

The network itself is NOT restated: the reference calls HF transformers' `Sam2Model`
(reference core/image/detection.py:494-509, loaded at core/ml/model_manager.py:982-1010), and that
library (transformers 5.15.0) is importable in this image, so the oracle runs the real third-party
implementation on CPU fp32 with seeded weights.  Two small pieces around it ARE restated because
`Sam2Processor` needs torchvision, which is absent:

  preprocess()    Sam2ImageProcessor (image_processing_sam2.py:370-380): resize to SxS with
                  antialiased bilinear on uint8 levels, /255, ImageNet mean/std; boxes scaled by
                  S/W, S/H (processing_sam2.py:175-203).  PARITY UNPINNED for the rounding of the
                  uint8 resize (torchvision's integer kernel is not available to compare).
  post_process()  post_process_masks (image_processing_sam2.py:642-647): bilinear
                  (align_corners=False) to page size, > 0.0; the reference's extra `> 0.5` on the
                  boolean result is the identity.
"""
import numpy as np
import torch
import torch.nn.functional as F

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def make_config(size: str = "tiny_test"):
    from transformers import Sam2Config
    from transformers.models.sam2.configuration_sam2 import (Sam2HieraDetConfig, Sam2MaskDecoderConfig,
                                                              Sam2PromptEncoderConfig, Sam2VisionConfig)
    if size == "hiera_large":      # facebook/sam2.1-hiera-large
        bb = Sam2HieraDetConfig(hidden_size=144, num_attention_heads=2, blocks_per_stage=[2, 6, 36, 4],
                                embed_dim_per_stage=[144, 288, 576, 1152], num_attention_heads_per_stage=[2, 4, 8, 16],
                                window_size_per_stage=[8, 4, 16, 8], global_attention_blocks=[23, 33, 43],
                                window_positional_embedding_background_size=[7, 7])
        vc = Sam2VisionConfig(backbone_config=bb, backbone_channel_list=[1152, 576, 288, 144])
        return Sam2Config(vision_config=vc)
    if size == "large_dims_test":  # Hiera-L's channel widths, head counts (head dim 72) and windows on a 256x256 input with [1, 1, 2, 1] blocks:
        bb = Sam2HieraDetConfig(hidden_size=144, num_attention_heads=2, image_size=[256, 256], blocks_per_stage=[1, 1, 2, 1],      # the real GEMM K / N sizes and
                                embed_dim_per_stage=[144, 288, 576, 1152], num_attention_heads_per_stage=[2, 4, 8, 16],           # strides at a size the simulator runs
                                window_size_per_stage=[8, 4, 16, 8], global_attention_blocks=[3],
                                window_positional_embedding_background_size=[7, 7])
        vc = Sam2VisionConfig(backbone_config=bb, backbone_channel_list=[1152, 576, 288, 144], backbone_feature_sizes=[[64, 64], [32, 32], [16, 16]])
        return Sam2Config(vision_config=vc, prompt_encoder_config=Sam2PromptEncoderConfig(image_size=256), mask_decoder_config=Sam2MaskDecoderConfig())
    if size == "tiny_test":        # 256x256 input, every block flavour present (window / q-pool / global)
        bb = Sam2HieraDetConfig(hidden_size=16, num_attention_heads=1, image_size=[256, 256], blocks_per_stage=[1, 2, 3, 2],
                                embed_dim_per_stage=[16, 32, 64, 128], num_attention_heads_per_stage=[1, 2, 4, 8],
                                window_size_per_stage=[8, 4, 16, 8], global_attention_blocks=[4])
        vc = Sam2VisionConfig(backbone_config=bb, backbone_channel_list=[128, 64, 32, 16],
                              backbone_feature_sizes=[[64, 64], [32, 32], [16, 16]])
        return Sam2Config(vision_config=vc, prompt_encoder_config=Sam2PromptEncoderConfig(image_size=256),
                          mask_decoder_config=Sam2MaskDecoderConfig())
    if size == "small_test":       # 512x512 input, head_dim 24 / 72-like odd sizes, two global blocks
        bb = Sam2HieraDetConfig(hidden_size=24, num_attention_heads=1, image_size=[512, 512], blocks_per_stage=[2, 2, 4, 2],
                                embed_dim_per_stage=[24, 48, 96, 192], num_attention_heads_per_stage=[1, 2, 4, 8],
                                window_size_per_stage=[8, 4, 16, 8], global_attention_blocks=[5, 7])
        vc = Sam2VisionConfig(backbone_config=bb, backbone_channel_list=[192, 96, 48, 24],
                              backbone_feature_sizes=[[128, 128], [64, 64], [32, 32]])
        return Sam2Config(vision_config=vc, prompt_encoder_config=Sam2PromptEncoderConfig(image_size=512),
                          mask_decoder_config=Sam2MaskDecoderConfig())
    raise ValueError(size)


def make_model(size: str = "tiny_test", seed: int = 0):
    """HF Sam2Model with seeded weights.  HF zero-initialises the positional tables and
    no_memory_embedding; they are re-seeded so those code paths are exercised by the parity tests."""
    from transformers import Sam2Model
    torch.manual_seed(seed)
    cfg = make_config(size)
    m = Sam2Model(cfg).eval().float()
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        bbm = m.vision_encoder.backbone
        bbm.pos_embed.copy_(torch.randn(bbm.pos_embed.shape, generator=g) * 0.02)
        bbm.pos_embed_window.copy_(torch.randn(bbm.pos_embed_window.shape, generator=g) * 0.02)
        m.no_memory_embedding.copy_(torch.randn(m.no_memory_embedding.shape, generator=g) * 0.02)
        for name, p in m.named_parameters():      # widen the tiny default init so signals survive 48 blocks
            if p.dim() >= 2 and "embed" not in name:
                p.mul_(2.0)
    return m, cfg


def preprocess(page_u8: np.ndarray, boxes_xyxy: np.ndarray, size: int):
    """uint8 HWC page + [N,4] boxes in page pixels -> pixel_values [1,3,S,S] fp32, input_boxes [1,N,4]."""
    h, w, _ = page_u8.shape
    # Sam2ImageProcessorFast -> torchvision resize(antialias=True) on the uint8 tensor -> ATen's native uint8 kernel (taken on every
    # AVX2 host; torch's own kernel is called here, so this step IS the upstream arithmetic, not a restatement of it)
    x = torch.from_numpy(np.ascontiguousarray(page_u8)).permute(2, 0, 1)[None]
    x = F.interpolate(x, (size, size), mode="bilinear", antialias=True, align_corners=False)
    x = x.float() / 255.0
    mean = torch.tensor(IMAGENET_MEAN).view(1, 3, 1, 1)
    std = torch.tensor(IMAGENET_STD).view(1, 3, 1, 1)
    b = torch.as_tensor(boxes_xyxy, dtype=torch.float32).clone().reshape(-1, 2, 2)
    b[..., 0] = b[..., 0] * (size / w)
    b[..., 1] = b[..., 1] * (size / h)
    return (x - mean) / std, b.reshape(1, -1, 4)


def post_process(pred_masks: torch.Tensor, h: int, w: int) -> torch.Tensor:
    """pred_masks [1,N,1,hl,wl] -> bool [N,H,W]"""
    up = F.interpolate(pred_masks[0].float(), (h, w), mode="bilinear", align_corners=False)
    return (up > 0.0)[:, 0]


@torch.no_grad()
def run(model, page_u8: np.ndarray, boxes_xyxy: np.ndarray):
    size = model.config.prompt_encoder_config.image_size
    pv, ib = preprocess(page_u8, boxes_xyxy, size)
    out = model(pixel_values=pv, input_boxes=ib, multimask_output=False)
    h, w, _ = page_u8.shape
    return dict(pred_masks=out.pred_masks, iou_scores=out.iou_scores, masks=post_process(out.pred_masks, h, w),
                pixel_values=pv, input_boxes=ib)
