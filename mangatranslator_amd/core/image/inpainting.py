"""FLUX.1-Kontext region inpainting — host side of SURVEY.md §8 rows a6/a7.

Operator surface of the reference's `FluxKontextInpainter` (core/image/inpainting.py:91-102, 172, 209,
636-645): same constructor arguments, `.inpaint_mask(image_pil, mask_np, seed, verbose, ocr_params,
strict_mask_clipping, composite_clip_bbox) -> PIL.Image`, `.load_models()`, `.unload_models()`;
returning the *same object* means "nothing was inpainted" (callers rely on that,
reference core/outside_text_processor.py:902-905).

The region math — mask bbox + context padding, EDT feather alpha, growth to the nearest of the 17
preferred Kontext aspect ratios, 2-px quantisation, strict / clip-bbox masking, LANCZOS round trip and
the fp32 alpha composite with uint8 truncation — is pinned against the reference by
tests/golden/kontext_*.{json,npz}.  The diffusion itself is `self.pipeline`, an object with the
diffusers call shape `pipeline(image=, width=, height=, num_inference_steps=, guidance_scale=,
generator=, output_type="pt", max_area=, prompt_embeds=, pooled_prompt_embeds=).images[0]` — the
MI355X FLUX graph from `ModelManager.load_flux_kontext_sdnq()` once that model is built; without it
`load_models()` leaves `pipeline = None` and the page is returned untouched, as the reference does.
"""
import math
from typing import Dict, Optional, Tuple

import numpy as np
import torch
from PIL import Image
from scipy.ndimage import distance_transform_edt

from ...utils.logging import log_message
from ..caching import get_cache

BLUR_SCALE_FACTOR = 0.1
MIN_BLUR_RADIUS = 1
MAX_BLUR_RADIUS = 10
FLUX_GUIDANCE_SCALE = 2.5
CONTEXT_PADDING_RATIO = 0.5
MAX_CONTEXT_PADDING = 80

PREFERRED_KONTEXT_RESOLUTIONS = [
    (672, 1568), (688, 1504), (720, 1456), (752, 1392), (800, 1328), (832, 1248), (880, 1184), (944, 1104),
    (1024, 1024), (1104, 944), (1184, 880), (1248, 832), (1328, 800), (1392, 752), (1456, 720), (1504, 688),
    (1568, 672),
]


def nearest_preferred_resolution(width: int, height: int, table=PREFERRED_KONTEXT_RESOLUTIONS) -> Tuple[int, int]:
    """Entry whose aspect ratio is closest to width/height; ties go to the smaller (w, h) tuple
    (the reference takes `min` over (|dAR|, w, h) triples)."""
    ar = width / height
    _, w, h = min((abs(ar - w / h), w, h) for (w, h) in table)
    return w, h


def feather_alpha(mask: np.ndarray, blur_radius: int) -> np.ndarray:
    """1 inside the mask, linear ramp 1 -> 0 over `blur_radius` px of Euclidean distance outside it."""
    m = mask.astype(bool)
    if blur_radius <= 0:
        return m.astype(np.float32)
    alpha = np.zeros(m.shape, np.float32)
    rows, cols = np.flatnonzero(m.any(axis=1)), np.flatnonzero(m.any(axis=0))
    if rows.size == 0:
        return alpha
    # the ramp is zero beyond blur_radius of the mask, and every mask pixel lies inside this window, so the
    # Euclidean transform of the window equals the full-page transform wherever alpha is non-zero
    g = int(blur_radius) + 2
    y0, y1 = max(0, int(rows[0]) - g), min(m.shape[0], int(rows[-1]) + 1 + g)
    x0, x1 = max(0, int(cols[0]) - g), min(m.shape[1], int(cols[-1]) + 1 + g)
    mw = m[y0:y1, x0:x1]
    d_out = distance_transform_edt(~mw)
    aw = np.zeros(mw.shape, np.float32)
    aw[mw] = 1.0
    ramp = np.clip(1.0 - d_out / blur_radius, 0.0, 1.0)
    outside = d_out > 0
    aw[outside] = ramp[outside]
    alpha[y0:y1, x0:x1] = aw
    return alpha


def _grow_axis(lo: int, hi: int, limit: int, target: int) -> Tuple[int, int]:
    """Grow [lo, hi) to `target` long inside [0, limit): pinned to an edge it already touches,
    otherwise centred (extra pixel to the far side) and shifted back inside."""
    target = min(limit, target)
    if hi == limit:
        return limit - target, limit
    if lo == 0:
        return 0, target
    new_lo = max(0, lo - (target - (hi - lo)) // 2)
    new_hi = new_lo + target
    if new_hi > limit:
        new_lo, new_hi = limit - target, limit
    return new_lo, new_hi


def mask_region(mask: np.ndarray, padding: int, aspect: Optional[float], transpose: bool = False) -> Tuple[int, int, int, int]:
    """(x, y, w, h) of the padded mask bbox grown towards `aspect` (= w/h) along one axis."""
    H, W = mask.shape
    rows, cols = np.flatnonzero(mask.any(axis=1)), np.flatnonzero(mask.any(axis=0))
    x1, x2 = max(0, int(cols[0]) - padding), min(W, int(cols[-1]) + 1 + padding)
    y1, y2 = max(0, int(rows[0]) - padding), min(H, int(rows[-1]) + 1 + padding)
    w0, h0 = x2 - x1, y2 - y1
    if aspect is None:
        aspect = nearest_preferred_resolution(w0, h0)
        aspect = aspect[0] / aspect[1]
    req_w, req_h = math.ceil(h0 * aspect), math.floor(w0 / aspect)
    nx1, nx2, ny1, ny2 = x1, x2, y1, y2
    widen, heighten = req_w > w0, req_h > h0
    if (not transpose and widen) or (transpose and not heighten and widen):
        nx1, nx2 = _grow_axis(x1, x2, W, req_w)
    elif heighten:
        ny1, ny2 = _grow_axis(y1, y2, H, req_h)
    return nx1, ny1, nx2 - nx1, ny2 - ny1


def quantize_region(x: int, y: int, w: int, h: int, img_w: int, img_h: int, quant: int = 2) -> Tuple[int, int, int, int]:
    """Snap the crop to multiples of `quant` (Python banker's rounding, as the reference's round())."""
    qx1 = max(0, min(img_w, int(round(x / quant) * quant)))
    qy1 = max(0, min(img_h, int(round(y / quant) * quant)))
    qx2 = max(qx1 + 1, min(img_w, int(round((x + w) / quant) * quant)))
    qy2 = max(qy1 + 1, min(img_h, int(round((y + h) / quant) * quant)))
    return qx1, qy1, max(1, qx2 - qx1), max(1, qy2 - qy1)


def composite_u8(page: np.ndarray, patch: np.ndarray, alpha: np.ndarray, x: int, y: int) -> np.ndarray:
    """fp32 `patch*alpha + page*(1-alpha)` on the crop, truncated back to uint8 (reference :950-968).
    A destination with more channels than the patch (RGBA page) gets an opaque source alpha."""
    out = page.copy()
    h, w = patch.shape[:2]
    h, w = max(0, min(h, page.shape[0] - y)), max(0, min(w, page.shape[1] - x))
    if h == 0 or w == 0:
        return out
    dst = page[y:y + h, x:x + w].astype(np.float32) / np.float32(255.0)
    src = patch[:h, :w].astype(np.float32) / np.float32(255.0)
    if dst.ndim == 2:
        dst, src = dst[..., None], src[..., None] if src.ndim == 2 else src
    if src.shape[-1] < dst.shape[-1]:
        src = np.concatenate([src, np.ones(src.shape[:-1] + (dst.shape[-1] - src.shape[-1],), np.float32)], -1)
    src = src[..., :dst.shape[-1]]
    a = alpha[:h, :w].astype(np.float32)[..., None]
    blended = src * a + dst * (np.float32(1.0) - a)
    out[y:y + h, x:x + w] = (blended * np.float32(255.0)).astype(np.uint8).reshape(out[y:y + h, x:x + w].shape)
    return out



def _opaque_for_device_resize(crop: Image.Image) -> bool:
    """The device resize works on RGB bytes.  Pillow resizes an image with an alpha band on PREMULTIPLIED colours (reference and host path:
    resize first, convert afterwards), so convert-then-resize equals it only while every pixel is opaque; L / RGB crops have no alpha."""
    if crop.mode in ("RGB", "L"):
        return True
    if crop.mode == "RGBA":
        return crop.getchannel("A").getextrema()[0] == 255
    return False


class FluxKontextInpainter:
    def __init__(self, device: Optional[torch.device] = None, huggingface_token: str = "", num_inference_steps: int = 8,
                 residual_diff_threshold: float = 0.15, backend: str = "nunchaku", low_vram: bool = False,
                 sdcpp_cache_mode: str = "none", sdcpp_diffusion_quant: str = "", sdcpp_text_encoder_quant: str = ""):
        from ..ml.model_manager import get_model_manager
        self.manager = get_model_manager()
        self.DEVICE = device if device is not None else self.manager.device
        self.DTYPE = self.manager.dtype
        self.huggingface_token = huggingface_token
        self.num_inference_steps = num_inference_steps
        self.residual_diff_threshold = residual_diff_threshold
        self.backend = backend.lower()
        if self.backend not in ("nunchaku", "sdnq", "sdcpp"):
            raise ValueError(f"Invalid Kontext backend '{backend}'. Must be 'nunchaku', 'sdnq', or 'sdcpp'.")
        self.low_vram = low_vram
        self.sdcpp_cache_mode = sdcpp_cache_mode
        self.sdcpp_diffusion_quant = sdcpp_diffusion_quant
        self.sdcpp_text_encoder_quant = sdcpp_text_encoder_quant
        self.PREFERED_KONTEXT_RESOLUTIONS = list(PREFERRED_KONTEXT_RESOLUTIONS)
        self.pipeline = None
        self.guidance_scale = FLUX_GUIDANCE_SCALE
        self.prompt = "Remove all text."
        self.context_padding_ratio = CONTEXT_PADDING_RATIO
        self.max_context_padding = MAX_CONTEXT_PADDING
        self._prompt_embeds = None
        self.cache = get_cache()

    # ---- model lifecycle (delegated to the manager, as in the reference) ------------------------------
    def load_models(self):
        if self.pipeline is not None:
            return
        # every backend name maps onto the one MI355X-native FLUX graph; what the reference's backends differ in on this path is the
        # first-block cache: its nunchaku loader wraps the pipeline in apply_cache_on_pipe(residual_diff_threshold=) (model_manager.py:1159-1162,
        # inpainting.py:203), its SDNQ and sd.cpp loaders run every block of every step
        self.pipeline = self.manager.load_flux_kontext_sdnq(low_vram=self.low_vram, verbose=True)
        if self.pipeline is not None:
            self.manager.set_flux_residual_diff_threshold(self.residual_diff_threshold)
            # the first-block-cache threshold travels with every call (`_cache_threshold`), not on the pipeline object the manager shares between
            # inpainter instances: a second instance with another backend / threshold must not change what this one's memo keys record

    @property
    def _cache_threshold(self) -> float:
        """residual_diff_threshold of this instance's pipeline calls: the reference turns the first-block cache on for its nunchaku backend only"""
        return float(self.residual_diff_threshold) if self.backend == "nunchaku" else 0.0

    def unload_models(self):
        self.pipeline = None
        self._prompt_embeds = None
        self._tail = None               # the DeviceTail belongs to the unloaded pipeline's device / library
        self.manager.unload_flux_kontext_sdnq_models()

    # ---- geometry -----------------------------------------------------------------------------------
    def flux_kontext_image_scale(self, image_pil: Image.Image) -> Image.Image:
        w_in, h_in = image_pil.size
        if w_in == 0 or h_in == 0:
            return image_pil
        w_opt, h_opt = nearest_preferred_resolution(w_in, h_in, self.PREFERED_KONTEXT_RESOLUTIONS)
        if (w_in, h_in) == (w_opt, h_opt):
            return image_pil
        return image_pil.resize((w_opt, h_opt), Image.Resampling.LANCZOS)

    def compute_mask_bbox_aspect_ratio(self, mask_chw, padding, blur_radius, target_ar=None, transpose=False,
                                       preferred_resolutions=None, verbose=False):
        """-> (alpha[1,h,w] tensor, x, y, w, h) — same return shape as the reference (:327-495)."""
        m = mask_chw[0, 0] if mask_chw.dim() == 4 else mask_chw[0]
        mask = m.cpu().numpy() > 0
        H, W = mask.shape
        if not mask.any():
            return torch.zeros((1, H, W), dtype=mask_chw.dtype), 0, 0, W, H
        alpha = feather_alpha(mask, blur_radius)
        x, y, w, h = mask_region(mask, padding, None if preferred_resolutions else target_ar, transpose)
        return torch.from_numpy(alpha[y:y + h, x:x + w])[None].to(mask_chw.dtype), x, y, w, h

    def region_for_mask(self, mask_np: np.ndarray, strict_mask_clipping: bool = False,
                        composite_clip_bbox: Optional[Tuple[int, int, int, int]] = None):
        """Crop rectangle and composite alpha for one mask: (alpha[h,w] f32, x, y, w, h, padding, blur)."""
        mask = np.asarray(mask_np).astype(bool)
        img_h, img_w = mask.shape
        rows, cols = np.flatnonzero(mask.any(axis=1)), np.flatnonzero(mask.any(axis=0))
        side = max(int(cols[-1]) - int(cols[0]), int(rows[-1]) - int(rows[0]))
        padding = min(int(side * self.context_padding_ratio), self.max_context_padding)
        blur = max(MIN_BLUR_RADIUS, min(int(side * BLUR_SCALE_FACTOR), MAX_BLUR_RADIUS))
        alpha_full = feather_alpha(mask, blur)
        x, y, w, h = mask_region(mask, padding, None)
        qx, qy, qw, qh = quantize_region(x, y, w, h, img_w, img_h)
        # alpha of the un-quantised crop, shifted into the quantised one (zero where the crop grew)
        alpha = np.zeros((qh, qw), np.float32)
        sx0, sy0 = max(x, qx), max(y, qy)
        sx1, sy1 = min(x + w, qx + qw), min(y + h, qy + qh)
        if sx1 > sx0 and sy1 > sy0:
            alpha[sy0 - qy:sy1 - qy, sx0 - qx:sx1 - qx] = alpha_full[sy0:sy1, sx0:sx1]
        if strict_mask_clipping:
            alpha = alpha * mask[qy:qy + qh, qx:qx + qw].astype(np.float32)
        if composite_clip_bbox is not None:
            cx1, cy1, cx2, cy2 = composite_clip_bbox
            cx1, cx2 = max(0, min(img_w, cx1)), max(0, min(img_w, cx2))
            cy1, cy2 = max(0, min(img_h, cy1)), max(0, min(img_h, cy2))
            keep = np.zeros_like(alpha)
            ax0, ax1 = max(0, cx1 - qx), min(qw, cx2 - qx)
            ay0, ay1 = max(0, cy1 - qy), min(qh, cy2 - qy)
            if ax1 > ax0 and ay1 > ay0:
                keep[ay0:ay1, ax0:ax1] = alpha[ay0:ay1, ax0:ax1]
            alpha = keep
        return alpha, qx, qy, qw, qh, padding, blur

    # ---- the operator ---------------------------------------------------------------------------------
    def inpaint_mask(self, image_pil: Image.Image, mask_np: np.ndarray, seed: int = 1, verbose: bool = False,
                     ocr_params: Optional[Dict] = None, strict_mask_clipping: bool = False,
                     composite_clip_bbox: Optional[Tuple[int, int, int, int]] = None) -> Image.Image:
        mask = np.asarray(mask_np)
        if mask.dtype != bool:
            mask = mask.astype(bool)
        if not mask.any():
            return image_pil
        alpha, x, y, w, h, padding, blur = self.region_for_mask(mask, strict_mask_clipping, composite_clip_bbox)
        log_message(f"  - Optimized bbox found at ({x}, {y}) with size {w}x{h}", verbose=verbose)
        crop = image_pil.crop((x, y, x + w, y + h))
        key = self._memo_key(crop, mask[y:y + h, x:x + w], seed, (x, y, w, h), padding, blur, ocr_params, strict_mask_clipping, composite_clip_bbox)
        patch = self.cache.get_inpainted_image(key) if key is not None else None
        if patch is not None:
            log_message("  - Using cached inpainting patch", verbose=verbose)
            return Image.fromarray(composite_u8(np.asarray(image_pil), np.asarray(patch), alpha, x, y))
        tail = self._device_tail() if _opaque_for_device_resize(crop) else None      # translucent crops: the host path (premultiplied resize)
        if tail is not None:
            # LANCZOS to the preferred Kontext resolution, the pipeline, LANCZOS back and the composite without leaving HBM
            # (core/image/device_tail.py: Pillow's resize and the composite bit for bit)
            dev = tail.device
            crop_dev = torch.from_numpy(np.array(crop if crop.mode == "RGB" else crop.convert("RGB"))).to(dev)
            inf_w, inf_h = nearest_preferred_resolution(w, h, self.PREFERED_KONTEXT_RESOLUTIONS) if w and h else (w, h)
            scaled = tail.resize(crop_dev, (inf_w, inf_h), "lanczos")
            with self.manager.flux_inference_lock:
                self.load_models()
                if self.pipeline is None:
                    log_message("Warning: Flux Kontext pipeline not available. Skipping inpainting.", always_print=True)
                    return image_pil
                with torch.inference_mode():
                    gen = torch.Generator(device="cpu").manual_seed(seed)
                    out = self.pipeline(image=scaled, width=inf_w, height=inf_h, num_inference_steps=self.num_inference_steps,
                                        guidance_scale=self.guidance_scale, generator=gen, output_type="pt",
                                        max_area=inf_w * inf_h, residual_diff_threshold=self._cache_threshold, **self._prompt_kwargs())
                    img = torch.nan_to_num(out.images[0].float(), nan=0.0, posinf=1.0, neginf=0.0).clamp_(0, 1)
                    patch_dev = img.mul(255).round().to(torch.uint8).permute(1, 2, 0).contiguous()
            patch_dev = tail.resize(patch_dev, (w, h), "lanczos")
            if key is not None:
                self.cache.set_inpainted_image(key, Image.fromarray(patch_dev.cpu().numpy()))
            page = torch.from_numpy(np.array(image_pil)).to(dev)          # a copy: the composite is in place
            if page.dim() == 2:
                page = page[..., None]
            tail.composite(page.contiguous(), patch_dev, torch.from_numpy(np.ascontiguousarray(alpha, dtype=np.float32)).to(dev), x, y)
            out_np = page.cpu().numpy()
            return Image.fromarray(out_np[..., 0] if out_np.shape[2] == 1 else out_np, image_pil.mode)
        scaled = self.flux_kontext_image_scale(crop)
        inf_w, inf_h = scaled.size
        if scaled.mode == "RGBA":
            scaled = scaled.convert("RGB")
        with self.manager.flux_inference_lock:
            self.load_models()
            if self.pipeline is None:
                log_message("Warning: Flux Kontext pipeline not available. Skipping inpainting.", always_print=True)
                return image_pil
            with torch.inference_mode():
                gen = torch.Generator(device="cpu").manual_seed(seed)
                out = self.pipeline(image=scaled, width=inf_w, height=inf_h, num_inference_steps=self.num_inference_steps,
                                    guidance_scale=self.guidance_scale, generator=gen, output_type="pt",
                                    max_area=inf_w * inf_h, residual_diff_threshold=self._cache_threshold, **self._prompt_kwargs())
                # sanitise / quantise where the tensor lives (on the GPU these are microseconds; on the host 100 ms of fp32 passes
                # over 3 MP) and download the uint8 HWC image — the same IEEE operations in the same order, so the bytes are identical
                img = torch.nan_to_num(out.images[0].float(), nan=0.0, posinf=1.0, neginf=0.0).clamp_(0, 1)
                patch = Image.fromarray(img.mul(255).round().to(torch.uint8).permute(1, 2, 0).contiguous().cpu().numpy())
        patch = patch.resize((w, h), Image.Resampling.LANCZOS)
        if key is not None:
            self.cache.set_inpainted_image(key, patch)
        page = np.asarray(image_pil)
        return Image.fromarray(composite_u8(page, np.asarray(patch), alpha, x, y))

    device_tail = True          # False: the host path (PIL / numpy), the reference's own arithmetic

    def _device_tail(self):
        """the DeviceTail of the pipeline's device, or None (host path): needs a loaded pipeline made of libmtx_hip graphs"""
        if not self.device_tail:
            return None
        t = getattr(self, "_tail", None)
        if t is not None:
            return t
        self.load_models()
        pipe = self.pipeline
        lib = getattr(getattr(pipe, "transformer", None), "lib", None)
        if pipe is None or lib is None or not hasattr(pipe, "device"):
            return None
        from .device_tail import get_device_tail
        self._tail = get_device_tail(lib, pipe.device)
        return self._tail

    def _memo_key(self, crop, mask_crop, seed, bbox, padding, blur, ocr_params, strict_mask_clipping, composite_clip_bbox):
        """Key of the crop-sized patch in the stage memo (reference :781-827): crop pixels, a <= 64x64 bilinear signature of the mask
        (robust to one-pixel jitter of the detections), sampler settings and crop geometry.  None when seed == -1 (fresh noise)."""
        if not self.cache.should_use_inpaint_cache(seed):
            return None
        params = {"bbox": tuple(int(v) for v in bbox), "padding": int(padding), "blur": int(blur), "backend": self.backend}
        if self.backend == "sdcpp":
            params.update(sdcpp_cache=self.sdcpp_cache_mode, sdcpp_diffusion_quant=self.sdcpp_diffusion_quant,
                          sdcpp_text_encoder_quant=self.sdcpp_text_encoder_quant)
        if strict_mask_clipping:
            params["strict_clip"] = True
        if composite_clip_bbox is not None:
            params["clip_bbox"] = tuple(composite_clip_bbox)
        if ocr_params:
            params.update(ocr_params)
        signature = mask_crop
        if mask_crop.size > 0:
            size = (min(64, max(4, mask_crop.shape[0])), min(64, max(4, mask_crop.shape[1])))
            small = torch.nn.functional.interpolate(torch.from_numpy(mask_crop.astype(np.float32))[None, None], size=size, mode="bilinear", align_corners=False)
            signature = (small > 0.5).numpy().astype(np.uint8)[0, 0]
        return self.cache.get_inpaint_cache_key(crop, signature, seed, self.num_inference_steps, self.residual_diff_threshold,
                                                self.guidance_scale, self.prompt, params)

    def _prompt_kwargs(self) -> dict:
        enc = getattr(self.pipeline, "encode_prompt", None)
        if enc is None:
            return {}
        if self._prompt_embeds is None:
            res = enc(prompt=self.prompt, prompt_2=None, device=self.DEVICE)
            self._prompt_embeds = (res[0], res[1])
        return {"prompt_embeds": self._prompt_embeds[0], "pooled_prompt_embeds": self._prompt_embeds[1]}


# =====================================================================================================================
# FLUX.2-Klein — the reference's DEFAULT inpainter (core/config.py:136-144; class at core/image/inpainting.py:980-1665)
# =====================================================================================================================
KLEIN_PROMPT = (
    "Remove all text, including hand-drawn Japanese sound effects and onomatopoeia. "
    "Preserve character line art, screentones, panel borders, and background details "
    "exactly as they appear. Maintain the original contrast and shading, leaving every "
    "area where text or a sound effect was completely blank."
)


def _grow_to_minimum(lo: int, hi: int, limit: int, minimum: int) -> Tuple[int, int]:
    """[lo, hi) widened to min(minimum, limit): the missing pixels split half before / rest after, each side clipped to the page,
    and what clipping took away is recovered by pinning the span to the edge it hit (reference :1131-1163, one axis)"""
    want = min(minimum, limit)
    if hi - lo >= want:
        return lo, hi
    extra = want - (hi - lo)
    lo = max(0, lo - extra // 2)
    hi = min(limit, hi + extra - extra // 2)
    if hi - lo < want:
        if lo == 0:
            hi = min(limit, want)
        else:
            lo = max(0, limit - want)
    return lo, hi


SDCPP_DIFFUSION_QUANT_DEFAULT = "Q4_K_M"              # reference utils/model_metadata.py FLUX_SDCPP_QUANT_FILES[...]["diffusion_model"]["default"]
SDCPP_KLEIN_TEXT_ENCODER_QUANT_DEFAULT = "Q4_K_XL"    # ... ["llm"]["default"] (flux_klein_4b and flux_klein_9b alike)


class FluxKleinInpainter:
    """Same constructor, attributes and operator surface as the reference class (core/image/inpainting.py:980-1068, 1070-1103,
    1350-1665).  `backend` ("sdnq" / "sdcpp") and the sd.cpp quantisation names are accepted and carried into the stage-memo key,
    but every backend is served by the one MI355X graph pair `ModelManager.load_flux_klein_4b() / _9b()` builds (core/ml/flux2.py)."""

    KLEIN_MAX_STEPS = 12
    KLEIN_DEFAULT_STEPS = 4
    KLEIN_GUIDANCE_SCALE = 1.0
    KLEIN_PROMPT = KLEIN_PROMPT
    MIN_RESOLUTION = 64
    MAX_RESOLUTION = 2048
    RESOLUTION_MULTIPLE = 16
    MAX_INFERENCE_PIXELS = 4_000_000
    KLEIN_PADDING_MULTIPLIER = 2.0

    def __init__(self, variant: str = "4b", device: Optional[torch.device] = None, huggingface_token: str = "",
                 num_inference_steps: int = 4, low_vram: bool = False, luminance_correction: bool = True,
                 upscale_small_crops: bool = True, backend: str = "sdnq", sdcpp_cache_mode: str = "none",
                 sdcpp_diffusion_quant: str = "", sdcpp_text_encoder_quant: str = "", verbose: bool = False):
        self.variant = variant.lower()
        if self.variant not in ("9b", "4b"):
            raise ValueError(f"Invalid variant '{variant}'. Must be '9b' or '4b'.")
        self.backend = backend.lower()
        if self.backend not in ("sdnq", "sdcpp"):
            raise ValueError(f"Invalid Klein backend '{backend}'. Must be 'sdnq' or 'sdcpp'.")
        from ..ml.model_manager import get_model_manager
        self.num_inference_steps = num_inference_steps
        self.low_vram = low_vram
        self.luminance_correction = luminance_correction
        self.upscale_small_crops = upscale_small_crops
        self.sdcpp_cache_mode = sdcpp_cache_mode
        # empty quant names resolve to the reference's defaults (utils/model_metadata.py:105-122: Q4_K_M for every diffusion model,
        # Q4_K_XL for the Klein text encoders) BEFORE they enter the stage-memo key, so a patch remembered by either implementation
        # is found by the other (reference :1051-1057)
        self.sdcpp_diffusion_quant = sdcpp_diffusion_quant or SDCPP_DIFFUSION_QUANT_DEFAULT
        self.sdcpp_text_encoder_quant = sdcpp_text_encoder_quant or SDCPP_KLEIN_TEXT_ENCODER_QUANT_DEFAULT
        self.verbose = verbose
        self.manager = get_model_manager()
        self.DEVICE = device if device is not None else self.manager.device
        self.DTYPE = self.manager.dtype
        self.huggingface_token = huggingface_token
        self.cache = get_cache()
        self.pipeline = None
        self._prompt_embeds = None

    # ---- model lifecycle ----------------------------------------------------------------------------------------
    def load_models(self):
        if self.pipeline is not None:
            return
        if self.huggingface_token:
            self.manager.set_flux_hf_token(self.huggingface_token)
        load = self.manager.load_flux_klein_9b if self.variant == "9b" else self.manager.load_flux_klein_4b
        self.pipeline = load(low_vram=self.low_vram, verbose=self.verbose)

    def unload_models(self):
        self.pipeline = None
        self._prompt_embeds = None
        self._tail = None
        self.manager.unload_flux_klein_models()

    # ---- geometry (reference :1126-1163, 1258-1313) ----------------------------------------------------------------
    def _quantize_dimension(self, dim: int) -> int:
        dim = max(self.MIN_RESOLUTION, min(self.MAX_RESOLUTION, dim))
        return dim // self.RESOLUTION_MULTIPLE * self.RESOLUTION_MULTIPLE

    def _expand_bounds_to_min_size(self, x1, y1, x2, y2, img_w, img_h):
        x1, x2 = _grow_to_minimum(x1, x2, img_w, self.MIN_RESOLUTION)
        y1, y2 = _grow_to_minimum(y1, y2, img_h, self.MIN_RESOLUTION)
        return x1, y1, x2, y2

    def _inference_size(self, w: int, h: int) -> Tuple[int, int]:
        """size the crop is diffused at: ~1 MP when small crops are scaled up (the default), else capped at 4 MP; both sides
        multiples of 16 within [64, 2048], shaved 16 px at a time (wider side first) while the product still exceeds 4 MP"""
        pixels = w * h
        if pixels > 0 and self.upscale_small_crops:
            scale = math.sqrt(1_048_576 / pixels)
        elif pixels > self.MAX_INFERENCE_PIXELS:
            scale = math.sqrt(self.MAX_INFERENCE_PIXELS / pixels)
        else:
            scale = 1.0
        nw, nh = self._quantize_dimension(int(w * scale)), self._quantize_dimension(int(h * scale))
        while nw * nh > self.MAX_INFERENCE_PIXELS:
            if nw >= nh and nw > self.MIN_RESOLUTION:
                nw -= self.RESOLUTION_MULTIPLE
            elif nh > self.MIN_RESOLUTION:
                nh -= self.RESOLUTION_MULTIPLE
            else:
                break
        return nw, nh

    def _prepare_image_for_inference(self, image_pil: Image.Image, verbose: bool = False):
        ow, oh = image_pil.size
        nw, nh = self._inference_size(ow, oh)
        if (nw, nh) != (ow, oh):
            log_message(f"  - Scaling {ow}x{oh} -> {nw}x{nh}", verbose=verbose)
            image_pil = image_pil.resize((nw, nh), Image.Resampling.LANCZOS)
        return image_pil, ow, oh

    def region_for_mask(self, mask: np.ndarray):
        """(x, y, w, h, padding, blur) of the crop Klein works on: mask bbox + doubled context padding, at least 64 px a side,
        sides floored to multiples of 16 and the crop slid back inside the page (reference :1390-1421)"""
        img_h, img_w = mask.shape
        rows, cols = np.flatnonzero(mask.any(axis=1)), np.flatnonzero(mask.any(axis=0))
        x_min, x_max, y_min, y_max = int(cols[0]), int(cols[-1]), int(rows[0]), int(rows[-1])
        side = max(x_max - x_min, y_max - y_min)
        padding = int(min(int(side * CONTEXT_PADDING_RATIO), MAX_CONTEXT_PADDING) * self.KLEIN_PADDING_MULTIPLIER)
        blur = max(MIN_BLUR_RADIUS, min(int(side * BLUR_SCALE_FACTOR), MAX_BLUR_RADIUS))
        x1, y1 = max(0, x_min - padding), max(0, y_min - padding)
        x2, y2 = min(img_w, x_max + 1 + padding), min(img_h, y_max + 1 + padding)
        x1, y1, x2, y2 = self._expand_bounds_to_min_size(x1, y1, x2, y2, img_w, img_h)
        w, h = min(self._quantize_dimension(x2 - x1), img_w), min(self._quantize_dimension(y2 - y1), img_h)
        if x1 + w > img_w:
            x1 = max(0, img_w - w)
        if y1 + h > img_h:
            y1 = max(0, img_h - h)
        return x1, y1, w, h, padding, blur

    @staticmethod
    def _crop_alpha(mask_crop: np.ndarray, blur: int) -> np.ndarray:
        """composite weight inside the crop: 1 on the mask, linear ramp to 0 over `blur` px of Euclidean distance (measured
        INSIDE the crop, unlike the Kontext class)"""
        if blur <= 0:
            return mask_crop.astype(np.float32)
        d_out = distance_transform_edt(~mask_crop)
        alpha = np.zeros(mask_crop.shape, np.float32)
        alpha[mask_crop] = 1.0
        ramp = np.clip(1.0 - d_out / blur, 0.0, 1.0)
        outside = d_out > 0
        alpha[outside] = ramp[outside]
        return alpha

    # ---- luminance match (reference :1165-1256) ------------------------------------------------------------------------
    def _compute_luminance_stats(self, image_np: np.ndarray, mask_np: np.ndarray, lab: Optional[np.ndarray] = None) -> Tuple[float, float]:
        """`lab`: the image's Lab conversion when the caller already has it (the match below needs it again)"""
        if not np.any(mask_np):
            return 127.5, 30.0
        from .color import rgb_to_lab_u8
        l_values = (rgb_to_lab_u8(image_np) if lab is None else lab)[:, :, 0][mask_np].astype(np.float32)
        return float(np.mean(l_values)), float(np.std(l_values)) + 1e-6

    def _match_luminance(self, generated_pil: Image.Image, original_crop_pil: Image.Image, mask_crop_np: np.ndarray,
                         verbose: bool = False) -> Image.Image:
        """affine remap of the patch's L channel (mean / std of the crop's unmasked pixels as reference, gain clamped to
        [0.5, 2]) applied on the masked pixels only, plus a shift of a / b when their context means drifted by more than 1.
        Each image is converted to Lab once (the reference converts each twice: once for the statistics, once for the remap)."""
        from .color import lab_to_rgb_u8, rgb_to_lab_u8
        context = ~mask_crop_np
        if not np.any(context) or not np.any(mask_crop_np):
            return generated_pil
        o_lab8 = rgb_to_lab_u8(np.asarray(original_crop_pil))
        g_lab8 = rgb_to_lab_u8(np.asarray(generated_pil))
        o_mean, o_std = self._compute_luminance_stats(None, context, lab=o_lab8)
        g_mean, g_std = self._compute_luminance_stats(None, context, lab=g_lab8)
        if abs(o_mean - g_mean) < 1.3 and abs(o_std - g_std) < 2.0:
            return generated_pil
        gain = max(0.5, min(2.0, o_std / g_std))
        log_message(f"  - Luminance correction: mean {g_mean:.1f}->{o_mean:.1f}, std {g_std:.1f}->{o_std:.1f} (scale={gain:.2f})", verbose=verbose)
        lab = g_lab8.astype(np.float32)
        lab[:, :, 0][mask_crop_np] = np.clip((lab[:, :, 0][mask_crop_np] - g_mean) * gain + o_mean, 0, 255)
        for ch in (1, 2):
            shift = float(np.mean(o_lab8[:, :, ch][context].astype(np.float32))) - float(np.mean(lab[:, :, ch][context]))
            if abs(shift) > 1.0:
                lab[:, :, ch][mask_crop_np] = np.clip(lab[:, :, ch][mask_crop_np] + shift, 0, 255)
        return Image.fromarray(lab_to_rgb_u8(lab.astype(np.uint8)))

    # ---- the operator ---------------------------------------------------------------------------------------------
    def inpaint_mask(self, image_pil: Image.Image, mask_np: np.ndarray, seed: int = 1, verbose: bool = False,
                     strict_mask_clipping: bool = False, composite_clip_bbox: Optional[Tuple[int, int, int, int]] = None,
                     ocr_params: Optional[Dict] = None) -> Image.Image:
        mask = np.asarray(mask_np)
        if mask.dtype != bool:
            mask = mask.astype(bool)
        if not mask.any():
            return image_pil
        log_message(f"  - Flux.2 Klein {self.variant.upper()} inpainting...", verbose=verbose)
        img_h, img_w = mask.shape
        x, y, w, h, padding, blur = self.region_for_mask(mask)
        if w <= 0 or h <= 0:
            log_message(f"  - Region has invalid size ({w}x{h}), skipping", verbose=verbose)
            return image_pil
        log_message(f"  - Processing region at ({x}, {y}) size {w}x{h}", verbose=verbose)
        crop = image_pil.crop((x, y, x + w, y + h))
        mask_crop = mask[y:y + h, x:x + w]
        key = self._memo_key(crop, mask_crop, seed, (x, y, w, h), padding, blur, ocr_params, strict_mask_clipping, composite_clip_bbox)
        patch = self.cache.get_inpainted_image(key) if key is not None else None
        if patch is not None:
            log_message("  - Using cached inpainting patch", verbose=verbose)
        clip_rect = None                                   # composite_clip_bbox in crop coordinates
        if composite_clip_bbox is not None:
            cx1, cy1, cx2, cy2 = composite_clip_bbox
            cx1, cx2 = max(0, min(img_w, cx1)), max(0, min(img_w, cx2))
            cy1, cy2 = max(0, min(img_h, cy1)), max(0, min(img_h, cy2))
            clip_rect = (max(0, cx1 - x), max(0, cy1 - y), min(w, cx2 - x), min(h, cy2 - y))
        generated_now = patch is None
        tail = self._device_tail() if patch is None and _opaque_for_device_resize(crop) else None     # translucent crops: the host path
        if tail is not None:
            # the whole chain around the pipeline stays in HBM (core/image/device_tail.py): the feather weight from the mask (exact EDT in
            # a window of the blur radius), LANCZOS to the inference size, the pipeline, LANCZOS back, the luminance match and the
            # composite — weight, resize and composite bit for bit, the Lab leg within a level
            result, patch = self._inpaint_on_device(tail, image_pil, crop, mask_crop, (blur, strict_mask_clipping, clip_rect), x, y, w, h, seed,
                                                    verbose)
            if result is None:
                return image_pil
            if key is not None:
                self.cache.set_inpainted_image(key, patch)
            return result
        alpha = self._crop_alpha(mask_crop, blur)
        if strict_mask_clipping:
            alpha = alpha * mask_crop.astype(np.float32)
        if clip_rect is not None:
            ax0, ay0, ax1, ay1 = clip_rect
            keep = np.zeros_like(alpha)
            if ax1 > ax0 and ay1 > ay0:
                keep[ay0:ay1, ax0:ax1] = alpha[ay0:ay1, ax0:ax1]
            alpha = keep
        if patch is None:
            scaled, _, _ = self._prepare_image_for_inference(crop, verbose=verbose)
            inf_w, inf_h = scaled.size
            if scaled.mode == "RGBA":
                scaled = scaled.convert("RGB")
            log_message("  - Running inference...", verbose=verbose)
            with self.manager.flux_inference_lock:
                self.load_models()
                if self.pipeline is None:
                    log_message(f"Warning: Flux Klein {self.variant.upper()} pipeline unavailable.", always_print=True)
                    return image_pil
                with torch.inference_mode():
                    # the reference seeds a generator on ITS device (:1568), so its noise comes from that backend's Philox stream; this
                    # pipeline draws the initial latents on the host (bit-stable across boxes and ranks).  Same seed -> same noise within
                    # either implementation, never across them: pixel parity with a checkpoint is judged on the SAME latents (the
                    # pipelines accept `latents=`, which the parity tests use)
                    gen = torch.Generator(device="cpu").manual_seed(seed)
                    out = self.pipeline(**self._prompt_kwargs(), image=scaled, height=inf_h, width=inf_w,
                                        guidance_scale=self.KLEIN_GUIDANCE_SCALE, num_inference_steps=self.num_inference_steps,
                                        generator=gen)
                    patch = out.images[0]
            if (inf_w, inf_h) != (w, h):
                patch = patch.resize((w, h), Image.Resampling.LANCZOS)
            if self.luminance_correction:
                patch = self._match_luminance(patch, crop, mask_crop, verbose=verbose)
        result = Image.fromarray(composite_u8(np.asarray(image_pil), np.asarray(patch), alpha, x, y))
        if generated_now and key is not None:
            self.cache.set_inpainted_image(key, patch)
        return result

    # ---- the same operator with the image arithmetic on the device ------------------------------------------------------------
    device_tail = True          # False: the host path (PIL / numpy), the reference's own arithmetic

    def _device_tail(self):
        """the DeviceTail of this inpainter's device, or None (host path): needs a pipeline that lives on a GPU (or the test simulator)
        and takes / returns device tensors, and an RGB crop"""
        if not self.device_tail:
            return None
        t = getattr(self, "_tail", None)
        if t is not None:
            return t
        self.load_models()
        pipe = self.pipeline
        lib = getattr(getattr(pipe, "transformer", None), "lib", None)
        if pipe is None or lib is None or not hasattr(pipe, "device"):
            return None
        from .device_tail import get_device_tail
        self._tail = get_device_tail(lib, pipe.device)
        return self._tail

    def _inpaint_on_device(self, tail, image_pil, crop, mask_crop, feather, x, y, w, h, seed, verbose):
        """`feather` = (blur, strict, clip rectangle in crop coordinates or None): what the composite weight is made from"""
        dev = tail.device
        mask_dev = torch.from_numpy(np.ascontiguousarray(mask_crop, dtype=np.uint8)).to(dev)
        crop_rgb = crop if crop.mode == "RGB" else crop.convert("RGB")
        crop_dev = torch.from_numpy(np.array(crop_rgb)).to(dev)
        inf_w, inf_h = self._inference_size(w, h)
        if (inf_w, inf_h) != (w, h):
            log_message(f"  - Scaling {w}x{h} -> {inf_w}x{inf_h}", verbose=verbose)
        scaled = tail.resize(crop_dev, (inf_w, inf_h), "lanczos")
        log_message("  - Running inference...", verbose=verbose)
        with self.manager.flux_inference_lock:
            self.load_models()
            if self.pipeline is None:
                log_message(f"Warning: Flux Klein {self.variant.upper()} pipeline unavailable.", always_print=True)
                return None, None
            with torch.inference_mode():
                gen = torch.Generator(device="cpu").manual_seed(seed)
                out = self.pipeline(**self._prompt_kwargs(), image=scaled, height=inf_h, width=inf_w, guidance_scale=self.KLEIN_GUIDANCE_SCALE,
                                    num_inference_steps=self.num_inference_steps, generator=gen, output_type="pt").images[0]
                patch = out.mul(255).round().to(torch.uint8).permute(1, 2, 0).contiguous()       # what the "pil" output type holds
        if (inf_w, inf_h) != (w, h):
            patch = tail.resize(patch, (w, h), "lanczos")
        if self.luminance_correction:
            patch = tail.match_luminance(patch, crop_dev, mask_dev, log=lambda m: log_message(m, verbose=verbose))
        page = torch.from_numpy(np.array(image_pil)).to(dev)          # a copy: the composite is in place
        if page.dim() == 2:
            page = page[..., None]
        blur, strict, clip_rect = feather
        tail.composite(page.contiguous(), patch, tail.feather(mask_dev, blur, strict, clip_rect), x, y)
        out_np = page.cpu().numpy()
        result = Image.fromarray(out_np[..., 0] if out_np.shape[2] == 1 else out_np, image_pil.mode)
        return result, Image.fromarray(patch.cpu().numpy())

    def _memo_key(self, crop, mask_crop, seed, bbox, padding, blur, ocr_params, strict_mask_clipping, composite_clip_bbox):
        """stage-memo key of the crop-sized patch (reference :1433-1489); None when seed == -1"""
        if not self.cache.should_use_inpaint_cache(seed):
            return None
        params = {"bbox": tuple(int(v) for v in bbox), "padding": padding, "blur": blur, "variant": self.variant,
                  "lum_corr": self.luminance_correction, "upscale_small": self.upscale_small_crops,
                  "max_pixels": self.MAX_INFERENCE_PIXELS, "min_size": (self.MIN_RESOLUTION, self.MIN_RESOLUTION), "backend": self.backend}
        if self.backend == "sdcpp":
            params.update(sdcpp_cache=self.sdcpp_cache_mode, sdcpp_diffusion_quant=self.sdcpp_diffusion_quant,
                          sdcpp_text_encoder_quant=self.sdcpp_text_encoder_quant)
        if strict_mask_clipping:
            params["strict_clip"] = True
        if composite_clip_bbox is not None:
            params["clip_bbox"] = tuple(composite_clip_bbox)
        if ocr_params:
            params.update(ocr_params)
        signature = mask_crop
        if mask_crop.size > 0:
            size = (min(64, max(4, mask_crop.shape[0])), min(64, max(4, mask_crop.shape[1])))
            small = torch.nn.functional.interpolate(torch.from_numpy(mask_crop.astype(np.float32))[None, None], size=size, mode="bilinear", align_corners=False)
            signature = (small > 0.5).numpy().astype(np.uint8)[0, 0]
        return self.cache.get_inpaint_cache_key(crop, signature, seed, self.num_inference_steps, 0.0, self.KLEIN_GUIDANCE_SCALE,
                                                self.KLEIN_PROMPT, params)

    def _prompt_kwargs(self) -> dict:
        enc = getattr(self.pipeline, "encode_prompt", None)
        if enc is None:
            return {}
        if self._prompt_embeds is None:
            self._prompt_embeds = enc(prompt=self.KLEIN_PROMPT, device=self.DEVICE)[0]
        return {"prompt_embeds": self._prompt_embeds}
