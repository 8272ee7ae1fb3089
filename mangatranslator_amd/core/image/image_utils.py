"""Final upscale operator: `upscale_image`, `upscale_image_to_dimension`,
`image_to_tensor`, `tensor_to_image` with the reference's signatures (core/image/image_utils.py:351-548).
The model call goes to the RCAN graph on libmtx_hip; PIL handles the final exact-size LANCZOS.
Plus the page writer of row f2, `save_image_with_compression` (reference :59-170), and the bubble-crop half of row f4:
`process_bubble_image_cached`, `resize_to_min_side`, `pil_to_cv2` / `cv2_to_pil` (reference :678-728, :569-595, :20-55)."""
import io
import os
from pathlib import Path

import numpy as np
import torch
from PIL import Image

from ...utils.exceptions import ImageProcessingError
from ...utils.logging import log_message
from ..caching import get_cache
from ..ml.model_manager import get_model_manager


def save_image_with_compression(image: Image.Image, output_path, jpeg_quality: int = 95, png_compression: int = 2, verbose: bool = False) -> bool:
    """JPEG (alpha composited on white, quality clamped to 1..100), PNG (oxipng level 0..6 when the optimiser is installed, else
    the reference's own Pillow fallback: compress_level + optimize), lossless WEBP; unknown extensions become .png (reference :59-170)."""
    output_path = Path(output_path)
    ext = output_path.suffix.lower()
    fmt, opts = None, {}
    if ext in (".jpg", ".jpeg"):
        fmt = "JPEG"
        if image.mode in ("RGBA", "LA"):
            background = Image.new("RGB", image.size, (255, 255, 255))
            background.paste(image, mask=image.split()[-1])
            image = background
        elif image.mode != "RGB":
            image = image.convert("RGB")
        opts["quality"] = max(1, min(jpeg_quality, 100))
    elif ext == ".png":
        fmt = "PNG"
    elif ext == ".webp":
        fmt, opts = "WEBP", {"lossless": True}
    else:
        log_message(f"Warning: Unknown output extension '{ext}'. Saving as PNG.", always_print=True)
        fmt, output_path = "PNG", output_path.with_suffix(".png")
    level = min(6, max(0, int(png_compression)))
    log_message(f"Saving {fmt} image to {output_path}", verbose=verbose)
    try:
        os.makedirs(output_path.parent, exist_ok=True)
        if fmt == "PNG":
            data = None
            try:
                import oxipng                                        # optional Rust optimiser, as in the reference
                buffer = io.BytesIO()
                image.save(buffer, format="PNG")
                data = oxipng.optimize_from_memory(buffer.getvalue(), level=level, optimize_alpha=True)
            except ImportError:
                pass
            except Exception as e:                                   # oxipng.PngError
                log_message(f"oxipng optimization failed: {e}. Falling back to Pillow save.", always_print=True)
            if data is None:
                data = _native_png(image, level)
            if data is not None:
                with open(output_path, "wb") as f:
                    f.write(data)
            else:                                                    # neither optimiser: the reference's own Pillow fallback
                image.save(str(output_path), format="PNG", compress_level=level, optimize=True)
        else:
            image.save(str(output_path), format=fmt, **opts)
        return True
    except Exception as e:
        log_message(f"Error saving image to {output_path}: {e}", always_print=True)
        raise ImageProcessingError(f"Failed to save image to {output_path}") from e


_OXIPNG_TO_ZLIB = {0: 1, 1: 4, 2: 6, 3: 7, 4: 8, 5: 9, 6: 9}
PNG_THREADS = max(1, min(8, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)))


def _native_png(image: Image.Image, level: int):
    """The page as a PNG file image from the native writer (csrc/host_png.cpp: colour-type reductions, per-row filter choice, stripes
    deflated on several threads — what oxipng does for the reference, image_utils.py:140-150), or None when the kernel library cannot
    be loaded or the mode is not one of L / LA / RGB / RGBA.  Lossless: a reader decodes the page's pixels exactly; like with oxipng the
    decoded MODE follows the file (an opaque RGBA page comes back as RGB, a grey one as L)."""
    if image.mode not in ("L", "LA", "RGB", "RGBA"):
        return None
    try:
        from ...hip.lib import get_library
        lib = get_library()
    except Exception:      # noqa: BLE001 — no kernel library on this host: the caller falls back to Pillow
        return None
    a = np.ascontiguousarray(np.asarray(image))
    h, w = a.shape[:2]
    c = 1 if a.ndim == 2 else a.shape[2]
    cap = h * (w * c + 1) + h * (w * c + 1) // 500 + (1 << 20)
    buf = np.empty(cap, np.uint8)
    n = lib.mtx_host_png_encode(a.ctypes.data, w, h, c, _OXIPNG_TO_ZLIB[level], PNG_THREADS, 1, buf.ctypes.data, cap)
    if n > cap:
        buf = np.empty(n, np.uint8)
        n = lib.mtx_host_png_encode(a.ctypes.data, w, h, c, _OXIPNG_TO_ZLIB[level], PNG_THREADS, 1, buf.ctypes.data, n)
    if n <= 0:
        return None
    return buf[:n].tobytes()


def pil_to_cv2(pil_image: Image.Image) -> np.ndarray:
    """RGB(A) PIL image -> BGR(A) ndarray, other modes as they are (reference :20-36; the channel flip is all cv2 did there)"""
    a = np.array(pil_image)
    if a.ndim == 3 and a.shape[2] == 3:
        return np.ascontiguousarray(a[..., ::-1])
    if a.ndim == 3 and a.shape[2] == 4:
        return np.ascontiguousarray(a[..., [2, 1, 0, 3]])
    return a


def cv2_to_pil(cv2_image: np.ndarray) -> Image.Image:
    """BGR(A) ndarray -> RGB(A) PIL image (reference :39-55)"""
    if cv2_image.ndim == 3 and cv2_image.shape[2] == 3:
        return Image.fromarray(np.ascontiguousarray(cv2_image[..., ::-1]))
    if cv2_image.ndim == 3 and cv2_image.shape[2] == 4:
        return Image.fromarray(np.ascontiguousarray(cv2_image[..., [2, 1, 0, 3]]))
    return Image.fromarray(cv2_image)


def convert_image_to_target_mode(pil_image: Image.Image, target_mode: str, verbose: bool = False) -> Image.Image:
    """Page into its working mode (reference :598-676).  To RGB: anything carrying transparency (RGBA, LA, palette with a transparency
    entry) is flattened onto white through its alpha, not stripped of it; everything else is a plain convert.  To RGBA: plain convert."""
    if pil_image.mode == target_mode:
        return pil_image
    if target_mode == "RGBA":
        log_message(f"Converting {pil_image.mode} to RGBA", verbose=verbose)
        return pil_image.convert("RGBA")
    if target_mode != "RGB":
        return pil_image
    transparent = pil_image.mode in ("RGBA", "LA") or (pil_image.mode == "P" and "transparency" in pil_image.info)
    if not transparent:
        log_message(f"Converting {pil_image.mode} to RGB", verbose=verbose)
        return pil_image.convert("RGB")
    log_message(f"Converting {pil_image.mode} to RGB (flattening transparency)", verbose=verbose)
    try:
        alpha = (pil_image if pil_image.mode != "P" else pil_image.convert("RGBA")).getchannel("A")
        flat = Image.new("RGB", pil_image.size, (255, 255, 255))
        flat.paste(pil_image, mask=alpha)
        return flat
    except Exception as e:      # noqa: BLE001 — the reference's second and third attempts
        log_message(f"Warning: Paste failed, trying alpha_composite: {e}", verbose=verbose)
        try:
            white = Image.new("RGBA", pil_image.size, (255, 255, 255, 255))
            return Image.alpha_composite(white, pil_image if pil_image.mode == "RGBA" else pil_image.convert("RGBA")).convert("RGB")
        except Exception as e2:      # noqa: BLE001
            log_message(f"Warning: Alpha composite failed, using simple convert: {e2}", verbose=verbose)
            return pil_image.convert("RGB")


def resize_to_max_side(image: Image.Image, max_side: int, verbose: bool = False) -> Image.Image:
    """LANCZOS resize so that the longer side is exactly `max_side` (reference :551-566)"""
    w, h = image.size
    if max(w, h) == max_side:
        return image
    scale = max_side / max(w, h)
    size = (max(1, int(round(w * scale))), max(1, int(round(h * scale))))
    log_message(f"Resizing to max-side {max_side}: {w}x{h} -> {size[0]}x{size[1]}", verbose=verbose)
    return image.resize(size, Image.LANCZOS)


def resize_to_min_side(image: Image.Image, min_side: int, verbose: bool = False) -> Image.Image:
    """LANCZOS resize so that the shorter side is exactly `min_side` (reference :569-595)"""
    w, h = image.size
    if w <= 0 or h <= 0:
        msg = f"Invalid image dimensions: {w}x{h}. Cannot resize 0x0 images."
        log_message(msg, always_print=True)
        raise ImageProcessingError(msg)
    if min(w, h) == min_side:
        return image
    scale = min_side / min(w, h)
    size = (max(1, int(round(w * scale))), max(1, int(round(h * scale))))
    log_message(f"Resizing to min-side {min_side}: {w}x{h} -> {size[0]}x{size[1]}", verbose=verbose)
    return image.resize(size, Image.LANCZOS)


def image_to_tensor(image: Image.Image, device: torch.device) -> torch.Tensor:
    if image.mode != "RGB":
        image = image.convert("RGB")
    arr = np.asarray(image).astype(np.float32) / 255.0
    return torch.from_numpy(arr).permute(2, 0, 1).unsqueeze(0).to(device)


def tensor_to_image(tensor: torch.Tensor) -> Image.Image:
    arr = (tensor.squeeze(0).permute(1, 2, 0).clamp(0, 1).cpu().numpy() * 255).astype(np.uint8)
    return Image.fromarray(arr)


def _upscale_image(model, image: Image.Image, device: torch.device) -> Image.Image:
    fast = getattr(model, "upscale_u8", None)
    if fast is not None:      # uint8 page in, uint8 page out: both conversions fused into the plan
        rgb = image if image.mode == "RGB" else image.convert("RGB")
        return Image.fromarray(fast(torch.from_numpy(np.array(rgb))).cpu().numpy())
    with torch.no_grad():
        return tensor_to_image(model(image_to_tensor(image, device)))


def upscale_image_to_dimension(model, image: Image.Image, target: int, device, mode: str, model_type: str = "model",
                               verbose: bool = False) -> Image.Image:
    """Model passes until max(w, h) (mode "max") or min(w, h) (mode "min") reaches `target` (reference :377-500; the
    reference's PNG round trips between passes only free host memory and are not reproduced).  Results are remembered by the stage
    memo (core/caching.py) under (pixels, target, mode, model type), like the reference :406-419, 499."""
    if mode not in {"max", "min"}:
        raise ImageProcessingError("mode must be 'max' or 'min'")
    if image.width <= 0 or image.height <= 0:
        msg = f"Invalid image dimensions: {image.width}x{image.height}. Cannot upscale 0x0 images."
        log_message(msg, always_print=True)
        raise ImageProcessingError(msg)
    cache = get_cache()
    key = cache.get_upscale_dimension_cache_key(image, target, mode, model_type)
    remembered = cache.get_upscaled_image(key)
    if remembered is not None:
        log_message("  - Using cached upscaled image", verbose=verbose)
        return remembered
    met = (lambda w, h: max(w, h) >= target) if mode == "max" else (lambda w, h: min(w, h) >= target)
    current = image
    while not met(current.width, current.height):
        log_message(f"Upscaling from {current.width}x{current.height}...", verbose=verbose)
        nxt = _upscale_image(model, current, device)
        if nxt.width <= current.width and nxt.height <= current.height:
            raise ImageProcessingError("upscale model did not enlarge the image")          # a 1x model would loop forever
        current = nxt
        log_message(f"...to {current.width}x{current.height}", verbose=verbose)
    cache.set_upscaled_image(key, current, verbose)
    return current


def upscale_image(image: Image.Image, factor: float, model_type: str = "model", verbose: bool = False) -> Image.Image:
    if factor == 1.0:
        return image
    cache = get_cache()
    with cache.pixels_scope():            # the page is digested once for this key and the one upscale_image_to_dimension builds
        key = cache.get_upscale_cache_key(image, factor, model_type)
        remembered = cache.get_upscaled_image(key)
        if remembered is not None:
            log_message("  - Using cached upscaled image", verbose=verbose)
            return remembered
        manager = get_model_manager()
        model = manager.load_upscale_lite() if model_type == "model_lite" else manager.load_upscale()
        log_message(f"Upscaling image by {factor}x...", verbose=verbose)
        tw, th = int(image.width * factor), int(image.height * factor)
        up = upscale_image_to_dimension(model, image, max(tw, th), manager.device, "max", model_type, verbose)
    result = up.resize((tw, th), Image.LANCZOS)
    cache.set_upscaled_image(key, result)
    return result


def process_bubble_image_cached(bubble_image_pil: Image.Image, upscale_model, device, target_min_side: int = 200, mode: str = "min",
                                model_type: str = "model", verbose: bool = False) -> Image.Image:
    """One bubble crop for the OCR / translation request: model passes until the shorter side reaches `target_min_side`, then the exact
    LANCZOS fit; the finished crop is remembered in the stage memo (reference :678-728)."""
    cache = get_cache()
    with cache.pixels_scope():
        key = cache.get_bubble_processing_cache_key(bubble_image_pil, target_min_side, mode, model_type)
        remembered = cache.get_upscaled_image(key)
        if remembered is not None:
            log_message("  - Using cached bubble processing result", verbose=verbose)
            return remembered
        up = upscale_image_to_dimension(upscale_model, bubble_image_pil, target_min_side, device, mode, model_type, verbose)
    fitted = resize_to_min_side(up, target_min_side, verbose)
    cache.set_upscaled_image(key, fitted, verbose)
    return fitted


def calculate_centroid_expansion_box(cleaned_mask: np.ndarray, padding_pixels: float = 4.0, verbose: bool = False):
    """Largest centred rectangle inside a cleaned bubble that keeps `padding_pixels` clear of the outline — consumed by the reference's
    text renderer (core/text/text_renderer.py:162), which sits outside the MI355X hot path; restated here (reference
    core/image/image_utils.py:173-345) so the module's import surface is whole when it is served in place of the reference's.

    Distance to the outline = exact Euclidean transform of the mask with a one-pixel zero frame (page borders count as outline); anchor =
    centre of mass of the pixels at least `padding_pixels` deep, or the deepest pixel when the centre of mass sits in a neck (less than
    70 % of the maximum depth); the box is twice the shortest clear run (left / right, up / down) from the anchor, minus one pixel per side.
    -> ((x, y, w, h), (cx, cy)); raises ImageProcessingError when no such box exists."""
    from scipy.ndimage import distance_transform_edt
    if cleaned_mask is None or not np.any(cleaned_mask):
        raise ImageProcessingError("Invalid or empty mask provided")
    try:
        h, w = cleaned_mask.shape
        framed = np.zeros((h + 2, w + 2), bool)
        framed[1:-1, 1:-1] = np.asarray(cleaned_mask) != 0
        depth = distance_transform_edt(framed)[1:-1, 1:-1].astype(np.float32)
        safe = depth >= padding_pixels
        if not safe.any():
            log_message(f"Safe area calculation failed: padding {padding_pixels:.0f}px too large", verbose=verbose, always_print=True)
            raise ImageProcessingError("Failed to create safe area mask")
        ys, xs = np.nonzero(safe)
        cx_f, cy_f = float(xs.sum() / xs.size), float(ys.sum() / ys.size)
        deepest = float(depth.max())
        peak_y, peak_x = np.unravel_index(int(np.argmax(depth)), depth.shape)          # first maximum in raster order
        px, py = max(0, min(int(round(cx_f)), w - 1)), max(0, min(int(round(cy_f)), h - 1))
        if depth[py, px] < deepest * 0.70:
            log_message(f"Centroid in constricted region (dist={depth[py, px]:.1f} vs max={deepest:.1f}). Moving anchor to pole of inaccessibility.", verbose=verbose)
            cx_f, cy_f = float(peak_x), float(peak_y)
        cx, cy = int(round(cx_f)), int(round(cy_f))
        if not (0 <= cy < h and 0 <= cx < w and safe[cy, cx]):
            d = np.sqrt((ys - cy_f) ** 2 + (xs - cx_f) ** 2)
            k = int(np.argmin(d))
            cy, cx = int(ys[k]), int(xs[k])
            cx_f, cy_f = float(cx), float(cy)
        row, col = safe[cy], safe[:, cx]
        gaps = np.flatnonzero(~row[:cx]);  left = cx - (int(gaps.max()) if gaps.size else 0)
        gaps = np.flatnonzero(~row[cx:]);  right = int(gaps.min()) if gaps.size else w - cx
        gaps = np.flatnonzero(~col[:cy]);  up = cy - (int(gaps.max()) if gaps.size else 0)
        gaps = np.flatnonzero(~col[cy:]);  down = int(gaps.min()) if gaps.size else h - cy
        half_w, half_h = min(left, right), min(up, down)
        bw = 2 * max(0, half_w - 1 if half_w > 1 else half_w)
        bh = 2 * max(0, half_h - 1 if half_h > 1 else half_h)
        if bw <= 0 or bh <= 0:
            log_message(f"Invalid safe area dimensions: {bw:.0f}x{bh:.0f}", verbose=verbose, always_print=True)
            raise ImageProcessingError("Failed to create safe area mask")
        bx, by = int(round(cx_f - bw / 2.0)), int(round(cy_f - bh / 2.0))
        if bx >= 0 and by >= 0 and bx + bw <= w and by + bh <= h:
            log_message(f"Safe area: {bw:.0f}x{bh:.0f} at ({cx_f:.0f}, {cy_f:.0f})", verbose=verbose)
            return (bx, by, bw, bh), (cx_f, cy_f)
        log_message(f"Safe area validation failed: exceeds bounds {w}x{h}", verbose=verbose, always_print=True)
        raise ImageProcessingError("Failed to create safe area mask")
    except ImageProcessingError:
        pass
    except Exception as e:
        log_message(f"Safe area calculation failed: {e}", verbose=verbose, always_print=True)
    raise ImageProcessingError("Safe area calculation failed")
