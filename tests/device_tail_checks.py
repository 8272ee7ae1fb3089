"""The device tail of the inpainting stage (core/image/device_tail.py, csrc/pagetail.hip) against the host arithmetic it replaces: Pillow's own
`Image.resize`, `composite_u8` and `FluxKleinInpainter._match_luminance`.  Integer parts bit for bit; the float Lab -> RGB leg within one level."""
import math

import numpy as np
import torch
from PIL import Image

from mangatranslator_amd.core.image import inpainting as ip
from mangatranslator_amd.core.image.device_tail import DeviceTail, pil_resample_tables


def _dev(lib):
    return torch.device("cpu") if lib.is_simulator else torch.device("cuda:0")


def check_resize(lib, cases, seed=0):
    dev = _dev(lib)
    tail = DeviceTail(lib, dev)
    rng = np.random.default_rng(seed)
    for (w, h, c), (nw, nh), filt in cases:
        a = rng.integers(0, 256, (h, w, c) if c > 1 else (h, w), dtype=np.uint8)
        # smooth structure as well as noise: ringing of the negative LANCZOS lobes, clipping at 0 / 255
        yy, xx = np.mgrid[0:h, 0:w]
        a = np.where(((xx // 7 + yy // 5) % 2 == 0)[..., None] if c > 1 else ((xx // 7 + yy // 5) % 2 == 0), a, (a // 32) * 255 // 7).astype(np.uint8)
        pil = Image.fromarray(a, {1: "L", 3: "RGB"}[c])
        ref = np.asarray(pil.resize((nw, nh), {"lanczos": Image.Resampling.LANCZOS, "bilinear": Image.Resampling.BILINEAR, "bicubic": Image.Resampling.BICUBIC}[filt]))
        got = tail.resize(torch.from_numpy(a.reshape(h, w, c)).to(dev), (nw, nh), filt).cpu().numpy().reshape(ref.shape)
        bad = int((got != ref).sum())
        assert bad == 0, f"{w}x{h}x{c} -> {nw}x{nh} {filt}: {bad} of {ref.size} bytes differ (max {np.abs(got.astype(int) - ref.astype(int)).max()})"


def check_composite(lib, seed=0):
    dev = _dev(lib)
    tail = DeviceTail(lib, dev)
    rng = np.random.default_rng(seed)
    # page_c 1 = an "L" page under an RGB patch (the patch is read with ITS pixel stride, its first channel is blended)
    for page_c, (ph, pw), (h, w), (x, y) in ((3, (90, 120), (40, 56), (30, 20)), (4, (64, 64), (32, 48), (16, 40)), (3, (50, 70), (30, 30), (60, 35)),
                                             (1, (64, 64), (32, 48), (16, 40)), (2, (40, 50), (20, 20), (35, 10))):
        page = rng.integers(0, 256, (ph, pw, page_c), dtype=np.uint8)
        patch = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        alpha = np.clip(rng.normal(0.5, 0.5, (h, w)), 0, 1).astype(np.float32)
        alpha[:5] = 1.0; alpha[-5:] = 0.0
        ref = ip.composite_u8(page[..., 0] if page_c == 1 else page, patch, alpha, x, y).reshape(page.shape)
        got = tail.composite(torch.from_numpy(page.copy()).to(dev), torch.from_numpy(patch).to(dev), torch.from_numpy(alpha).to(dev), x, y).cpu().numpy()
        assert np.array_equal(got, ref), f"composite on a {page_c}-channel page: {(got != ref).sum()} bytes differ"


def check_feather(lib, sizes=((61, 83), (40, 40), (97, 30)), radii=(1, 2, 3, 7, 10), seed=0):
    """the composite weight from the mask on the device against `FluxKleinInpainter._crop_alpha` (scipy's exact EDT + the float64 ramp,
    rounded to float32), the strict variant and the clip rectangle of `inpaint_mask`: EQUAL as float32 bit patterns.  Masks: blobs, thin
    lines, single pixels, pixels on the crop's border, a nearly full and an empty crop."""
    dev = _dev(lib)
    tail = DeviceTail(lib, dev)
    rng = np.random.default_rng(seed)
    n = 0
    for (h, w) in sizes:
        yy, xx = np.mgrid[0:h, 0:w]
        masks = {
            "blobs": ((xx - w * 0.3) ** 2 / 90.0 + (yy - h * 0.4) ** 2 / 40.0 < 1.0) | ((xx - w * 0.8) ** 2 + (yy - h * 0.75) ** 2 < 30.0),
            "lines and points": (yy == h // 2) & (xx % 9 < 5) | (xx == 3) & (yy % 7 == 0) | (xx == w - 1) & (yy == h - 1) | (xx == 0) & (yy == 0),
            "noise": rng.random((h, w)) < 0.004,
            "nearly full": ~((xx == w // 2) & (yy == h // 3)),
            "empty": np.zeros((h, w), bool),
        }
        for name, m in masks.items():
            m = np.ascontiguousarray(m)
            md = torch.from_numpy(m.astype(np.uint8)).to(dev)
            for R in radii:
                for strict, clip in ((False, None), (True, None), (False, (w // 5, h // 6, w - 7, h - 2)), (False, (5, 5, 5, 9))):
                    if name == "empty":
                        ref = np.zeros((h, w), np.float32)       # scipy's EDT of an all-background crop is not what the operator ever sees (it returns early)
                    else:
                        ref = ip.FluxKleinInpainter._crop_alpha(m, R)
                    if strict:
                        ref = ref * m.astype(np.float32)
                    if clip is not None:
                        keep = np.zeros_like(ref)
                        x0, y0, x1, y1 = clip
                        if x1 > x0 and y1 > y0:
                            keep[y0:y1, x0:x1] = ref[y0:y1, x0:x1]
                        ref = keep
                    got = tail.feather(md, R, strict, clip).cpu().numpy()
                    assert got.dtype == np.float32 and np.array_equal(got.view(np.uint32), ref.astype(np.float32).view(np.uint32)), \
                        f"{name} {w}x{h} R={R} strict={strict} clip={clip}: {(got != ref).sum()} weights differ (max {np.abs(got - ref).max()})"
                    n += 1
    return n


def check_luminance(lib, h=96, w=128, seed=0):
    """the luminance match of a patch that is darker and flatter than its surroundings (the case the reference corrects), and of one that
    needs no correction (returned untouched)"""
    dev = _dev(lib)
    tail = DeviceTail(lib, dev)
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    crop = np.clip(np.stack([150 + 60 * np.sin(xx / 9.0), 140 + 50 * np.cos(yy / 7.0), 120 + 40 * np.sin((xx + yy) / 11.0)], -1) + rng.normal(0, 6, (h, w, 3)), 0, 255).astype(np.uint8)
    mask = np.zeros((h, w), bool)
    mask[h // 4: 3 * h // 4, w // 4: 3 * w // 4] = True
    gen = np.clip(crop.astype(np.float32) * 0.6 + 25 + rng.normal(0, 3, (h, w, 3)), 0, 255).astype(np.uint8)      # darker, flatter, a colour cast
    gen[..., 2] = np.clip(gen[..., 2].astype(int) + 14, 0, 255).astype(np.uint8)
    inp = ip.FluxKleinInpainter.__new__(ip.FluxKleinInpainter)
    ref = np.asarray(inp._match_luminance(Image.fromarray(gen), Image.fromarray(crop), mask))
    assert not np.array_equal(ref, gen), "the test patch must need a correction"
    got = tail.match_luminance(torch.from_numpy(gen).to(dev), torch.from_numpy(crop).to(dev), torch.from_numpy(mask.astype(np.uint8)).to(dev)).cpu().numpy()
    d = np.abs(got.astype(int) - ref.astype(int))
    mse = float((d.astype(np.float64) ** 2).mean())
    psnr = 99.0 if mse == 0 else 10 * math.log10(255.0 ** 2 / mse)
    assert d.max() <= 1 and psnr >= 60.0, f"luminance match: max |diff| {d.max()}, PSNR {psnr:.1f} dB, {(d > 0).mean():.2%} of bytes differ"
    same = tail.match_luminance(torch.from_numpy(crop).to(dev), torch.from_numpy(crop).to(dev), torch.from_numpy(mask.astype(np.uint8)).to(dev))
    assert np.array_equal(same.cpu().numpy(), crop)
    return psnr, float((d > 0).mean())


class StandInPipeline:
    """deterministic stand-in with the Klein pipeline's call shape on both intakes (PIL in / PIL out, device tensor in / float tensor out):
    mirrors the image, darkens and flattens it (so the luminance match has work to do).  The real pipeline cannot serve here: its GroupNorm
    statistics are summed with fp32 atomics, two runs differ in the last bit and seeded random weights amplify that to several levels."""

    def __init__(self, lib, device):
        import types
        self.transformer = types.SimpleNamespace(lib=lib)
        self.device = torch.device(device)
        self.calls = []

    def __call__(self, image=None, height=None, width=None, output_type="pil", generator=None, **kw):
        import types
        a = image if torch.is_tensor(image) else torch.from_numpy(np.asarray(image.convert("RGB")).copy())
        assert tuple(a.shape[:2]) == (height, width)
        self.calls.append((output_type, tuple(a.shape)))
        # integer arithmetic only: the SAME bytes whether the image came in on the host or on the device (float ops differ in the last bit
        # between the two, which a LANCZOS pass and a Lab round trip then turn into a level or two)
        noise = torch.randint(0, 5, a.shape, generator=generator, dtype=torch.int32).to(a.device)
        u8 = ((a.flip(1).to(torch.int32) * 141 >> 8) + 30 + noise).clamp(0, 255).to(torch.uint8)
        if output_type == "pt":
            return types.SimpleNamespace(images=[(u8.float() / 255.0).permute(2, 0, 1).contiguous()])      # x / 255 * 255 rounds back to x
        return types.SimpleNamespace(images=[Image.fromarray(u8.cpu().numpy())])


def check_klein_operator(lib, page_hw=(300, 400), mask_box=(120, 150, 200, 260), page_mode="RGB", translucent=False):
    """`FluxKleinInpainter.inpaint_mask` with its image arithmetic on the device against the SAME operator on the host path (PIL / numpy):
    same stand-in pipeline, same seed.  Pixels outside the crop identical; inside within one level (the float Lab -> RGB leg); the
    remembered patch likewise; both LANCZOS passes (crop -> ~1 MP inference size and back), the luminance match and the composite run."""
    import threading
    import types
    from mangatranslator_amd.core.caching import get_cache
    dev = _dev(lib)
    H, W = page_hw
    rng = np.random.default_rng(4)
    yy, xx = np.mgrid[0:H, 0:W]
    page = np.clip(np.stack([140 + 70 * np.sin(xx / 13.0), 130 + 60 * np.cos(yy / 9.0), 120 + 50 * np.sin((xx + 2 * yy) / 17.0)], -1) + rng.normal(0, 5, (H, W, 3)), 0, 255).astype(np.uint8)
    if page_mode == "RGBA":
        page = np.concatenate([page, np.full((H, W, 1), 255, np.uint8)], -1)
        if translucent:          # Pillow resizes RGBA on premultiplied colours: the device path must step aside for such a crop
            page[mask_box[0] - 10: mask_box[0] + 30, mask_box[1]: mask_box[1] + 50, 3] = 90
    mask = np.zeros((H, W), bool)
    y0, x0, y1, x1 = mask_box
    mask[y0:y1, x0:x1] = True
    outs, pipes = [], []
    for on_device in (True, False):
        pipe = StandInPipeline(lib, dev)
        inp = ip.FluxKleinInpainter.__new__(ip.FluxKleinInpainter)
        inp.variant, inp.backend, inp.num_inference_steps, inp.low_vram = "4b", "sdnq", 1, False
        inp.luminance_correction, inp.upscale_small_crops, inp.verbose = True, True, False
        inp.sdcpp_cache_mode, inp.sdcpp_diffusion_quant, inp.sdcpp_text_encoder_quant = "none", "Q4_K_M", "Q4_K_XL"
        inp.manager = types.SimpleNamespace(flux_inference_lock=threading.Lock(), device=dev)
        inp.DEVICE, inp.DTYPE, inp.huggingface_token = dev, torch.bfloat16, ""
        inp.cache = get_cache()
        inp.cache.reset()
        inp.pipeline, inp._prompt_embeds = pipe, torch.zeros(4, 8)
        inp.load_models = lambda: None
        inp.device_tail = on_device
        res = inp.inpaint_mask(Image.fromarray(page, page_mode), mask, seed=7)
        assert res.mode == page_mode
        outs.append(np.asarray(res))
        pipes.append(pipe)
        x, y, w, h, _, _ = inp.region_for_mask(mask)
    if translucent:
        assert pipes[0].calls[0][0] == "pil" and np.array_equal(outs[0], outs[1]), "a translucent crop must take the host path, byte for byte"
        return 0.0, pipes[0].calls[0][1]
    assert pipes[0].calls[0][0] == "pt" and pipes[1].calls[0][0] == "pil" and pipes[0].calls[0][1] == pipes[1].calls[0][1]
    a, b = outs
    d = np.abs(a.astype(int) - b.astype(int))
    outside = np.ones((H, W), bool)
    outside[y:y + h, x:x + w] = False
    assert d[outside].max() == 0 and not np.array_equal(a, page), "pixels outside the crop changed, or nothing was inpainted"
    assert d.max() <= 1, f"device tail vs host path: max |diff| {d.max()}"
    return float((d > 0).mean()), pipes[0].calls[0][1]


def check_kontext_operator(lib, page_hw=(300, 400), mask_box=(120, 150, 200, 260)):
    """`FluxKontextInpainter.inpaint_mask`, device tail vs host path, same stand-in pipeline and seed: no Lab leg here, so the pages must be
    IDENTICAL (Pillow's LANCZOS both ways and the composite are bit-exact on the device)"""
    import threading
    import types
    from mangatranslator_amd.core.caching import get_cache
    dev = _dev(lib)
    H, W = page_hw
    rng = np.random.default_rng(5)
    yy, xx = np.mgrid[0:H, 0:W]
    page = np.clip(np.stack([140 + 70 * np.sin(xx / 13.0), 130 + 60 * np.cos(yy / 9.0), 120 + 50 * np.sin((xx + 2 * yy) / 17.0)], -1) + rng.normal(0, 5, (H, W, 3)), 0, 255).astype(np.uint8)
    mask = np.zeros((H, W), bool)
    y0, x0, y1, x1 = mask_box
    mask[y0:y1, x0:x1] = True
    outs = []
    for on_device in (True, False):
        pipe = StandInPipeline(lib, dev)
        inp = ip.FluxKontextInpainter.__new__(ip.FluxKontextInpainter)
        inp.PREFERED_KONTEXT_RESOLUTIONS = list(ip.PREFERRED_KONTEXT_RESOLUTIONS)
        inp.context_padding_ratio, inp.max_context_padding = ip.CONTEXT_PADDING_RATIO, ip.MAX_CONTEXT_PADDING
        inp.num_inference_steps, inp.guidance_scale, inp.residual_diff_threshold, inp.backend, inp.prompt = 1, 2.5, 0.15, "sdnq", "Remove all text."
        inp.sdcpp_cache_mode, inp.sdcpp_diffusion_quant, inp.sdcpp_text_encoder_quant = "none", "", ""
        inp.manager = types.SimpleNamespace(flux_inference_lock=threading.Lock(), device=dev)
        inp.DEVICE, inp.DTYPE = dev, torch.bfloat16
        inp.cache = get_cache()
        inp.cache.reset()
        inp.pipeline, inp._prompt_embeds, inp._pooled = pipe, None, None
        inp.load_models = lambda: None
        inp._prompt_kwargs = lambda: {}
        inp.device_tail = on_device
        outs.append(np.asarray(inp.inpaint_mask(Image.fromarray(page), mask, seed=3)))
    assert not np.array_equal(outs[0], page)
    assert np.array_equal(outs[0], outs[1]), f"{(outs[0] != outs[1]).sum()} bytes differ between the device tail and the host path"
