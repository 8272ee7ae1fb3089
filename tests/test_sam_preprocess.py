"""CPU tier: SAM-2.1's page pre-processing is the installed torch's own uint8 antialiased resize, bit for bit (VERDICT r03 missing #4:
every SAM input pixel passes through it, and the restatement was unpinned).  HF's Sam2ImageProcessorFast calls torchvision's
`resize(..., antialias=True)` on the uint8 image, which dispatches to ATen's fixed-point uint8 kernel (reference call site
core/image/detection.py:494-495); that kernel is present here, so the tap tables (`aten_aa_bilinear_tables`) and the device kernel
that applies them are compared with it directly."""
import numpy as np
import torch
import torch.nn.functional as F

from mangatranslator_amd.core.image.device_tail import aten_aa_bilinear_tables


def _torch_resize(img, oh, ow):
    t = torch.from_numpy(np.ascontiguousarray(img)).permute(2, 0, 1)[None]
    return F.interpolate(t, (oh, ow), mode="bilinear", antialias=True, align_corners=False)[0].permute(1, 2, 0).numpy()


def _numpy_pass(img, n_out, axis):
    b, taps, ksize, bits = aten_aa_bilinear_tables(img.shape[axis], n_out)
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    idx = np.clip(b[:, :1] + np.arange(ksize)[None, :], 0, src.shape[0] - 1)
    valid = np.arange(ksize)[None, :] < b[:, 1:2]
    acc = np.einsum("ok,ok...->o...", np.where(valid, taps, 0).astype(np.int64), src[idx]) + (1 << (bits - 1))
    return np.moveaxis(np.clip(acc >> bits, 0, 255).astype(np.uint8), 0, axis)


def test_tables_reproduce_the_aten_uint8_kernel_over_many_sizes():
    rng = np.random.default_rng(3)
    pairs = [((1536, 1024), (1024, 1024)), ((768, 512), (256, 256)), ((96, 64), (40, 40)), ((64, 100), (64, 64)), ((50, 70), (64, 64)), ((37, 91), (17, 23)),
             ((300, 200), (256, 256)), ((120, 260), (256, 256)), ((1, 9), (4, 4)), ((9, 1), (3, 5))]
    pairs += [((int(a), int(b)), (int(c), int(d))) for a, b, c, d in zip(rng.integers(2, 400, 50), rng.integers(2, 400, 50), rng.integers(1, 300, 50), rng.integers(1, 300, 50))]
    for (h, w), (oh, ow) in pairs:
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        img[: h // 3] = np.where(np.arange(w)[None, :, None] % 7 < 3, 255, 0)           # hard edges (a manga page's strokes)
        got = _numpy_pass(_numpy_pass(img, ow, 1), oh, 0)                            # horizontal pass first, uint8 in between
        assert np.array_equal(got, _torch_resize(img, oh, ow)), ((h, w), (oh, ow))


def test_device_preprocess_is_the_torch_resize(emu_lib):
    """the SAM pre-plan on the simulator: the uint8 image the normalisation reads equals torch's resize, hence the network input equals
    (resize / 255 - mean) / std rounded to the storage type"""
    import sam2_checks as sc
    from oracle import sam2_ref as sr
    from mangatranslator_amd.core.ml.sam2 import Sam2Hip, IMAGENET_MEAN, IMAGENET_STD
    model, cfg = sr.make_model("tiny_test", 0)
    hip = Sam2Hip(model.state_dict(), cfg, device="cpu", lib=emu_lib)
    for (h, w) in ((300, 200), (120, 260), (256, 256)):
        page = sc.make_page(h, w, 1)
        pre = hip._pre_plan(h, w)
        pre.page.copy_(torch.from_numpy(page))
        pre.run()
        S = hip._encoder().img.h
        want = (torch.from_numpy(_torch_resize(page, S, S)).float() / 255.0 - torch.tensor(IMAGENET_MEAN)) / torch.tensor(IMAGENET_STD)
        got = hip._encoder().img.t[0, :, :, :3].float()
        assert torch.equal(got, want.to(hip.tdt).float()), (h, w)
        pv, _ = sr.preprocess(page, np.zeros((1, 4), np.float32), S)
        assert torch.equal(got, pv[0].permute(1, 2, 0).to(hip.tdt).float())         # and the oracle's preprocess is the same thing
