"""`upscale_image` / `upscale_image_to_dimension` operator logic (reference core/image/image_utils.py:377-548) with a stand-in 2x model."""
import numpy as np
import pytest
import torch
from PIL import Image

from mangatranslator_amd.core.image import image_utils as iu
from mangatranslator_amd.utils.exceptions import ImageProcessingError


class Nearest2x:
    calls = 0

    def __call__(self, t):
        Nearest2x.calls += 1
        return t.repeat_interleave(2, -1).repeat_interleave(2, -2)


def test_passes_until_target():
    img = Image.fromarray((np.random.default_rng(0).random((30, 50, 3)) * 255).astype(np.uint8))
    Nearest2x.calls = 0
    out = iu.upscale_image_to_dimension(Nearest2x(), img, 180, torch.device("cpu"), "max")
    assert out.size == (200, 120) and Nearest2x.calls == 2          # 50 -> 100 -> 200 >= 180
    Nearest2x.calls = 0
    out = iu.upscale_image_to_dimension(Nearest2x(), img, 100, torch.device("cpu"), "min")
    assert out.size == (200, 120) and Nearest2x.calls == 2          # min side 30 -> 60 -> 120 >= 100
    assert iu.upscale_image_to_dimension(Nearest2x(), img, 40, torch.device("cpu"), "max") is img     # already met: untouched
    assert np.array_equal(np.asarray(out)[::4, ::4], np.asarray(img))


def test_errors():
    img = Image.new("RGB", (8, 8))
    with pytest.raises(ImageProcessingError):
        iu.upscale_image_to_dimension(Nearest2x(), img, 100, torch.device("cpu"), "avg")
    with pytest.raises(ImageProcessingError):
        iu.upscale_image_to_dimension(lambda t: t, img, 100, torch.device("cpu"), "max")


def test_tensor_round_trip_truncates_like_the_reference():
    t = torch.tensor([[[[0.999, 0.5, -0.2, 1.4]]]]).repeat(1, 3, 1, 1)
    assert np.asarray(iu.tensor_to_image(t))[0, :, 0].tolist() == [254, 127, 0, 255]          # astype(uint8) truncation after clamp
    img = Image.fromarray(np.arange(48, dtype=np.uint8).reshape(4, 4, 3))
    assert torch.equal(iu.image_to_tensor(img, torch.device("cpu")) * 255, torch.from_numpy(np.asarray(img)).permute(2, 0, 1)[None].float())


def _fake_upscaler(t):
    up = t.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)
    yy = torch.arange(up.shape[2], dtype=torch.float32).view(1, 1, -1, 1)
    xx = torch.arange(up.shape[3], dtype=torch.float32).view(1, 1, 1, -1)
    return up * 0.9 + ((yy * 3 + xx * 5) % 17) / 255.0


def test_upscale_image_matches_reference_flow(monkeypatch):
    """goldens from the reference's `upscale_image` (core/image/image_utils.py:351-548) with the same stand-in 2x model: pass count,
    the u8 truncation between passes, final exact-size LANCZOS, RGBA / L inputs (tests/golden/make_goldens.py gen_upscale)"""
    import json
    import types
    from pathlib import Path
    g = Path(__file__).resolve().parent / "golden"
    gold = json.loads((g / "upscale_flow.json").read_text())
    arr = np.load(g / "upscale_flow_arrays.npz")
    mgr = types.SimpleNamespace(load_upscale=lambda *a, **k: _fake_upscaler, load_upscale_lite=lambda *a, **k: _fake_upscaler, device=torch.device("cpu"))
    monkeypatch.setattr(iu, "get_model_manager", lambda: mgr)
    for name, c in gold.items():
        a = arr[f"{name}_in"]
        img = Image.fromarray(a[..., 0] if c["mode"] == "L" else a, c["mode"])
        res = iu.upscale_image(img, c["factor"], model_type=c["model_type"])
        assert res.mode == c["out_mode"] and list(res.size) == c["out_size"], name
        assert np.array_equal(np.asarray(res), arr[f"{name}_out"]), name


@pytest.mark.parametrize("mode", ["RGB", "RGBA", "L", "LA"])
def test_saved_pages_decode_to_the_same_pixels(tmp_path, mode):
    """The reference writes PNG through oxipng (core/image/image_utils.py:140-150; a Rust optimiser that re-filters, reduces colour type
    and, with `optimize_alpha=True`, may rewrite the colour of fully transparent pixels); without that wheel the file BYTES cannot be the
    reference's.  What is pinned instead is what a reader of the file gets: every format's decoded pixels — PNG and lossless WEBP equal
    the page exactly (all channels, transparent pixels included: a superset of what oxipng guarantees), JPEG equals Pillow's own encode
    of the page flattened on white at the clamped quality."""
    rng = np.random.default_rng(11)
    bands = len(mode)
    a = rng.integers(0, 256, (37, 53, bands) if bands > 1 else (37, 53), dtype=np.uint8)
    if "A" in mode:
        a[:10, :, -1] = 0                                   # a fully transparent strip with non-trivial colour under it
        a[10:20, :, -1] = 255
    img = Image.fromarray(a, mode)
    for level in (0, 2, 6, 9):
        p = tmp_path / f"l{level}.png"
        assert iu.save_image_with_compression(img, p, png_compression=level)
        back = Image.open(p)
        assert back.mode == mode and np.array_equal(np.asarray(back), a)
    p = tmp_path / "x.webp"
    iu.save_image_with_compression(img, p)
    back = Image.open(p)
    want = np.asarray(img.convert("RGBA" if "A" in mode else "RGB"))
    got = np.asarray(back.convert("RGBA" if "A" in mode else "RGB"))
    if "A" in mode:                                       # lossless WEBP (as in the reference: lossless=True, exact off) keeps alpha and every VISIBLE pixel
        vis = want[..., 3] > 0
        assert np.array_equal(got[..., 3], want[..., 3]) and np.array_equal(got[vis], want[vis])
    else:
        assert np.array_equal(got, want)
    p = tmp_path / "x.jpg"
    iu.save_image_with_compression(img, p, jpeg_quality=300)       # clamped to 100
    flat = img
    if "A" in mode:
        flat = Image.new("RGB", img.size, (255, 255, 255))
        flat.paste(img, mask=img.split()[-1])
    elif mode != "RGB":
        flat = img.convert("RGB")
    import io
    buf = io.BytesIO()
    flat.save(buf, format="JPEG", quality=100)
    assert np.array_equal(np.asarray(Image.open(p)), np.asarray(Image.open(io.BytesIO(buf.getvalue()))))
    p = tmp_path / "x.bmp"                                          # unknown extension -> .png
    iu.save_image_with_compression(img, p)
    assert not p.exists() and np.array_equal(np.asarray(Image.open(p.with_suffix(".png"))), a)


def test_native_png_writer_reduces_like_oxipng_and_stays_lossless(tmp_path):
    """csrc/host_png.cpp behind `save_image_with_compression`: opaque alpha is dropped, R == G == B becomes grey (the decoded mode follows
    the file, as with the reference's oxipng pass), every decoded pixel equals the page; stripes of a tall image concatenate into one
    valid zlib stream (Pillow decodes it); all compression levels"""
    if iu._native_png(Image.new("L", (4, 4)), 2) is None:
        pytest.skip("kernel library not built here")
    rng = np.random.default_rng(3)
    g = rng.integers(0, 256, (700, 300), dtype=np.uint8)
    rgb = rng.integers(0, 256, (700, 300, 3), dtype=np.uint8)
    cases = {"grey RGBA, opaque": (np.dstack([g, g, g, np.full_like(g, 255)]), "RGBA", "L"),
             "grey RGB": (np.dstack([g, g, g]), "RGB", "L"),
             "colour RGBA, opaque": (np.dstack([rgb, np.full_like(g, 255)]), "RGBA", "RGB"),
             "grey RGBA, one transparent pixel": (np.dstack([g, g, g, np.where(np.arange(g.size).reshape(g.shape) == 777, 0, 255).astype(np.uint8)]), "RGBA", "LA"),
             "colour RGBA with alpha": (np.dstack([rgb, g]), "RGBA", "RGBA"), "L": (g, "L", "L"), "LA opaque": (np.dstack([g, np.full_like(g, 255)]), "LA", "L")}
    for name, (arr, mode, want_mode) in cases.items():
        img = Image.fromarray(arr, mode)
        for level in (0, 2, 6):
            p = tmp_path / "x.png"
            iu.save_image_with_compression(img, p, png_compression=level)
            back = Image.open(p)
            back.load()
            assert back.mode == want_mode, (name, level, back.mode)
            assert np.array_equal(np.asarray(back.convert(mode)), arr), (name, level)
    big = Image.fromarray(rng.integers(0, 256, (3000, 1200, 3), dtype=np.uint8).astype(np.uint8) // 64 * 64, "RGB")      # > 8 stripes' worth of rows
    iu.save_image_with_compression(big, tmp_path / "big.png")
    assert np.array_equal(np.asarray(Image.open(tmp_path / "big.png")), np.asarray(big))
