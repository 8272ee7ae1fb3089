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
