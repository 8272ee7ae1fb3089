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
