"""GPU tier: RT-DETR-v2 through the C ABI on gfx950 vs HF RTDetrV2ForObjectDetection."""
import pytest

import rtdetr_checks as rc

pytestmark = pytest.mark.gpu


def test_rtdetr_tiny(hip_lib):
    rc.check_raw(hip_lib, "cuda:0", hw=(128, 160))


def test_rtdetr_r50_640(hip_lib):
    """the real geometry: ResNet-50-vd, d = 256, 300 queries, 6 decoder layers, 640 x 640"""
    rc.check_raw(hip_lib, "cuda:0", size="r50", hw=(640, 640), tol=4e-2)


def test_rtdetr_call_shape(hip_lib):
    rc.check_call_shape(hip_lib, "cuda:0")
