"""GPU tier: YOLOv8-seg on MI355X through the C ABI vs the torch fp32 oracle."""
import pytest

import yolo_checks as yc

pytestmark = pytest.mark.gpu


def test_yolo_n(hip_lib):
    yc.check_yolo(hip_lib, "cuda:0", "n", 384, 256, 256, mask_tol=0.03)


def test_yolo_m_page(hip_lib):
    """YOLOv8m-seg geometry (the reference's yolo_1 detector) on a 768x512 page at imgsz 800."""
    be, mm = yc.check_yolo(hip_lib, "cuda:0", "m", 768, 512, 800, seed=1, n_det=20, mask_tol=0.03)
    print(f"yolov8m-seg: box err {be:.3f} px, worst mask mismatch {mm:.4%}")
