"""GPU tier: YOLOv8-seg on MI355X through the C ABI vs the torch fp32 oracle."""
import pytest

import yolo_checks as yc
from parity_log import record

pytestmark = pytest.mark.gpu


def test_yolo_n(hip_lib):
    yc.check_yolo(hip_lib, "cuda:0", "n", 384, 256, 256, mask_tol=0.03)


def test_yolo_m_page(hip_lib):
    """YOLOv8m-seg geometry (the reference's yolo_1 detector) on a 768x512 page at imgsz 800."""
    be, mm = yc.check_yolo(hip_lib, "cuda:0", "m", 768, 512, 800, seed=1, n_det=20, mask_tol=0.03)
    print(f"yolov8m-seg: box err {be:.3f} px, worst mask mismatch {mm:.4%}")
    record("yolo.v8m.768x512.imgsz800", box_err_px=be, worst_mask_mismatch_frac=mm)


def test_yolo_m_bench_geometry(hip_lib):
    """the bench's detector call: 1024 x 1536 page letterboxed to 1088 x 1600 (imgsz 1600, reference detection.py:1337-1345)"""
    be, mm = yc.check_yolo(hip_lib, "cuda:0", "m", 1536, 1024, 1600, seed=2, n_det=20, mask_tol=0.03)
    print(f"yolov8m-seg @1088x1600: box err {be:.3f} px, worst mask mismatch {mm:.4%}")
    record("yolo.v8m.1024x1536.imgsz1600", box_err_px=be, worst_mask_mismatch_frac=mm)


def test_yolo_m_page_2048x3072(hip_lib):
    """BASELINE config 5 page: the same 1088 x 1600 network input, retina masks resized to 2048 x 3072"""
    be, mm = yc.check_yolo(hip_lib, "cuda:0", "m", 3072, 2048, 1600, seed=3, n_det=20, mask_tol=0.03)
    print(f"yolov8m-seg 2048x3072: box err {be:.3f} px, worst mask mismatch {mm:.4%}")
    record("yolo.v8m.2048x3072.imgsz1600", box_err_px=be, worst_mask_mismatch_frac=mm)
