"""GPU tier: YOLO11 / YOLO12 graphs through the C ABI on gfx950 vs the fp32 CPU oracle at the geometry the reference runs them at
(panel detector YOLO11-L and OSB text detector YOLO12x at imgsz 640; the default bubble detector, a YOLO11-seg, at imgsz 1600)."""
import pytest

import yolo11_checks as yc
from parity_log import record

pytestmark = pytest.mark.gpu


def test_yolo11n_and_12n_small(hip_lib):
    yc.check(hip_lib, "cuda:0", "11", "n", True, h=192, w=128, imgsz=128, seed=2)
    yc.check(hip_lib, "cuda:0", "12", "n", False, h=192, w=128, imgsz=128, seed=3)


def test_yolo11l_panel_detector_geometry(hip_lib):
    be, ce = yc.check(hip_lib, "cuda:0", "11", "l", False, h=1536, w=1024, imgsz=640, seed=5, nc=4, n_det=20)
    record("yolo11.l.detect.1024x1536.imgsz640", class_abs_err=ce, **yc.stats)


def test_yolo12x_osb_text_detector_geometry(hip_lib):
    be, ce = yc.check(hip_lib, "cuda:0", "12", "x", False, h=1536, w=1024, imgsz=640, seed=6, n_det=20)
    record("yolo12.x.detect.1024x1536.imgsz640", class_abs_err=ce, **yc.stats)


def test_yolo11m_seg_bubble_detector_geometry(hip_lib):
    """`yolo_2` call: imgsz 1600 on a 1024 x 1536 page (1088 x 1600 letterbox), retina masks"""
    be, ce = yc.check(hip_lib, "cuda:0", "11", "m", True, h=1536, w=1024, imgsz=1600, seed=7, n_det=20)
    record("yolo11.m.seg.1024x1536.imgsz1600", box_err_px=be, class_abs_err=ce)


def test_detector_batcher_matches_single_calls(hip_lib):
    """cross-page batches through one detector graph (core/ml/detector_batch.py) at the product geometry — YOLO11-L and YOLO12x, 1024 x 1536 pages at
    imgsz 640 — 4 pages in one replay, and 6 pages from their own threads in batches of 4: decoded rows and boxes are the one-page call's bytes"""
    s1 = yc.check_batched(hip_lib, "cuda:0", family="11", scale="l", h=1536, w=1024, imgsz=640, pages=4, batch=4)
    s2 = yc.check_batched(hip_lib, "cuda:0", family="12", scale="x", h=1536, w=1024, imgsz=640, pages=3, batch=4, seed=1)
    s3 = yc.check_batched(hip_lib, "cuda:0", family="11", scale="n", h=1536, w=1024, imgsz=640, pages=6, batch=4, seed=2, threads=True)
    record("detector_batch.640px", yolo11l=s1, yolo12x=s2, yolo11n_threads=s3)
