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


def test_rtdetr_batcher_matches_single_calls(hip_lib):
    """RT-DETR's backbone + encoder shared by the pages of a batch at the real geometry (R50, 640 x 640, graph replays): 3 pages in one batch of 3, 5 pages
    from their own threads in batches of 3 — boxes, scores and classes are the one-page call's bytes"""
    rc.check_batched(hip_lib, "cuda:0", size="r50", imgsz=640, pages=3, batch=3, graph=True, conf=0.0)       # (seeded weights: every query passes, 300 boxes per page compared)
    rc.check_batched(hip_lib, "cuda:0", size="r50", imgsz=640, pages=5, batch=3, seed=2, threads=True, graph=True, conf=0.0)
    rc.check_batched(hip_lib, "cuda:0", pages=4, batch=2, seed=3)
