"""CPU tier: the YOLO11 / YOLO12 graphs (core/ml/yolo11.py) on the kernel simulator vs the fp32 oracle, nano scale on a small page."""
import yolo11_checks as yc


def test_yolo11n_detect(emu_lib):
    yc.check(emu_lib, "cpu", "11", "n", False)


def test_yolo11n_seg(emu_lib):
    yc.check(emu_lib, "cpu", "11", "n", True, h=192, w=128, imgsz=128, seed=2)


def test_yolo12n_detect(emu_lib):
    """A2C2f blocks with area attention (4 areas at P4): 64 x 96 letterbox -> 4 x 6 = 24 positions at P4, 6 per area"""
    yc.check(emu_lib, "cpu", "12", "n", False, seed=3)


def test_detector_batcher_matches_single_calls(emu_lib):
    """cross-page batches through one detector graph (core/ml/detector_batch.py): YOLO11 and YOLO12 (area attention), 3 pages in a batch of 4,
    4 pages in batches of 2 (both buffer sets in flight), and 5 pages submitted and collected from their own threads (sets reused)"""
    yc.check_batched(emu_lib, "cpu", family="11")
    yc.check_batched(emu_lib, "cpu", family="12", pages=4, batch=2, seed=1)
    yc.check_batched(emu_lib, "cpu", family="11", pages=5, batch=2, seed=2, threads=True)


def test_detector_batcher_survives_a_failed_submit(emu_lib):
    """a submit that fails after it took its slot (here: the launch of a full batch raises) gives the slot back — the next pages go through and get the
    one-page call's results"""
    import numpy as np
    import pytest
    import torch
    from mangatranslator_amd.core.ml.detector_batch import DetectorBatcher
    from mangatranslator_amd.core.ml.yolo11 import Yolo11Hip
    from oracle import yolo11_ref as yr
    net = yr.make_model("11", "n", 1, False, seed=4)
    hip = Yolo11Hip(net.state_dict(), device="cpu", lib=emu_lib)
    pages = [yc.make_page(96, 64, 9 + i) for i in range(2)]
    want = [hip(p, conf=0.05, imgsz=64)[0] for p in pages]
    bat = DetectorBatcher(hip, batch=1)
    real, state = bat._run, {"fail": True}

    def flaky(b):
        if state.pop("fail", False):
            raise RuntimeError("injected launch failure")
        return real(b)
    bat._run = flaky
    with pytest.raises(RuntimeError, match="injected"):
        bat.submit(pages[0], conf=0.05, imgsz=64)
    for p, w in zip(pages, want):          # both buffer sets are usable afterwards
        got = bat(p, conf=0.05, imgsz=64)[0]
        assert (w.boxes is None) == (got.boxes is None)
        if w.boxes is not None:
            assert torch.equal(w.boxes.xyxy, got.boxes.xyxy) and torch.equal(w.boxes.conf, got.boxes.conf)
    assert all(s.filled == 0 and not s.launched for s in bat._sets[(96, 64, 64)])
