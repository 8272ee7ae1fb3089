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
