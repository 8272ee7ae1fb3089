"""CPU tier: the YOLO11 / YOLO12 graphs (core/ml/yolo11.py) on the kernel simulator vs the fp32 oracle, nano scale on a small page."""
import yolo11_checks as yc


def test_yolo11n_detect(emu_lib):
    yc.check(emu_lib, "cpu", "11", "n", False)


def test_yolo11n_seg(emu_lib):
    yc.check(emu_lib, "cpu", "11", "n", True, h=192, w=128, imgsz=128, seed=2)


def test_yolo12n_detect(emu_lib):
    """A2C2f blocks with area attention (4 areas at P4): 64 x 96 letterbox -> 4 x 6 = 24 positions at P4, 6 per area"""
    yc.check(emu_lib, "cpu", "12", "n", False, seed=3)
