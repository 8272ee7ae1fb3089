"""CPU tier: RT-DETR-v2 graphs on the kernel simulator vs HF RTDetrV2ForObjectDetection (tiny geometry)."""
import rtdetr_checks as rc


def test_rtdetr_tiny(emu_lib):
    rc.check_raw(emu_lib, "cpu", hw=(64, 96))


def test_rtdetr_call_shape(emu_lib):
    rc.check_call_shape(emu_lib, "cpu")
