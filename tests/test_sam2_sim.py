"""CPU tier: the whole SAM-2.1 graph (pre-process, Hiera encoder, FPN neck, two-way mask decoder,
stability selection, fused upsample+threshold) on the kernel simulator vs HF transformers'
Sam2Model on CPU fp32 (the reference's own third-party implementation)."""
import numpy as np

import sam2_checks as sc
from mangatranslator_amd.core.ml.sam2 import window_order


def test_window_order_roundtrip():
    o = window_order(8, 8, 4)
    assert sorted(o.tolist()) == list(range(64))
    assert o[:16].tolist() == [0, 1, 2, 3, 8, 9, 10, 11, 16, 17, 18, 19, 24, 25, 26, 27]


def test_sam2_tiny(emu_lib):
    err, mism = sc.check_sam2(emu_lib, "cpu", "tiny_test", h=300, w=200, n_boxes=3, seed=0)
    assert err < 0.05


def test_sam2_tiny_one_box_landscape(emu_lib):
    sc.check_sam2(emu_lib, "cpu", "tiny_test", h=120, w=260, n_boxes=1, seed=3)
