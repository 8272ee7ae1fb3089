"""CPU tier: YOLOv8n-seg graph (letterbox, backbone, PAN neck, Segment head, decode, NMS, retina masks)
on the kernel simulator vs the torch fp32 oracle."""
import numpy as np

import yolo_checks as yc
from mangatranslator_amd.core.ml import yolo


def test_letterbox_params_match_oracle():
    from oracle import yolo_ref as yr
    for h, w, s in ((1536, 1024, 1600), (1150, 800, 640), (300, 900, 640), (640, 640, 640)):
        a, b = yolo.letterbox_params(h, w, s), yr.letterbox_params(h, w, s)
        assert all(a[k] == b[k] for k in ("nh", "nw", "top", "left", "H", "W"))
        assert a["H"] % 32 == 0 and a["W"] % 32 == 0
    assert yolo.letterbox_params(1536, 1024, 1600)["W"] == 1088        # [1,3,1600,1088]


def test_nms_order_and_ties():
    b = np.array([[0, 0, 10, 10], [1, 1, 11, 11], [20, 20, 30, 30], [0, 0, 10, 10]], np.float32)
    assert yolo.nms_xyxy(b, np.array([0.9, 0.8, 0.7, 0.9], np.float32), 0.5) == [0, 2]


def test_yolo_n_small(emu_lib):
    yc.check_yolo(emu_lib, "cpu", "n", 160, 96, 128, mask_tol=0.05)
