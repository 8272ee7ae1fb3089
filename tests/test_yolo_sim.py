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


def test_result_objects_have_the_ultralytics_surface(emu_lib):
    """what the operators call on a detector result (reference core/image/detection.py:525-556, conjoined assembly): len(boxes),
    len(masks), masks[i].xy[0] (outline polygon of the largest blob), masks.data"""
    import numpy as np
    import torch
    from mangatranslator_amd.core.image.cleaning import _polygon_mask
    from mangatranslator_amd.core.image.conjoined import fallback_to_yolo_mask
    from mangatranslator_amd.core.ml.yolo import _Boxes, _Masks
    import types
    yy, xx = np.mgrid[0:40, 0:50]
    blob = ((xx - 20) / 12.0) ** 2 + ((yy - 18) / 9.0) ** 2 <= 1
    m = blob.astype(np.uint8)
    m[30:33, 40:44] = 1                                   # a smaller second blob: the outline is the largest one's
    masks = _Masks(torch.from_numpy(np.stack([m, np.zeros_like(m)])), emu_lib)
    boxes = _Boxes(torch.zeros(2, 4), torch.ones(2), torch.zeros(2))
    assert len(masks) == 2 and len(boxes) == 2
    pts = masks[0].xy[0]
    assert pts.dtype == np.float32 and pts.shape[1] == 2 and np.array_equal(_polygon_mask(pts, 40, 50) > 0, blob)
    assert masks[1].xy[0].shape == (0, 2)
    res = types.SimpleNamespace(masks=masks, boxes=boxes, orig_shape=(40, 50))
    assert fallback_to_yolo_mask(res, 0, "points") == pts.tolist()
    assert np.array_equal(fallback_to_yolo_mask(res, 0, "binary") > 0, m > 0)
    assert fallback_to_yolo_mask(res, 5, "points") is None
