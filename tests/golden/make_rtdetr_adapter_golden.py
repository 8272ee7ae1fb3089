"""Golden for the secondary detector's call boundary: the REFERENCE's `RTDetrYOLOAdapter` (core/ml/rtdetr_adapter.py:36-140) around HF's
own `RTDetrImageProcessor` (PIL backend here: torchvision is absent, and must NOT be stubbed for transformers to pick it) and a seeded
tiny `RTDetrV2ForObjectDetection` — what the YOLO-shaped call returns for a BGR ndarray and for a PIL page at two thresholds.  Pins the
pre/post-processing restated in oracle/rtdetr_ref.py (`predict`) and mirrored by RTDetrHip.

    python tests/golden/make_rtdetr_adapter_golden.py        # rewrites tests/golden/rtdetr_adapter.json
"""
import importlib
import importlib.machinery
import json
import sys
import types
from pathlib import Path
from unittest.mock import MagicMock

import numpy as np
import torch
from PIL import Image

HERE = Path(__file__).resolve().parent
REF = "/root/reference"
stub = MagicMock(name="cv2")
stub.__spec__ = importlib.machinery.ModuleSpec("cv2", None)
sys.modules.setdefault("cv2", stub)
for pkg in ["core", "core.ml"]:
    m = types.ModuleType(pkg)
    m.__path__ = [REF + "/" + pkg.replace(".", "/")]
    sys.modules[pkg] = m
sys.path.insert(0, REF)
sys.path.insert(0, str(HERE.parent.parent))

from transformers import RTDetrImageProcessor  # noqa: E402
from oracle import rtdetr_ref  # noqa: E402

adapter_mod = importlib.import_module("core.ml.rtdetr_adapter")
adapter_mod.cv2 = types.SimpleNamespace(cvtColor=lambda a, code: np.ascontiguousarray(a[..., ::-1]), COLOR_BGR2RGB=4)

if __name__ == "__main__":
    model, cfg = rtdetr_ref.make_model("tiny_test", seed=5)
    rtdetr_ref.spread_class_scores(model)
    names = {0: "bubble", 1: "text_bubble", 2: "text_free"}
    ad = adapter_mod.RTDetrYOLOAdapter(model, RTDetrImageProcessor(), torch.device("cpu"), names=names)
    page = (np.random.default_rng(21).random((150, 110, 3)) * 255).astype(np.uint8)          # BGR, as the operators hand it over
    out = {"page_seed": 21, "page_shape": [150, 110, 3], "model_seed": 5, "imgsz": 64, "runs": {}}
    lo = ad(page, conf=0.05, device="cpu", imgsz=64)[0].boxes.conf.tolist()
    mid = (lo[9] + lo[10]) / 2                           # a threshold inside the score range: exactly ten detections pass
    for tag, src, conf in [("bgr_lo", page, 0.05), ("bgr_mid", page, mid), ("pil_lo", Image.fromarray(np.ascontiguousarray(page[..., ::-1])), 0.05)]:
        r = ad(src, conf=conf, device="cpu", imgsz=64)[0]
        out["runs"][tag] = dict(conf=conf, xyxy=r.boxes.xyxy.tolist(), scores=r.boxes.conf.tolist(), cls=r.boxes.cls.tolist(),
                                names={str(k): v for k, v in r.names.items()})
    json.dump(out, open(HERE / "rtdetr_adapter.json", "w"))
    for k, v in out["runs"].items():
        print(k, len(v["scores"]), v["scores"][:3], v["cls"][:6])
