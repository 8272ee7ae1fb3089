"""Golden for the panel operator (host half): the REFERENCE `detect_panels` (core/image/detection.py:1817-1915) on canned
detector outputs.      python tests/golden/make_panel_goldens.py      # rewrites tests/golden/panels.json"""
import json
import sys
import types
from pathlib import Path

import numpy as np
import torch
from PIL import Image

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
import make_goldens as mg  # noqa: E402

CASES = dict(          # name -> (boxes, classes, names or None)
    frames_and_text=([[10.5, 20.5, 110.49, 90.51], [1.5, 2.5, 3.5, 4.5], [200.2, 10.7, 390.9, 280.1], [0.0, 0.0, 399.6, 299.4]], [0, 1, 0, 2], {0: "Frame", 1: "text", 2: "face"}),
    no_frame_class=([[10.5, 20.5, 110.49, 90.51], [1.5, 2.5, 3.5, 4.5]], [3, 1], {1: "text", 3: "panel"}),
    no_names=([[7.5, 8.5, 9.5, 10.5]], [5], None),
    empty=([], [], {0: "frame"}),
    boxes_none=(None, None, {0: "frame"}),
)


def main():
    det = mg.detection
    det.cv2 = types.SimpleNamespace(cvtColor=lambda a, code: np.ascontiguousarray(a[..., ::-1]), COLOR_RGB2BGR=4, COLOR_BGR2RGB=4)
    img = Image.fromarray(np.zeros((300, 400, 3), np.uint8))
    out = {}
    for name, (boxes, classes, names) in CASES.items():
        b = None if boxes is None else types.SimpleNamespace(xyxy=torch.tensor(boxes, dtype=torch.float32).reshape(-1, 4), cls=torch.tensor(classes, dtype=torch.float32))
        model = lambda *a, _b=b, **k: [types.SimpleNamespace(boxes=_b)]
        holder = types.SimpleNamespace(__call__=None)

        class M:
            def __call__(self, *a, **k):
                return model(*a, **k)
        m = M()
        if names is not None:
            m.names = names
        det.get_model_manager = lambda m=m: types.SimpleNamespace(load_yolo_panel=lambda verbose=False: m)
        out[name] = [list(p) for p in det.detect_panels(Path("page.png"), confidence=0.25, device="cpu", image_override=img)]

    class Boom:
        names = {0: "frame"}

        def __call__(self, *a, **k):
            raise RuntimeError("kernel fault")
    det.get_model_manager = lambda: types.SimpleNamespace(load_yolo_panel=lambda verbose=False: Boom())
    out["model_raises"] = det.detect_panels(Path("page.png"), image_override=img)

    def no_model(verbose=False):
        raise RuntimeError("download failed")
    det.get_model_manager = lambda: types.SimpleNamespace(load_yolo_panel=no_model)
    try:
        det.detect_panels(Path("page.png"), image_override=img)
        out["loader_raises"] = None
    except Exception as e:
        out["loader_raises"] = [type(e).__name__, str(e)]
    json.dump(dict(cases={k: dict(boxes=v[0], classes=v[1], names=v[2]) for k, v in CASES.items()}, results=out), open(HERE / "panels.json", "w"), indent=0)
    print(out)


if __name__ == "__main__":
    main()
