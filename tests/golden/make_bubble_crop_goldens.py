"""Golden crops for SURVEY.md §8 row f4 (bubble crops for the OCR / translation request): the REFERENCE
`prepare_bubble_images_for_translation` (core/services/translation.py:2097-2258) with `process_bubble_image_cached` /
`resize_to_min_side` (core/image/image_utils.py:569-595, 678-728) run here on a small synthetic page.  Stand-ins: the 2x model
(make_goldens.fake_upscaler), cv2.cvtColor (channel flips) and cv2.imencode (hands the array it is given to the recorder).

    python tests/golden/make_bubble_crop_goldens.py      # rewrites tests/golden/bubble_crops.json
"""
import json
import sys
import types
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
import make_goldens as mg  # noqa: E402
import make_cache_goldens  # noqa: E402,F401  (fontTools stubs + the reference UnifiedCache import)
from core.caching import UnifiedCache  # noqa: E402
from core.image import image_utils as refu  # noqa: E402
from core.services import translation as reft  # noqa: E402

RGB2BGR, RGBA2BGRA, BGR2RGB, BGRA2RGBA = 1, 2, 3, 4


def cvt(a, code):
    a = np.asarray(a)
    return np.ascontiguousarray(a[..., ::-1] if code in (RGB2BGR, BGR2RGB) else a[..., [2, 1, 0, 3]])


def scene(channels):
    """page + detections, rebuilt identically by tests/test_bubble_crops.py"""
    H, W = 160, 200
    yy, xx = np.mgrid[0:H, 0:W]
    page = np.stack([(xx * 5 + yy * 3) % 256, (xx * 2 + yy * 7) % 256, (xx * yy // 5) % 256] + ([np.full((H, W), 255)] if channels == 4 else []), -1).astype(np.uint8)

    def ellipse(cx, cy, a, b):
        return (((xx - cx) / a) ** 2 + ((yy - cy) / b) ** 2 <= 1).astype(np.uint8) * 255
    m0 = ellipse(40, 40, 30, 22)                       # reaches 4 px past its box on the left
    m1, m2 = ellipse(110, 50, 28, 30), ellipse(150, 60, 26, 32)
    m2 = np.where(m1 > 0, 0, m2).astype(np.uint8)      # conjoined pair, disjoint masks
    dets = [dict(bbox=(14, 20, 70, 60), sam_mask=m0, confidence=0.9),
            dict(bbox=(82, 20, 138, 80), sam_mask=m1, conjoined_neighbor_bboxes=[(124, 28, 176, 92)], confidence=0.8),
            dict(bbox=(124, 28, 176, 92), sam_mask=np.repeat(m2[..., None], 3, axis=2), conjoined_neighbor_bboxes=[(82, 20, 138, 80), (1, 2, 3, 4)], confidence=0.7),
            dict(bbox=(20, 100, 90, 150), confidence=0.6),                                    # no mask at all
            dict(bbox=(120, 110, 150, 140), sam_mask=np.zeros((H, W), np.uint8), confidence=0.5)]   # empty mask: the box alone
    return page, dets


def main():
    shim = types.SimpleNamespace(cvtColor=cvt, COLOR_RGB2BGR=RGB2BGR, COLOR_RGBA2BGRA=RGBA2BGRA, COLOR_BGR2RGB=BGR2RGB, COLOR_BGRA2RGBA=BGRA2RGBA)
    refu.cv2 = shim
    captured = []
    reft.cv2 = types.SimpleNamespace(imencode=lambda ext, arr: (captured.append((ext, np.asarray(arr).copy())) or True, b"x"))
    cache = UnifiedCache()
    refu.get_cache = lambda: cache
    passes = [0]

    def model(t):
        passes[0] += 1
        return mg.fake_upscaler(t)
    import hashlib
    out = {}
    for name, (channels, method, min_side, whiteout, mime) in dict(
            model=(3, "model", 120, True, "image/png"), model_again=(3, "model", 120, True, "image/png"), lite_no_whiteout=(3, "model_lite", 90, False, "image/jpeg"),
            lanczos=(4, "lanczos", 100, True, "image/png"), none=(4, "none", 100, True, "image/jpeg")).items():
        page, dets = scene(channels)
        captured.clear()
        n0 = passes[0]
        res = reft.prepare_bubble_images_for_translation(dets, page, model, torch.device("cpu"), mime, min_side, method, whiteout)
        assert len(res) == len(dets) == len(captured)
        out[name] = dict(channels=channels, method=method, min_side=min_side, whiteout=whiteout, mime=mime, passes=passes[0] - n0,
                         ext=[e for e, _ in captured], shapes=[list(a.shape) for _, a in captured],
                         keys=[sorted(k for k in r if k not in ("sam_mask",)) for r in res], memo=cache.get_cache_stats()["upscale"],
                         sha256=[hashlib.sha256(a.tobytes()).hexdigest() for _, a in captured])      # of the BGR(A) array handed to the encoder
    json.dump(out, open(HERE / "bubble_crops.json", "w"), indent=0)
    print({k: (v["passes"], v["shapes"], v["memo"]) for k, v in out.items()})


if __name__ == "__main__":
    main()
