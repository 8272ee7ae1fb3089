"""Golden for the page mode rule: the REFERENCE `convert_image_to_target_mode` and `resize_to_max_side`
(core/image/image_utils.py:598-676, 551-566) on small images of every mode a page can arrive in.
    python tests/golden/make_mode_goldens.py      # rewrites tests/golden/page_modes.json"""
import hashlib
import json
import sys
from pathlib import Path

import numpy as np
from PIL import Image

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
import make_goldens as mg  # noqa: E402,F401
from core.image import image_utils as refu  # noqa: E402


def sources():
    """the inputs, rebuilt identically by tests/test_page_modes.py"""
    rng = np.random.default_rng(77)
    rgba = rng.integers(0, 256, (9, 7, 4), dtype=np.uint8)
    rgba[..., 3] = np.where(rng.random((9, 7)) < 0.3, 0, np.where(rng.random((9, 7)) < 0.5, 255, rgba[..., 3]))
    out = dict(rgba=Image.fromarray(rgba, "RGBA"), rgb=Image.fromarray(rgba[..., :3].copy(), "RGB"),
               la=Image.fromarray(rgba[..., [0, 3]].copy(), "LA"), l=Image.fromarray(rgba[..., 1].copy(), "L"))
    pal = out["rgb"].convert("P", palette=Image.ADAPTIVE, colors=16)
    out["p"] = pal
    pt = pal.copy(); pt.info["transparency"] = 3
    out["p_transparent"] = pt
    out["cmyk"] = out["rgb"].convert("CMYK")
    out["one_bit"] = out["l"].convert("1")
    out["i16"] = Image.fromarray((rgba[..., 0].astype(np.uint16) * 200), "I;16")
    return out


def main():
    res = {}
    for name, im in sources().items():
        for target in ("RGB", "RGBA"):
            try:
                o = refu.convert_image_to_target_mode(im, target)
                res[f"{name}->{target}"] = dict(mode=o.mode, size=list(o.size), same_object=o is im, sha256=hashlib.sha256(o.tobytes()).hexdigest())
            except Exception as e:      # noqa: BLE001
                res[f"{name}->{target}"] = dict(error=type(e).__name__)
    im = sources()["rgb"]
    res["max_side"] = {str(t): dict(size=list(refu.resize_to_max_side(im, t).size), same_object=refu.resize_to_max_side(im, t) is im,
                                    sha256=hashlib.sha256(refu.resize_to_max_side(im, t).tobytes()).hexdigest()) for t in (9, 4, 30, 1)}
    json.dump(res, open(HERE / "page_modes.json", "w"), indent=0)
    print({k: (v.get("mode"), v.get("error")) for k, v in res.items() if "->" in k})


if __name__ == "__main__":
    main()
