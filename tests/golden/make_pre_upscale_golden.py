"""Golden for the initial-upscale rule of the page flow: the REFERENCE `_resolve_pre_upscale_factor` / `_apply_pre_upscale_if_needed`
(core/pipeline.py:602-635).      python tests/golden/make_pre_upscale_golden.py      # rewrites tests/golden/pre_upscale.json"""
import json
import sys
import types
from pathlib import Path

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
import make_goldens as mg  # noqa: E402,F401
from core import pipeline as refp  # noqa: E402

T = types.SimpleNamespace
CASES = [None, T(enabled=False, factor=3.0), T(enabled=True, factor=None), T(enabled=True, factor=1.0), T(enabled=True, factor=1.01),
         T(enabled=True, factor=1.011), T(enabled=True, factor=2), T(enabled=True, factor=7.99), T(enabled=True, factor=8.5), T(enabled=True, factor=0.2),
         T(enabled=True, factor="2.5")]

if __name__ == "__main__":
    out = [dict(cfg=None if c is None else dict(enabled=c.enabled, factor=c.factor), factor=refp._resolve_pre_upscale_factor(c)) for c in CASES]
    calls = []
    refp.upscale_image = lambda image, factor, model_type="model", verbose=False: (calls.append((factor, model_type)), "up")[1]
    r1 = refp._apply_pre_upscale_if_needed("img", T(preprocessing=T(enabled=True, factor=2.0), output=T(image_upscale_model="model")))
    r2 = refp._apply_pre_upscale_if_needed("img", T(preprocessing=T(enabled=True, factor=3.0)))
    r3 = refp._apply_pre_upscale_if_needed("img", T(preprocessing=T(enabled=False, factor=3.0), output=T(image_upscale_model="model")))
    json.dump(dict(resolve=out, apply=[list(r1), list(r2), list(r3)], calls=[list(c) for c in calls]), open(HERE / "pre_upscale.json", "w"))
    print(out, calls)
