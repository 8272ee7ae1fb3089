"""diagnostic (GPU): per-page output digests of the page stack run (a) page by page without the harness, (b) harness front_workers=1,
(c) front_workers=2 on two instance sets (twice), (d) front_workers=2 with every front half on instance set 0"""
import contextlib
import hashlib
import sys
import tempfile
import types
from pathlib import Path

import numpy as np
import torch
from PIL import Image

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from oracle import rtdetr_ref, sam2_ref, yolo_ref                      # noqa: E402  (seeded weights only)
from mangatranslator_amd.core import pipeline                          # noqa: E402
from mangatranslator_amd.core.ml import model_manager as mm            # noqa: E402
from mangatranslator_amd.core.ml.rtdetr import RTDetrHip               # noqa: E402
from mangatranslator_amd.core.ml.sam2 import Sam2Hip                   # noqa: E402
from mangatranslator_amd.core.ml.yolo import YoloSegHip                # noqa: E402
from mangatranslator_amd.hip.lib import get_library                    # noqa: E402
from mangatranslator_amd.utils.synthetic_pages import make_page        # noqa: E402
from mangatranslator_amd.utils import logging as mlog                  # noqa: E402
import test_page_vision_gpu as tpg                                     # noqa: E402

lib = get_library(); lib.init(0)
dev = torch.device("cuda:0")
mgr = mm.get_model_manager(); mgr.device = dev
ynet = yolo_ref.make_model("n", 1, seed=3)
with torch.no_grad():
    for l in range(3):
        ynet.model[22].cv3[l][2].weight.mul_(0.05); ynet.model[22].cv3[l][2].bias.fill_(-1.0)
        ynet.model[22].cv2[l][2].weight.mul_(0.1)
rmodel, rcfg = rtdetr_ref.make_model("tiny_test", seed=5)
smodel, scfg = sam2_ref.make_model("tiny_test", seed=2)
sets = []
for r in range(2):
    yolo = YoloSegHip(ynet.state_dict(), device=dev, lib=lib, names={0: "speech_bubble"})
    rtdetr = RTDetrHip(rmodel.state_dict(), rcfg, device=dev, lib=lib, names={0: "bubble", 1: "text_bubble", 2: "text_free"})
    sam = Sam2Hip(smodel.state_dict(), scfg, device=dev, lib=lib)
    sets.append((yolo, rtdetr, sam))
    for mt, obj in [(mm.ModelType.YOLO_SPEECH_BUBBLE, yolo), (mm.ModelType.RTDETR_CONJOINED_BUBBLE, rtdetr),
                    (mm.ModelType.SAM2, (mm._Sam2ProcessorShim(), mm._Sam2ModelShim(sam, torch.bfloat16)))]:
        mgr.models[mt if r == 0 else (mt, r)] = obj
W, H, n = 512, 768, 6
tmp = Path(tempfile.mkdtemp())
root = tmp / "in"; root.mkdir()
for i in range(n):
    pg, _b, _r = make_page(20 + i, W, H, bubbles=8, osb_regions=0)
    Image.fromarray(pg).save(root / f"p{i}.png")
    if i == 0:
        sets[0][0](np.ascontiguousarray(pg[..., ::-1]), conf=0.0, imgsz=640, max_det=1)
        plan0, _ = next(iter(sets[0][0]._plans.values()))
        sc = plan0.decoded[:, 4].float().sort(descending=True).values
        conf = float(sc[min(12, len(sc) - 1)])
cfg = tpg._config(conf, None)
cfg.outside_text.enabled = False
cfg.output = types.SimpleNamespace(upscale_final_image=False, image_upscale_factor=1.0, image_upscale_model="model_lite", output_format="png", jpeg_quality=95, png_compression=2)
cfg.verbose = False

def dig(img):
    return hashlib.sha256(np.asarray(img.convert("RGBA")).tobytes()).hexdigest()[:8]


infos = {}
real_back = pipeline.process_page_vision_back


def back_rec(state):
    out, info = real_back(state)
    key = Path(str(state["image_path"])).name
    infos.setdefault(key, []).append((len(info["bubbles"]), hashlib.sha256(b"".join(np.ascontiguousarray(b["sam_mask"]).tobytes() for b in info["bubbles"])).hexdigest()[:8],
                                      hashlib.sha256(repr([tuple(b["bbox"]) for b in info["bubbles"]]).encode()).hexdigest()[:8]))
    return out, info
pipeline.process_page_vision_back = back_rec

rows = {}
for rep in range(2):
    rows[f"page-by-page #{rep}"] = [dig(pipeline.process_page_vision(Image.open(root / f"p{i}.png").convert("RGBA"), cfg, root / f"p{i}.png")[0]) for i in range(n)]


def harness(tag, **kw):
    out = tmp / tag.replace(" ", "_").replace("#", "")
    res = pipeline.batch_vision_images(root, cfg, out, **kw) if "front_context" not in kw else None
    if res is None:
        def front(page, path):
            return pipeline.process_page_vision_front(page, cfg, path, "PNG", False)
        res = pipeline.batch_process_images(root, cfg, out, process_front=front, process_back=lambda s: pipeline.process_page_vision_back(s)[0], **kw)
    assert res["success_count"] == n, res
    rows[tag] = [dig(Image.open(out / f"p{i}_translated.png")) for i in range(n)]


harness("harness fw=1 #0", front_workers=1)
harness("harness fw=2 replicas #0", front_workers=2)
harness("harness fw=2 replicas #1", front_workers=2)
harness("harness fw=1 #1", front_workers=1)
harness("harness fw=2 same set #0", front_workers=2, front_context=lambda slot: contextlib.nullcontext())
harness("harness fw=2 replicas #2", front_workers=2)
for k, v in rows.items():
    print("DIAG", f"{k:28s}", " ".join(v))
for k in sorted(infos):
    print("DIAG", k, infos[k])
