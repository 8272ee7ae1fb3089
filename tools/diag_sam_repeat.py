"""diagnostic (GPU): the same page + boxes through one Sam2Hip instance several times — which stage's output changes between calls?"""
import hashlib
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from oracle import sam2_ref                                           # noqa: E402 (seeded weights only)
from mangatranslator_amd.core.ml.sam2 import Sam2Hip                   # noqa: E402
from mangatranslator_amd.hip.lib import get_library                    # noqa: E402
from mangatranslator_amd.utils.synthetic_pages import make_page        # noqa: E402

lib = get_library(); lib.init(0)
dev = torch.device("cuda:0")
smodel, scfg = sam2_ref.make_model("tiny_test", seed=2)
W, H = 512, 768
pg, _b, _r = make_page(23, W, H, bubbles=8, osb_regions=0)
rng = np.random.default_rng(0)


def d(t):
    return hashlib.sha256(t.detach().float().cpu().numpy().tobytes()).hexdigest()[:6]


for graph in (True, False):
    sam = Sam2Hip(smodel.state_dict(), scfg, device=dev, lib=lib, graph=graph)
    for nb in (28, 23, 28):
        r = np.random.default_rng(nb)
        bx = np.stack([r.integers(0, 300, nb), r.integers(0, 500, nb), r.integers(0, 300, nb) + 330, r.integers(0, 500, nb) + 560], 1).astype(np.float32)
        bx[:, 2] = bx[:, 0] + 25 + (np.arange(nb) % 5) * 30
        bx[:, 3] = bx[:, 1] + 32 + (np.arange(nb) % 7) * 20
        rows = []
        for rep in range(6):
            if rep == 3:
                junk = [torch.randn(int(s), device=dev) for s in rng.integers(1000, 3_000_000, 12)]      # stir the allocator
                del junk
            masks, low, iou, sel = sam.segment(pg, bx, return_logits=True)
            torch.cuda.synchronize()
            enc, dec = sam._encoder(), sam._decoder(nb)
            rows.append((d(enc.img.t), d(enc.src.t), d(enc.feat_s0.t), d(enc.feat_s1.t), d(dec.tok0), d(dec.logits), d(iou), d(sel), d(low), d(masks)))
        print("DIAG graph", graph, "n", nb)
        for row in rows:
            print("DIAG   img %s src %s s0 %s s1 %s tok0 %s logits %s iou %s sel %s low %s masks %s" % row)
