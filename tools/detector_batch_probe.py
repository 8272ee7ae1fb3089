"""What would a batch of B pages through one 640-px detector graph cost?  Proxy without a batched builder: the same network's plan at 1x, 2x and 4x the
letterboxed pixels (640 x 448 -> 640 x 896 -> 1280 x 896), hipGraph replays timed alone.  Convolutions see exactly a batch's work; YOLO12's area attention
grows with the square of the pixels per area, so its figure is an upper bound on a real batch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mangatranslator_amd.hip.lib import get_library
from mangatranslator_amd.core.ml.yolo11 import Yolo11Hip
from oracle import yolo11_ref as y11

lib = get_library(); lib.init(0)
for name, fam, size, seed in (("panel YOLO11-L", "11", "l", 1), ("outside-text YOLO12x", "12", "x", 2)):
    m = Yolo11Hip(y11.make_model(fam, size, 1, False, seed=seed).state_dict(), device="cuda:0", lib=lib)
    out = []
    with m._lane.enter():
        for H, W in ((640, 448), (640, 896), (1280, 896)):
            plan = m._build(dict(H=H, W=W))
            plan.run(graph=True); torch.cuda.synchronize()
            out.append((H, W, len(plan.ops), plan.time(20, graph=True)))
    base = out[0][3]
    print(name + ": " + "  ".join(f"{H}x{W}: {ms:.2f} ms ({ms / base:.2f}x for {H * W / (640 * 448):.0f}x the pixels, {n} launches)" for H, W, n, ms in out), flush=True)
