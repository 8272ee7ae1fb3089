import sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from mangatranslator_amd.hip.lib import get_library
from mangatranslator_amd.core.ml.yolo import YoloSegHip
from mangatranslator_amd.utils.synthetic_pages import make_page
from oracle import yolo_ref as yr
lib = get_library(); lib.init(0)
net = yr.make_model("n", 1, 0)
with torch.no_grad():
    for l in range(3):
        net.model[22].cv3[l][2].weight.mul_(0.05); net.model[22].cv3[l][2].bias.fill_(-1.0); net.model[22].cv2[l][2].weight.mul_(0.1)
h, w, imgsz = 384, 256, 256
page, _, _ = make_page(0, w, h, bubbles=3); bgr = np.ascontiguousarray(page[..., ::-1])
x, lp = yr.letterbox(bgr, imgsz)
pred, proto = net(x)
hip = YoloSegHip(net.state_dict(), device="cuda:0", lib=lib)
out = hip(bgr, conf=0.3, imgsz=imgsz)
plan, _ = hip._plans[(h, w, imgsz)]
dec = plan.decoded.cpu()
e = (dec[:, :4] - pred[0, :4].t()).abs()
print("lp", lp, "anchors", dec.shape, pred.shape)
A = [(lp["H"] // s) * (lp["W"] // s) for s in (8, 16, 32)]
o = 0
for a_ in A:
    print("level", a_, "max", e[o:o + a_].max().item(), "median", e[o:o + a_].median().item()); o += a_
i = int(e.max(1).values.argmax())
print("worst anchor", i, dec[i, :4], pred[0, :4, i])
seg = net.model[22]
hb = plan.dbg["head0"].t.float().cpu()[0]
print("head0 shape", hb.shape, "box logits absmax", hb[..., :64].abs().max().item())
