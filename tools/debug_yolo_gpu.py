import sys
from pathlib import Path
import numpy as np, torch, torch.nn.functional as F
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from mangatranslator_amd.hip.lib import get_library
from mangatranslator_amd.core.ml.yolo import YoloSegHip
from mangatranslator_amd.utils.synthetic_pages import make_page
from oracle import yolo_ref as yr
lib = get_library(); lib.init(0)
net = yr.make_model("n", 1, 0)
h, w, imgsz = 384, 256, 256
page, _, _ = make_page(0, w, h, bubbles=3); bgr = np.ascontiguousarray(page[..., ::-1])
x, lp = yr.letterbox(bgr, imgsz)
m = net.model
with torch.no_grad():
    t = m[1](m[0](x)); p2 = m[2](t); p3 = m[4](m[3](p2)); p4 = m[6](m[5](p3)); p5 = m[9](m[8](m[7](p4)))
    up = lambda t: F.interpolate(t, scale_factor=2.0, mode="nearest")
    h4 = m[12](torch.cat([up(p5), p4], 1)); h3 = m[15](torch.cat([up(h4), p3], 1))
    n4 = m[18](torch.cat([m[16](h3), h4], 1)); n5 = m[21](torch.cat([m[19](n4), p5], 1))
ref = dict(p2=p2, p3=p3, p4=p4, p5=p5, h4=h4, h3=h3, n4=n4, n5=n5)
hip = YoloSegHip(net.state_dict(), device="cuda:0", lib=lib, graph=False)
for trial in range(2):
    out = hip(bgr, conf=0.5, imgsz=imgsz)
    plan, _ = hip._plans[(h, w, imgsz)]
    torch.cuda.synchronize()
    for k, r in ref.items():
        a = plan.dbg[k].torch().float().cpu()[0].permute(2, 0, 1)
        e = (a - r[0]).abs().max().item() / r.abs().max().item()
        print(trial, k, "rel err %.4f" % e, "nan" if torch.isnan(a).any() else "")
with torch.no_grad():
    pred, proto = net(x)
    seg = net.model[22]
    feats = [h3, n4, n5]
for graph in (False, True):
    hip = YoloSegHip(net.state_dict(), device="cuda:0", lib=lib, graph=graph)
    out = hip(bgr, conf=0.5, imgsz=imgsz)
    plan, _ = hip._plans[(h, w, imgsz)]
    torch.cuda.synchronize()
    for l in range(3):
        hb = plan.dbg[f"head{l}"].t.float().cpu()[0]
        with torch.no_grad():
            rb = seg.cv2[l](feats[l])[0].permute(1, 2, 0); rc = seg.cv3[l](feats[l])[0].permute(1, 2, 0); rm = seg.cv4[l](feats[l])[0].permute(1, 2, 0)
        print("graph", graph, "level", l, "box %.4f" % ((hb[..., :64] - rb).abs().max() / rb.abs().max()).item(),
              "cls %.4f" % ((hb[..., 64:65] - rc).abs().max() / rc.abs().max()).item(),
              "coef %.4f" % ((hb[..., 72:104] - rm).abs().max() / rm.abs().max()).item(), "pad", hb[..., 65:72].abs().max().item())
    dec = plan.decoded.cpu()
    print("graph", graph, "decoded box err", (dec[:, :4] - pred[0, :4].t()).abs().max().item(), "cls err", (dec[:, 4] - pred[0, 4]).abs().max().item(),
          "nan", torch.isnan(dec).any().item())
    pr = plan.proto.t[0].float().cpu().permute(2, 0, 1)
    print("proto rel err", ((pr - proto[0]).abs().max() / proto[0].abs().max()).item())
