"""CPU oracle for the YOLOv8-seg speech-bubble detector.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the reference calls an ultralytics model object
(`model(image_cv, conf=, device=, verbose=False, imgsz=, retina_masks=True)[0]`, reference
core/image/detection.py:1337-1345; loaded at core/ml/model_manager.py:711-743 with
`ultralytics>=8.3.94`, un-pinned and NOT installed here, and no `.pt` checkpoint is present).  This
file restates the published YOLOv8-seg architecture and predictor post-processing
(ultralytics cfg/models/v8/yolov8-seg.yaml; nn/modules/{conv,block,head}.py; utils/ops.py):

  backbone Conv-Conv-C2f-Conv-C2f-Conv-C2f-Conv-C2f-SPPF, PAN neck with nearest upsample + concat,
  Segment head (DFL box branch, class branch, 32 mask coefficients, Proto net), fused Conv+BN (the
  predictor runs `model.fuse()`), SiLU;
  LetterBox(auto, stride 32, pad 114) -> /255 -> net -> conf filter -> class-offset NMS (IoU 0.7,
  max_det 300) -> scale_boxes -> retina masks (coeff @ proto, crop letterbox padding, bilinear to the
  page, crop to box, > 0).

State-dict keys follow ultralytics' fused module tree (`model.{i}.conv.weight`, `model.{i}.m.{k}.cv1.conv.*`,
`model.22.cv2.{l}.2.weight`, `model.22.proto.upsample.weight`, ...).
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

SCALES = {"n": (0.33, 0.25, 1024), "s": (0.33, 0.50, 1024), "m": (0.67, 0.75, 768), "l": (1.0, 1.0, 512), "x": (1.0, 1.25, 512)}


def make_divisible(x, d=8):
    return int(math.ceil(x / d) * d)


def arch(scale="m", nc=1):
    d, w, mc = SCALES[scale]
    ch = lambda c: make_divisible(min(c, mc) * w, 8)
    dep = lambda n: max(round(n * d), 1)
    return dict(c=[ch(64), ch(128), ch(256), ch(512), ch(1024)], n=[dep(3), dep(6), dep(6), dep(3)], nh=dep(3), nc=nc, nm=32,
                npr=ch(256), reg_max=16)


class Conv(nn.Module):
    def __init__(self, c1, c2, k=1, s=1, act=True):
        super().__init__()
        self.conv = nn.Conv2d(c1, c2, k, s, k // 2, bias=True)
        self.act = act

    def forward(self, x):
        y = self.conv(x)
        return F.silu(y) if self.act else y


class Bottleneck(nn.Module):
    def __init__(self, c, shortcut):
        super().__init__()
        self.cv1, self.cv2, self.add = Conv(c, c, 3), Conv(c, c, 3), shortcut

    def forward(self, x):
        y = self.cv2(self.cv1(x))
        return x + y if self.add else y


class C2f(nn.Module):
    def __init__(self, c1, c2, n, shortcut):
        super().__init__()
        self.c = c2 // 2
        self.cv1, self.cv2 = Conv(c1, 2 * self.c, 1), Conv((2 + n) * self.c, c2, 1)
        self.m = nn.ModuleList(Bottleneck(self.c, shortcut) for _ in range(n))

    def forward(self, x):
        y = list(self.cv1(x).chunk(2, 1))
        for m in self.m:
            y.append(m(y[-1]))
        return self.cv2(torch.cat(y, 1))


class SPPF(nn.Module):
    def __init__(self, c1, c2):
        super().__init__()
        self.cv1, self.cv2 = Conv(c1, c1 // 2, 1), Conv(c1 // 2 * 4, c2, 1)

    def forward(self, x):
        y = [self.cv1(x)]
        for _ in range(3):
            y.append(F.max_pool2d(y[-1], 5, 1, 2))
        return self.cv2(torch.cat(y, 1))


class Proto(nn.Module):
    def __init__(self, c1, c_, c2):
        super().__init__()
        self.cv1 = Conv(c1, c_, 3)
        self.upsample = nn.ConvTranspose2d(c_, c_, 2, 2, 0, bias=True)
        self.cv2, self.cv3 = Conv(c_, c_, 3), Conv(c_, c2, 1)

    def forward(self, x):
        return self.cv3(self.cv2(self.upsample(self.cv1(x))))


class Segment(nn.Module):
    def __init__(self, ch, nc, nm, npr, reg_max):
        super().__init__()
        self.nc, self.nm, self.reg_max = nc, nm, reg_max
        c2, c3, c4 = max(16, ch[0] // 4, reg_max * 4), max(ch[0], min(nc, 100)), max(ch[0] // 4, nm)
        br = lambda cin, cm, cout: nn.Sequential(Conv(cin, cm, 3), Conv(cm, cm, 3), nn.Conv2d(cm, cout, 1))
        self.cv2 = nn.ModuleList(br(x, c2, 4 * reg_max) for x in ch)
        self.cv3 = nn.ModuleList(br(x, c3, nc) for x in ch)
        self.cv4 = nn.ModuleList(br(x, c4, nm) for x in ch)
        self.proto = Proto(ch[0], npr, nm)

    def forward(self, xs):
        proto = self.proto(xs[0])
        outs, anchors, strides = [], [], []
        for i, x in enumerate(xs):
            b, _, h, w = x.shape
            outs.append(torch.cat([self.cv2[i](x), self.cv3[i](x), self.cv4[i](x)], 1).flatten(2))
            sy, sx = torch.meshgrid(torch.arange(h, dtype=torch.float32) + 0.5, torch.arange(w, dtype=torch.float32) + 0.5, indexing="ij")
            anchors.append(torch.stack([sx, sy], -1).view(-1, 2))
            strides.append(torch.full((h * w,), float(8 * 2 ** i)))
        y = torch.cat(outs, 2)
        anc, st = torch.cat(anchors).t(), torch.cat(strides)
        box, cls, mc = y.split([4 * self.reg_max, self.nc, self.nm], 1)
        b, _, a = box.shape
        dist = (box.view(b, 4, self.reg_max, a).softmax(2) * torch.arange(self.reg_max, dtype=torch.float32).view(1, 1, -1, 1)).sum(2)
        lt, rb = dist.chunk(2, 1)
        xyxy = torch.cat([anc[None] - lt, anc[None] + rb], 1) * st
        return torch.cat([xyxy, cls.sigmoid(), mc], 1), proto     # boxes kept as xyxy (letterboxed pixels)


class YoloV8Seg(nn.Module):
    def __init__(self, a):
        super().__init__()
        c, n = a["c"], a["n"]
        self.a = a
        m = nn.ModuleList()
        m += [Conv(3, c[0], 3, 2), Conv(c[0], c[1], 3, 2), C2f(c[1], c[1], n[0], True), Conv(c[1], c[2], 3, 2), C2f(c[2], c[2], n[1], True),
              Conv(c[2], c[3], 3, 2), C2f(c[3], c[3], n[2], True), Conv(c[3], c[4], 3, 2), C2f(c[4], c[4], n[3], True), SPPF(c[4], c[4]),
              nn.Identity(), nn.Identity(), C2f(c[4] + c[3], c[3], a["nh"], False), nn.Identity(), nn.Identity(), C2f(c[3] + c[2], c[2], a["nh"], False),
              Conv(c[2], c[2], 3, 2), nn.Identity(), C2f(c[2] + c[3], c[3], a["nh"], False), Conv(c[3], c[3], 3, 2), nn.Identity(),
              C2f(c[3] + c[4], c[4], a["nh"], False), Segment([c[2], c[3], c[4]], a["nc"], a["nm"], a["npr"], a["reg_max"])]
        self.model = m

    @torch.no_grad()
    def forward(self, x):
        m = self.model
        x = m[1](m[0](x))
        p2 = m[2](x)
        p3 = m[4](m[3](p2))
        p4 = m[6](m[5](p3))
        p5 = m[9](m[8](m[7](p4)))
        up = lambda t: F.interpolate(t, scale_factor=2.0, mode="nearest")
        h4 = m[12](torch.cat([up(p5), p4], 1))
        h3 = m[15](torch.cat([up(h4), p3], 1))
        n4 = m[18](torch.cat([m[16](h3), h4], 1))
        n5 = m[21](torch.cat([m[19](n4), p5], 1))
        return m[22]([h3, n4, n5])


def make_model(scale="n", nc=1, seed=0):
    torch.manual_seed(seed)
    net = YoloV8Seg(arch(scale, nc)).eval().float()
    with torch.no_grad():
        for name, p in net.named_parameters():
            if p.dim() == 4:
                fan = p.shape[1] * p.shape[2] * p.shape[3]
                p.normal_(0, 1.6 / math.sqrt(fan))
            else:
                p.normal_(0, 0.1)
        seg = net.model[22]
        for l in range(3):      # keep a handful of anchors above the confidence threshold
            seg.cv3[l][2].bias.fill_(-2.0)
    return net


def letterbox_params(h, w, imgsz, stride=32):
    r = min(imgsz / h, imgsz / w)
    nh, nw = int(round(h * r)), int(round(w * r))
    dw, dh = (imgsz - nw) % stride / 2, (imgsz - nh) % stride / 2
    top, left = int(round(dh - 0.1)), int(round(dw - 0.1))
    bottom, right = int(round(dh + 0.1)), int(round(dw + 0.1))
    return dict(r=r, nh=nh, nw=nw, top=top, left=left, H=nh + top + bottom, W=nw + left + right, dw=dw, dh=dh)


def letterbox(img_bgr: np.ndarray, imgsz: int):
    h, w = img_bgr.shape[:2]
    lp = letterbox_params(h, w, imgsz)
    t = torch.from_numpy(img_bgr[..., ::-1].copy()).permute(2, 0, 1)[None].float()
    if (lp["nh"], lp["nw"]) != (h, w):
        t = F.interpolate(t, (lp["nh"], lp["nw"]), mode="bilinear", align_corners=False).round()
    canvas = torch.full((1, 3, lp["H"], lp["W"]), 114.0)
    canvas[:, :, lp["top"]:lp["top"] + lp["nh"], lp["left"]:lp["left"] + lp["nw"]] = t
    return canvas / 255.0, lp


def nms(boxes: np.ndarray, scores: np.ndarray, iou_thres: float):
    """Greedy NMS in score order (torchvision.ops.nms semantics: suppress IoU > thres)."""
    order = np.argsort(-scores, kind="stable")
    keep = []
    area = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
    alive = np.ones(len(boxes), bool)
    for i in order:
        if not alive[i]:
            continue
        keep.append(int(i))
        xx1, yy1 = np.maximum(boxes[i, 0], boxes[:, 0]), np.maximum(boxes[i, 1], boxes[:, 1])
        xx2, yy2 = np.minimum(boxes[i, 2], boxes[:, 2]), np.minimum(boxes[i, 3], boxes[:, 3])
        inter = np.clip(xx2 - xx1, 0, None) * np.clip(yy2 - yy1, 0, None)
        iou = inter / (area[i] + area - inter)
        alive &= ~(iou > iou_thres)
    return keep


def postprocess(pred: torch.Tensor, proto: torch.Tensor, lp, orig_hw, conf=0.6, iou=0.7, max_det=300, nc=1):
    """pred [4+nc+nm, A] (xyxy, letterboxed px), proto [nm, mh, mw] -> boxes[N,4] page px, conf, cls, masks[N,H,W] bool"""
    h0, w0 = orig_hw
    p = pred.t()
    scores, cls = p[:, 4:4 + nc].max(1)
    sel = scores > conf
    p, scores, cls = p[sel], scores[sel], cls[sel]
    if p.shape[0] == 0:
        return np.zeros((0, 4), np.float32), np.zeros(0, np.float32), np.zeros(0, np.int64), np.zeros((0, h0, w0), bool)
    boxes = p[:, :4].numpy().astype(np.float32)
    off = cls.numpy().astype(np.float32)[:, None] * 7680.0
    keep = nms(boxes + off, scores.numpy(), iou)[:max_det]
    boxes, scores, cls, coef = boxes[keep], scores.numpy()[keep], cls.numpy()[keep], p[keep, 4 + nc:]
    # scale_boxes: remove the letterbox padding, divide by gain, clip
    gain = min(lp["H"] / h0, lp["W"] / w0)
    padw, padh = round((lp["W"] - w0 * gain) / 2 - 0.1), round((lp["H"] - h0 * gain) / 2 - 0.1)
    pb = boxes.copy()
    pb[:, [0, 2]] = (pb[:, [0, 2]] - padw) / gain
    pb[:, [1, 3]] = (pb[:, [1, 3]] - padh) / gain
    pb[:, [0, 2]] = pb[:, [0, 2]].clip(0, w0)
    pb[:, [1, 3]] = pb[:, [1, 3]].clip(0, h0)
    # process_mask_native: coefficients @ prototypes, crop the padding at mask resolution, bilinear to the page
    nm, mh, mw = proto.shape
    m = (coef @ proto.view(nm, -1)).view(-1, mh, mw)
    gm = min(mh / h0, mw / w0)
    pw, ph = (mw - w0 * gm) / 2, (mh - h0 * gm) / 2
    top, left = int(round(ph - 0.1)), int(round(pw - 0.1))
    bottom, right = mh - int(round(ph + 0.1)), mw - int(round(pw + 0.1))
    m = F.interpolate(m[None, :, top:bottom, left:right], (h0, w0), mode="bilinear", align_corners=False)[0]
    xs, ys = torch.arange(w0)[None, None, :], torch.arange(h0)[None, :, None]
    bt = torch.from_numpy(pb)
    inside = (xs >= bt[:, 0, None, None]) & (xs < bt[:, 2, None, None]) & (ys >= bt[:, 1, None, None]) & (ys < bt[:, 3, None, None])
    masks = (m * inside).gt(0.0)
    return pb, scores, cls, masks.numpy(), dict(roi=(top, left, bottom - top, right - left))


@torch.no_grad()
def predict(net, img_bgr: np.ndarray, imgsz=640, conf=0.6):
    x, lp = letterbox(img_bgr, imgsz)
    pred, proto = net(x)
    out = postprocess(pred[0], proto[0], lp, img_bgr.shape[:2], conf=conf, nc=net.a["nc"])
    return dict(boxes=out[0], conf=out[1], cls=out[2], masks=out[3], pred=pred[0], proto=proto[0], lp=lp)
