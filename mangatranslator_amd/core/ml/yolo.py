"""YOLOv8-seg speech-bubble detector on libmtx_hip.

The model object `ModelManager.load_yolo_speech_bubble()` returns (reference
core/ml/model_manager.py:711-743 loads an ultralytics `YOLO`; the operator calls
`model(image_cv, conf=, device=, verbose=False, imgsz=, retina_masks=True)[0]` and reads
`.boxes.xyxy/.conf/.cls`, `.masks.data`, `.orig_shape`, `.names`, reference
core/image/detection.py:1337-1351, 525-556).

Weights: ultralytics' fused module tree as a flat state dict (`model.{i}.conv.weight`, ...; BatchNorm
already folded as the ultralytics predictor does with `model.fuse()`); un-fused `bn.*` entries are
folded at load.  The architecture (width/depth/nc) is derived from tensor shapes for the YOLOv8-seg
family.

Graph: every `torch.cat` of the reference network is free — producers write straight into channel
slices of the consumer's NHWC buffer (C2f branches, SPPF pyramid, PAN concats, the three head
branches); ConvTranspose2d(2,2) = 1x1 conv + pixel-shuffle store; DFL/sigmoid/anchor decode is one
kernel; retina masks = one GEMM (prototypes x coefficients) + fused padding-crop / bilinear / box-crop
/ threshold writing page-resolution bitmasks.  NMS runs on the few surviving candidates on the host in
the reference's exact order.
"""
import threading
from types import SimpleNamespace

import numpy as np
import torch

from ...hip import abi
from ...hip.lib import get_library
from ...hip.plan import AsyncLane, LaneTicket, PlanBuilder, PlanCache, result_tensors
from ...utils.exceptions import ModelError


def fold_batchnorm(sd: dict) -> dict:
    """conv.weight + bn.{weight,bias,running_mean,running_var} -> conv.{weight,bias} (eps 1e-3, ultralytics)."""
    out = dict(sd)
    for k in list(sd):
        if k.endswith(".bn.weight"):
            base = k[:-len(".bn.weight")]
            w = sd[base + ".conv.weight"].float()
            g, b = sd[base + ".bn.weight"].float(), sd[base + ".bn.bias"].float()
            mu, var = sd[base + ".bn.running_mean"].float(), sd[base + ".bn.running_var"].float()
            s = g / torch.sqrt(var + 1e-3)
            out[base + ".conv.weight"] = w * s.view(-1, 1, 1, 1)
            out[base + ".conv.bias"] = b - mu * s
            for suf in (".bn.weight", ".bn.bias", ".bn.running_mean", ".bn.running_var", ".bn.num_batches_tracked"):
                out.pop(base + suf, None)
    return out


def derive_arch(sd: dict) -> dict:
    try:
        c = [sd[f"model.{i}.conv.weight"].shape[0] for i in (0, 1, 3, 5, 7)]
        count = lambda i: len({k.split(".")[3] for k in sd if k.startswith(f"model.{i}.m.")})
        n = [count(2), count(4), count(6), count(8)]
        nh = count(12)
        nc = sd["model.22.cv3.0.2.weight"].shape[0]
        nm = sd["model.22.cv4.0.2.weight"].shape[0]
        reg_max = sd["model.22.cv2.0.2.weight"].shape[0] // 4
        npr = sd["model.22.proto.cv1.conv.weight"].shape[0]
    except KeyError as e:
        raise ModelError(f"not a YOLOv8-seg state dict (missing {e})") from e
    return dict(c=c, n=n, nh=nh, nc=int(nc), nm=int(nm), npr=int(npr), reg_max=int(reg_max))


def letterbox_params(h, w, imgsz, stride=32):
    """ultralytics LetterBox(auto=True): scale to fit imgsz, pad each side to a stride multiple."""
    r = min(imgsz / h, imgsz / w)
    nh, nw = int(round(h * r)), int(round(w * r))
    dw, dh = (imgsz - nw) % stride / 2, (imgsz - nh) % stride / 2
    top, left = int(round(dh - 0.1)), int(round(dw - 0.1))
    bottom, right = int(round(dh + 0.1)), int(round(dw + 0.1))
    return dict(r=r, nh=nh, nw=nw, top=top, left=left, H=nh + top + bottom, W=nw + left + right)


def nms_xyxy(boxes: np.ndarray, scores: np.ndarray, iou_thres: float):
    order = np.argsort(-scores, kind="stable")
    area = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
    alive = np.ones(len(boxes), bool)
    keep = []
    for i in order:
        if not alive[i]:
            continue
        keep.append(int(i))
        iw = np.clip(np.minimum(boxes[i, 2], boxes[:, 2]) - np.maximum(boxes[i, 0], boxes[:, 0]), 0, None)
        ih = np.clip(np.minimum(boxes[i, 3], boxes[:, 3]) - np.maximum(boxes[i, 1], boxes[:, 1]), 0, None)
        inter = iw * ih
        alive &= ~(inter / (area[i] + area - inter) > iou_thres)
    return keep


class YoloSegHip:
    def __init__(self, state_dict: dict, device="cuda", names=None, lib=None, graph: bool = True):
        self.lib = lib if lib is not None else get_library()
        self.device = torch.device(device)
        self.dtype, self.tdt = abi.F16, torch.float16
        sd = fold_batchnorm({k: v.detach().float().cpu() for k, v in state_dict.items()})
        self.a = self._derive(sd)
        self.names = names or {i: f"class{i}" for i in range(self.a["nc"])}
        self._graph = graph and not self.lib.is_simulator
        self._lane = AsyncLane(self.device, self.lib.is_simulator)
        self._plans = PlanCache(8)
        self._mask_plans = PlanCache(8)
        self._pack(sd)

    def _derive(self, sd):
        return derive_arch(sd)

    # ---- weights ------------------------------------------------------------------------------------
    def _pack(self, sd):
        self.W = {}

        def put(name, w, b, cout_pad=0, cin_pad=0):
            co, ci, kh, kw = w.shape
            ci_p = max((ci + 7) // 8 * 8, cin_pad)
            co_p = max(co, cout_pad)
            wt = torch.zeros(co_p, kh * kw, ci_p)
            wt[:co, :, :ci] = w.permute(0, 2, 3, 1).reshape(co, kh * kw, ci)
            bt = torch.zeros(co_p)
            bt[:co] = b
            self.W[name] = (wt.to(self.device, self.tdt).contiguous(), bt.to(self.device).contiguous(), co_p, kh)

        for k in sd:
            if k.endswith(".conv.weight"):
                base = k[:-len(".conv.weight")]
                put(base, sd[k], sd[base + ".conv.bias"])
        for l in range(3):
            for br, pad in (("cv2", 0), ("cv3", 8), ("cv4", 0)):
                base = f"model.22.{br}.{l}.2"
                put(base, sd[base + ".weight"], sd[base + ".bias"], cout_pad=pad)
        up = sd["model.22.proto.upsample.weight"]                      # [Cin, Cout, 2, 2]
        ci, co = up.shape[:2]
        w1 = up.permute(2, 3, 1, 0).reshape(4 * co, ci, 1, 1)
        put("model.22.proto.upsample", w1, sd["model.22.proto.upsample.bias"].repeat(4))

    # ---- graph --------------------------------------------------------------------------------------
    def _conv(self, pb, x, name, stride=1, act=abi.ACT_SILU, out=None, res=None, label=None):
        w, b, co, k = self.W[name]
        return pb.conv2d(x, w, b, co, ksize=k, stride=stride, act=act, out=out, res=res, label=label or name)

    def _box_logits_f32(self, pb, t, name, nb):
        """the box branch's last 1x1 convolution as an fp32-output GEMM of the pixel rows: the 4 * reg_max DFL logits of a level never round to 16
        bits (the reference's detector is fp32 end to end, core/image/detection.py:1337-1345; its boxes feed IoU / IoA index decisions, and 16-bit
        logits alone cost 0.03 bin = 1 px at stride 32 — VERDICT r05 #7).  -> fp32 [h * w, nb]"""
        w, b, co, k = self.W[name]
        assert k == 1 and co == nb
        rows = t.n * t.h * t.w
        out = pb.buf((rows, nb), torch.float32)
        pb.gemm(t, w, rows, nb, t.c, lda=t.ld, ldw=int(w.shape[-1]), out=out, ldc=nb, bias=b, out_f32=True, label=name + ".f32")
        return out

    def _c2f(self, pb, x, i, n, shortcut, out=None):
        c2 = self.W[f"model.{i}.cv2"][2]
        c = c2 // 2
        cat = pb.act(x.n, x.h, x.w, (2 + n) * c)
        self._conv(pb, x, f"model.{i}.cv1", out=cat.slice(0, 2 * c))
        for k in range(n):
            y = cat.slice((1 + k) * c, c)
            t = self._conv(pb, y, f"model.{i}.m.{k}.cv1")
            self._conv(pb, t, f"model.{i}.m.{k}.cv2", out=cat.slice((2 + k) * c, c), res=y if shortcut else None)
        return self._conv(pb, cat, f"model.{i}.cv2", out=out)

    def _build(self, lp):
        a, c, n = self.a, self.a["c"], self.a["n"]
        pb = PlanBuilder(self.lib, self.device, self.dtype)
        H, W = lp["H"], lp["W"]
        if H % 32 or W % 32:
            raise ModelError(f"letterboxed input {W}x{H} must be a multiple of 32")
        img = pb.act(1, H, W, 8)
        x = self._conv(pb, img, "model.0", 2)
        x = self._conv(pb, x, "model.1", 2)
        p2 = self._c2f(pb, x, 2, n[0], True)
        h8, w8, h16, w16, h32, w32 = H // 8, W // 8, H // 16, W // 16, H // 32, W // 32
        cat15 = pb.act(1, h8, w8, c[3] + c[2])
        cat12 = pb.act(1, h16, w16, c[4] + c[3])
        cat18 = pb.act(1, h16, w16, c[2] + c[3])
        cat21 = pb.act(1, h32, w32, c[3] + c[4])
        x = self._conv(pb, p2, "model.3", 2)
        p3 = self._c2f(pb, x, 4, n[1], True, out=cat15.slice(c[3], c[2]))
        x = self._conv(pb, p3, "model.5", 2)
        p4 = self._c2f(pb, x, 6, n[2], True, out=cat12.slice(c[4], c[3]))
        x = self._conv(pb, p4, "model.7", 2)
        x = self._c2f(pb, x, 8, n[3], True)
        # SPPF: three chained 5x5 max-pools written into slices of one buffer
        ch = c[4] // 2
        sp = pb.act(1, h32, w32, 4 * ch)
        self._conv(pb, x, "model.9.cv1", out=sp.slice(0, ch))
        for k in range(3):
            pb.ew(abi.EW_MAXPOOL, sp.slice(k * ch, ch), out=sp.slice((k + 1) * ch, ch), i0=5, i1=1, label=f"sppf.pool{k}")
        p5 = self._conv(pb, sp, "model.9.cv2", out=cat21.slice(c[3], c[4]))
        # PAN neck
        pb.ew(abi.EW_UPSAMPLE2X, p5, out=cat12.slice(0, c[4]), label="up.p5")
        h4 = self._c2f(pb, cat12, 12, a["nh"], False, out=cat18.slice(c[2], c[3]))
        pb.ew(abi.EW_UPSAMPLE2X, h4, out=cat15.slice(0, c[3]), label="up.h4")
        h3 = self._c2f(pb, cat15, 15, a["nh"], False)
        self._conv(pb, h3, "model.16", 2, out=cat18.slice(0, c[2]))
        n4 = self._c2f(pb, cat18, 18, a["nh"], False)
        self._conv(pb, n4, "model.19", 2, out=cat21.slice(0, c[3]))
        n5 = self._c2f(pb, cat21, 21, a["nh"], False)
        # Segment head: [4*reg_max | nc padded to 8 | nm] per level
        nb, ncp, nm = 4 * a["reg_max"], 8, a["nm"]
        heads, box32 = [], []
        for l, f in enumerate((h3, n4, n5)):
            hb = pb.act(1, f.h, f.w, nb + ncp + nm)
            for br, off, cw in (("cv2", 0, nb), ("cv3", nb, ncp), ("cv4", nb + ncp, nm)):
                t = self._conv(pb, f, f"model.22.{br}.{l}.0")
                t = self._conv(pb, t, f"model.22.{br}.{l}.1")
                if br == "cv2":
                    box32.append(self._box_logits_f32(pb, t, f"model.22.cv2.{l}.2", nb))      # (hb's first nb channels stay unwritten: the decode reads the fp32 logits)
                else:
                    self._conv(pb, t, f"model.22.{br}.{l}.2", act=abi.ACT_NONE, out=hb.slice(off, cw))
            heads.append(hb)
        anchors = sum(hb.h * hb.w for hb in heads)
        decoded = pb.buf((anchors, 4 + a["nc"] + nm), torch.float32)
        pb.yolo_decode(heads, [8, 16, 32], a["nc"], nm, a["reg_max"], decoded, cls_off=nb, mc_off=nb + ncp, box_f32=box32)
        # Proto
        t = self._conv(pb, h3, "model.22.proto.cv1")
        w, b, co, _ = self.W["model.22.proto.upsample"]
        t = pb.conv2d(t, w, b, co, ksize=1, pixel_shuffle=2, label="proto.upsample")
        t = self._conv(pb, t, "model.22.proto.cv2")
        proto = self._conv(pb, t, "model.22.proto.cv3")
        plan = pb.build()
        plan.img, plan.decoded, plan.proto = img, decoded, proto
        plan.dbg = dict(p2=p2, p3=p3, p4=p4, p5=p5, h4=h4, h3=h3, n4=n4, n5=n5, head0=heads[0], head1=heads[1], head2=heads[2])
        return plan

    def _mask_plan(self, nd, mh, mw, roi, h0, w0):
        """retina-mask plan for up to `nd` detections (callers round the count up to a multiple of 8, so a stream of pages with 5..30
        bubbles shares four plans); kept in a cache of its own — per-page keys must never evict the detection plans and their hipGraphs"""
        key = (nd, mh, mw, roi, h0, w0)
        if key not in self._mask_plans:
            pb = PlanBuilder(self.lib, self.device, self.dtype)
            nm = self.a["nm"]
            coef = pb.buf((nd, nm), self.tdt)
            proto = pb.buf((mh * mw, nm), self.tdt)
            boxes = pb.buf((nd, 4), torch.float32)
            logits = pb.gemm(proto, coef, mh * mw, nd, nm, out_f32=True, label="masks.gemm")
            masks = pb.buf((nd, h0, w0), torch.uint8)
            pb.resize_threshold(logits, masks, nd, mh, mw, h0, w0, 0.0, abi.F32, pix_stride=nd, batch_stride=1, roi=roi, crop_xyxy=boxes)
            plan = pb.build()
            plan.coef, plan.protoflat, plan.boxes, plan.masks = coef, proto, boxes, masks
            self._mask_plans[key] = plan
        return self._mask_plans[key]

    # ---- the ultralytics call shape -------------------------------------------------------------------
    @torch.no_grad()
    def __call__(self, image_bgr, conf=0.25, device=None, verbose=False, imgsz=640, retina_masks=True, iou=0.7, max_det=300):
        return self.collect(self.submit(image_bgr, conf=conf, imgsz=imgsz, iou=iou, max_det=max_det))

    @torch.no_grad()
    def submit(self, image_bgr, conf=0.25, imgsz=640, iou=0.7, max_det=300, **_kw):
        """first half of a call: upload, letterbox and the network's graph replay are queued on this model's own stream (`AsyncLane`)
        and the call returns at once with a ticket for `collect`.  The model stays busy until the ticket is collected."""
        # a page that is already on the device (uint8 [H, W, 3] BGR tensor: the caller uploaded it once for all of its detectors) is
        # used in place; a host image is uploaded here
        on_device = torch.is_tensor(image_bgr) and image_bgr.device.type == self.device.type and self.device.type != "cpu"
        img = image_bgr[..., :3] if on_device else np.ascontiguousarray(np.asarray(image_bgr)[..., :3])
        h0, w0 = int(img.shape[0]), int(img.shape[1])
        lp = letterbox_params(h0, w0, imgsz)
        key = (h0, w0, imgsz)
        self._lane.acquire()
        try:
            if on_device:
                self._lane.adopt(image_bgr)
            with self._lane.enter():
                if key not in self._plans:
                    plan = self._build(lp)
                    pre = PlanBuilder(self.lib, self.device, self.dtype)
                    page = pre.buf((h0, w0, 3), torch.uint8)
                    pre.letterbox(page, plan.img, h0, w0, lp["nh"], lp["nw"], lp["top"], lp["left"])
                    pp = pre.build()
                    pp.page = page
                    self._plans[key] = (plan, pp)
                plan, pp = self._plans[key]
                pp.page.copy_(img if on_device else torch.from_numpy(img).to(self.device, non_blocking=True))
                pp.run()
                plan.run(graph=self._graph)
        except BaseException:
            self._lane.release()
            raise
        return LaneTicket(self._lane, plan=plan, lp=lp, hw=(h0, w0), conf=conf, iou=iou, max_det=max_det)

    @torch.no_grad()
    def collect(self, ticket):
        """second half: candidates above `conf` come to the host, NMS, boxes back in page coordinates, retina masks"""
        try:
            with self._lane.resume():
                res = self._finish(**ticket)
            self._lane.hand_over(*result_tensors(res))
            return res
        finally:
            ticket.close() if isinstance(ticket, LaneTicket) else self._lane.release()

    def _finish(self, plan, lp, hw, conf, iou, max_det):
        h0, w0 = hw
        nc, nm = self.a["nc"], self.a["nm"]
        dec = plan.decoded
        scores, cls = dec[:, 4:4 + nc].max(1)
        idx = torch.nonzero(scores > conf).flatten()
        cand = dec[idx].cpu().numpy()
        sc, cl = scores[idx].cpu().numpy(), cls[idx].cpu().numpy()
        res = SimpleNamespace(orig_shape=(h0, w0), names=self.names, boxes=None, masks=None)
        if len(cand) == 0:
            return [res]
        keep = nms_xyxy(cand[:, :4] + cl[:, None].astype(np.float32) * 7680.0, sc, iou)[:max_det]
        cand, sc, cl = cand[keep], sc[keep], cl[keep]
        gain = min(lp["H"] / h0, lp["W"] / w0)
        padw, padh = round((lp["W"] - w0 * gain) / 2 - 0.1), round((lp["H"] - h0 * gain) / 2 - 0.1)
        pb_ = cand[:, :4].astype(np.float32).copy()
        pb_[:, [0, 2]] = ((pb_[:, [0, 2]] - padw) / gain).clip(0, w0)
        pb_[:, [1, 3]] = ((pb_[:, [1, 3]] - padh) / gain).clip(0, h0)
        boxes_t = torch.from_numpy(pb_).to(self.device)
        res.boxes = _Boxes(boxes_t, torch.from_numpy(sc).to(self.device), torch.from_numpy(cl.astype(np.float32)).to(self.device))
        if nm == 0:                    # detect-only head (panel / outside-text detectors): boxes are the whole result
            return [res]
        # retina masks
        mh, mw = plan.proto.h, plan.proto.w
        gm = min(mh / h0, mw / w0)
        pw, ph = (mw - w0 * gm) / 2, (mh - h0 * gm) / 2
        top, left = int(round(ph - 0.1)), int(round(pw - 0.1))
        roi = (top, left, mh - int(round(ph + 0.1)) - top, mw - int(round(pw + 0.1)) - left)
        nk = len(keep)
        mp = self._mask_plan((nk + 7) // 8 * 8, mh, mw, roi, h0, w0)
        mp.coef.zero_()
        mp.coef[:nk].copy_(torch.from_numpy(cand[:, 4 + nc:]).to(self.device, self.tdt))
        mp.protoflat.copy_(plan.proto.t.view(mh * mw, nm))
        mp.boxes.zero_()                                  # rows past nk: an empty crop box, masks of zeros nobody reads
        mp.boxes[:nk].copy_(boxes_t)
        mp.run()
        res.masks = _Masks(mp.masks[:nk].clone(), self.lib)
        return [res]


class _Boxes:
    """the slice of ultralytics `Boxes` the operators read: `.xyxy / .conf / .cls` tensors on the device, `len()`"""

    def __init__(self, xyxy, conf, cls):
        self.xyxy, self.conf, self.cls = xyxy, conf, cls

    def __len__(self):
        return int(self.xyxy.shape[0])


class _Masks:
    """the slice of ultralytics `Masks` the operators read (reference core/image/detection.py:525-556): `.data [N, H, W]` on the
    device, `len()`, and `masks[i].xy[0]` — the instance's outline polygon (largest external contour, float32 (x, y) pixel
    coordinates), traced natively on the host (`mtx_host_mask_outline`) only when somebody asks for it"""

    def __init__(self, data, lib):
        self.data, self._lib = data, lib

    def __len__(self):
        return int(self.data.shape[0])

    def __getitem__(self, i):
        if not -len(self) <= i < len(self):
            raise IndexError(i)
        m = np.ascontiguousarray((self.data[i] > 0).to(torch.uint8).cpu().numpy())
        h, w = m.shape
        cap = 4 * (h + w)
        buf = np.empty((cap, 2), np.int32)
        n = self._lib.mtx_host_mask_outline(m.ctypes.data, w, h, buf.ctypes.data, cap)
        if n > cap:
            buf = np.empty((n, 2), np.int32)
            n = self._lib.mtx_host_mask_outline(m.ctypes.data, w, h, buf.ctypes.data, n)
        if n < 0:
            raise RuntimeError("mtx_host_mask_outline failed")
        return SimpleNamespace(xy=[buf[:n].astype(np.float32)], data=self.data[i:i + 1])
