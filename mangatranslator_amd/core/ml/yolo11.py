"""YOLO11 / YOLO12 detectors on libmtx_hip: the reference's DEFAULT bubble detector `yolo_2`
(manga109-segmentation-bubble, YOLO11-seg; core/ml/model_manager.py:120-125, 183-190, called at core/image/detection.py:1337-1351), the
panel detector (YOLO11-L; model_manager.py:809-838, detection.py:1867-1873) and the outside-text detector (AnimeText YOLO12x;
model_manager.py:780-808, detection.py:144-150, ocr_detection.py:425-431).  Same call shape and result object as `YoloSegHip`
(ultralytics `model(img, conf=, imgsz=, retina_masks=)[0]`), same letterbox / decode / NMS / retina-mask path; what differs is the
network, restated in oracle/yolo11_ref.py.

Graph notes (on top of core/ml/yolo.py):
  * C3k2 / C3k / A2C2f write their branches into channel slices of one buffer like C2f does; only the PAN concatenations of two
    producers that live elsewhere are copies.
  * C2PSA attention: the 1x1 qkv conv is re-packed at load so that all heads' q, k, v land in three contiguous column blocks
    (q and k zero-padded from key_dim to head_dim: the dot products do not change), attention reads them through strides; the depthwise
    3x3 positional conv on v (MTX_EW_DWCONV) is added to the attention output, `proj` carries the block's residual in its epilogue.
  * YOLO12 area attention: the `area` chunks of the flattened sequence are the batch dimension of one attention launch (batch stride =
    N / area rows); depthwise 7x7 positional conv; the learnt `gamma` residual is one gated-residual element-wise op; an MLP width that is
    not a multiple of 8 (int(1.2 c)) is zero-padded.
  * the class branch of the head is depthwise 3x3 + 1x1 twice.
"""
import torch

from ...hip import abi
from ...hip.plan import Act, PlanBuilder
from ...utils.exceptions import ModelError
from .yolo import YoloSegHip


def _count(sd, prefix):
    return len({k[len(prefix):].split(".")[0] for k in sd if k.startswith(prefix)})


class Yolo11Hip(YoloSegHip):
    def _derive(self, sd):
        if "model.6.m.0.0.attn.qkv.conv.weight" in sd:
            family, hi = "12", 21
        elif "model.10.m.0.attn.qkv.conv.weight" in sd:
            family, hi = "11", 23
        else:
            raise ModelError("not a YOLO11 / YOLO12 state dict (no C2PSA at model.10 and no A2C2f at model.6)")
        try:
            c = [sd[f"model.{i}.conv.weight"].shape[0] for i in (0, 1, 3, 5, 7)]
            nc = int(sd[f"model.{hi}.cv3.0.2.weight"].shape[0])
            seg = f"model.{hi}.cv4.0.2.weight" in sd
            nm = int(sd[f"model.{hi}.cv4.0.2.weight"].shape[0]) if seg else 0
            reg_max = int(sd[f"model.{hi}.cv2.0.2.weight"].shape[0]) // 4
        except KeyError as e:
            raise ModelError(f"incomplete YOLO{family} state dict (missing {e})") from e
        return dict(family=family, head=hi, c=c, nc=nc, nm=nm, seg=seg, reg_max=reg_max)

    # ---- weights ------------------------------------------------------------------------------------------------------
    def _pack(self, sd):
        self.W, self.DW, self.sd_shapes = {}, {}, {k: tuple(v.shape) for k, v in sd.items()}
        hi = self.a["head"]

        def put(name, w, b, cout_pad=0, cin_pad=0):
            co, ci, kh, kw = w.shape
            ci_p = max((ci + 7) // 8 * 8, cin_pad)
            co_p = max((co + 7) // 8 * 8, cout_pad)
            wt = torch.zeros(co_p, kh * kw, ci_p)
            wt[:co, :, :ci] = w.permute(0, 2, 3, 1).reshape(co, kh * kw, ci)
            bt = torch.zeros(co_p)
            bt[:co] = b
            self.W[name] = (wt.to(self.device, self.tdt).contiguous(), bt.to(self.device).contiguous(), co_p, kh)

        def put_dw(name, w, b):          # depthwise [C, 1, k, k] -> taps [k*k, C]
            c, _, kh, kw = w.shape
            self.DW[name] = (w.reshape(c, kh * kw).t().contiguous().to(self.device, self.tdt), b.to(self.device).contiguous(), kh)

        for k in sd:
            if not k.endswith(".conv.weight"):
                continue
            base = k[:-len(".conv.weight")]
            w, b = sd[k], sd[base + ".conv.bias"]
            if w.shape[1] == 1 and w.shape[0] > 1 and (base.endswith(".pe") or ".cv3." in base):       # depthwise convs: attention pe, head DWConv
                put_dw(base, w, b)
            elif base.endswith(".attn.qkv"):
                self._put_qkv(base, w, b)
            else:
                put(base, w, b)
        for l in range(3):
            for br, pad in (("cv2", 0), ("cv3", 8)) + ((("cv4", 0),) if self.a["seg"] else ()):
                base = f"model.{hi}.{br}.{l}.2"
                put(base, sd[base + ".weight"], sd[base + ".bias"], cout_pad=pad)
        if self.a["seg"]:
            up = sd[f"model.{hi}.proto.upsample.weight"]                      # [Cin, Cout, 2, 2]
            ci, co = up.shape[:2]
            put(f"model.{hi}.proto.upsample", up.permute(2, 3, 1, 0).reshape(4 * co, ci, 1, 1), sd[f"model.{hi}.proto.upsample.bias"].repeat(4))
        # MLP widths that are not a multiple of 8 (YOLO12: int(1.2 c)): the consumer's input channels are padded to match
        for name in list(self.W):
            if name.endswith(".mlp.1") or name.endswith(".ffn.1"):
                prod = self.W[name[:-1] + "0"]
                w, b, co, k = self.W[name]
                if w.shape[2] != prod[2]:
                    wp = torch.zeros(w.shape[0], w.shape[1], prod[2], dtype=w.dtype, device=w.device)
                    wp[:, :, : w.shape[2]] = w
                    self.W[name] = (wp.contiguous(), b, co, k)
        for k, v in sd.items():
            if k.endswith(".gamma"):
                self.W[k] = v.to(self.device, self.tdt).contiguous()

    def _put_qkv(self, base, w, b):
        """1x1 qkv conv re-packed to [Q heads | K heads | V heads] output columns, q / k zero-padded to the head dim"""
        cout, cin = w.shape[:2]
        w2, geo = w.reshape(cout, cin), None
        if self.a["family"] == "12" or ".attn.qkv" in base and cout == 3 * cin:      # area attention: per head q | k | v of head_dim 32
            hd = 32
            nh = cin // hd
            kd = hd
        if not (cout == 3 * cin and self.a["family"] == "12"):                         # PSA attention: per head q(kd) | k(kd) | v(hd), kd = hd / 2
            nh = max(cin // 64, 1)
            hd = cin // nh
            kd = (cout - cin) // (2 * nh)
        per = 2 * kd + hd
        wq, wk, wv = torch.zeros(nh * hd, cin), torch.zeros(nh * hd, cin), torch.zeros(nh * hd, cin)
        bq, bk, bv = torch.zeros(nh * hd), torch.zeros(nh * hd), torch.zeros(nh * hd)
        for h in range(nh):
            o = h * per
            wq[h * hd: h * hd + kd], bq[h * hd: h * hd + kd] = w2[o: o + kd], b[o: o + kd]
            wk[h * hd: h * hd + kd], bk[h * hd: h * hd + kd] = w2[o + kd: o + 2 * kd], b[o + kd: o + 2 * kd]
            wv[h * hd: (h + 1) * hd], bv[h * hd: (h + 1) * hd] = w2[o + 2 * kd: o + per], b[o + 2 * kd: o + per]
        wt = torch.cat([wq, wk, wv]).view(3 * nh * hd, 1, cin)
        self.W[base] = (wt.to(self.device, self.tdt).contiguous(), torch.cat([bq, bk, bv]).to(self.device).contiguous(), 3 * nh * hd, 1)
        self.W[base + "#geo"] = (nh, hd, kd)

    # ---- blocks ---------------------------------------------------------------------------------------------------------
    def _has(self, key):
        return key in self.sd_shapes

    def _bottleneck(self, pb, x, p, shortcut, out=None):
        t = self._conv(pb, x, p + ".cv1")
        return self._conv(pb, t, p + ".cv2", out=out, res=x if shortcut else None)

    def _c3k(self, pb, x, p, out=None):
        c_ = self.W[p + ".cv1"][2]
        cat = pb.act(x.n, x.h, x.w, 2 * c_)
        y = self._conv(pb, x, p + ".cv1")
        n = _count(self.sd_shapes, p + ".m.")
        for j in range(n):
            y = self._bottleneck(pb, y, f"{p}.m.{j}", True, out=cat.slice(0, c_) if j == n - 1 else None)
        self._conv(pb, x, p + ".cv2", out=cat.slice(c_, c_))
        return self._conv(pb, cat, p + ".cv3", out=out)

    def _c3k2(self, pb, x, i, out=None):
        p = f"model.{i}"
        n = _count(self.sd_shapes, p + ".m.")
        c = self.W[p + ".cv1"][2] // 2
        cat = pb.act(x.n, x.h, x.w, (2 + n) * c)
        self._conv(pb, x, p + ".cv1", out=cat.slice(0, 2 * c))
        for k in range(n):
            y, dst = cat.slice((1 + k) * c, c), cat.slice((2 + k) * c, c)
            if self._has(f"{p}.m.{k}.cv3.conv.weight"):
                self._c3k(pb, y, f"{p}.m.{k}", out=dst)
            else:
                self._bottleneck(pb, y, f"{p}.m.{k}", True, out=dst)
        return self._conv(pb, cat, p + ".cv2", out=out)

    def _attention(self, pb, x, p, area=1):
        """x + proj(attn(x) + pe(v)) — PSABlock / ABlock attention half; x is a dense [1, h, w, C] activation"""
        nh, hd, kd = self.W[p + ".qkv#geo"]
        C, N = nh * hd, x.h * x.w
        qkv = self._conv(pb, x, p + ".qkv", act=abi.ACT_NONE)                          # [1, h, w, 3C]: Q | K | V blocks
        o = pb.act(x.n, x.h, x.w, C)
        if N % area:
            raise ModelError(f"area attention: {N} positions do not split into {area} areas")
        nb = N // area
        # (a batch of images: image b's areas are launch batches b * area .. — an image is `area` consecutive chunks of nb positions)
        pb.attention(qkv.t, qkv.t, qkv.t, o.t, x.n * area, nh, nb, nb, hd, (nb * 3 * C, 3 * C, hd), (nb * 3 * C, 3 * C, hd), (nb * 3 * C, 3 * C, hd),
                     (nb * C, C, hd), float(kd) ** -0.5, k_off=C, v_off=2 * C, label=p + ".sdpa")
        w, b, k = self.DW[p + ".pe"]
        pe = pb.dwconv(qkv.slice(2 * C, C), w, b, k, label=p + ".pe")
        s = pb.ew(abi.EW_ADD, o, b=pe, label=p + ".add_pe")
        return self._conv(pb, s, p + ".proj", act=abi.ACT_NONE, res=x)

    def _mlp(self, pb, x, p0, p1, out=None):
        return self._conv(pb, self._conv(pb, x, p0), p1, act=abi.ACT_NONE, res=x, out=out)

    def _c2psa(self, pb, x, i):
        p = f"model.{i}"
        c = self.W[p + ".cv1"][2] // 2
        cat = self._conv(pb, x, p + ".cv1")
        n = _count(self.sd_shapes, p + ".m.")
        b = pb.ew(abi.EW_COPY, cat.slice(c, c), label=p + ".b")                      # dense copy of the attention half
        for k in range(n):
            b = self._attention(pb, b, f"{p}.m.{k}.attn")
            b = self._mlp(pb, b, f"{p}.m.{k}.ffn.0", f"{p}.m.{k}.ffn.1", out=cat.slice(c, c) if k == n - 1 else None)
        return self._conv(pb, cat, p + ".cv2")

    def _a2c2f(self, pb, x, i, area):
        p = f"model.{i}"
        c_ = self.W[p + ".cv1"][2]
        n = _count(self.sd_shapes, p + ".m.")
        cat = pb.act(x.n, x.h, x.w, (1 + n) * c_)
        y = self._conv(pb, x, p + ".cv1", out=cat.slice(0, c_))
        for r in range(n):
            dst = cat.slice((1 + r) * c_, c_)
            if self._has(f"{p}.m.{r}.0.attn.qkv.conv.weight"):
                t = pb.ew(abi.EW_COPY, y, label=f"{p}.m.{r}.in") if y.ld != y.c else y
                for j in range(2):
                    t = self._attention(pb, t, f"{p}.m.{r}.{j}.attn", area)
                    t = self._mlp(pb, t, f"{p}.m.{r}.{j}.mlp.0", f"{p}.m.{r}.{j}.mlp.1", out=dst if j == 1 else None)
            else:
                self._c3k(pb, y, f"{p}.m.{r}", out=dst)
            y = dst
        out = self._conv(pb, cat, p + ".cv2")
        if p + ".gamma" in self.W:
            g = self.W[p + ".gamma"]
            res = pb.act(x.n, x.h, x.w, out.c)
            e = abi.EwArgs()
            e.a, e.b, e.s, e.y = out.ptr, x.ptr, g.data_ptr(), res.ptr
            e.n, e.h, e.w, e.c = 1, 1, x.n * x.h * x.w, out.c
            e.lda, e.ldb, e.ldy, e.lds = out.ld, x.ld, res.ld, 0
            e.kind, e.act, e.act_param, e.i0, e.i1, e.dtype = abi.EW_GATE_RES, 0, 0.0, 0, 0, self.dtype
            pb._add(abi.OP_EW, e, p + ".gamma_res")
            return res
        return out

    def _cat(self, pb, parts, label):
        c = sum(t.c for t in parts)
        cat = pb.act(parts[0].n, parts[0].h, parts[0].w, c)
        off = 0
        for t in parts:
            pb.ew(abi.EW_COPY, t, out=cat.slice(off, t.c), label=label)
            off += t.c
        return cat

    def _sppf(self, pb, x, i):
        ch = self.W[f"model.{i}.cv1"][2]
        sp = pb.act(x.n, x.h, x.w, 4 * ch)
        self._conv(pb, x, f"model.{i}.cv1", out=sp.slice(0, ch))
        for k in range(3):
            pb.ew(abi.EW_MAXPOOL, sp.slice(k * ch, ch), out=sp.slice((k + 1) * ch, ch), i0=5, i1=1, label=f"sppf.pool{k}")
        return self._conv(pb, sp, f"model.{i}.cv2")

    def _build(self, lp, batch=1):
        """batch > 1: `batch` letterboxed images through one graph (core/ml/detector_batch.py) — every activation carries the image index as its
        outermost dimension, every kernel treats images independently (same tiles, same arithmetic per image as the batch-1 plan), and the
        head is decoded image by image into plan.decoded[b]"""
        a = self.a
        pb = PlanBuilder(self.lib, self.device, self.dtype)
        H, W = lp["H"], lp["W"]
        if H % 32 or W % 32:
            raise ModelError(f"letterboxed input {W}x{H} must be a multiple of 32")
        img = pb.act(batch, H, W, 8)
        up = lambda t, lab: pb.ew(abi.EW_UPSAMPLE2X, t, label=lab)
        x = self._conv(pb, self._conv(pb, img, "model.0", 2), "model.1", 2)
        p3 = self._c3k2(pb, self._conv(pb, self._c3k2(pb, x, 2), "model.3", 2), 4)
        if a["family"] == "11":
            p4 = self._c3k2(pb, self._conv(pb, p3, "model.5", 2), 6)
            p5 = self._c2psa(pb, self._sppf(pb, self._c3k2(pb, self._conv(pb, p4, "model.7", 2), 8), 9), 10)
            h4 = self._c3k2(pb, self._cat(pb, [up(p5, "up.p5"), p4], "cat12"), 13)
            h3 = self._c3k2(pb, self._cat(pb, [up(h4, "up.h4"), p3], "cat15"), 16)
            n4 = self._c3k2(pb, self._cat(pb, [self._conv(pb, h3, "model.17", 2), h4], "cat18"), 19)
            n5 = self._c3k2(pb, self._cat(pb, [self._conv(pb, n4, "model.20", 2), p5], "cat21"), 22)
        else:
            p4 = self._a2c2f(pb, self._conv(pb, p3, "model.5", 2), 6, 4)
            p5 = self._a2c2f(pb, self._conv(pb, p4, "model.7", 2), 8, 1)
            h4 = self._a2c2f(pb, self._cat(pb, [up(p5, "up.p5"), p4], "cat10"), 11, 1)
            h3 = self._a2c2f(pb, self._cat(pb, [up(h4, "up.h4"), p3], "cat13"), 14, 1)
            n4 = self._a2c2f(pb, self._cat(pb, [self._conv(pb, h3, "model.15", 2), h4], "cat16"), 17, 1)
            n5 = self._c3k2(pb, self._cat(pb, [self._conv(pb, n4, "model.18", 2), p5], "cat19"), 20)
        hi, nb, ncp, nm = a["head"], 4 * a["reg_max"], 8, a["nm"]
        heads, box32 = [], []
        for l, f in enumerate((h3, n4, n5)):
            hb = pb.act(batch, f.h, f.w, nb + ncp + nm)
            t = self._conv(pb, self._conv(pb, f, f"model.{hi}.cv2.{l}.0"), f"model.{hi}.cv2.{l}.1")
            box32.append(self._box_logits_f32(pb, t, f"model.{hi}.cv2.{l}.2", nb))       # (hb's first nb channels stay unwritten: the decode reads the fp32 logits)
            t = f
            for s in range(2):            # (depthwise 3x3, SiLU) -> (1x1, SiLU), twice
                w, b, k = self.DW[f"model.{hi}.cv3.{l}.{s}.0"]
                t = self._conv(pb, pb.dwconv(t, w, b, k, act=abi.ACT_SILU, label=f"head.cv3.{l}.{s}.dw"), f"model.{hi}.cv3.{l}.{s}.1")
            self._conv(pb, t, f"model.{hi}.cv3.{l}.2", act=abi.ACT_NONE, out=hb.slice(nb, ncp))
            if nm:
                t = self._conv(pb, self._conv(pb, f, f"model.{hi}.cv4.{l}.0"), f"model.{hi}.cv4.{l}.1")
                self._conv(pb, t, f"model.{hi}.cv4.{l}.2", act=abi.ACT_NONE, out=hb.slice(nb + ncp, nm))
            heads.append(hb)
        anchors = sum(hb.h * hb.w for hb in heads)
        if batch == 1:
            decoded = pb.buf((anchors, 4 + a["nc"] + nm), torch.float32)
            pb.yolo_decode(heads, [8, 16, 32], a["nc"], nm, a["reg_max"], decoded, cls_off=nb, mc_off=nb + ncp, box_f32=box32)
        else:
            if nm:
                raise ModelError("batched plans exist for the detect-only heads (panel / outside-text detectors)")
            decoded = pb.buf((batch, anchors, 4 + a["nc"] + nm), torch.float32)
            for b in range(batch):
                pb.yolo_decode(heads, [8, 16, 32], a["nc"], nm, a["reg_max"], decoded[b], cls_off=nb, mc_off=nb + ncp, box_f32=box32, image=b,
                               label=f"yolo_decode.{b}")
        proto = None
        if nm:
            t = self._conv(pb, h3, f"model.{hi}.proto.cv1")
            w, b, co, _ = self.W[f"model.{hi}.proto.upsample"]
            t = pb.conv2d(t, w, b, co, ksize=1, pixel_shuffle=2, label="proto.upsample")
            proto = self._conv(pb, self._conv(pb, t, f"model.{hi}.proto.cv2"), f"model.{hi}.proto.cv3")
        plan = pb.build()
        plan.img, plan.decoded, plan.proto = img, decoded, proto
        plan.dbg = dict(p3=p3, p4=p4, p5=p5, h3=h3, n4=n4, n5=n5)
        return plan
