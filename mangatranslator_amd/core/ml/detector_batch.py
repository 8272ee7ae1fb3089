"""Cross-page batches through a 640-pixel detector graph ("batch B pages through each 640-px detector graph").

The panel and outside-text detectors the reference runs on every page (`core/image/detection.py:1867-1873`, `core/image/ocr_detection.py:425-431`:
`model(img, imgsz=640)`, one page per call) are launch-latency-bound on an MI355X: YOLO11-L at 640 x 448 is 192 launches in 4.5 ms, and the same
graph over FOUR times the pixels takes 5.2 ms (`profiles/r06_detector_batch_probe.log`).  The reference has no cross-page batch (its batch mode is
threads over one GPU, `core/pipeline.py:2343`); this wrapper forms one without changing a caller: it keeps the model's `submit` / `collect` /
`__call__` shape, and the front halves of several pages — each on its own thread, as `batch_process_images` and `bench.py` run them — share ONE
wrapper per detector.

  submit(page)   takes the next free slot of the batch that is being filled: the page is copied into the slot's buffer and letterboxed on the
                 wrapper's stream right away; the network itself is NOT queued yet.  The batch is launched when its last slot is taken ...
  collect(t)     ... or when the first of its tickets is collected (a page never waits for pages that may not come).  Each ticket then runs
                 the model's own per-image second half (`_finish`: candidates, NMS, page coordinates) on its slice of the decoded rows.

Two buffer sets alternate, so the next batch fills while the previous one is collected.  Every kernel of a batched plan treats the images
independently — same tiles, same arithmetic per image as the one-image plan — so a page's boxes do not depend on what else was in its batch
(`tests/test_yolo11_sim.py`, `tests/test_yolo11_gpu.py`: byte-identical to the one-image call)."""
import threading
from types import SimpleNamespace

import numpy as np
import torch

from mangatranslator_amd.hip.plan import PlanBuilder, PlanCache, result_tensors

from .yolo import letterbox_params


class _Batch:
    """one buffer set: the batched network plan, a page buffer + letterbox plan per slot, and the bookkeeping of a fill / launch / collect cycle"""

    def __init__(self, model, key, lp, size):
        h0, w0, _ = key
        self.plan = model._build(lp, batch=size)
        self.pre = []
        for b in range(size):
            pre = PlanBuilder(model.lib, model.device, model.dtype)
            page = pre.buf((h0, w0, 3), torch.uint8)
            slot = type(self.plan.img)(self.plan.img.t[b:b + 1], 1, self.plan.img.h, self.plan.img.w, self.plan.img.c, self.plan.img.c0)
            pre.letterbox(page, slot, h0, w0, lp["nh"], lp["nw"], lp["top"], lp["left"])
            pp = pre.build()
            pp.page = page
            self.pre.append(pp)
        self.size, self.lp = size, lp
        self.filled = self.collected = 0
        self.launched = False
        # the first graph replay of a plan CAPTURES it, and a capture must not run beside other threads' GPU work (another front half's allocation or
        # synchronising read ends it: "hipStreamBeginCapture: the operation cannot be performed in the present state", first hardware run).  Sets
        # are built where a model's other plans are — on the first page of a size, i.e. during the sequential set-up pages of a batch run.
        with model._lane.enter():
            self.plan.run(graph=model._graph)
        if model._lane.on:
            model._lane.stream.synchronize()

    def close(self):
        for p in [self.plan] + self.pre:
            if hasattr(p, "close"):
                p.close()


class BatchTicket(dict):
    """what `DetectorBatcher.submit` returns; collecting (or closing) it gives its slot back"""

    def __init__(self, owner, batch, slot, **fields):
        super().__init__(**fields)
        self._owner, self.batch, self.slot, self._open = owner, batch, slot, True

    def close(self):
        if self._open:
            self._open = False
            self._owner._slot_done(self.batch)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):
        try:
            self.close()
        except Exception:      # noqa: BLE001 — interpreter shutdown
            pass


class DetectorBatcher:
    """`model` is a detect-only Yolo11Hip (panel / outside-text detector); `batch` pages share a graph replay.  Thread-safe; one instance is shared
    by the front halves that run side by side.  Pages of another size than the batch being filled start their own batch."""

    WAIT_S = 120.0
    # `model(page)` — the reference's call shape: submit and collect in one go — would launch every batch with one page.  With `peers` > 1 front
    # halves sharing the wrapper the call lingers this long for companions before it launches what it has (a third of one replay; pages that
    # left a batch together reach the next detector together, so after the first few pages the wait is rarely used up)
    LINGER_S = 0.0015

    def __init__(self, model, batch: int = 4, peers: int = 1):
        if getattr(model, "a", None) is not None and model.a["nm"]:
            raise ValueError("DetectorBatcher: detect-only heads (the segmentation head's masks are per page)")
        self.model, self.batch, self.peers = model, max(1, int(batch)), max(1, int(peers))
        self.names = model.names
        self._lane = model._lane
        self._cv = threading.Condition()
        self._sets = PlanCache(4)           # (h0, w0, imgsz) -> [_Batch, _Batch]
        self._filling = {}                  # key -> the _Batch that is taking slots
        self.stats = dict(pages=0, launches=0)

    # ---- the reference's call shape ----------------------------------------------------------------------------------------------------
    def __call__(self, image_bgr, conf=0.25, device=None, verbose=False, imgsz=640, iou=0.7, max_det=300, **_kw):
        ticket = self.submit(image_bgr, conf=conf, imgsz=imgsz, iou=iou, max_det=max_det)
        if self.peers > 1:
            with self._cv:
                want = min(ticket.batch.size, self.peers)
                self._cv.wait_for(lambda: ticket.batch.launched or ticket.batch.filled >= want, timeout=self.LINGER_S)
        return self.collect(ticket)

    @torch.no_grad()
    def submit(self, image_bgr, conf=0.25, imgsz=640, iou=0.7, max_det=300, **_kw):
        m = self.model
        on_device = torch.is_tensor(image_bgr) and image_bgr.device.type == m.device.type and m.device.type != "cpu"
        img = image_bgr[..., :3] if on_device else np.ascontiguousarray(np.asarray(image_bgr)[..., :3])
        h0, w0 = int(img.shape[0]), int(img.shape[1])
        key = (h0, w0, imgsz)
        with self._cv:
            b = self._take_slot(key, letterbox_params(h0, w0, imgsz))
            slot = b.filled
            b.filled += 1
            try:
                if on_device:
                    self._lane.adopt(image_bgr)
                with self._lane.enter():
                    pp = b.pre[slot]
                    pp.page.copy_(img if on_device else torch.from_numpy(img).to(m.device, non_blocking=True))
                    pp.run()
                    if b.filled == b.size:
                        self._launch(b, key)
            except BaseException:
                self._slot_done(b)           # the slot was taken but no ticket will ever give it back: count it as dropped, or the buffer set stays busy for good
                raise
            self.stats["pages"] += 1
            self._cv.notify_all()
        return BatchTicket(self, b, slot, key=key, hw=(h0, w0), conf=conf, iou=iou, max_det=max_det)

    @torch.no_grad()
    def collect(self, ticket):
        b = ticket.batch
        try:
            with self._cv:
                if not b.launched:          # nobody else is coming in time: the batch goes with the slots it has
                    with self._lane.enter():
                        self._launch(b, ticket["key"])
            with self._lane.resume():
                view = SimpleNamespace(decoded=b.plan.decoded[ticket.slot] if b.size > 1 else b.plan.decoded, proto=None)      # (a one-image plan has no image axis)
                res = self.model._finish(view, b.lp, ticket["hw"], ticket["conf"], ticket["iou"], ticket["max_det"])
            self._lane.hand_over(*result_tensors(res))
            return res
        finally:
            ticket.close()

    # ---- internals (called with the condition held) ---------------------------------------------------------------------------------------
    def _take_slot(self, key, lp):
        b = self._filling.get(key)
        if b is not None and not b.launched and b.filled < b.size:
            return b
        if key not in self._sets:
            self._sets[key] = [self._new_set(key, lp), self._new_set(key, lp)]
        sets = self._sets[key]
        ok = self._cv.wait_for(lambda: any(s.filled == 0 for s in sets), timeout=self.WAIT_S)
        if not ok:
            raise RuntimeError(f"detector batch busy for {self.WAIT_S:.0f} s: tickets of an earlier batch were neither collected nor closed")
        b = next(s for s in sets if s.filled == 0)
        self._filling[key] = b
        return b

    def _new_set(self, key, lp):
        return _Batch(self.model, key, lp, self.batch)

    def _launch(self, b, key):
        self._run(b)
        b.launched = True
        self.stats["launches"] += 1
        if self._filling.get(key) is b:
            del self._filling[key]

    def _run(self, b):
        b.plan.run(graph=self.model._graph)

    def _slot_done(self, b):
        with self._cv:
            b.collected += 1
            if b.collected >= b.filled:          # every page that took a slot has its results (or dropped its ticket): the buffer set is free again
                b.filled = b.collected = 0
                b.launched = False
                self._cv.notify_all()


class _RTDetrBatch:
    """one buffer set of the RT-DETR batcher: the batched backbone + encoder plan, ONE decoder plan that the images of a batch go through one after
    the other on the wrapper's stream, and the per-slot results of the last launch"""

    def __init__(self, model, size, hw):
        H, W = hw
        self.enc = model._build(H, W, batch=size)
        self.dec = model._build_decoder(self.enc.shapes, self.enc.S)
        self.size = size
        self.filled = self.collected = 0
        self.launched = False
        self.meta = [None] * size          # per slot: (original height, width, confidence)
        self.out = [None] * size           # per slot after the launch: the ticket fields of RTDetrHip.submit
        with model._lane.enter():          # capture both graphs now (see _Batch)
            self.enc.run(graph=model._graph)
            self.dec.run(graph=model._graph)
        if model._lane.on:
            model._lane.stream.synchronize()

    def close(self):
        for p in (self.enc, self.dec):
            if hasattr(p, "close"):
                p.close()


class RTDetrBatcher(DetectorBatcher):
    """The same sharing for the RT-DETR-v2 secondary detector (`core/ml/rtdetr.py`): the backbone + hybrid encoder — two thirds of its GPU time, all
    small convolutions — runs once for the pages of a batch (`RTDetrHip._build(H, W, batch=B)`); query selection and the six decoder layers then
    follow image by image on the same stream with the model's own arithmetic (`_enqueue`'s second half), so each page's boxes are the one-page call's."""

    def __init__(self, model, batch: int = 4, peers: int = 1):
        super().__init__(model, batch, peers)

    @torch.no_grad()
    def submit(self, source, conf: float = 0.35, imgsz=None, **_kw):
        from PIL import Image
        m = self.model
        if isinstance(source, Image.Image):
            pil = source.convert("RGB") if source.mode != "RGB" else source
        elif isinstance(source, np.ndarray):
            arr = source
            if arr.ndim == 2:
                arr = np.stack([arr] * 3, -1)
            pil = Image.fromarray(np.ascontiguousarray(arr[..., :3][..., ::-1]))      # cv2 BGR -> RGB, as the adapter does
        else:
            pil = Image.open(source).convert("RGB")
        ow, oh = pil.size
        size = int(imgsz) if imgsz is not None else 640
        img = np.asarray(pil.resize((size, size), resample=Image.Resampling.BILINEAR))        # RTDetrImageProcessor: resize + 1/255 (host side, like the model's own submit)
        key = (size, size, "rtdetr")
        with self._cv:
            b = self._take_slot(key, None)
            slot = b.filled
            b.filled += 1
            b.meta[slot] = (oh, ow, float(conf))
            try:
                with self._lane.enter():
                    b.enc.src[slot].copy_(torch.from_numpy(np.array(img, dtype=np.uint8)).to(m.device, non_blocking=True))
                    if b.filled == b.size:
                        self._launch(b, key)
            except BaseException:
                self._slot_done(b)
                raise
            self.stats["pages"] += 1
            self._cv.notify_all()
        return BatchTicket(self, b, slot, key=key)

    @torch.no_grad()
    def collect(self, ticket):
        from .rtdetr import _Boxes
        b = ticket.batch
        try:
            with self._cv:
                if not b.launched:
                    with self._lane.enter():
                        self._launch(b, ticket["key"])
            with self._lane.resume():
                t = b.out[ticket.slot]
                keep = t["keep"]
                res = [SimpleNamespace(boxes=_Boxes(t["xyxy"][keep].float(), t["top_s"][keep].float(), t["labels"][keep].float()), names=self.model.names,
                                       orig_shape=t["hw"], masks=None)]
            self._lane.hand_over(*result_tensors(res))
            return res
        finally:
            ticket.close()

    def _new_set(self, key, lp):
        return _RTDetrBatch(self.model, self.batch, key[:2])

    def _run(self, b):
        """on the wrapper's stream: the batched encoder graph, then per filled slot what `RTDetrHip._enqueue` / `submit` queue for one image"""
        m, a, d = self.model, b.enc, b.dec
        cfg = m.cfg
        nc, Q, S = cfg.num_labels, cfg.num_queries, a.S
        a.run(graph=m._graph)
        for slot in range(b.filled):
            rows = slice(slot * S, (slot + 1) * S)
            top = a.scores[rows, :nc].max(-1).values.topk(Q, dim=0).indices
            d.mem.copy_(a.mem[rows])
            d.h0.copy_(a.om[rows].index_select(0, top))
            d.ref_logit.copy_((a.boxes[rows] + a.anchors).index_select(0, top))
            d.run(graph=m._graph)
            logits, boxes = d.logits[:, :nc].clone(), d.boxes[:, :4].clone()
            oh, ow, conf = b.meta[slot]
            scores = logits.sigmoid()
            k = min(Q, scores.numel())
            top_s, idx = scores.flatten().topk(k)
            labels, qi = idx % nc, idx // nc
            cx, cy, w, h = boxes[qi].unbind(-1)
            scale = torch.tensor([ow, oh, ow, oh], dtype=boxes.dtype).to(boxes.device, non_blocking=True)
            xyxy = torch.stack([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], -1) * scale
            b.out[slot] = dict(xyxy=xyxy, top_s=top_s, labels=labels, keep=top_s > conf, hw=(oh, ow), logits=logits, boxes=boxes)
