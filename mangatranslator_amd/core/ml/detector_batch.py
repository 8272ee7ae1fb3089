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
        if model.a["nm"]:
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
            if on_device:
                self._lane.adopt(image_bgr)
            with self._lane.enter():
                pp = b.pre[slot]
                pp.page.copy_(img if on_device else torch.from_numpy(img).to(m.device, non_blocking=True))
                pp.run()
                if b.filled == b.size:
                    self._launch(b, key)
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
                view = SimpleNamespace(decoded=b.plan.decoded[ticket.slot], proto=None)
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
            self._sets[key] = [_Batch(self.model, key, lp, self.batch), _Batch(self.model, key, lp, self.batch)]
        sets = self._sets[key]
        ok = self._cv.wait_for(lambda: any(s.filled == 0 for s in sets), timeout=self.WAIT_S)
        if not ok:
            raise RuntimeError(f"detector batch busy for {self.WAIT_S:.0f} s: tickets of an earlier batch were neither collected nor closed")
        b = next(s for s in sets if s.filled == 0)
        self._filling[key] = b
        return b

    def _launch(self, b, key):
        b.plan.run(graph=self.model._graph)
        b.launched = True
        self.stats["launches"] += 1
        if self._filling.get(key) is b:
            del self._filling[key]

    def _slot_done(self, b):
        with self._cv:
            b.collected += 1
            if b.collected >= b.filled:          # every page that took a slot has its results (or dropped its ticket): the buffer set is free again
                b.filled = b.collected = 0
                b.launched = False
                self._cv.notify_all()
