"""Stage memo against the reference `UnifiedCache` (core/caching.py:12-658): every key byte-identical, and the
store behaviour (capacity 1 / 20, recency, page change) step by step — golden made by tests/golden/make_cache_goldens.py."""
import json
import threading
from pathlib import Path

import numpy as np
import torch
from PIL import Image

from mangatranslator_amd.core import caching

GOLD = json.loads((Path(__file__).resolve().parent / "golden" / "cache_keys.json").read_text())


def _images():
    rng = np.random.default_rng(2024)
    rgb = Image.fromarray(rng.integers(0, 256, (48, 32, 3), dtype=np.uint8), "RGB")
    rgba = Image.fromarray(rng.integers(0, 256, (40, 24, 4), dtype=np.uint8), "RGBA")
    gray = Image.fromarray(rng.integers(0, 256, (16, 20), dtype=np.uint8), "L")
    return dict(rgb=rgb, rgba=rgba, gray=gray, pal=rgb.convert("P"))


ARRS = dict(empty=np.zeros((0, 4), np.float32), boxes=np.asarray([[1.5, 2, 30, 40], [5, 6, 70, 80.25]], np.float32),
            mask=(np.arange(64 * 48).reshape(64, 48) % 7 == 0).astype(np.uint8), i64=np.arange(6).reshape(2, 3))


def test_digests_and_keys():
    c, im = caching.UnifiedCache(), _images()
    assert {k: c._hash_image(v) for k, v in im.items()} == GOLD["hash_image"]
    assert {k: c._hash_numpy(v) for k, v in ARRS.items()} == GOLD["hash_numpy"]
    for tag, path, conf, key in GOLD["yolo"]:
        assert c.get_yolo_cache_key(im[tag], path, conf) == key
    for tag, boxes, seg, cj, cc, key in GOLD["sam"]:
        b = dict(tensor=torch.from_numpy(ARRS["boxes"]), list=ARRS["boxes"].tolist(), empty=torch.tensor([]))[boxes]
        assert c.get_sam_cache_key(im[tag], b, seg, cj, cc) == key
    for tag, f, mt, key in GOLD["upscale"]:
        assert c.get_upscale_cache_key(im[tag], f, mt) == key
    for tag, t, mode, mt, key in GOLD["upscale_dim"]:
        assert c.get_upscale_dimension_cache_key(im[tag], t, mode, mt) == key
    for tag, t, mode, mt, key in GOLD["bubble_proc"]:
        assert c.get_bubble_processing_cache_key(im[tag], t, mode, mt) == key
    for tag, seed, steps, thr, gs, prompt, extra, key in GOLD["inpaint"]:
        if extra:
            extra = {k: tuple(v) if isinstance(v, list) else v for k, v in extra.items()}       # json turned the bbox tuples into lists
        assert c.get_inpaint_cache_key(im[tag], ARRS["mask"], seed, steps, thr, gs, prompt, extra) == key
    assert [[s, c.should_use_inpaint_cache(s)] for s in (-1, 0, 1, 42)] == GOLD["should_use"]
    assert len(c._hash_dict({"a": 1})) == 16


def test_store_trace():
    c, im = caching.UnifiedCache(), _images()
    pages = dict(page_a=im["rgb"], page_b=im["rgba"])
    it = iter(GOLD["trace"])

    def expect(op, *a):
        g_op, g_args, g_ret, g_stats = next(it)
        assert g_op == op
        r = getattr(c, op)(*a)
        assert (list(r) if isinstance(r, (list, tuple)) else r) == g_ret, (op, a)
        assert c.get_cache_stats() == g_stats, (op, a)

    expect("get_yolo_detection", "k1"); expect("set_yolo_detection", "k1", "det1"); expect("get_yolo_detection", "k1")
    expect("set_yolo_detection", "k2", "det2"); expect("get_yolo_detection", "k1")
    expect("set_sam_masks", "s1", [1, 2]); expect("get_sam_masks", "s1")
    for i in range(22):
        c.set_upscaled_image(f"u{i}", i)
    expect("get_upscaled_image", "u0"); expect("get_upscaled_image", "u1"); expect("get_upscaled_image", "u2")
    c.set_upscaled_image("u22", 22)
    expect("get_upscaled_image", "u3"); expect("get_upscaled_image", "u2")
    for i in range(21):
        c.set_inpainted_image(f"p{i}", i)
    expect("get_inpainted_image", "p0"); expect("get_inpainted_image", "p20")
    for page in (pages["page_a"], pages["page_a"].copy(), pages["page_b"]):
        g_op, _, g_ret, g_stats = next(it)
        assert g_op == "set_current_image" and c.set_current_image(page) == g_ret and c.get_cache_stats() == g_stats
    expect("set_yolo_detection", "k3", "det3"); expect("clear_yolo_cache"); expect("set_sam_masks", "s2", 5); expect("clear_all")
    assert next(it, None) is None


def test_global_instance_is_shared_across_threads():
    seen = []
    ts = [threading.Thread(target=lambda: seen.append(caching.get_cache())) for _ in range(8)]
    [t.start() for t in ts]; [t.join() for t in ts]
    assert all(s is seen[0] for s in seen) and seen[0] is caching.get_cache()


def test_inpaint_memo_flow():
    """`inpaint_mask` with the memo: same key as the reference computes for the same page / mask / settings, the pipeline runs once per
    key, a hit composites the same bytes, seed -1 never remembers."""
    import hashlib
    from test_host_goldens import _inpainter
    inp = _inpainter()
    calls = []
    plain = inp.pipeline
    inp.pipeline = lambda **kw: (calls.append(kw["image"].size), plain(**kw))[1]
    keys = []
    real = inp.cache.get_inpaint_cache_key
    inp.cache.get_inpaint_cache_key = lambda *a: (keys.append(real(*a)), keys[-1])[1]
    masks = np.load(Path(__file__).resolve().parent / "golden" / "cache_inpaint_masks.npz")
    for ci, row in enumerate(GOLD["inpaint_flow"]):
        h, w = row["h"], row["w"]
        m = np.unpackbits(masks[f"mask{ci}"])[: h * w].reshape(h, w).astype(bool)
        yy, xx = np.mgrid[0:h, 0:w]
        page = Image.fromarray(np.stack([(xx * 3 + yy) % 256, (xx + yy * 2) % 256, (xx * yy // 7) % 256], -1).astype(np.uint8))
        for ev in row["events"]:
            nk, nc = len(keys), len(calls)
            out = inp.inpaint_mask(page, m, seed=row["seed"], ocr_params=row["ocr"], strict_mask_clipping=row["strict"], composite_clip_bbox=row["clip"])
            assert (keys[-1] if len(keys) > nk else None) == ev["key"], f"case {ci}"
            assert (len(calls) > nc) == ev["pipeline_ran"]
            assert hashlib.sha256(np.asarray(out).tobytes()).hexdigest()[:16] == ev["out_sha"]
            assert inp.cache.get_cache_stats()["inpaint"] == ev["stats"]


def test_detection_memo_flow(emu_lib, monkeypatch):
    """`detect_speech_bubbles` over a script of calls: the primary detector is remembered per (page, model, confidence), RT-DETR always
    runs, finished SAM detections per (page, prompt boxes, seg model, conjoined settings) — and a hit hands back the remembered list
    itself, also when only the detector confidence changed.  Run counts and identities from the reference with its real memo."""
    import types
    from mangatranslator_amd.core.image import detection
    import mangatranslator_amd.hip.lib as libmod
    inp = json.loads((Path(__file__).resolve().parent / "golden" / "detection_flow.json").read_text())["inputs"]
    H, W = inp["H"], inp["W"]
    counts = dict(primary=0, secondary=0, sam=0)

    class Boxes:
        def __init__(self, xyxy, conf, cls):
            self.xyxy, self.conf, self.cls = (torch.tensor(v, dtype=torch.float32) for v in (xyxy, conf, cls))

        def __len__(self):
            return len(self.xyxy)

    class Model:
        def __init__(self, tag, result, names):
            self.tag, self.result, self.names = tag, result, names

        def __call__(self, *a, **k):
            counts[self.tag] += 1
            return [self.result]

    names2 = {int(k): v for k, v in inp["names"].items()}
    pm = Model("primary", types.SimpleNamespace(boxes=Boxes(inp["primary"], inp["pconf"], [0] * 6), masks=None, orig_shape=(H, W)), {0: "speech_bubble"})
    sm = Model("secondary", types.SimpleNamespace(boxes=Boxes(inp["secondary"], inp["sconf"], inp["scls"]), names=names2), names2)

    class Inputs(dict):
        def to(self, *a, **k):
            return self

    class Proc:
        def __call__(self, image, input_boxes=None, return_tensors="pt"):
            return Inputs(boxes=torch.as_tensor(input_boxes, dtype=torch.float32).reshape(-1, 4), original_sizes=torch.tensor([[H, W]]))

        def post_process_masks(self, pred, sizes, **kw):
            return [pred]

    def sam(multimask_output=False, **inputs):
        counts["sam"] += 1
        yy, xx = np.mgrid[0:H, 0:W]
        ms = [((xx - (x0 + x1) / 2) / ((x1 - x0) / 2 * 1.08)) ** 2 + ((yy - (y0 + y1) / 2) / ((y1 - y0) / 2 * 1.08)) ** 2 <= 1.0 for x0, y0, x1, y1 in inputs["boxes"].tolist()]
        return types.SimpleNamespace(pred_masks=torch.from_numpy(np.stack(ms))[:, None].float())

    def no_osb(*a, **k):
        raise RuntimeError("not staged")
    mgr = types.SimpleNamespace(load_yolo_speech_bubble=lambda *a, **k: pm, load_rtdetr_conjoined_bubble=lambda *a, **k: sm,
                                load_sam2=lambda *a, **k: (Proc(), sam), load_yolo_osbtext=no_osb, device="cpu")
    monkeypatch.setattr(detection, "get_model_manager", lambda: mgr)
    monkeypatch.setattr(libmod, "_lib", emu_lib)
    a = Image.fromarray((np.random.default_rng(3).random((H, W, 3)) * 255).astype(np.uint8))
    pages = dict(a=a, b=a.transpose(Image.FLIP_LEFT_RIGHT))
    results = []
    for row in GOLD["detection_memo"]:
        page, seg, conf, cconf = row["call"]
        dets, _ = detection.detect_speech_bubbles(Path("page.png"), "yolo_2", confidence=conf, device="cpu", seg_model=seg, conjoined_detection=True,
                                                  conjoined_confidence=cconf, image_override=pages[page])
        assert counts == row["counts"], row["call"]
        assert next((i for i, r in enumerate(results) if r is dets), None) == row["same_list_as_call"]
        results.append(dets)
        stats = caching.get_cache().get_cache_stats()
        assert len(dets) == row["n"] and (stats["yolo"], stats["sam"]) == (row["stats"]["yolo"], row["stats"]["sam"])


def test_upscale_memo_flow(monkeypatch):
    import types
    from mangatranslator_amd.core.image import image_utils as iu
    from test_image_utils import _fake_upscaler
    passes = [0]

    def model(t):
        passes[0] += 1
        return _fake_upscaler(t)
    mgr = types.SimpleNamespace(load_upscale=lambda *a, **k: model, load_upscale_lite=lambda *a, **k: model, device=torch.device("cpu"))
    monkeypatch.setattr(iu, "get_model_manager", lambda: mgr)
    rng = np.random.default_rng(9)
    pages = dict(a=Image.fromarray(rng.integers(0, 256, (20, 14, 3), dtype=np.uint8)), b=Image.fromarray(rng.integers(0, 256, (12, 18, 3), dtype=np.uint8)))
    results = []
    for row in GOLD["upscale_memo"]:
        page, factor, mt = row["call"]
        r = iu.upscale_image(pages[page], factor, mt)
        assert passes[0] == row["passes"] and list(r.size) == row["size"], row["call"]
        assert next((i for i, x in enumerate(results) if x is r), None) == row["same_object_as_call"]
        results.append(r)
        assert caching.get_cache().get_cache_stats()["upscale"] == row["stats"]


def test_pixels_scope_digests_an_object_once(monkeypatch):
    c, im = caching.UnifiedCache(), _images()
    n = [0]
    plain = caching.UnifiedCache._digest_image
    monkeypatch.setattr(caching.UnifiedCache, "_digest_image", staticmethod(lambda image: (n.__setitem__(0, n[0] + 1), plain(image))[1]))
    k1 = c.get_yolo_cache_key(im["rgb"], "a.pt", 0.6), c.get_sam_cache_key(im["rgb"], [[1, 2, 3, 4]], "sam2")
    assert n[0] == 2
    with c.pixels_scope():
        k2 = c.get_yolo_cache_key(im["rgb"], "a.pt", 0.6), c.get_sam_cache_key(im["rgb"], [[1, 2, 3, 4]], "sam2")
        with c.pixels_scope():                                   # nested (upscale_image -> upscale_image_to_dimension): same memo
            c.get_upscale_cache_key(im["rgb"], 2.0)
        c.get_upscale_cache_key(im["rgb"].copy(), 2.0)           # another object with the same pixels is digested on its own
    assert k1 == k2 and n[0] == 4
    c.get_upscale_cache_key(im["rgb"], 2.0)                      # outside a scope nothing is remembered
    assert n[0] == 5
    seen = []
    t = threading.Thread(target=lambda: seen.append(getattr(c._tls, "digests", None)))
    with c.pixels_scope():
        t.start(); t.join()
    assert seen == [None]                                        # scopes are per thread
