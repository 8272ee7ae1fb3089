"""Golden vectors for the stage memo: keys and store behaviour of the REFERENCE `UnifiedCache`
(core/caching.py:12-658), produced by importing it here (third-party modules its package pulls in are stubbed).

    python tests/golden/make_cache_goldens.py        # rewrites tests/golden/cache_keys.json
"""
import importlib.machinery
import importlib.util
import json
import sys
import types
from pathlib import Path
from unittest.mock import MagicMock

import numpy as np
import torch
from PIL import Image

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
import make_goldens as mg  # noqa: E402  (sets up the stubbed import of the reference package; its __main__ part does not run)
for name in ["fontTools", "fontTools.ttLib"]:
    try:
        if importlib.util.find_spec(name) is not None:
            continue
    except (ImportError, ValueError):
        pass
    stub = MagicMock(name=name)
    stub.__spec__ = importlib.machinery.ModuleSpec(name, None)
    sys.modules.setdefault(name, stub)
from core.caching import UnifiedCache  # noqa: E402


def images():
    """the shared inputs, rebuilt identically by tests/test_caching.py"""
    rng = np.random.default_rng(2024)
    rgb = Image.fromarray(rng.integers(0, 256, (48, 32, 3), dtype=np.uint8), "RGB")
    rgba = Image.fromarray(rng.integers(0, 256, (40, 24, 4), dtype=np.uint8), "RGBA")
    gray = Image.fromarray(rng.integers(0, 256, (16, 20), dtype=np.uint8), "L")
    pal = rgb.convert("P")
    return dict(rgb=rgb, rgba=rgba, gray=gray, pal=pal)


def main():
    c = UnifiedCache()
    im = images()
    out = dict(hash_image={k: c._hash_image(v) for k, v in im.items()})
    arrs = dict(empty=np.zeros((0, 4), np.float32), boxes=np.asarray([[1.5, 2, 30, 40], [5, 6, 70, 80.25]], np.float32),
                mask=(np.arange(64 * 48).reshape(64, 48) % 7 == 0).astype(np.uint8), i64=np.arange(6).reshape(2, 3))
    out["hash_numpy"] = {k: c._hash_numpy(v) for k, v in arrs.items()}
    out["yolo"] = [[tag, path, conf, c.get_yolo_cache_key(im[tag], path, conf)]
                   for tag, path, conf in [("rgb", "models/yolo/a.pt", 0.6), ("rgb", "models/yolo/a.pt", 0.6004), ("rgb", "models/yolo/a.pt", 0.6006),
                                           ("rgba", "b.pt", 0.35), ("gray", "", 1.0)]]
    out["sam"] = []
    for tag, boxes, seg, cj, cc in [("rgb", "tensor", "sam2", True, 0.35), ("rgb", "list", "sam2", True, 0.35), ("rgb", "tensor", "yolo", False, 0.5),
                                    ("rgba", "empty", "sam3", True, 0.35), ("rgb", "tensor", "other", True, 0.3504)]:
        b = dict(tensor=torch.from_numpy(arrs["boxes"]), list=arrs["boxes"].tolist(), empty=torch.tensor([]))[boxes]
        out["sam"].append([tag, boxes, seg, cj, cc, c.get_sam_cache_key(im[tag], b, seg, cj, cc)])
    out["upscale"] = [[tag, f, mt, c.get_upscale_cache_key(im[tag], f, mt)] for tag, f, mt in [("rgb", 2.0, "model"), ("rgb", 1.5004, "model_lite"), ("pal", 4, "model")]]
    out["upscale_dim"] = [[tag, t, mode, mt, c.get_upscale_dimension_cache_key(im[tag], t, mode, mt)]
                          for tag, t, mode, mt in [("rgb", 2048, "max", "model"), ("gray", 640, "min", "model_lite")]]
    out["bubble_proc"] = [[tag, t, mode, mt, c.get_bubble_processing_cache_key(im[tag], t, mode, mt)] for tag, t, mode, mt in [("rgb", 512, "min", "model")]]
    out["inpaint"] = []
    for tag, seed, steps, thr, gs, prompt, extra in [
            ("rgb", 1, 20, 0.12, 2.5, "Remove all text.", None),
            ("rgb", 1, 20, 0.12, 2.5, "Remove all text.", {"bbox": (8, 16, 64, 48), "padding": 1.5, "blur": 3, "backend": "sdnq"}),
            ("rgba", 7, 8, 0.1204, 2.504, "x", {"bbox": (0, 0, 16, 16), "padding": 2, "blur": 0, "backend": "sdnq", "strict_clip": True, "clip_bbox": (1, 2, 3, 4), "min_size": 200})]:
        out["inpaint"].append([tag, seed, steps, thr, gs, prompt, extra, c.get_inpaint_cache_key(im[tag], arrs["mask"], seed, steps, thr, gs, prompt, extra)])
    out["should_use"] = [[s, c.should_use_inpaint_cache(s)] for s in (-1, 0, 1, 42)]

    # store behaviour: a script of operations, the stats and lookups after each
    trace = []
    c = UnifiedCache()
    def step(op, *a):
        r = getattr(c, op)(*a)
        if isinstance(r, Image.Image):
            r = ["image", r.size]
        trace.append([op, [x if not isinstance(x, Image.Image) else "image" for x in a], r, c.get_cache_stats()])
    page_a, page_b = im["rgb"], im["rgba"]
    step("get_yolo_detection", "k1")
    step("set_yolo_detection", "k1", "det1")
    step("get_yolo_detection", "k1")
    step("set_yolo_detection", "k2", "det2")                 # size 1: k1 evicted
    step("get_yolo_detection", "k1")
    step("set_sam_masks", "s1", [1, 2])
    step("get_sam_masks", "s1")
    for i in range(22):
        c.set_upscaled_image(f"u{i}", i)
    step("get_upscaled_image", "u0")
    step("get_upscaled_image", "u1")
    step("get_upscaled_image", "u2")                          # touched: survives the next put
    c.set_upscaled_image("u22", 22)
    step("get_upscaled_image", "u3")
    step("get_upscaled_image", "u2")
    for i in range(21):
        c.set_inpainted_image(f"p{i}", i)
    step("get_inpainted_image", "p0")
    step("get_inpainted_image", "p20")
    trace.append(["set_current_image", ["page_a"], c.set_current_image(page_a), c.get_cache_stats()])
    trace.append(["set_current_image", ["page_a"], c.set_current_image(page_a.copy()), c.get_cache_stats()])
    trace.append(["set_current_image", ["page_b"], c.set_current_image(page_b), c.get_cache_stats()])
    step("set_yolo_detection", "k3", "det3")
    step("clear_yolo_cache")
    step("set_sam_masks", "s2", 5)
    step("clear_all")
    out["trace"] = trace
    out["inpaint_flow"] = inpaint_flow()
    out["detection_memo"] = detection_memo()
    out["upscale_memo"] = upscale_memo()
    out["osb_memo"] = osb_memo()
    json.dump(out, open(HERE / "cache_keys.json", "w"), indent=0)
    np.savez_compressed(HERE / "cache_inpaint_masks.npz", **MASKS)
    print("wrote cache_keys.json", len(json.dumps(out)))


MASKS = {}

DETECT_SCRIPT = [                     # (page, seg_model, confidence, conjoined_confidence) — shared with tests/test_caching.py through the json
    ("a", "sam2", 0.6, 0.35), ("a", "sam2", 0.6, 0.35), ("a", "yolo", 0.6, 0.35), ("a", "sam2", 0.5, 0.35), ("a", "sam2", 0.5, 0.5),
    ("b", "sam2", 0.5, 0.5), ("a", "sam2", 0.5, 0.5), ("a", "sam2", 0.5, 0.5)]


def detection_memo():
    """reference `detect_speech_bubbles` (core/image/detection.py:1263-1816) with the REAL memo on the canned rig of make_goldens.py: how
    often each model runs over a script of calls, and which calls hand back the remembered list itself"""
    mg.gen_detection_flow()                                  # leaves the canned model manager + cv2 shim installed in the module
    det = mg.detection
    mgr = det.get_model_manager()
    mgr.load_yolo_osbtext = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("not staged"))
    counts = dict(primary=0, secondary=0, sam=0)
    pm, sm = mgr.load_yolo_speech_bubble(), mgr.load_rtdetr_conjoined_bubble()
    pf, sf = pm.fn, sm.fn
    pm.fn = lambda *a, **k: (counts.__setitem__("primary", counts["primary"] + 1), pf(*a, **k))[1]
    sm.fn = lambda *a, **k: (counts.__setitem__("secondary", counts["secondary"] + 1), sf(*a, **k))[1]
    sam = mgr.load_sam2()[1]
    plain = type(sam).__call__
    type(sam).__call__ = lambda self, **k: (counts.__setitem__("sam", counts["sam"] + 1), plain(self, **k))[1]
    cache = UnifiedCache()
    det.get_cache = lambda: cache
    inp = mg.detection_flow_inputs()
    rng = np.random.default_rng(3)
    a = Image.fromarray((rng.random((inp["H"], inp["W"], 3)) * 255).astype(np.uint8))
    pages = dict(a=a, b=a.transpose(Image.FLIP_LEFT_RIGHT))
    rows, results = [], []
    for page, seg, conf, cconf in DETECT_SCRIPT:
        dets, _ = det.detect_speech_bubbles(Path("page.png"), "yolo_2", confidence=conf, device="cpu", seg_model=seg, conjoined_detection=True,
                                            conjoined_confidence=cconf, image_override=pages[page])
        same_as = next((i for i, r in enumerate(results) if r is dets), None)
        results.append(dets)
        rows.append(dict(call=[page, seg, conf, cconf], counts=dict(counts), same_list_as_call=same_as, n=len(dets), stats=cache.get_cache_stats()))
    return rows


def osb_memo():
    """reference `OutsideTextDetector.detect_outside_text` (core/image/ocr_detection.py:189-540) with the REAL memo on the canned rig of
    make_goldens.gen_osb, over the script of tests/test_osb_regions.py: how often each detector runs"""
    mg.gen_osb()
    from core.image import ocr_detection as ref
    mgr = ref.get_model_manager()
    bub, sec, osb = mgr.load_yolo_speech_bubble(), mgr.load_rtdetr_conjoined_bubble(), mgr.load_yolo_osbtext()
    bub.calls = sec.calls = osb.calls = 0
    state = dict(ok=True)

    def load_osb(token=None):
        if not state["ok"]:
            raise RuntimeError("gated repo")
        return osb

    class Paths(dict):
        def __missing__(self, k):
            return repr(k)                                       # one distinct path per model type
    mgr.load_yolo_osbtext, mgr.model_paths = load_osb, Paths()
    cache = UnifiedCache()
    ref.get_cache = lambda: cache
    det = ref.OutsideTextDetector(device="cpu")
    inp = mg.osb_inputs()
    img = Image.fromarray((np.random.default_rng(5).random((inp["H"], inp["W"], 3)) * 255).astype(np.uint8))
    provided = [dict(bbox=inp["bubbles"][0]), inp["bubbles"][1], dict(bbox=None), [1, 2, 3]]
    rows = []

    def run(tag, **kw):
        res = det.detect_outside_text("page.png", image_override=img, **kw)
        rows.append(dict(tag=tag, calls=[bub.calls, sec.calls, osb.calls], n=len(res), yolo_slot=cache.get_cache_stats()["yolo"]))
    run("all_models")
    run("all_models_again")
    run("provided", existing_bubbles=provided, text_free_boxes=[inp["secondary"][1]])
    run("text_free_only", existing_bubbles=provided, text_free_only=True)
    state["ok"] = False
    run("osb_unavailable", min_area_ignore_ratio=0.01)
    state["ok"] = True
    run("empty_list", existing_bubbles=[])
    run("other_confidence", confidence=0.5)
    return rows


def upscale_memo():
    """reference `upscale_image` / `upscale_image_to_dimension` (core/image/image_utils.py:377-548) with the REAL memo and the stand-in 2x
    model: model passes over a script of calls, and which calls return a remembered image object"""
    mg.gen_upscale()                                         # installs the stand-in model manager in the reference module
    from core.image import image_utils as refu
    cache = UnifiedCache()
    refu.get_cache = lambda: cache
    mgr = refu.get_model_manager()
    passes = [0]
    models = {}
    for name in ("load_upscale", "load_upscale_lite"):
        model = getattr(mgr, name)()
        models[name] = (lambda m: (lambda t: (passes.__setitem__(0, passes[0] + 1), m(t))[1]))(model)
    mgr.load_upscale = lambda *a, **k: models["load_upscale"]
    mgr.load_upscale_lite = lambda *a, **k: models["load_upscale_lite"]
    rng = np.random.default_rng(9)
    a = Image.fromarray(rng.integers(0, 256, (20, 14, 3), dtype=np.uint8))
    b = Image.fromarray(rng.integers(0, 256, (12, 18, 3), dtype=np.uint8))
    pages = dict(a=a, b=b)
    rows, results = [], []
    for page, factor, mt in [("a", 2.0, "model"), ("a", 2.0, "model"), ("a", 3.0, "model"), ("a", 2.0, "model_lite"), ("b", 1.5, "model"), ("a", 4.0, "model"),
                             ("a", 1.0, "model"), ("b", 1.5, "model")]:
        r = refu.upscale_image(pages[page], factor, mt)
        same_as = next((i for i, x in enumerate(results) if x is r), None)
        results.append(r)
        rows.append(dict(call=[page, factor, mt], passes=passes[0], size=list(r.size), same_object_as_call=same_as, stats=cache.get_cache_stats()["upscale"]))
    return rows


def inpaint_flow():
    """reference `FluxKontextInpainter.inpaint_mask` (core/image/inpainting.py:636-977) with the REAL memo and the stand-in pipeline of
    make_goldens.py: the key each call computes, whether the pipeline ran, and that a hit composites the same bytes."""
    import hashlib
    import threading
    ref = mg.inpainting
    inp = ref.FluxKontextInpainter.__new__(ref.FluxKontextInpainter)
    inp.context_padding_ratio, inp.max_context_padding = ref.CONTEXT_PADDING_RATIO, ref.MAX_CONTEXT_PADDING
    inp.PREFERED_KONTEXT_RESOLUTIONS = [(672, 1568), (688, 1504), (720, 1456), (752, 1392), (800, 1328), (832, 1248), (880, 1184), (944, 1104),
                                        (1024, 1024), (1104, 944), (1184, 880), (1248, 832), (1328, 800), (1392, 752), (1456, 720), (1504, 688), (1568, 672)]
    inp.backend, inp.low_vram, inp.DEVICE = "sdnq", True, torch.device("cpu")
    inp.num_inference_steps, inp.residual_diff_threshold, inp.guidance_scale, inp.prompt = 4, 0.12, 2.5, "Remove all text."
    calls = []
    def pipeline(**kw):
        calls.append(kw["image"].size)
        return mg.fake_pipeline(**kw)
    inp.pipeline = pipeline
    inp.load_models = lambda *a, **k: None
    inp._get_prompt_embeddings = lambda *a, **k: (None, None)
    inp.manager = types.SimpleNamespace(flux_inference_lock=threading.Lock())
    ref._flux_prompt_kwargs = lambda a, b: {}
    ref._pipeline_execution_device = lambda p, d: d
    cache = UnifiedCache()
    keys = []
    real_key = cache.get_inpaint_cache_key
    cache.get_inpaint_cache_key = lambda *a: (keys.append(real_key(*a)) or keys[-1])
    inp.cache = cache
    rng = np.random.default_rng(5)
    rows = []
    for ci in range(6):
        h, w = int(rng.integers(220, 420)), int(rng.integers(220, 420))
        m = mg.rand_mask(rng, h, w, ci % 4)
        yy, xx = np.mgrid[0:h, 0:w]
        page = Image.fromarray(np.stack([(xx * 3 + yy) % 256, (xx + yy * 2) % 256, (xx * yy // 7) % 256], -1).astype(np.uint8))
        ys, xs = np.where(m)
        strict = bool(ci % 2)
        clip = None if ci % 3 else [int(xs.min()) - 3, int(ys.min()) - 3, int(xs.max()) + 9, int(ys.max()) + 9]
        ocr = {"min_size": 200} if ci == 2 else None
        seed = -1 if ci == 5 else 1 + ci
        row = dict(h=h, w=w, kind=ci % 4, strict=strict, clip=clip, ocr=ocr, seed=seed, events=[])
        MASKS[f"mask{ci}"] = np.packbits(m)
        for rep in range(2):
            n_keys, n_calls = len(keys), len(calls)
            out = inp.inpaint_mask(page, m, seed=seed, ocr_params=ocr, strict_mask_clipping=strict, composite_clip_bbox=clip)
            row["events"].append(dict(key=keys[-1] if len(keys) > n_keys else None, pipeline_ran=len(calls) > n_calls,
                                      out_sha=hashlib.sha256(np.asarray(out).tobytes()).hexdigest()[:16], stats=cache.get_cache_stats()["inpaint"]))
        rows.append(row)
    return rows


if __name__ == "__main__":
    main()
