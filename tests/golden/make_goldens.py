"""Generates the golden vectors for the host-side integer logic of the hot path by IMPORTING THE
REFERENCE (read-only, /root/reference) in this container — SURVEY.md Appendix C harness: heavy
third-party deps are stubbed, only pure-Python reference functions are executed.  The reference
cannot travel to the GPU box, so the vectors are committed as small fixtures next to this script:

    python tests/golden/make_goldens.py        # rewrites tests/golden/*.json / *.npz

Covered (reference file:line):
  box hygiene      core/image/detection.py:219-254 (_deduplicate_primary_boxes), :257-295
                   (_remove_contained_boxes), :345-400 (_categorize_detections), :403-472
                   (_detect_overlapping_primaries)
  wave scheduling  core/batch_coordinator.py:78-153 (bboxes_overlap, expanded_mask_bbox,
                   partition_non_overlapping_waves)
  FLUX host math   core/image/inpainting.py:327-495 (compute_mask_bbox_aspect_ratio), :636-977
                   (inpaint_mask with a deterministic stand-in pipeline: crop geometry, bbox
                   quantisation, preferred-resolution choice, LANCZOS round trip, alpha composite)
  harness          core/pipeline.py:133-142 (_natural_path_sort_key), :2027-2064 (_resolve_output_path),
                   core/scaling.py:64-96 (scale_kernel)
"""
import json
import sys
import types
from pathlib import Path
from unittest.mock import MagicMock

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
REF = "/root/reference"

import importlib.machinery
import importlib.util
for name in ["cv2", "spandrel", "ultralytics", "oxipng", "diffusers", "sdnq", "skia", "uharfbuzz", "manga_ocr",
             "pythainlp", "pythainlp.tokenize", "gradio", "torchvision"]:
    try:
        if importlib.util.find_spec(name) is not None:
            continue
    except (ImportError, ValueError):
        pass
    stub = MagicMock(name=name)
    stub.__spec__ = importlib.machinery.ModuleSpec(name, None)
    sys.modules.setdefault(name, stub)
for pkg in ["core", "core.image", "core.ml", "core.text", "core.services"]:
    m = types.ModuleType(pkg)
    m.__path__ = [REF + "/" + pkg.replace(".", "/")]
    sys.modules[pkg] = m
sys.path.insert(0, REF)
# core.ml.model_manager imports HF processors that need torchvision (absent here); the functions under
# test never touch it, so the module is replaced by a stub
_mm = MagicMock(name="core.ml.model_manager")
_mm.__spec__ = importlib.machinery.ModuleSpec("core.ml.model_manager", None)
sys.modules["core.ml.model_manager"] = _mm

from core import batch_coordinator, scaling  # noqa: E402
from core.image import detection, inpainting  # noqa: E402
from PIL import Image  # noqa: E402


def rng_boxes(rng, n, w=1024, h=1536, nest=True):
    b = []
    for _ in range(n):
        x0, y0 = rng.uniform(0, w * 0.8), rng.uniform(0, h * 0.8)
        b.append([x0, y0, x0 + rng.uniform(30, 300), y0 + rng.uniform(30, 300)])
    if nest and n >= 4:
        b[1] = [b[0][0] + 5, b[0][1] + 5, b[0][2] - 5, b[0][3] - 5]          # nested
        b[2] = [b[0][0] + 1, b[0][1] - 2, b[0][2] + 2, b[0][3] + 1]          # near-duplicate
        b[3] = [b[0][0] + 0.4 * (b[0][2] - b[0][0]), b[0][1], b[0][2] + 80, b[0][3]]  # partial overlap
    return np.asarray(b, dtype=np.float32)


def gen_boxes():
    rng = np.random.default_rng(7)
    cases = []
    for ci in range(24):
        n = int(rng.integers(0, 14))
        boxes = rng_boxes(rng, n, nest=(ci % 3 != 0))
        conf = rng.uniform(0.3, 0.99, n).astype(np.float32)
        if n >= 5 and ci % 4 == 0:
            conf[4] = conf[0]                                                  # tie
        tb, tc = torch.from_numpy(boxes).reshape(-1, 4), torch.from_numpy(conf)
        _, keep = detection._deduplicate_primary_boxes(tb, tc, 0.7)
        _, kept_idx = detection._remove_contained_boxes(tb, None, 0.9)
        m = int(rng.integers(0, 10))
        sec = rng_boxes(rng, m, nest=False)
        if m >= 2 and n >= 1:   # two halves of primary 0 -> conjoined
            x0, y0, x1, y1 = boxes[0]
            sec[0] = [x0 + 2, y0 + 2, (x0 + x1) / 2, y1 - 2]
            sec[1] = [(x0 + x1) / 2, y0 + 2, x1 - 2, y1 - 2]
        conj, simple = ([], list(range(n)))
        if n > 0 and m > 0:
            conj, simple = detection._categorize_detections(tb, torch.from_numpy(sec).reshape(-1, 4))
        groups, upd = detection._detect_overlapping_primaries(tb, list(simple)) if n > 0 else ([], [])
        cases.append(dict(boxes=boxes.tolist(), conf=conf.tolist(), secondary=sec.tolist(),
                          dedup_keep=[int(k) for k in keep], contained_keep=[int(i) for _, i in kept_idx],
                          conjoined=[[int(p), [int(s) for s in ss]] for p, ss in conj], simple=[int(s) for s in simple],
                          synthetic_groups=[[int(x) for x in g] for g in groups], simple_after=[int(s) for s in upd]))
    return cases


def rand_mask(rng, h, w, kind):
    m = np.zeros((h, w), bool)
    if kind == 0:
        x0, y0 = int(rng.integers(0, w - 40)), int(rng.integers(0, h - 40))
        m[y0:y0 + int(rng.integers(8, 120)), x0:x0 + int(rng.integers(8, 160))] = True
    elif kind == 1:
        yy, xx = np.mgrid[0:h, 0:w]
        cx, cy = rng.uniform(0.2, 0.8) * w, rng.uniform(0.2, 0.8) * h
        m = ((xx - cx) / rng.uniform(20, 90)) ** 2 + ((yy - cy) / rng.uniform(15, 70)) ** 2 <= 1
    elif kind == 2:
        m[0:int(rng.integers(5, 60)), 0:int(rng.integers(5, 80))] = True            # flush top-left
    else:
        m[h - int(rng.integers(5, 60)):, w - int(rng.integers(5, 80)):] = True      # flush bottom-right
    return m


COORD_MASKS = {}


def gen_coordinator():
    rng = np.random.default_rng(11)
    out = []
    for ci in range(16):
        h, w = int(rng.integers(200, 500)), int(rng.integers(200, 500))
        m = rand_mask(rng, h, w, ci % 4)
        bbox = batch_coordinator.expanded_mask_bbox(m, (w, h))
        out.append(dict(h=h, w=w, seed=int(ci), kind=ci % 4, bbox=list(bbox) if bbox else None))
        COORD_MASKS[f"m{ci}"] = np.packbits(m)
    waves_cases = []
    for ci in range(12):
        n = int(rng.integers(0, 9))
        items = []
        for _ in range(n):
            if rng.uniform() < 0.15:
                items.append(None)
            else:
                x0, y0 = int(rng.integers(0, 300)), int(rng.integers(0, 300))
                items.append([x0, y0, x0 + int(rng.integers(10, 150)), y0 + int(rng.integers(10, 150))])
        idx = list(range(n))
        waves = batch_coordinator.partition_non_overlapping_waves(idx, lambda i: tuple(items[i]) if items[i] else None)
        waves_cases.append(dict(bboxes=items, waves=waves))
    return dict(expanded=out, waves=waves_cases)


class _FakeOut:
    def __init__(self, img):
        self.images = [img]


def fake_pipeline(**kw):
    """Deterministic stand-in for FluxKontextPipeline: inverted, slightly blurred input at the requested size."""
    img = kw["image"].convert("RGB").resize((kw["width"], kw["height"]), Image.BILINEAR)
    t = torch.from_numpy(np.asarray(img, dtype=np.float32) / 255.0).permute(2, 0, 1)
    return _FakeOut(1.0 - 0.9 * t)


def gen_inpaint():
    rng = np.random.default_rng(5)
    inp = inpainting.FluxKontextInpainter.__new__(inpainting.FluxKontextInpainter)
    # only the attributes inpaint_mask touches
    inp.context_padding_ratio = inpainting.CONTEXT_PADDING_RATIO
    inp.max_context_padding = inpainting.MAX_CONTEXT_PADDING
    inp.PREFERED_KONTEXT_RESOLUTIONS = [(672, 1568), (688, 1504), (720, 1456), (752, 1392), (800, 1328), (832, 1248),
                                        (880, 1184), (944, 1104), (1024, 1024), (1104, 944), (1184, 880), (1248, 832),
                                        (1328, 800), (1392, 752), (1456, 720), (1504, 688), (1568, 672)]
    inp.backend = "sdnq"
    inp.low_vram = True
    inp.DEVICE = torch.device("cpu")
    inp.num_inference_steps = 4
    inp.residual_diff_threshold = 0.12
    inp.guidance_scale = 2.5
    inp.prompt = "Remove all text."
    inp.pipeline = fake_pipeline
    inp.load_models = lambda *a, **k: None
    inp._get_prompt_embeddings = lambda *a, **k: (None, None)
    inp.manager = types.SimpleNamespace(flux_inference_lock=__import__("threading").Lock())
    inp.cache = types.SimpleNamespace(should_use_inpaint_cache=lambda seed: False)
    inpainting._flux_prompt_kwargs = lambda a, b: {}
    inpainting._pipeline_execution_device = lambda p, d: d
    geo, arrays = [], {}
    for ci in range(8):
        h, w = int(rng.integers(220, 420)), int(rng.integers(220, 420))
        m = rand_mask(rng, h, w, ci % 4)
        yy, xx = np.mgrid[0:h, 0:w]
        page = np.stack([(xx * 3 + yy) % 256, (xx + yy * 2) % 256, (xx * yy // 7) % 256], -1).astype(np.uint8)
        mt = torch.from_numpy(m.astype(np.float32))[None, None]
        ys, xs = np.where(m)
        bw, bh = int(xs.max()) - int(xs.min()), int(ys.max()) - int(ys.min())
        padding = min(int(max(bw, bh) * inp.context_padding_ratio), inp.max_context_padding)
        blur = max(inpainting.MIN_BLUR_RADIUS, min(int(max(bw, bh) * inpainting.BLUR_SCALE_FACTOR), inpainting.MAX_BLUR_RADIUS))
        alpha, x, y, ww, hh = inp.compute_mask_bbox_aspect_ratio(mask_chw=mt, padding=padding, blur_radius=blur,
                                                                 preferred_resolutions=inp.PREFERED_KONTEXT_RESOLUTIONS)
        strict = bool(ci % 2)
        clip = None if ci % 3 else [int(xs.min()) - 3, int(ys.min()) - 3, int(xs.max()) + 9, int(ys.max()) + 9]
        out = inp.inpaint_mask(Image.fromarray(page), m, seed=1, strict_mask_clipping=strict, composite_clip_bbox=clip)
        geo.append(dict(h=h, w=w, kind=ci % 4, padding=padding, blur=blur, bbox=[x, y, ww, hh], strict=strict, clip=clip))
        arrays[f"mask{ci}"] = np.packbits(m)
        arrays[f"alpha{ci}"] = alpha.numpy().astype(np.float32)
        o = np.asarray(out)
        assert np.array_equal(np.delete(o.reshape(-1, 3), [], 0).shape, page.reshape(-1, 3).shape)
        outside = np.ones((h, w), bool); outside[y:y + hh, x:x + ww] = False
        # the crop can shift by the 2-px quantisation; store a generous window and require the rest untouched
        y0, y1, x0, x1 = max(0, y - 4), min(h, y + hh + 4), max(0, x - 4), min(w, x + ww + 4)
        outside[y0:y1, x0:x1] = False
        assert np.array_equal(o[outside], page[outside])
        arrays[f"out{ci}"] = o[y0:y1, x0:x1]
        geo[-1]["window"] = [y0, y1, x0, x1]
    consts = dict(context_padding_ratio=inp.context_padding_ratio, max_context_padding=inp.max_context_padding,
                  blur_scale=inpainting.BLUR_SCALE_FACTOR, min_blur=inpainting.MIN_BLUR_RADIUS, max_blur=inpainting.MAX_BLUR_RADIUS)
    return geo, arrays, consts


def gen_harness():
    from core import pipeline
    names = ["ch2/001.jpg", "ch2/010.jpg", "ch10/001.jpg", "P1.png", "p2.png", "p10.png", "a/b/3.webp", "a/b/12.webp",
             "a/B/2.webp", "007.png", "7.png", "x1y20.png", "x1y3.png", "X1y3.PNG", "z.jpeg"]
    order = sorted(names, key=lambda s: pipeline._natural_path_sort_key(Path(s)))
    res = []
    for fmt in ("png", "jpeg", "auto", "bogus"):
        for preserve in (False, True):
            cfg = types.SimpleNamespace(output=types.SimpleNamespace(output_format=fmt))
            import tempfile
            with tempfile.TemporaryDirectory() as td:
                outp, disp, err = pipeline._resolve_output_path(Path("/in/ch1/p01.JPG"), Path("/in"), Path(td), cfg, preserve)
                res.append(dict(fmt=fmt, preserve=preserve, out=str(outp.relative_to(td)), display=disp, error_key=err))
    kernels = [[list(scaling.scale_kernel((a, b), s)) for s in (None, 0.5, 1.0, 1.254, 2.5, 6.3)] for a, b in ((7, 7), (5, 5), (3, 9))]
    return dict(names=names, order=order, resolve=res, scale_kernel=kernels)


def gen_conjoined():
    """conjoined-bubble mask partition: core/image/detection.py:971-1035 (_split_conjoined_mask) with everything it calls
    (:568-579, :582-665, :668-929, :932-968, :317-342, :793-827) and :1075-1260 (_build_segmentation_detections).
    cv2 is absent here: the ONE cv2 call on this path, cv2.distanceTransform(inv_seed, DIST_L2, 5) at :954, is served by
    the oracle's restatement (oracle/cleaning_ref.distance_transform_l2_5x5); every other line executed is the reference's."""
    sys.path.insert(0, str(HERE.parent.parent))
    from oracle.cleaning_ref import distance_transform_l2_5x5
    detection.cv2 = types.SimpleNamespace(distanceTransform=lambda img, dt, ms: distance_transform_l2_5x5(np.asarray(img)), DIST_L2=2)
    H, W = 120, 168
    yy, xx = np.mgrid[0:H, 0:W]
    ell = lambda cx, cy, a, b: ((xx - cx) / a) ** 2 + ((yy - cy) / b) ** 2 <= 1.0
    scen = []
    # (parent mask, child boxes, text boxes or None)
    scen.append((ell(50, 60, 40, 35) | ell(110, 60, 42, 36), [[10.2, 24.7, 92.5, 96.1], [66.4, 22.0, 153.3, 97.9]], None))                      # side by side
    scen.append((ell(80, 35, 50, 28) | ell(84, 82, 52, 30), [[28.0, 6.0, 132.0, 66.0], [30.5, 48.2, 137.0, 113.0]], None))                       # stacked
    scen.append((ell(55, 45, 42, 34) | ell(112, 78, 44, 33), [[12.0, 10.0, 99.0, 80.0], [66.0, 44.0, 158.0, 112.0]], None))                      # diagonal, same-sign offsets
    scen.append((ell(112, 42, 42, 32) | ell(56, 80, 44, 32), [[68.0, 9.0, 156.0, 76.0], [10.0, 46.0, 102.0, 113.0]], None))                      # diagonal, opposite signs
    scen.append((ell(50, 60, 40, 35) | ell(110, 60, 42, 36), [[10.2, 24.7, 92.5, 96.1], [66.4, 22.0, 153.3, 97.9]],
                 [[20, 45, 70, 75, 0.9], [22, 47, 60, 70, 0.8], [96, 40, 140, 80, 0.9]]))                                                          # text-safe, nested text box
    scen.append((ell(50, 60, 40, 35) | ell(110, 60, 42, 36), [[10.2, 24.7, 92.5, 96.1], [66.4, 22.0, 153.3, 97.9]],
                 [[30, 45, 84, 75, 0.9], [74, 40, 140, 80, 0.9]]))                                                                                  # texts overlap the cut: no feasible offset
    scen.append((ell(50, 60, 40, 35) | ell(110, 60, 42, 36), [[10.2, 24.7, 92.5, 96.1], [66.4, 22.0, 153.3, 97.9]],
                 [[60, 50, 100, 70, 0.9], [20, 40, 50, 60, 0.7], [110, 45, 140, 75, 0.7]]))                                                        # one ambiguous text box
    scen.append((ell(40, 60, 32, 40) | ell(84, 60, 30, 42) | ell(128, 60, 32, 40), [[6.0, 18.0, 70.0, 102.0], [52.0, 16.0, 116.0, 104.0], [96.0, 18.0, 162.0, 102.0]], None))   # three in a row
    scen.append((ell(50, 60, 40, 35), [[10.0, 24.0, 60.0, 96.0], [120.0, 10.0, 160.0, 40.0]], None))                                             # second child outside the parent: nearest-pixel seed
    scen.append((np.zeros((H, W), bool), [[10.0, 24.0, 60.0, 96.0], [50.0, 20.0, 120.0, 90.0]], None))                                            # empty parent
    scen.append((ell(80, 60, 60, 45), [[20.0, 15.0, 140.0, 105.0]], None))                                                                        # single child
    scen.append((ell(60, 60, 45, 40) | ell(100, 62, 45, 40), [[14.5, 19.5, 105.5, 100.5], [54.5, 21.5, 146.0, 102.5]],
                 [[25, 50, 55, 70, 0.9], [105, 50, 135, 72, 0.9]]))                                                                                  # text-safe offset far from the midline
    arrays, meta = {}, []
    for k, (pm, boxes, texts) in enumerate(scen):
        tb = [torch.tensor(b, dtype=torch.float32) for b in boxes]
        tx = np.asarray(texts, np.float32) if texts is not None else None
        grp = detection._get_group_osb_text_boxes(tx, torch.tensor([0.0, 0.0, float(W), float(H)])) if tx is not None else None
        out = detection._split_conjoined_mask(pm.astype(np.uint8) * 255, tb, osb_text_boxes=grp)
        arrays[f"parent_{k}"] = np.packbits(pm)
        arrays[f"out_{k}"] = np.packbits(np.stack([np.asarray(o) > 0 for o in out]) if out else np.zeros((0, H, W), bool))
        meta.append(dict(boxes=boxes, texts=texts, n_out=len(out), arrangement=detection._detect_group_arrangement(tb),
                         group_texts=(np.asarray(grp).tolist() if grp is not None else None),
                         match=({str(i): np.asarray(v).tolist() for i, v in detection._match_text_boxes_to_bubbles(grp, tb).items()} if grp is not None else None),
                         rects=[[int(v) for v in np.nonzero(detection._build_rect_mask_from_box(t, H, W).any(0))[0][[0, -1]]] +
                                [int(v) for v in np.nonzero(detection._build_rect_mask_from_box(t, H, W).any(1))[0][[0, -1]]] for t in tb]))
    # assembly: one simple bubble with a SAM mask, one simple without (rect fallback), one conjoined pair, one synthetic group
    pm = ell(50, 60, 40, 35) | ell(110, 60, 42, 36)
    primary = torch.tensor([[8.0, 20.0, 156.0, 100.0], [20.3, 5.2, 60.8, 30.9], [100.0, 100.0, 150.0, 118.0], [10.0, 100.0, 60.0, 119.0], [40.0, 98.0, 95.0, 119.0]], dtype=torch.float32)
    secondary = torch.tensor([[10.2, 24.7, 92.5, 96.1], [66.4, 22.0, 153.3, 97.9]], dtype=torch.float32)
    ns = types.SimpleNamespace
    pres = ns(boxes=ns(conf=torch.tensor([0.9, 0.8, 0.7, 0.65, 0.6]), cls=torch.tensor([0.0, 0.0, 0.0, 0.0, 0.0]), __len__=None), masks=None)
    pres.boxes = type("B", (), {"conf": torch.tensor([0.9, 0.8, 0.7, 0.65, 0.6]), "cls": torch.zeros(5), "__len__": lambda self: 5})()
    sres = ns(names={0: "bubble"})
    sres.boxes = type("B", (), {"conf": torch.tensor([0.55, 0.45]), "cls": torch.zeros(2), "__len__": lambda self: 2})()
    model = ns(names={0: "speech_bubble"})
    sam = [pm.astype(np.uint8) * 255, ell(40, 18, 18, 11).astype(np.uint8) * 255, None, None, None]
    synth = [dict(parent_mask=(ell(35, 110, 24, 8) | ell(68, 109, 26, 9)).astype(np.uint8) * 255, parent_box=[10.0, 98.0, 95.0, 119.0], member_indices=[3, 4])]
    dets = detection._build_segmentation_detections(primary, primary, [("primary", i) for i in range(5)], pres, model, secondary,
                                                    [("secondary", 0), ("secondary", 1)], sres, [1, 2], [(0, [0, 1])], H, W, 0.35,
                                                    osb_text_boxes_np=None, sam_masks=sam, synthetic_conjoined_groups=synth)
    arrays["assembly_masks"] = np.packbits(np.stack([np.asarray(d["sam_mask"]) > 0 for d in dets]))
    arrays["assembly_sam0"], arrays["assembly_sam1"], arrays["assembly_synth"] = np.packbits(sam[0] > 0), np.packbits(sam[1] > 0), np.packbits(synth[0]["parent_mask"] > 0)
    assembly = dict(primary=primary.tolist(), secondary=secondary.tolist(),
                    dets=[dict(bbox=list(d["bbox"]), confidence=float(d["confidence"]), cls=d["class"],
                               neighbors=[list(b) for b in d.get("conjoined_neighbor_bboxes", [])] if "conjoined_neighbor_bboxes" in d else None) for d in dets])
    return dict(H=H, W=W, scenarios=meta, assembly=assembly), arrays


def detection_flow_inputs():
    """canned detector outputs for the operator-flow golden (shared with tests/test_detection_flow.py through the json)"""
    primary = [[20.0, 30.0, 190.0, 120.0],      # P0: one box over two touching bubbles (conjoined parent)
               [210.0, 20.0, 290.0, 90.0],      # P1: simple
               [30.0, 150.0, 120.0, 230.0],     # P2, P3: mutually overlapping primaries -> synthetic group
               [95.0, 155.0, 185.0, 232.0],
               [220.0, 150.0, 300.0, 220.0],    # P4: marked text_free by the secondary model
               [212.0, 22.0, 288.0, 88.0]]      # P5: duplicate of P1 (lower confidence)
    pconf = [0.9, 0.85, 0.8, 0.75, 0.7, 0.65]
    secondary = [[22.0, 32.0, 108.0, 118.0], [100.0, 34.0, 188.0, 119.0],          # S0, S1: the two halves of P0 (bubble)
                 [222.0, 152.0, 298.0, 218.0],                                          # S2: text_free over P4
                 [130.0, 250.0, 200.0, 300.0],                                          # S3: a bubble the primary missed
                 [40.0, 50.0, 90.0, 100.0],                                             # S4: text_bubble (ignored)
                 [30.0, 40.0, 100.0, 110.0]]                                            # S5: bubble nested in S0 (contained removal)
    sconf = [0.8, 0.78, 0.6, 0.5, 0.9, 0.4]
    scls = [0, 0, 2, 0, 1, 0]
    osb_text = [[35.0, 45.0, 95.0, 105.0], [112.0, 50.0, 180.0, 100.0],          # one text block in each half of the conjoined P0
                [250.0, 60.0, 300.0, 100.0],                                          # sticks out of P1 -> P1 grows
                [60.0, 170.0, 110.0, 210.0], [125.0, 175.0, 175.0, 215.0],            # P2 / P3 texts (the first reaches into the overlap)
                [5.0, 280.0, 60.0, 310.0]]                                            # belongs to no bubble
    return dict(H=320, W=320, primary=primary, pconf=pconf, secondary=secondary, sconf=sconf, scls=scls,
                names={"0": "bubble", "1": "text_bubble", "2": "text_free"}, osb_text=osb_text, osb_conf=[0.9, 0.88, 0.8, 0.7, 0.75, 0.6])


def gen_detection_flow():
    """the whole operator: core/image/detection.py:1263-1816 `detect_speech_bubbles` driven by canned detector / SAM outputs
    (the models themselves have their own parity tests).  cv2 here: cvtColor at image load (a channel flip) and the
    distanceTransform of the conjoined partition (the oracle's restatement, as in gen_conjoined)."""
    sys.path.insert(0, str(HERE.parent.parent))
    from oracle.cleaning_ref import distance_transform_l2_5x5
    inp = detection_flow_inputs()
    H, W = inp["H"], inp["W"]
    detection.cv2 = types.SimpleNamespace(distanceTransform=lambda img, dt, ms: distance_transform_l2_5x5(np.asarray(img)), DIST_L2=2,
                                          cvtColor=lambda a, code: np.ascontiguousarray(a[..., ::-1]), COLOR_RGB2BGR=4, COLOR_BGR2RGB=4)

    class Boxes:
        def __init__(self, xyxy, conf, cls):
            self.xyxy, self.conf, self.cls = torch.tensor(xyxy, dtype=torch.float32), torch.tensor(conf, dtype=torch.float32), torch.tensor(cls, dtype=torch.float32)

        def __len__(self):
            return len(self.xyxy)

    primary_model = lambda *a, **k: [types.SimpleNamespace(boxes=Boxes(inp["primary"], inp["pconf"], [0] * 6), masks=None, orig_shape=(H, W))]
    primary_model_obj = types.SimpleNamespace(names={0: "speech_bubble"}, __call__=None)

    class Callable_:
        def __init__(self, fn, names):
            self.fn, self.names = fn, names

        def __call__(self, *a, **k):
            return self.fn(*a, **k)

    pm = Callable_(primary_model, {0: "speech_bubble"})
    sm = Callable_(lambda *a, **k: [types.SimpleNamespace(boxes=Boxes(inp["secondary"], inp["sconf"], inp["scls"]), names={int(k_): v for k_, v in inp["names"].items()})],
                   {int(k_): v for k_, v in inp["names"].items()})
    prompts_seen = []

    class Inputs(dict):
        def to(self, *a, **k):
            return self

    class Proc:
        def __call__(self, image, input_boxes=None, return_tensors="pt"):
            return Inputs(boxes=torch.as_tensor(input_boxes, dtype=torch.float32).reshape(-1, 4), original_sizes=torch.tensor([[H, W]]))

        def post_process_masks(self, pred, sizes, **kw):
            return [pred]

    class Sam:
        dtype = torch.float32

        def __call__(self, multimask_output=False, **inputs):
            bx = inputs["boxes"]
            prompts_seen.append(bx.tolist())
            yy, xx = np.mgrid[0:H, 0:W]
            ms = []
            for x0, y0, x1, y1 in bx.tolist():   # an ellipse around the prompt box, slightly larger than it (so the box clip matters)
                cx, cy, a, b = (x0 + x1) / 2, (y0 + y1) / 2, (x1 - x0) / 2 * 1.08, (y1 - y0) / 2 * 1.08
                ms.append(((xx - cx) / a) ** 2 + ((yy - cy) / b) ** 2 <= 1.0)
            return types.SimpleNamespace(pred_masks=torch.from_numpy(np.stack(ms))[:, None].float())

    sam = Sam()
    mgr = types.SimpleNamespace(load_yolo_speech_bubble=lambda *a, **k: pm, load_rtdetr_conjoined_bubble=lambda *a, **k: sm,
                                load_sam2=lambda *a, **k: (Proc(), sam), device="cpu")
    detection.get_model_manager = lambda: mgr
    none = lambda *a, **k: None
    detection.get_cache = lambda: types.SimpleNamespace(get_yolo_cache_key=none, get_yolo_detection=none, set_yolo_detection=none,
                                                        get_sam_cache_key=none, get_sam_masks=none, set_sam_masks=none)
    rng = np.random.default_rng(3)
    img = Image.fromarray((rng.random((H, W, 3)) * 255).astype(np.uint8))
    out = {}
    for seg in ("sam2", "yolo"):
        dets, text_free = detection.detect_speech_bubbles(Path("page.png"), "yolo_2", confidence=0.6, device="cpu", seg_model=seg,
                                                          conjoined_detection=True, image_override=img)
        out[seg] = dict(text_free=[[float(v) for v in b] for b in text_free],
                        dets=[dict(bbox=list(d["bbox"]), confidence=float(d["confidence"]), cls=d["class"],
                                   neighbors=[list(b) for b in d["conjoined_neighbor_bboxes"]] if "conjoined_neighbor_bboxes" in d else None) for d in dets])
        DET_MASKS[seg] = np.packbits(np.stack([np.asarray(d["sam_mask"]) > 0 for d in dets]))
    out["prompts"] = prompts_seen[0]
    # osb_text_verification: the OSB text model's boxes grow the bubble boxes (:120-198) and steer the conjoined cuts; the reference
    # reads them back from its detection cache (:298-314), so the stub cache really stores here
    store = {}
    om = Callable_(lambda *a, **k: [types.SimpleNamespace(boxes=Boxes(inp["osb_text"], inp["osb_conf"], [0] * len(inp["osb_text"])))], {0: "text"})

    class Paths(dict):
        def __missing__(self, k):
            return "osb.pt"

    mgr.load_yolo_osbtext = lambda *a, **k: om
    mgr.model_paths = Paths()
    detection.get_cache = lambda: types.SimpleNamespace(get_yolo_cache_key=lambda im, path, conf: (str(path), conf), get_yolo_detection=store.get,
                                                        set_yolo_detection=store.__setitem__, get_sam_cache_key=none, get_sam_masks=none, set_sam_masks=none)
    n_before = len(prompts_seen)
    dets, text_free = detection.detect_speech_bubbles(Path("page.png"), "yolo_2", confidence=0.6, device="cpu", seg_model="sam2",
                                                      conjoined_detection=True, image_override=img, osb_text_verification=True)
    out["sam2_osb_verify"] = dict(text_free=[[float(v) for v in b] for b in text_free], prompts=prompts_seen[n_before],
                                  dets=[dict(bbox=list(d["bbox"]), confidence=float(d["confidence"]), cls=d["class"],
                                             neighbors=[list(b) for b in d["conjoined_neighbor_bboxes"]] if "conjoined_neighbor_bboxes" in d else None) for d in dets])
    DET_MASKS["sam2_osb_verify"] = np.packbits(np.stack([np.asarray(d["sam_mask"]) > 0 for d in dets]))
    return dict(inputs=inp, results=out)


DET_MASKS = {}


def osb_inputs():
    """canned detector outputs for the OSB region golden (shared with tests/test_osb_regions.py through the json)"""
    return dict(
        W=400, H=300,
        bubbles=[[20.3, 20.1, 140.2, 110.6], [200.4, 30.2, 330.1, 120.7]], bconf=[0.9, 0.8],
        secondary=[[40.2, 200.4, 120.6, 270.1],        # S0 bubble the primary missed
                   [250.5, 180.2, 380.3, 260.8],       # S1 text_free (a narration box)
                   [205.1, 32.3, 328.2, 118.4],        # S2 text_free that is the bubble B1
                   [50.0, 40.0, 90.0, 70.0]],          # S3 text_bubble (ignored)
        sconf=[0.7, 0.6, 0.55, 0.9], scls=[0, 2, 2, 1],
        names={"0": "bubble", "1": "text_bubble", "2": "text_free"},
        osb=[[30.4, 30.6, 100.2, 80.3],                # T0 inside B0 -> belongs to a bubble
             [210.7, 40.2, 300.9, 100.4],              # T1 inside B1, but B1 is a text_free region -> kept
             [150.2, 130.6, 190.8, 150.3],             # T2 free-standing
             [152.1, 132.2, 188.3, 148.9],             # T3 nested in T2 -> removed
             [130.5, 60.1, 160.2, 90.7],               # T4 a third of it inside B0 -> belongs to B0
             [135.3, 95.2, 185.6, 125.4],              # T5 5 % inside B0, centre outside -> kept
             [160.9, 140.2, 200.1, 160.8],             # T6 close to T2
             [45.6, 205.3, 110.2, 260.9],              # T7 inside the missed bubble S0
             [300.2, 10.5, 300.9, 25.0]],              # T8 zero width after int() -> dropped by get_text_masks, shifting the pairing
        oconf=[0.91, 0.82, 0.73, 0.64, 0.55, 0.86, 0.77, 0.68, 0.95])


OSB_MASKS = {}


def gen_osb():
    """core/image/ocr_detection.py:189-808 — `detect_outside_text`, `get_text_masks`, `_group_text_boxes_spatially` driven by
    canned detector outputs.  cv2 there is only the RGB<->BGR flip at image load."""
    from core.image import ocr_detection as ref
    inp = osb_inputs()
    W, H = inp["W"], inp["H"]
    ref.cv2 = types.SimpleNamespace(cvtColor=lambda a, code: np.ascontiguousarray(a[..., ::-1]), COLOR_RGB2BGR=4, COLOR_BGR2RGB=4)

    class Boxes:
        def __init__(self, xyxy, conf, cls):
            self.xyxy, self.conf, self.cls = torch.tensor(xyxy, dtype=torch.float32).reshape(-1, 4), torch.tensor(conf, dtype=torch.float32), torch.tensor(cls, dtype=torch.float32)

        def __len__(self):
            return len(self.xyxy)

    class Model:
        def __init__(self, boxes, names):
            self.boxes, self.names, self.calls = boxes, names, 0

        def __call__(self, *a, **k):
            self.calls += 1
            return [types.SimpleNamespace(boxes=self.boxes)]

    names = {int(k): v for k, v in inp["names"].items()}
    bub = Model(Boxes(inp["bubbles"], inp["bconf"], [0, 0]), {0: "speech_bubble"})
    sec = Model(Boxes(inp["secondary"], inp["sconf"], inp["scls"]), names)
    osb = Model(Boxes(inp["osb"], inp["oconf"], [0] * len(inp["osb"])), {0: "text"})
    state = dict(osb_ok=True)

    def load_osb(token=None):
        if not state["osb_ok"]:
            raise RuntimeError("gated repo")
        return osb

    class Paths(dict):
        def __missing__(self, k):
            return "model.pt"

    mgr = types.SimpleNamespace(load_yolo_speech_bubble=lambda *a, **k: bub, load_rtdetr_conjoined_bubble=lambda *a, **k: sec,
                                load_yolo_osbtext=load_osb, model_paths=Paths(), device="cpu")
    none = lambda *a, **k: None
    ref.get_model_manager = lambda: mgr
    ref.get_cache = lambda: types.SimpleNamespace(get_yolo_cache_key=none, get_yolo_detection=none, set_yolo_detection=none)
    ref.get_best_device = lambda: "cpu"
    det = ref.OutsideTextDetector(device="cpu")
    img = Image.fromarray((np.random.default_rng(5).random((H, W, 3)) * 255).astype(np.uint8))
    ser = lambda res: [dict(bbox=[float(v) for v in b], conf=float(c)) for b, c in res]
    out = {}
    res_a = det.detect_outside_text("page.png", image_override=img)
    out["detect_all_models"] = ser(res_a)
    provided = [dict(bbox=inp["bubbles"][0]), inp["bubbles"][1], dict(bbox=None), [1, 2, 3]]
    out["detect_provided_bubbles"] = ser(det.detect_outside_text("page.png", image_override=img, existing_bubbles=provided, text_free_boxes=[inp["secondary"][1]]))
    out["detect_text_free_only"] = ser(det.detect_outside_text("page.png", image_override=img, existing_bubbles=provided, text_free_only=True))
    state["osb_ok"] = False
    out["detect_osb_model_unavailable"] = ser(det.detect_outside_text("page.png", image_override=img, min_area_ignore_ratio=0.01))
    state["osb_ok"] = True
    out["detect_no_bubbles_given_empty_list"] = ser(det.detect_outside_text("page.png", image_override=img, existing_bubbles=[]))

    def ser_groups(tag, groups):
        rows = []
        for gi, g in enumerate(groups or []):
            OSB_MASKS[f"{tag}_{gi}_combined"] = np.packbits(g["combined_mask"])
            OSB_MASKS[f"{tag}_{gi}_individual"] = np.packbits(np.stack(g["individual_masks"]))
            rows.append(dict(bbox=g["bbox"], original_bbox=g["original_bbox"], mask_indices=[int(i) for i in g["mask_indices"]],
                             confidence=float(g["confidence"]), n_individual=len(g["individual_masks"])))
        return rows

    # the un-filtered list keeps the degenerate box T8, so the pairing shift of get_text_masks is exercised too
    order = [0, 1, 2, 8, 3, 4, 5, 6, 7]          # the degenerate box in the middle
    raw = [(np.asarray(inp["osb"][i], np.float32), float(inp["oconf"][i])) for i in order]
    masks = {}
    for tag, (ew, eh, ratio, results) in dict(plain=(0.0, 0.0, 0.02, res_a), grown=(0.1, 0.2, 0.06, res_a), wide=(0.5, 0.5, 0.3, res_a),
                                              raw_shift=(0.05, 0.05, 0.08, raw)).items():
        groups, _ = det.get_text_masks("page.png", ew, eh, ratio, image_override=img, existing_results=results)
        masks[tag] = dict(args=[ew, eh, ratio], source="raw" if results is raw else "detect_all_models", raw_order=order, groups=ser_groups(tag, groups))
    big = Image.new("RGB", (2000, 1800), "white")
    far = [(np.asarray([10.5, 20.5, 110.2, 90.9], np.float32), 0.9), (np.asarray([1700.1, 1650.2, 1900.7, 1760.3], np.float32), 0.7),
           (np.asarray([60.0, 60.0, 200.0, 130.0], np.float32), 0.5)]
    groups, _ = det.get_text_masks("page.png", 0.0, 0.0, 2.0, image_override=big, existing_results=far)
    masks["too_large"] = dict(args=[0.0, 0.0, 2.0], size=[2000, 1800], results=[dict(bbox=[float(v) for v in b], conf=c) for b, c in far],
                              groups=ser_groups("too_large", groups))
    out["empty"] = det.get_text_masks("page.png", image_override=img, existing_results=[])[0]
    return dict(inputs=inp, provided=[p if not isinstance(p, dict) else p for p in provided], detect=out, masks=masks)


def gen_osb_stage():
    """core/outside_text_processor.py:217-1691 — `prepare_outside_text_work` + `finish_outside_text_work` on the synthetic page of
    tests/golden/osb_page.py.  Stand-ins: the inpainter (deterministic), cv2.dilate (a zero-padded maximum filter == cv2's default
    constant border), and the cv2 calls of the rendered-text colour probe (:1096-1165: identity / no contours — that probe only feeds
    the text renderer); `_build_outside_text_data` / `_apply_inpaint_render_metadata` (translation payload) are replaced by no-ops."""
    from scipy import ndimage
    sys.path.insert(0, str(HERE))
    import osb_page
    from core import outside_text_processor as ref
    from core.image import ocr_detection as refdet

    refdet.cv2 = types.SimpleNamespace(cvtColor=lambda a, code: np.ascontiguousarray(a[..., ::-1]), COLOR_RGB2BGR=4, COLOR_BGR2RGB=4)
    ref.cv2 = types.SimpleNamespace(
        dilate=lambda img, k, iterations=1: ndimage.maximum_filter(np.asarray(img), size=k.shape[0], mode="constant", cval=0),
        cvtColor=lambda a, code: np.asarray(a), morphologyEx=lambda a, op, k: a, erode=lambda a, k, iterations=1: a,
        findContours=lambda a, m, c: ([], None), contourArea=lambda c: 0.0, drawContours=lambda *a, **k: None,
        COLOR_RGB2LAB=0, COLOR_RGB2HSV=1, MORPH_CLOSE=3, RETR_EXTERNAL=0, CHAIN_APPROX_SIMPLE=2, FILLED=-1)
    ref._build_outside_text_data = lambda **kw: []
    ref._apply_inpaint_render_metadata = lambda *a, **k: None
    ref.FluxKontextInpainter = osb_page.StandInInpainter

    class Boxes:
        def __init__(self, xyxy, conf):
            self.xyxy, self.conf, self.cls = torch.tensor(xyxy, dtype=torch.float32).reshape(-1, 4), torch.tensor(conf, dtype=torch.float32), torch.zeros(len(conf))

    osb_model = lambda *a, **k: [types.SimpleNamespace(boxes=Boxes(osb_page.OSB, osb_page.OSB_CONF))]

    def boom(*a, **k):
        raise RuntimeError("bubbles are provided: no bubble detector may run")

    class Paths(dict):
        def __missing__(self, k):
            return "model.pt"

    mgr = types.SimpleNamespace(load_yolo_speech_bubble=boom, load_rtdetr_conjoined_bubble=boom, load_yolo_osbtext=lambda token=None: osb_model,
                                model_paths=Paths(), device="cpu")
    none = lambda *a, **k: None
    refdet.get_model_manager = lambda: mgr
    refdet.get_cache = lambda: types.SimpleNamespace(get_yolo_cache_key=none, get_yolo_detection=none, set_yolo_detection=none)
    refdet.get_best_device = lambda: "cpu"
    page = osb_page.make_page()
    out = {}
    for tag, method, over in [("flux", "flux_kontext", {}), ("opencv", "opencv", {}), ("none", "none", {}),
                              ("flux_no_coordinator", "flux_kontext", dict(_no_coord=True)),
                              ("min_area", "flux_kontext", dict(min_area_ignore_ratio=0.012, osb_render_expansion_narrow_multiplier=1.0))]:
        over = dict(over)
        coord = None if over.pop("_no_coord", False) else batch_coordinator.BatchRequestCoordinator(2)
        cfg = osb_page.make_config(coord, method, **over)
        osb_page.StandInInpainter.calls = []
        work = ref.prepare_outside_text_work(page, cfg, "page.png", "PNG", bubble_data=osb_page.bubble_data(), text_free_boxes=osb_page.TEXT_FREE,
                                             panels=osb_page.PANELS)
        prep = dict(results=[[[int(v) for v in b], float(c)] for b, c in work.outside_text_results],
                    raw=[[[float(v) for v in b], float(c)] for b, c in work.raw_outside_text_results],
                    colors={",".join(map(str, k)): bool(v) for k, v in work.original_text_colors.items()},
                    groups=[dict(bbox=g["bbox"], original_bbox=g["original_bbox"], mask_indices=[int(i) for i in g["mask_indices"]]) for g in work.mask_groups],
                    bubble_mask_sum=int(work.total_bubble_mask.sum()))
        STAGE_ARRAYS[f"{tag}_bubble_mask"] = np.packbits(work.total_bubble_mask)
        final, _ = ref.finish_outside_text_work(work)
        STAGE_ARRAYS[f"{tag}_final"] = np.asarray(final.convert("RGB"))
        out[tag] = dict(method=method, over=over, prepare=prep, calls=sorted(osb_page.StandInInpainter.calls, key=lambda c: c["seed"]))
    return out


STAGE_ARRAYS = {}

CLEAN_CASES = {      # name -> (page kwargs, operator kwargs, neighbours?, RGBA page?)
    "light": (dict(seed=0), dict(), False, False),
    "dark": (dict(seed=1, dark=True), dict(), False, False),
    "otsu": (dict(seed=2), dict(use_otsu_threshold=True), False, False),
    "scaled": (dict(seed=3), dict(processing_scale=1.5, roi_shrink_px=4), False, False),
    "colored": (dict(seed=4), dict(inpaint_colored_bubbles=True, inpaint_method="none"), False, False),
    # colored bubbles repainted by the configured FLUX inpainter (stand-in class): one by one, and in coordinator waves; BGRA page too
    "colored_flux": (dict(seed=4, gradient=True), dict(inpaint_colored_bubbles=True, inpaint_method="flux_klein_4b", flux_seed=5, flux_num_inference_steps=3, thresholding_value=120), False, False),
    "colored_flux_waves": (dict(seed=4, gradient=True), dict(inpaint_colored_bubbles=True, inpaint_method="flux_kontext", flux_seed=0, _coordinator=2, thresholding_value=120), False, True),
    "colored_flux_fails": (dict(seed=4, gradient=True), dict(inpaint_colored_bubbles=True, inpaint_method="flux_klein_9b", flux_seed=7, _fail=True, thresholding_value=120), False, False),
    "neighbors": (dict(seed=5), dict(), True, False),
    "border_rgba": (dict(seed=6, touch_border=True), dict(thresholding_value=180), False, True),
    "retry": (dict(seed=7, dark=True), dict(thresholding_value=250), False, False),       # nothing passes 250 on a dark bubble -> Otsu retry
}
CLEAN_ARRAYS = {}


def gen_cleaning():
    """core/image/cleaning.py:210-1140 — `clean_speech_bubbles` (incl. `process_single_bubble`, `_build_adaptive_shrink_mask`, the Otsu
    retry and the grouped flat fill) run on the synthetic bubble pages of tests/cleaning_checks.py with every cv2 primitive served by
    the restatement in oracle/cleaning_ref.py (tests/golden/cv2_shim.py): pins the reference's control flow around the primitives."""
    sys.path.insert(0, str(HERE.parent.parent)); sys.path.insert(0, str(HERE.parent)); sys.path.insert(0, str(HERE))
    import cv2_shim
    import cleaning_checks
    from core.image import cleaning as refc
    from core.image import image_utils as refu
    refc.cv2 = cv2_shim.namespace
    refu.cv2 = cv2_shim.namespace
    sys.path.insert(0, str(HERE.parent))
    from standin_inpainter import StandInInpainter        # deterministic stand-in for both FLUX inpainter classes (records its calls)
    refc.FluxKleinInpainter = StandInInpainter
    refc.FluxKontextInpainter = StandInInpainter
    out = {}
    for name, (pkw, okw, neigh, rgba) in CLEAN_CASES.items():
        okw = dict(okw)
        StandInInpainter.calls.clear()
        StandInInpainter.fail = bool(okw.pop("_fail", False))
        nco = okw.pop("_coordinator", 0)
        if nco:
            okw["request_coordinator"] = batch_coordinator.BatchRequestCoordinator(nco)
        page, masks, bboxes = cleaning_checks.make_page(**pkw)
        dets = []
        for i, (m, bb) in enumerate(zip(masks, bboxes)):
            d = {"bbox": tuple(int(v) for v in bb), "confidence": 0.9, "class": "bubble", "sam_mask": m}
            if neigh:
                d["conjoined_neighbor_bboxes"] = [tuple(int(v) for v in bboxes[1 - i])]
            dets.append(d)
        rgb = page[..., ::-1]
        pil = Image.fromarray(np.dstack([rgb, np.full(rgb.shape[:2], 255, np.uint8)]) if rgba else np.ascontiguousarray(rgb))
        cleaned, info = refc.clean_speech_bubbles(pil, None, pre_computed_detections=dets, **okw)
        okw.pop("request_coordinator", None)
        if nco:
            okw["_coordinator"] = nco
        if StandInInpainter.fail:
            okw["_fail"] = True
        CLEAN_ARRAYS[f"{name}_cleaned"] = np.asarray(cleaned)
        CLEAN_ARRAYS[f"{name}_masks"] = np.packbits(np.stack([b["mask"] > 0 for b in info])) if info else np.zeros(0, np.uint8)
        out[name] = dict(page=pkw, op={k: v for k, v in okw.items()}, neighbors=neigh, rgba=rgba,
                         bubbles=[dict(bbox=[int(v) for v in b["bbox"]], color=[int(v) for v in b["color"]], is_colored=bool(b["is_colored"]),
                                       text_bbox=[int(v) for v in b["text_bbox"]] if b.get("text_bbox") is not None else None,
                                       is_sam=bool(b["is_sam"]), inpainted=bool(b.get("inpainted", False))) for b in info],
                         inpaint_calls=sorted(StandInInpainter.calls, key=lambda c: c["seed"]))
    return out


UPSCALE_ARRAYS = {}


def fake_upscaler(t):
    """deterministic stand-in for the RCAN model: nearest-neighbour 2x plus a position-dependent tint (float32 in, float32 out)"""
    up = t.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)
    yy = torch.arange(up.shape[2], dtype=torch.float32).view(1, 1, -1, 1)
    xx = torch.arange(up.shape[3], dtype=torch.float32).view(1, 1, 1, -1)
    return up * 0.9 + ((yy * 3 + xx * 5) % 17) / 255.0


def gen_upscale():
    """core/image/image_utils.py:351-548 — `upscale_image` / `upscale_image_to_dimension` / tensor conversions with a stand-in 2x
    model: pass count, u8 truncation between passes, the final exact-size LANCZOS resize, RGBA / L inputs."""
    from core.image import image_utils as refu
    none = lambda *a, **k: None
    refu.get_cache = lambda: types.SimpleNamespace(get_upscale_cache_key=none, get_upscale_dimension_cache_key=none, get_upscaled_image=none, set_upscaled_image=none)
    mgr = types.SimpleNamespace(load_upscale=lambda *a, **k: fake_upscaler, load_upscale_lite=lambda *a, **k: fake_upscaler, device=torch.device("cpu"))
    refu.get_model_manager = lambda: mgr
    rng = np.random.default_rng(11)
    out = {}
    for name, (mode, size, factor, mtype) in dict(x2=("RGB", (23, 17), 2.0, "model"), x1_5=("RGB", (23, 17), 1.5, "model_lite"),
                                                  x3=("RGB", (9, 14), 3.0, "model"), x1=("RGB", (8, 8), 1.0, "model"),
                                                  rgba=("RGBA", (12, 10), 2.5, "model"), gray=("L", (10, 12), 2.0, "model")).items():
        arr = (rng.random((size[1], size[0], {"RGB": 3, "RGBA": 4, "L": 1}[mode])) * 255).astype(np.uint8)
        img = Image.fromarray(arr[..., 0] if mode == "L" else arr, mode)
        res = refu.upscale_image(img, factor, model_type=mtype)
        UPSCALE_ARRAYS[f"{name}_in"] = arr
        UPSCALE_ARRAYS[f"{name}_out"] = np.asarray(res)
        out[name] = dict(mode=mode, size=list(size), factor=factor, model_type=mtype, out_mode=res.mode, out_size=list(res.size))
    return out


BATCH_TREE = ["P1.png", "p2.png", "p10.png", "ch2/001.jpg", "ch2/010.jpg", "ch10/001.jpg", "ch10/notes.txt", "x.webp", "cover.JPEG", "thumbs.db"]
BATCH_FAIL = ["p2.png", "010.jpg"]


def gen_batch():
    """core/pipeline.py:2481-2733 `batch_translate_images` (sequential branch) with `translate_and_render` replaced by a recorder that
    fails for two files: page list and order, output paths, results dict, failed_paths.txt — for the three output formats and both
    directory modes."""
    import tempfile
    from core import pipeline as refp
    refp._build_previous_context_images = lambda *a, **k: None
    refp._build_previous_context_texts = lambda *a, **k: None
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        root = Path(tmp) / "in"
        for rel in BATCH_TREE:
            f = root / rel
            f.parent.mkdir(parents=True, exist_ok=True)
            if f.suffix.lower() in (".png", ".jpg", ".jpeg", ".webp"):
                Image.new("RGB", (8, 8), (200, 10, 10)).save(f)
            else:
                f.write_text("x")
        for tag, preserve, fmt in [("flat_png", False, "png"), ("tree_auto", True, "auto"), ("flat_jpeg", False, "jpeg"), ("tree_png", True, "png")]:
            seen = []
            odir = Path(tmp) / f"out_{tag}"

            def fake(img_path, config, output_path, **kw):
                seen.append([str(Path(img_path).relative_to(root)), str(Path(output_path).relative_to(odir))])
                if Path(img_path).name in BATCH_FAIL:
                    raise RuntimeError(f"boom: {Path(img_path).name}")
            refp.translate_and_render = fake
            cfg = types.SimpleNamespace(parallel_requests=1, verbose=False, retry_failed_once=False,
                                        output=types.SimpleNamespace(output_format=fmt, jpeg_quality=95, png_compression=2))
            res = refp.batch_translate_images(root, cfg, odir, preserve_structure=preserve)
            ff = res.get("failed_paths_file")
            out[tag] = dict(preserve=preserve, fmt=fmt, seen=seen, success_count=res["success_count"], error_count=res["error_count"],
                            errors=res["errors"], failed=[str(Path(p_).relative_to(root.resolve())) for p_ in res["failed_image_paths"]],
                            failed_file_lines=[str(Path(l).relative_to(root.resolve())) for l in Path(ff).read_text().split()] if ff else None,
                            failed_file_name=Path(ff).name if ff else None)
    return dict(tree=BATCH_TREE, fail=BATCH_FAIL, runs=out)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "cleaning":          # regenerate the cleaning-flow fixtures only
        json.dump(gen_cleaning(), open(HERE / "cleaning_flow.json", "w"))
        np.savez_compressed(HERE / "cleaning_flow_arrays.npz", **CLEAN_ARRAYS)
        raise SystemExit(0)
    json.dump(gen_upscale(), open(HERE / "upscale_flow.json", "w"))
    np.savez_compressed(HERE / "upscale_flow_arrays.npz", **UPSCALE_ARRAYS)
    json.dump(gen_cleaning(), open(HERE / "cleaning_flow.json", "w"))
    np.savez_compressed(HERE / "cleaning_flow_arrays.npz", **CLEAN_ARRAYS)
    json.dump(gen_batch(), open(HERE / "batch_harness.json", "w"))
    json.dump(gen_osb_stage(), open(HERE / "osb_stage.json", "w"))
    np.savez_compressed(HERE / "osb_stage_arrays.npz", **STAGE_ARRAYS)
    json.dump(gen_osb(), open(HERE / "osb_regions.json", "w"))
    np.savez_compressed(HERE / "osb_regions_masks.npz", **OSB_MASKS)
    json.dump(gen_detection_flow(), open(HERE / "detection_flow.json", "w"))
    np.savez_compressed(HERE / "detection_flow_masks.npz", **DET_MASKS)
    cj_meta, cj_arrays = gen_conjoined()
    json.dump(cj_meta, open(HERE / "conjoined.json", "w"))
    np.savez_compressed(HERE / "conjoined.npz", **cj_arrays)
    json.dump(gen_boxes(), open(HERE / "box_hygiene.json", "w"))
    json.dump(gen_coordinator(), open(HERE / "batch_coordinator.json", "w"))
    np.savez_compressed(HERE / "batch_coordinator_masks.npz", **COORD_MASKS)
    geo, arrays, consts = gen_inpaint()
    json.dump(dict(cases=geo, consts=consts), open(HERE / "kontext_geometry.json", "w"))
    np.savez_compressed(HERE / "kontext_arrays.npz", **arrays)
    json.dump(gen_harness(), open(HERE / "harness.json", "w"))
    print("golden vectors written to", HERE)
