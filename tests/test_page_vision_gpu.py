"""GPU tier: one synthetic page through `process_page_vision` with every stage on the real HIP path (libmtx_hip through the C ABI) —
YOLO-seg detector, RT-DETR-v2 secondary detector, SAM-2.1 (processor / model shims of the manager), the OSB stage with the FLUX
Kontext pipeline, bubble cleaning kernels, RCAN final upscale — small seeded geometries of each network, loaded through the
ModelManager slots the operators ask for.  Checks that the operators and the model objects fit together end to end (call shapes,
result objects, devices, modes) and that the page leaves every stage changed the way that stage changes it."""
import types

import numpy as np
import pytest
import torch
from PIL import Image

pytestmark = pytest.mark.gpu


def _config(yolo_conf, coordinator):
    return types.SimpleNamespace(
        device=torch.device("cuda:0"), yolo_model_path=None, upscaling_only=False, request_coordinator=coordinator,
        preprocessing=types.SimpleNamespace(auto_scale=True),
        detection=types.SimpleNamespace(confidence=yolo_conf, seg_model="sam2", conjoined_detection=True, conjoined_confidence=0.35,
                                        use_osb_text_verification=False, bubble_detector_model="yolo_1"),
        cleaning=types.SimpleNamespace(thresholding_value=200, use_otsu_threshold=False, roi_shrink_px=5, inpaint_colored_bubbles=False),
        outside_text=types.SimpleNamespace(
            enabled=True, enable_page_number_filtering=False, min_area_ignore_ratio=0.0, seed=1, huggingface_token="",
            inpainting_method="flux_kontext", flux_backend="sdnq", flux_low_vram=False, flux_num_inference_steps=2,
            flux_group_regions=False, flux_residual_diff_threshold=0.15, osb_confidence=0.5, osb_text_free_only=False,
            bbox_expansion_percent_width=0.1, bbox_expansion_percent_height=0.1, osb_render_expansion_narrow_multiplier=1.0,
            osb_render_expansion_tiny_multiplier=1.0, text_box_proximity_ratio=0.02),
        output=types.SimpleNamespace(upscale_final_image=True, image_upscale_factor=2.0, image_upscale_model="model_lite"))


def test_page_through_all_stages(hip_lib, monkeypatch):
    import flux_checks
    from oracle import rcan_ref, rtdetr_ref, sam2_ref, yolo_ref
    from mangatranslator_amd.core import pipeline
    from mangatranslator_amd.core.batch_coordinator import BatchRequestCoordinator
    from mangatranslator_amd.core.ml import flux as fx
    from mangatranslator_amd.core.ml import model_manager as mm
    from mangatranslator_amd.core.ml.rcan import RCANUpscaler
    from mangatranslator_amd.core.ml.rtdetr import RTDetrHip
    from mangatranslator_amd.core.ml.sam2 import Sam2Hip
    from mangatranslator_amd.core.ml.yolo import YoloSegHip
    from mangatranslator_amd.utils.synthetic_pages import make_page
    dev = torch.device("cuda:0")
    mgr = mm.get_model_manager()
    monkeypatch.setattr(mgr, "device", dev)

    ynet = yolo_ref.make_model("n", 1, seed=3)
    with torch.no_grad():
        for l in range(3):
            ynet.model[22].cv3[l][2].weight.mul_(0.05); ynet.model[22].cv3[l][2].bias.fill_(-1.0)
            ynet.model[22].cv2[l][2].weight.mul_(0.1)
    yolo = YoloSegHip(ynet.state_dict(), device=dev, lib=hip_lib, names={0: "speech_bubble"})
    rmodel, rcfg = rtdetr_ref.make_model("tiny_test", seed=5)
    rtdetr = RTDetrHip(rmodel.state_dict(), rcfg, device=dev, lib=hip_lib, names={0: "bubble", 1: "text_bubble", 2: "text_free"})
    smodel, scfg = sam2_ref.make_model("tiny_test", seed=2)
    sam = Sam2Hip(smodel.state_dict(), scfg, device=dev, lib=hip_lib)
    t, v = flux_checks.models(seed=4)
    dit, vae = flux_checks.hip_models(t, v, hip_lib, dev)
    flux = fx.FluxKontextHip(dit, vae)
    g = torch.Generator().manual_seed(9)
    flux.set_prompt_embeds(torch.randn(16, t.cfg["joint_dim"], generator=g), torch.randn(t.cfg["pooled_dim"], generator=g))
    rcan = RCANUpscaler(rcan_ref.make_state_dict(n_feats=64, n_resgroups=2, n_resblocks=2, unshuffle=1, seed=7), device=dev, lib=hip_lib)
    for slot, obj in [(mm.ModelType.YOLO_SPEECH_BUBBLE, yolo), (mm.ModelType.RTDETR_CONJOINED_BUBBLE, rtdetr),
                      (mm.ModelType.SAM2, (mm._Sam2ProcessorShim(), mm._Sam2ModelShim(sam, torch.bfloat16))),
                      (mm.ModelType.FLUX_KONTEXT_SDNQ_PIPELINE, flux), (mm.ModelType.UPSCALE_LITE, rcan)]:
        monkeypatch.setitem(mgr.models, slot, obj)

    W, H = 512, 768
    pg, boxes, regions = make_page(3, W, H, bubbles=8, osb_regions=1)
    page = Image.fromarray(pg).convert("RGBA")
    # seeded weights have arbitrary scores: a threshold that lets about a dozen anchors through (as bench.py does)
    yolo(np.ascontiguousarray(pg[..., ::-1]), conf=0.0, imgsz=640, max_det=1)
    plan0, _ = next(iter(yolo._plans.values()))
    sc = plan0.decoded[:, 4].float().sort(descending=True).values
    conf = float(sc[min(12, len(sc) - 1)])
    cfg = _config(conf, BatchRequestCoordinator(1))
    x0, y0, x1, y1 = regions[0]
    calls0 = flux.calls

    # the OSB text boxes come from the generator (the OSB text network is not built: the manager's loader raises and the
    # detector falls back to text_free boxes, which here are appended to what the secondary detector reports)
    from mangatranslator_amd.core import outside_text_processor as otp
    real = otp.prepare_outside_text_work          # the OSB stage's prepare half (the page's front half calls it: core/pipeline.py)

    def prepare_with_ground_truth(p, c, path, fmt, verbose=False, bubble_data=None, text_free_boxes=None, panels=None):
        return real(p, c, path, fmt, verbose=verbose, bubble_data=bubble_data, text_free_boxes=list(text_free_boxes or []) + [[x0 + 5.0, y0 + 5.0, x1 - 5.0, y1 - 5.0]], panels=panels)
    monkeypatch.setattr(otp, "prepare_outside_text_work", prepare_with_ground_truth)

    out, info = pipeline.process_page_vision(page, cfg)
    assert out.mode == "RGBA" and out.size == (2 * W, 2 * H)
    assert len(info["bubbles"]) >= 1, "the calibrated threshold lets detections through"
    for b in info["bubbles"]:
        assert b["sam_mask"].shape == (H, W) and b["sam_mask"].dtype == np.uint8 and len(b["bbox"]) == 4
    assert flux.calls >= calls0 + 1, "the text block on the gradient went through the Kontext pipeline"
    assert info["processing_scale"] == pytest.approx((W * H / 1e6) ** 0.5)
    a = np.asarray(out.convert("RGB"))
    assert a.std() > 5 and np.isfinite(a).all()

    # the same page with SAM switched off: masks come from the detector's own instance masks (`results.masks`), and the cleaning
    # chain accepts those detections
    from mangatranslator_amd.core.image import cleaning, detection
    dets, _tf = detection.detect_speech_bubbles("page.png", None, conf, device=dev, seg_model="none", conjoined_detection=True,
                                                image_override=page, bubble_detector_model="yolo_1")
    assert len(dets) >= 1 and all(d["sam_mask"].shape == (H, W) for d in dets)
    cleaned, cinfo = cleaning.clean_speech_bubbles(page, None, pre_computed_detections=dets, device=dev, processing_scale=info["processing_scale"])
    assert cleaned.shape == (H, W, 4) and isinstance(cinfo, list)


def test_batch_with_two_front_halves_in_flight(hip_lib, monkeypatch, tmp_path):
    """`batch_vision_images(front_workers=2)` on the HIP path: two pages' detect stages (YOLO-seg, RT-DETR-v2, SAM-2.1) run at once on two
    instance sets served by `ModelManager.front_replica`, cleaning follows in the back half; the written pages equal those of the
    one-front-half run byte for byte"""
    from oracle import rtdetr_ref, sam2_ref, yolo_ref
    from mangatranslator_amd.core import pipeline
    from mangatranslator_amd.core.ml import model_manager as mm
    from mangatranslator_amd.core.ml.rtdetr import RTDetrHip
    from mangatranslator_amd.core.ml.sam2 import Sam2Hip
    from mangatranslator_amd.core.ml.yolo import YoloSegHip
    from mangatranslator_amd.utils.synthetic_pages import make_page
    dev = torch.device("cuda:0")
    mgr = mm.get_model_manager()
    monkeypatch.setattr(mgr, "device", dev)
    ynet = yolo_ref.make_model("n", 1, seed=3)
    with torch.no_grad():
        for l in range(3):
            ynet.model[22].cv3[l][2].weight.mul_(0.05); ynet.model[22].cv3[l][2].bias.fill_(-1.0)
            ynet.model[22].cv2[l][2].weight.mul_(0.1)
    rmodel, rcfg = rtdetr_ref.make_model("tiny_test", seed=5)
    smodel, scfg = sam2_ref.make_model("tiny_test", seed=2)
    sets = []
    for r in range(2):
        yolo = YoloSegHip(ynet.state_dict(), device=dev, lib=hip_lib, names={0: "speech_bubble"})
        rtdetr = RTDetrHip(rmodel.state_dict(), rcfg, device=dev, lib=hip_lib, names={0: "bubble", 1: "text_bubble", 2: "text_free"})
        sam = Sam2Hip(smodel.state_dict(), scfg, device=dev, lib=hip_lib, precision="high")      # what the batch run asks the manager for (a "fast" instance would be unloaded and re-read from disk)
        sets.append((yolo, rtdetr, sam))
        for mt, obj in [(mm.ModelType.YOLO_SPEECH_BUBBLE, yolo), (mm.ModelType.RTDETR_CONJOINED_BUBBLE, rtdetr),
                        (mm.ModelType.SAM2, (mm._Sam2ProcessorShim(), mm._Sam2ModelShim(sam, torch.bfloat16)))]:
            monkeypatch.setitem(mgr.models, mt if r == 0 else (mt, r), obj)
    W, H = 512, 768
    root = tmp_path / "in"
    root.mkdir()
    n = 6
    for i in range(n):
        pg, _boxes, _regions = make_page(20 + i, W, H, bubbles=8, osb_regions=0)
        Image.fromarray(pg).save(root / f"p{i}.png")
        if i == 0:
            sets[0][0](np.ascontiguousarray(pg[..., ::-1]), conf=0.0, imgsz=640, max_det=1)
            plan0, _ = next(iter(sets[0][0]._plans.values()))
            sc = plan0.decoded[:, 4].float().sort(descending=True).values
            conf = float(sc[min(12, len(sc) - 1)])
    cfg = _config(conf, None)
    cfg.outside_text.enabled = False
    cfg.output = types.SimpleNamespace(upscale_final_image=False, image_upscale_factor=1.0, image_upscale_model="model_lite", output_format="png",
                                       jpeg_quality=95, png_compression=2)
    cfg.verbose = False
    used = {0: 0, 1: 0}
    real_front = pipeline.process_page_vision_front

    def counting_front(*a, **k):
        used[mgr.current_front_replica()] += 1
        return real_front(*a, **k)
    monkeypatch.setattr(pipeline, "process_page_vision_front", counting_front)
    two = pipeline.batch_vision_images(root, cfg, tmp_path / "two", front_workers=2)
    assert two["success_count"] == n and two["io"]["pages_in_flight"] == 3 and used[0] > 0 and used[1] > 0
    assert mgr.models[mm.ModelType.SAM2][1].hip is sets[0][2], "the batch kept the staged SAM instance (no reload, no fallback to the detector's masks)"
    one = pipeline.batch_vision_images(root, cfg, tmp_path / "one", front_workers=1)
    assert one["success_count"] == n
    changed = 0
    for i in range(n):
        a = np.asarray(Image.open(tmp_path / "two" / f"p{i}_translated.png").convert("RGBA"))
        b = np.asarray(Image.open(tmp_path / "one" / f"p{i}_translated.png").convert("RGBA"))
        assert np.array_equal(a, b), i
        changed += int(not np.array_equal(a[..., :3], np.asarray(Image.open(root / f"p{i}.png").convert("RGB"))))
    assert changed >= 1, "cleaning changed at least one page (the calibrated threshold lets detections through)"


def test_batch_front_halves_share_detector_batches(hip_lib, monkeypatch, tmp_path):
    """`batch_vision_images(front_workers=2)` hands the panel detector out behind ONE DetectorBatcher (core/ml/detector_batch.py): the front halves
    that run side by side share its graph replays, and every page's panels — and the written page — equal the one-front-half run's (no wrapper, one
    page per replay) exactly"""
    from oracle import rtdetr_ref, sam2_ref, yolo11_ref, yolo_ref
    from mangatranslator_amd.core import pipeline
    from mangatranslator_amd.core.ml import model_manager as mm
    from mangatranslator_amd.core.ml.detector_batch import DetectorBatcher
    from mangatranslator_amd.core.ml.rtdetr import RTDetrHip
    from mangatranslator_amd.core.ml.sam2 import Sam2Hip
    from mangatranslator_amd.core.ml.yolo import YoloSegHip
    from mangatranslator_amd.core.ml.yolo11 import Yolo11Hip
    from mangatranslator_amd.utils.synthetic_pages import make_page
    dev = torch.device("cuda:0")
    mgr = mm.get_model_manager()
    monkeypatch.setattr(mgr, "device", dev)
    ynet = yolo_ref.make_model("n", 1, seed=3)
    with torch.no_grad():
        for l in range(3):
            ynet.model[22].cv3[l][2].weight.mul_(0.05); ynet.model[22].cv3[l][2].bias.fill_(-1.0)
            ynet.model[22].cv2[l][2].weight.mul_(0.1)
    rmodel, rcfg = rtdetr_ref.make_model("tiny_test", seed=5)
    smodel, scfg = sam2_ref.make_model("tiny_test", seed=2)
    for r in range(2):
        yolo = YoloSegHip(ynet.state_dict(), device=dev, lib=hip_lib, names={0: "speech_bubble"})
        rtdetr = RTDetrHip(rmodel.state_dict(), rcfg, device=dev, lib=hip_lib, names={0: "bubble", 1: "text_bubble", 2: "text_free"})
        sam = Sam2Hip(smodel.state_dict(), scfg, device=dev, lib=hip_lib, precision="high")      # what the batch run asks the manager for (a "fast" instance would be unloaded and re-read from disk)
        if r == 0:
            first_yolo = yolo
        for mt, obj in [(mm.ModelType.YOLO_SPEECH_BUBBLE, yolo), (mm.ModelType.RTDETR_CONJOINED_BUBBLE, rtdetr),
                        (mm.ModelType.SAM2, (mm._Sam2ProcessorShim(), mm._Sam2ModelShim(sam, torch.bfloat16)))]:
            monkeypatch.setitem(mgr.models, mt if r == 0 else (mt, r), obj)
    panel = Yolo11Hip(yolo11_ref.make_model("11", "n", 1, False, seed=7).state_dict(), device=dev, lib=hip_lib, names={0: "frame"})
    monkeypatch.setitem(mgr.models, mm.ModelType.YOLO_PANEL, panel)
    monkeypatch.setattr(mgr, "_batchers", {})
    W, H = 512, 768
    root = tmp_path / "in"
    root.mkdir()
    n = 8
    for i in range(n):
        pg, _boxes, _regions = make_page(40 + i, W, H, bubbles=8, osb_regions=0)
        Image.fromarray(pg).save(root / f"p{i}.png")
        if i == 0:
            first_yolo(np.ascontiguousarray(pg[..., ::-1]), conf=0.0, imgsz=640, max_det=1)
            plan0, _ = next(iter(first_yolo._plans.values()))
            sc = plan0.decoded[:, 4].float().sort(descending=True).values
            conf = float(sc[min(12, len(sc) - 1)])
            panel(np.ascontiguousarray(pg[..., ::-1]), conf=0.0, imgsz=640, max_det=1)
            pplan, _ = next(iter(panel._plans.values()))
            ps = pplan.decoded[:, 4].float().sort(descending=True).values
            panel_conf = float(ps[len(ps) // 3])          # a third of the anchors pass on this page (the page flow may rescale pages: a threshold at the very top would pass nothing)
    cfg = _config(conf, None)
    cfg.detection.use_panel_sorting, cfg.detection.panel_confidence = True, panel_conf
    cfg.outside_text.enabled = False
    cfg.output = types.SimpleNamespace(upscale_final_image=False, image_upscale_factor=1.0, image_upscale_model="model_lite", output_format="png",
                                       jpeg_quality=95, png_compression=2)
    cfg.verbose = False
    seen = {}
    from pathlib import Path
    from mangatranslator_amd.core.image import detection as detection_mod
    real_panels = detection_mod.detect_panels

    def recording_panels(image_path, *a, **k):
        out = real_panels(image_path, *a, **k)
        seen.setdefault(mgr.detector_batch, {})[Path(str(image_path)).name] = out
        return out
    monkeypatch.setattr(detection_mod, "detect_panels", recording_panels)      # (the page flow imports it at call time)
    two = pipeline.batch_vision_images(root, cfg, tmp_path / "two", front_workers=2)
    wrapper = mgr._batchers.get(mm.ModelType.YOLO_PANEL)
    assert two["success_count"] == n and isinstance(wrapper, DetectorBatcher) and wrapper.model is panel
    assert wrapper.stats["pages"] == n and 1 <= wrapper.stats["launches"] <= n
    from mangatranslator_amd.core.ml.detector_batch import RTDetrBatcher
    rwrap = mgr._batchers.get(mm.ModelType.RTDETR_CONJOINED_BUBBLE)
    assert isinstance(rwrap, RTDetrBatcher) and rwrap.stats["pages"] == n and 1 <= rwrap.stats["launches"] <= n, "the secondary detector is shared the same way"
    assert mgr.detector_batch == 1, "the batch run restores the manager's setting"
    one = pipeline.batch_vision_images(root, cfg, tmp_path / "one", front_workers=1)
    assert one["success_count"] == n and wrapper.stats["pages"] == n, "one front half: the detector is called directly"
    assert set(seen) == {1, 2} and len(seen[1]) == n and len(seen[2]) == n
    assert sum(len(v or []) for v in seen[1].values()) > 0, "no page produced a panel: the comparison is empty"
    for name in seen[1]:
        assert seen[1][name] == seen[2][name], f"{name}: panels differ between the batched and the one-page call"
    for i in range(n):
        a = np.asarray(Image.open(tmp_path / "two" / f"p{i}_translated.png").convert("RGBA"))
        b = np.asarray(Image.open(tmp_path / "one" / f"p{i}_translated.png").convert("RGBA"))
        assert np.array_equal(a, b), i
    print(f"panel detector: {wrapper.stats['pages']} pages in {wrapper.stats['launches']} graph replays; RT-DETR: {rwrap.stats['pages']} in {rwrap.stats['launches']}")
