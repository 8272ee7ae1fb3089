"""CPU tier: RT-DETR-v2 graphs on the kernel simulator vs HF RTDetrV2ForObjectDetection (tiny geometry)."""
import rtdetr_checks as rc


def test_rtdetr_tiny(emu_lib):
    rc.check_raw(emu_lib, "cpu", hw=(64, 96))


def test_rtdetr_call_shape(emu_lib):
    rc.check_call_shape(emu_lib, "cpu")


def test_oracle_pre_post_matches_reference_adapter():
    """goldens from the REFERENCE's RTDetrYOLOAdapter around HF's RTDetrImageProcessor (tests/golden/make_rtdetr_adapter_golden.py): the
    restated resize / rescale / sigmoid / top-k over queries x classes / box scaling / threshold of oracle.rtdetr_ref.predict"""
    import json
    from pathlib import Path
    import numpy as np
    import torch
    from PIL import Image
    from oracle import rtdetr_ref
    g = json.loads((Path(__file__).resolve().parent / "golden" / "rtdetr_adapter.json").read_text())
    model, _ = rtdetr_ref.make_model("tiny_test", seed=g["model_seed"])
    rtdetr_ref.spread_class_scores(model)
    page = (np.random.default_rng(g["page_seed"]).random(tuple(g["page_shape"])) * 255).astype(np.uint8)
    pil = Image.fromarray(np.ascontiguousarray(page[..., ::-1]))
    for tag, run in g["runs"].items():
        xyxy, scores, labels = rtdetr_ref.predict(model, pil, conf=run["conf"], imgsz=g["imgsz"])
        assert len(scores) == len(run["scores"]), tag
        assert torch.allclose(scores, torch.tensor(run["scores"]), atol=2e-6), tag
        assert labels.tolist() == [int(c) for c in run["cls"]], tag
        assert torch.allclose(xyxy, torch.tensor(run["xyxy"]).reshape(-1, 4), atol=2e-3), tag
    assert len(g["runs"]["bgr_mid"]["scores"]) == 10 and g["runs"]["bgr_lo"]["xyxy"] == g["runs"]["pil_lo"]["xyxy"]


def test_rtdetr_batcher_matches_single_calls(emu_lib):
    """RT-DETR's backbone + encoder shared by the pages of a batch (core/ml/detector_batch.py RTDetrBatcher): 3 pages in a batch of 4, 4 pages in
    batches of 2, 5 pages from their own threads — boxes, scores and classes are the one-page call's bytes"""
    rc.check_batched(emu_lib, "cpu")
    rc.check_batched(emu_lib, "cpu", pages=4, batch=2, seed=2)
    rc.check_batched(emu_lib, "cpu", pages=5, batch=2, seed=3, threads=True)
