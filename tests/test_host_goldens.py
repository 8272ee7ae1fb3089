"""Host-side integer logic vs golden vectors generated from the reference (tests/golden/make_goldens.py).
Bit-exact: kept indices, groupings, bboxes, wave partitions, crop rectangles, alpha, composited pixels."""
import json
import types
from pathlib import Path

import numpy as np
import pytest
import torch
from PIL import Image

from mangatranslator_amd.core import batch_coordinator as bc
from mangatranslator_amd.core import pipeline as pl
from mangatranslator_amd.core import scaling
from mangatranslator_amd.core.image import box_ops

G = Path(__file__).resolve().parent / "golden"


def test_box_hygiene_matches_reference():
    cases = json.load(open(G / "box_hygiene.json"))
    assert len(cases) >= 20
    for c in cases:
        boxes = np.asarray(c["boxes"], np.float32).reshape(-1, 4)
        conf = np.asarray(c["conf"], np.float32)
        sec = np.asarray(c["secondary"], np.float32).reshape(-1, 4)
        assert box_ops.deduplicate_primary_boxes(boxes, conf, 0.7) == c["dedup_keep"]
        assert box_ops.remove_contained_boxes(boxes, 0.9) == c["contained_keep"]
        if len(boxes) and len(sec):
            conj, simple = box_ops.categorize_detections(boxes, sec)
            assert [[p, s] for p, s in conj] == c["conjoined"]
            assert simple == c["simple"]
        groups, rest = box_ops.detect_overlapping_primaries(boxes, c["simple"]) if len(boxes) else ([], [])
        assert groups == c["synthetic_groups"] and rest == c["simple_after"]


def test_box_hygiene_accepts_torch_and_edge_cases():
    assert box_ops.deduplicate_primary_boxes(torch.zeros(0, 4), torch.zeros(0)) == []
    assert box_ops.remove_contained_boxes(torch.tensor([[0., 0., 10., 10.]])) == [0]
    b = torch.tensor([[0., 0., 10., 10.], [0., 0., 10., 10.]])
    assert box_ops.deduplicate_primary_boxes(b, torch.tensor([0.5, 0.5])) == [0]       # tie keeps the first
    assert box_ops.remove_contained_boxes(b) == [1]                                      # order-dependent rule
    assert box_ops.categorize_detections(b[:1], torch.zeros(0, 4)) == ([], [0])


def test_batch_coordinator_matches_reference():
    g = json.load(open(G / "batch_coordinator.json"))
    masks = np.load(G / "batch_coordinator_masks.npz")
    for i, c in enumerate(g["expanded"]):
        m = np.unpackbits(masks[f"m{i}"])[: c["h"] * c["w"]].reshape(c["h"], c["w"]).astype(bool)
        got = bc.expanded_mask_bbox(m, (c["w"], c["h"]))
        assert (list(got) if got else None) == c["bbox"]
    assert bc.expanded_mask_bbox(np.zeros((5, 5), bool), (5, 5)) is None
    for c in g["waves"]:
        items = c["bboxes"]
        waves = bc.partition_non_overlapping_waves(range(len(items)), lambda i: tuple(items[i]) if items[i] else None)
        assert waves == c["waves"]
    assert not bc.bboxes_overlap((0, 0, 10, 10), (10, 0, 20, 10)) and bc.bboxes_overlap((0, 0, 10, 10), (9, 9, 20, 20))


def test_request_coordinator_semantics():
    co = bc.BatchRequestCoordinator(2)
    assert co.map_ordered([]) == [] and co.map_ordered([lambda: 7]) == [7]
    assert co.map_ordered([(lambda i=i: i * i) for i in range(6)]) == [0, 1, 4, 9, 16, 25]
    with co.slot():
        assert co.in_slot()
        with co.slot():          # same-thread re-entry does not take a second slot
            assert co.in_slot()
    assert not co.in_slot()


def _inpainter():
    from mangatranslator_amd.core.image import inpainting
    inp = inpainting.FluxKontextInpainter.__new__(inpainting.FluxKontextInpainter)
    inp.context_padding_ratio, inp.max_context_padding = inpainting.CONTEXT_PADDING_RATIO, inpainting.MAX_CONTEXT_PADDING
    inp.PREFERED_KONTEXT_RESOLUTIONS = list(inpainting.PREFERRED_KONTEXT_RESOLUTIONS)
    inp.num_inference_steps, inp.guidance_scale, inp.prompt = 4, 2.5, "Remove all text."
    inp.backend, inp.residual_diff_threshold = "sdnq", 0.12
    from mangatranslator_amd.core import caching
    inp.cache = caching.UnifiedCache()
    inp._prompt_embeds = None
    inp.manager = types.SimpleNamespace(flux_inference_lock=__import__("threading").Lock())
    inp.load_models = lambda: None

    class Out:
        def __init__(self, img):
            self.images = [img]

    def fake(**kw):   # the same stand-in pipeline the golden generator gave the reference
        img = kw["image"].convert("RGB").resize((kw["width"], kw["height"]), Image.BILINEAR)
        t = torch.from_numpy(np.asarray(img, dtype=np.float32) / 255.0).permute(2, 0, 1)
        return Out(1.0 - 0.9 * t)
    inp.pipeline = fake
    return inp


def test_kontext_region_and_composite_match_reference():
    g = json.load(open(G / "kontext_geometry.json"))
    arr = np.load(G / "kontext_arrays.npz")
    inp = _inpainter()
    assert g["consts"]["max_context_padding"] == inp.max_context_padding
    for i, c in enumerate(g["cases"]):
        h, w = c["h"], c["w"]
        mask = np.unpackbits(arr[f"mask{i}"])[: h * w].reshape(h, w).astype(bool)
        yy, xx = np.mgrid[0:h, 0:w]
        page = np.stack([(xx * 3 + yy) % 256, (xx + yy * 2) % 256, (xx * yy // 7) % 256], -1).astype(np.uint8)
        # un-quantised geometry + feather alpha (reference compute_mask_bbox_aspect_ratio)
        mt = torch.from_numpy(mask.astype(np.float32))[None, None]
        alpha, x, y, ww, hh = inp.compute_mask_bbox_aspect_ratio(mt, c["padding"], c["blur"], preferred_resolutions=inp.PREFERED_KONTEXT_RESOLUTIONS)
        assert [x, y, ww, hh] == c["bbox"]
        assert np.array_equal(alpha.numpy(), arr[f"alpha{i}"])
        # whole operator with the stand-in pipeline: bit-exact composited page
        out = np.asarray(inp.inpaint_mask(Image.fromarray(page), mask, seed=1, strict_mask_clipping=c["strict"],
                                          composite_clip_bbox=c["clip"]))
        y0, y1, x0, x1 = c["window"]
        assert np.array_equal(out[y0:y1, x0:x1], arr[f"out{i}"]), f"case {i}: composited crop differs"
        outside = np.ones((h, w), bool)
        outside[y0:y1, x0:x1] = False
        assert np.array_equal(out[outside], page[outside])


def test_inpaint_empty_mask_returns_same_object():
    inp = _inpainter()
    img = Image.new("RGB", (32, 32))
    assert inp.inpaint_mask(img, np.zeros((32, 32), bool)) is img


def test_harness_matches_reference(tmp_path):
    g = json.load(open(G / "harness.json"))
    assert sorted(g["names"], key=lambda s: pl._natural_path_sort_key(Path(s))) == g["order"]
    for r in g["resolve"]:
        cfg = types.SimpleNamespace(output=types.SimpleNamespace(output_format=r["fmt"]))
        out, disp, err = pl._resolve_output_path(Path("/in/ch1/p01.JPG"), Path("/in"), tmp_path, cfg, r["preserve"])
        assert str(out.relative_to(tmp_path)) == r["out"] and disp == r["display"] and err == r["error_key"]
    ks = [[list(scaling.scale_kernel((a, b), s)) for s in (None, 0.5, 1.0, 1.254, 2.5, 6.3)] for a, b in ((7, 7), (5, 5), (3, 9))]
    assert ks == g["scale_kernel"]


def test_page_sharding_partitions_every_page_once():
    pages = [f"ch{c}/{i:03d}.png" for c in (1, 2, 10) for i in range(1, 8)]
    shards = [pl.shard_pages(pages, r, 8) for r in range(8)]
    flat = [p for s in shards for p in s]
    assert sorted(flat) == sorted(pages) and len(set(flat)) == len(pages)
    assert shards[0][0] == "ch1/001.png" and max(len(s) for s in shards) - min(len(s) for s in shards) <= 1
    merged = pl.merge_batch_results([{"success_count": 2, "error_count": 1, "errors": {"a": "x"}, "failed_image_paths": ["p10.png"]},
                                     {"success_count": 3, "error_count": 0, "errors": {}, "failed_image_paths": ["p2.png"]}])
    assert merged["success_count"] == 5 and merged["failed_image_paths"] == ["p2.png", "p10.png"]


def test_gpu_numa_placement(tmp_path, monkeypatch):
    """core/device.py gpu_local_cpus on a made-up /sys: eight accelerators, four per NUMA node — every rank gets a quarter of its node's
    CPUs, nothing outside the process's affinity mask, and None when the tree is absent (containers)"""
    import os
    from mangatranslator_amd.core import device as dv
    root = tmp_path / "sys"
    for i in range(8):
        d = root / "bus" / "pci" / "devices" / f"0000:{0x05 + 0x10 * i:02x}:00.0"
        d.mkdir(parents=True)
        (d / "vendor").write_text("0x1002\n"); (d / "class").write_text("0x120000\n")
        (d / "local_cpulist").write_text("0-15,64-79\n" if i < 4 else "16-31,80-95\n")
    nic = root / "bus" / "pci" / "devices" / "0000:01:00.0"
    nic.mkdir(parents=True); (nic / "vendor").write_text("0x15b3\n"); (nic / "class").write_text("0x020000\n")
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(128)), raising=False)
    monkeypatch.setattr(dv.torch.cuda, "is_available", lambda: False)
    assert len(dv.gpu_pci_addresses(str(root))) == 8
    shares = [dv.gpu_local_cpus(i, 8, str(root)) for i in range(8)]
    assert shares[0] == list(range(0, 8)) and shares[1] == list(range(8, 16)) and shares[2] == list(range(64, 72)) and shares[3] == list(range(72, 80))
    assert shares[4] == list(range(16, 24)) and shares[7] == list(range(88, 96))
    assert not set(shares[0]) & set(shares[1]) and all(len(s) == 8 for s in shares)
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(0, 8)) | set(range(16, 20)), raising=False)
    assert dv.gpu_local_cpus(0, 8, str(root)) == [0, 1] and dv.gpu_local_cpus(5, 8, str(root)) == [17]
    assert dv.gpu_local_cpus(0, 8, str(tmp_path / "nowhere")) is None and dv.gpu_local_cpus(9, 8, str(root)) is None
    assert dv.pin_host_threads_to_gpu(0, 8, str(tmp_path / "nowhere"))["pinned"] is False
