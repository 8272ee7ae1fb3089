"""Conjoined-bubble mask partition vs golden vectors produced by the reference
(tests/golden/make_goldens.py gen_conjoined): bit-exact masks, boxes and metadata."""
import json
import types
from pathlib import Path

import numpy as np
import pytest
import torch

from mangatranslator_amd.core.image import conjoined as cj

G = Path(__file__).resolve().parent / "golden"
META = json.loads((G / "conjoined.json").read_text())
ARR = np.load(G / "conjoined.npz")
H, W = META["H"], META["W"]


def _unpack(name, n=None):
    bits = np.unpackbits(ARR[name])
    return bits[:H * W].reshape(H, W).astype(bool) if n is None else bits[:n * H * W].reshape(n, H, W).astype(bool)


@pytest.mark.parametrize("k", range(len(META["scenarios"])))
def test_split_conjoined_mask(emu_lib, k):
    sc = META["scenarios"][k]
    parent = _unpack(f"parent_{k}")
    boxes = [torch.tensor(b, dtype=torch.float32) for b in sc["boxes"]]
    texts = np.asarray(sc["texts"], np.float32) if sc["texts"] is not None else None
    assert cj.detect_group_arrangement(boxes) == sc["arrangement"]
    grp = cj.group_text_boxes(texts, torch.tensor([0.0, 0.0, float(W), float(H)])) if texts is not None else None
    if sc["group_texts"] is None:
        assert grp is None
    else:
        assert np.asarray(grp).tolist() == sc["group_texts"]
        got = {str(i): np.asarray(v).tolist() for i, v in cj.match_text_boxes_to_bubbles(grp, boxes).items()}
        assert got == sc["match"]
    for b, r in zip(boxes, sc["rects"]):
        m = cj.build_rect_mask_from_box(b, H, W)
        assert [int(v) for v in np.nonzero(m.any(0))[0][[0, -1]]] + [int(v) for v in np.nonzero(m.any(1))[0][[0, -1]]] == r
    out = cj.split_conjoined_mask(parent.astype(np.uint8) * 255, boxes, osb_text_boxes=grp, lib=emu_lib)
    assert len(out) == sc["n_out"]
    want = _unpack(f"out_{k}", sc["n_out"])
    for i, o in enumerate(out):
        assert o.dtype == np.uint8 and set(np.unique(o)) <= {0, 255}
        assert np.array_equal(o > 0, want[i]), f"scenario {k} child {i}: {((o > 0) != want[i]).sum()} px differ"
    if parent.any() and len(boxes) > 1:       # a partition: children are disjoint and cover the parent
        stack = np.stack([o > 0 for o in out])
        assert stack.sum(0).max() <= 1 and np.array_equal(stack.any(0), parent)


def test_build_segmentation_detections(emu_lib):
    a = META["assembly"]
    primary, secondary = torch.tensor(a["primary"], dtype=torch.float32), torch.tensor(a["secondary"], dtype=torch.float32)
    ns = types.SimpleNamespace
    pres, sres = ns(masks=None), ns(names={0: "bubble"})
    pres.boxes = type("B", (), {"conf": torch.tensor([0.9, 0.8, 0.7, 0.65, 0.6]), "cls": torch.zeros(5), "__len__": lambda self: 5})()
    sres.boxes = type("B", (), {"conf": torch.tensor([0.55, 0.45]), "cls": torch.zeros(2), "__len__": lambda self: 2})()
    sam = [_unpack("assembly_sam0").astype(np.uint8) * 255, _unpack("assembly_sam1").astype(np.uint8) * 255, None, None, None]
    synth = [dict(parent_mask=_unpack("assembly_synth").astype(np.uint8) * 255, parent_box=[10.0, 98.0, 95.0, 119.0], member_indices=[3, 4])]
    dets = cj.build_segmentation_detections(primary, primary, [("primary", i) for i in range(5)], pres, ns(names={0: "speech_bubble"}), secondary,
                                            [("secondary", 0), ("secondary", 1)], sres, [1, 2], [(0, [0, 1])], H, W, 0.35,
                                            sam_masks=sam, synthetic_conjoined_groups=synth, lib=emu_lib)
    assert len(dets) == len(a["dets"])
    want_masks = _unpack("assembly_masks", len(dets))
    for d, w, wm in zip(dets, a["dets"], want_masks):
        assert list(d["bbox"]) == w["bbox"] and d["class"] == w["cls"] and abs(d["confidence"] - w["confidence"]) < 1e-7
        assert ([list(b) for b in d["conjoined_neighbor_bboxes"]] if "conjoined_neighbor_bboxes" in d else None) == w["neighbors"]
        assert np.array_equal(np.asarray(d["sam_mask"]) > 0, wm)


def test_chamfer_native_matches_oracle(emu_lib):
    from oracle.cleaning_ref import distance_transform_l2_5x5
    rng = np.random.default_rng(0)
    img = (rng.random((37, 53)) < 0.97).astype(np.uint8)
    img[:, 0] = 1
    assert np.array_equal(cj.chamfer_distance(img, emu_lib), distance_transform_l2_5x5(img))
