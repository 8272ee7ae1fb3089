"""CPU tier: the product-side parameter inventories (mangatranslator_amd/utils/synthetic_checkpoints.py — what bench.py seeds its
stand-in checkpoints from) name and shape every parameter exactly as the CPU oracles' modules do, for every scale either side knows."""
import pytest
import torch

from mangatranslator_amd.utils import synthetic_checkpoints as sc


def _shapes(module):
    return {k: tuple(v.shape) for k, v in module.state_dict().items() if v.is_floating_point()}


@pytest.mark.parametrize("scale", ["n", "s", "m", "l", "x"])
@pytest.mark.parametrize("nc", [1, 3])
def test_yolov8_seg_inventory(scale, nc):
    from oracle import yolo_ref as yr
    want = _shapes(yr.YoloV8Seg(yr.arch(scale, nc)))
    got, head = sc.yolov8_seg_shapes(scale, nc)
    assert head == 22 and dict(got) == want


@pytest.mark.parametrize("family,scale,seg", [("11", "n", False), ("11", "s", True), ("11", "m", True), ("11", "l", False), ("11", "x", False),
                                              ("12", "n", False), ("12", "s", False), ("12", "m", False), ("12", "l", False), ("12", "x", False)])
def test_yolo11_family_inventory(family, scale, seg):
    from oracle import yolo11_ref as y11
    with torch.device("meta"):
        want = _shapes(y11.Yolo11(y11.arch(family, scale, 2, seg)))
    got, head = sc.yolo11_shapes(family, scale, 2, seg)
    assert head == (23 if family == "11" else 21) and dict(got) == want


def test_seeded_detector_fills_every_entry_and_tames_the_head():
    shapes, head = sc.yolo11_shapes("12", "n", 1, False)
    sd = sc.seeded_detector(shapes, head, seed=5, class_bias=-2.0, class_gain=0.05)
    assert list(sd) == list(shapes) and all(tuple(sd[k].shape) == tuple(shapes[k]) for k in shapes)
    assert all(torch.isfinite(v).all() for v in sd.values())
    assert float(sd[f"model.{head}.cv3.0.2.bias"][0]) == -2.0
    again = sc.seeded_detector(shapes, head, seed=5, class_bias=-2.0, class_gain=0.05)
    assert all(torch.equal(sd[k], again[k]) for k in sd)                    # same seed, same checkpoint (rank 0 / other ranks agree)


def test_rcan_inventory():
    from oracle import rcan_ref
    for kw in (dict(n_feats=64, n_resgroups=10, n_resblocks=20, unshuffle=1), dict(n_feats=48, n_resgroups=3, n_resblocks=5, reduction=8, unshuffle=2)):
        want = {k: tuple(v.shape) for k, v in rcan_ref.make_state_dict(seed=1, **kw).items()}
        got = {k: tuple(v.shape) for k, v in sc.rcan_state_dict(seed=1, **kw).items()}
        assert got == want


def test_hf_configurations_match_the_oracles():
    from oracle import rtdetr_ref, sam2_ref
    assert sc.sam2_hiera_large_config().to_dict() == sam2_ref.make_config("hiera_large").to_dict()
    assert sc.rtdetr_r50_config().to_dict() == rtdetr_ref.make_config("r50").to_dict()
    shapes = sc.sam2_shapes(sc.sam2_hiera_large_config())
    assert len(shapes) > 500 and shapes["vision_encoder.backbone.pos_embed"][1] == 144
