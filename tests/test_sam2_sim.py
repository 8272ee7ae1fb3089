"""CPU tier: the whole SAM-2.1 graph (pre-process, Hiera encoder, FPN neck, two-way mask decoder,
stability selection, fused upsample+threshold) on the kernel simulator vs HF transformers'
Sam2Model on CPU fp32 (the reference's own third-party implementation)."""
import numpy as np

import sam2_checks as sc
from mangatranslator_amd.core.ml.sam2 import window_order


def test_window_order_roundtrip():
    o = window_order(8, 8, 4)
    assert sorted(o.tolist()) == list(range(64))
    assert o[:16].tolist() == [0, 1, 2, 3, 8, 9, 10, 11, 16, 17, 18, 19, 24, 25, 26, 27]


def test_sam2_tiny(emu_lib):
    err, mism = sc.check_sam2(emu_lib, "cpu", "tiny_test", h=300, w=200, n_boxes=3, seed=0)
    assert err < 0.05


def test_sam2_tiny_one_box_landscape(emu_lib):
    sc.check_sam2(emu_lib, "cpu", "tiny_test", h=120, w=260, n_boxes=1, seed=3)


def test_sam2_tiny_f16_storage(emu_lib):
    """f16 storage (what `ModelManager.load_sam2` tries first): same graph, 3 more mantissa bits"""
    from mangatranslator_amd.hip import abi
    err, mism = sc.check_sam2(emu_lib, "cpu", "tiny_test", h=300, w=200, n_boxes=3, seed=0, dtype=abi.F16)
    assert err < 0.01


def test_sam2_tiny_high_precision(emu_lib):
    """precision "high": trunk / neck linears with hi + lo weight pairs (one GEMM over [x | x] x [W_hi | W_lo]), prompt encoder, two-way
    transformer and mask head as fp32 plans (csrc/f32ops.hip): same masks, and the logit error against the fp32 oracle drops by more
    than half on the same page and boxes (at Hiera-L depth the weights' rounding is a larger share of the total: DESIGN.md §3)"""
    from mangatranslator_amd.hip import abi
    sc.check_sam2(emu_lib, "cpu", "tiny_test", h=300, w=200, n_boxes=3, seed=0, dtype=abi.F16, calibrated=True)
    fast = sc.stats["logit_abs_err_rms"]
    sc.check_sam2(emu_lib, "cpu", "tiny_test", h=300, w=200, n_boxes=3, seed=0, dtype=abi.F16, calibrated=True, precision="high")
    high = sc.stats["logit_abs_err_rms"]
    assert high < 0.5 * fast, (fast, high)                     # logit units at std 5.9: rms 0.0117 -> 0.0041, page mask mismatch 6.8e-4 -> 2.7e-4
    assert sc.stats["decided_pixels_wrong"] == 0 and sc.stats["wrong_beyond_1_logit"] == 0


def test_sam2_small_high_precision(emu_lib):
    """the same at 512 x 512 input with head dim 24 and two global-attention blocks (dims that are no multiples of 64: other GEMM tile
    edges for the doubled K, other strides for the wide operands); logit rms 0.0212 -> 0.0065 at std 12.5"""
    from mangatranslator_amd.hip import abi
    sc.check_sam2(emu_lib, "cpu", "small_test", h=300, w=200, n_boxes=3, seed=0, dtype=abi.F16, calibrated=True)
    fast = sc.stats["logit_abs_err_rms"]
    sc.check_sam2(emu_lib, "cpu", "small_test", h=300, w=200, n_boxes=3, seed=0, dtype=abi.F16, calibrated=True, precision="high")
    assert sc.stats["logit_abs_err_rms"] < 0.5 * fast, (fast, sc.stats["logit_abs_err_rms"])
    assert sc.stats["decided_pixels_wrong"] == 0 and sc.stats["wrong_beyond_1_logit"] == 0


def test_sam2_hiera_large_dims_high_precision(emu_lib):
    """Hiera-L's own channel widths (144 / 288 / 576 / 1152), head dim 72, windows and decoder on a 256-pixel input with [1, 1, 2, 1] blocks:
    the GEMM shapes (K' = 288 ... 9216), operand strides and attention head size of the real model, at a size the simulator runs"""
    from mangatranslator_amd.hip import abi
    sc.check_sam2(emu_lib, "cpu", "large_dims_test", h=300, w=200, n_boxes=3, seed=0, dtype=abi.F16, calibrated=True)
    fast = sc.stats["logit_abs_err_rms"]
    sc.check_sam2(emu_lib, "cpu", "large_dims_test", h=300, w=200, n_boxes=3, seed=0, dtype=abi.F16, calibrated=True, precision="high")
    assert sc.stats["logit_abs_err_rms"] < 0.5 * fast, (fast, sc.stats["logit_abs_err_rms"])
    assert sc.stats["decided_pixels_wrong"] == 0 and sc.stats["wrong_beyond_1_logit"] == 0


def test_manager_loads_sam_in_f16_and_falls_back_to_bf16(emu_lib, tmp_path, monkeypatch):
    """reference model_manager.py:982-1010: (processor, model).  The loader takes f16 storage when the f16 and bf16 models agree on the
    load-time probe, bf16 otherwise (a checkpoint with an activation beyond 65504 saturates in f16 and must not be served that way)."""
    import torch
    from safetensors.torch import save_file
    import mangatranslator_amd.hip.lib as libmod
    from mangatranslator_amd.core.ml import model_manager as mm
    from mangatranslator_amd.hip import abi
    from oracle import sam2_ref as sr
    monkeypatch.setattr(libmod, "_lib", emu_lib)
    monkeypatch.setattr(mm, "_model_manager", None)
    monkeypatch.setattr(mm.ModelManager, "_instance", None)
    from mangatranslator_amd.core.ml.sam2 import Sam2Hip
    probe = Sam2Hip.probe_logits
    monkeypatch.setattr(Sam2Hip, "probe_logits", lambda self, size=96: probe(self, size))      # a smaller probe page: the simulator's time
    m = mm.get_model_manager()
    try:
        root = tmp_path / "sam"
        m.model_paths[mm.ModelType.SAM2] = root
        model, cfg = sr.make_model("tiny_test", 0)
        cfg.save_pretrained(str(root))
        sd = {k: v.contiguous() for k, v in model.state_dict().items()}
        save_file(sd, str(root / "model.safetensors"))
        assert m.sam_precision == "high" and m.sam_storage == "auto"          # the defaults (round 5: "high" measured on hardware)
        proc, shim = m.load_sam2()                                  # hi + lo trunk weights, fp32 mask decoder, on the storage type the probe settled on
        assert shim.hip.dtype == abi.F16 and shim.hip.high and shim.hip.ddtype == abi.F32
        with m.front_replica(1):
            assert m.load_sam2()[1].hip.high                         # a replica follows set 0
        m.sam_precision = "fast"                                    # changed after the load: the next call builds the other arithmetic
        proc, shim = m.load_sam2()
        assert shim.hip.dtype == abi.F16 and not shim.hip.high
        assert m.load_sam2()[1] is shim
        m.unload_model(mm.ModelType.SAM2)
        m.sam_storage = "bf16"                                      # the reference's own GPU dtype, no probe
        assert m.load_sam2()[1].hip.dtype == abi.BF16
        m.unload_model(mm.ModelType.SAM2)
        m.sam_storage = "auto"
        # blow up one MLP of the trunk: its hidden activations leave the f16 range
        key = next(k for k in sd if "backbone" in k and "mlp" in k and k.endswith("proj_in.weight"))
        sd[key] = sd[key] * 3.0e5
        save_file(sd, str(root / "model.safetensors"))
        proc, shim = m.load_sam2()
        assert shim.hip.dtype == abi.BF16
    finally:
        monkeypatch.setattr(mm.ModelManager, "_instance", None)
