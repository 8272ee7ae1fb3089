"""CPU tier: the pickle-free reader of ultralytics `.pt` checkpoints (core/ml/ultralytics_pt.py) and the loaders on top of it.

The test WRITES such a file the way ultralytics does — `torch.save({"model": <nn.Module of ultralytics.nn.* classes>.half(), "ema": ...,
"train_args": ...})` — with stand-in classes registered under the ultralytics module names only while the file is pickled; afterwards
those modules are gone from `sys.modules`, so `torch.load` could not read the file here, and the reader must (reference
core/ml/model_manager.py:711-743 does `YOLO(str(path))`)."""
import pickle
import sys
import types
from pathlib import Path

import numpy as np
import pytest
import torch
from torch import nn

from mangatranslator_amd.core.ml.ultralytics_pt import read_ultralytics_pt
from mangatranslator_amd.utils.exceptions import ModelError

STUB_MODULES = ("ultralytics", "ultralytics.nn", "ultralytics.nn.tasks", "ultralytics.nn.modules", "ultralytics.nn.modules.conv",
                "ultralytics.nn.modules.block", "ultralytics.nn.modules.head")


class _Stubs:
    """registers stand-in ultralytics classes for the duration of a `with` block"""

    def __enter__(self):
        self.mods = {}
        for name in STUB_MODULES:
            self.mods[name] = sys.modules[name] = types.ModuleType(name)

        def cls(module, name):
            c = type(name, (nn.Module,), {"__module__": module, "forward": lambda self, x: x})
            setattr(self.mods[module], name, c)
            return c

        self.Model = cls("ultralytics.nn.tasks", "SegmentationModel")
        self.Block = cls("ultralytics.nn.modules.block", "C2f")
        self.Conv = cls("ultralytics.nn.modules.conv", "Conv")
        self.Head = cls("ultralytics.nn.modules.head", "Segment")
        return self

    def __exit__(self, *exc):
        for name in STUB_MODULES:
            sys.modules.pop(name, None)

    def tree(self, sd, names):
        """nn.Module tree whose state_dict() is `sd` (keys like model.0.conv.weight), running statistics as buffers"""
        root = self.Model()
        for key, t in sd.items():
            parts = key.split(".")
            mod = root
            for i, part in enumerate(parts[:-1]):
                if part not in mod._modules:
                    kind = nn.Sequential if (i == 0 and part == "model") else (nn.BatchNorm2d if part == "bn" else (self.Conv if part in ("conv", "cv1", "cv2") else self.Block))
                    child = nn.BatchNorm2d(1) if kind is nn.BatchNorm2d else kind()
                    if kind is nn.BatchNorm2d:
                        child._parameters.clear(); child._buffers.clear()
                    mod.add_module(part, child)
                mod = mod._modules[part]
            leaf = parts[-1]
            if leaf in ("running_mean", "running_var", "num_batches_tracked"):
                mod.register_buffer(leaf, t.clone())
            else:
                mod.register_parameter(leaf, nn.Parameter(t.clone(), requires_grad=False))
        root.names = names
        root.task = "segment"
        root.yaml = {"nc": len(names), "scale": "n"}
        root.args = {"imgsz": 640}
        root.register_buffer("scratch_not_saved", torch.zeros(3), persistent=False)
        return root


def _write_pt(path, sd, names, ema=False):
    with _Stubs() as st:
        model = st.tree(sd, names).half()
        ckpt = {"epoch": -1, "best_fitness": None, "model": None if ema else model, "ema": model if ema else None, "updates": None, "optimizer": None,
                "train_args": {"data": Path("/data/x.yaml"), "imgsz": 640}, "train_metrics": {"fitness": np.float64(0.5)}, "date": "2026-01-01", "version": "8.3.0"}
        if ema:
            ckpt["model"] = st.tree({k: v * 0 for k, v in sd.items()}, names).half()     # ultralytics prefers the EMA weights
        torch.save(ckpt, str(path))
        ref = {k: v.float() for k, v in model.state_dict().items() if v.dtype.is_floating_point}
    return ref


def _yolo_sd(seed=1):
    from oracle import yolo_ref as yr
    net = yr.make_model("n", 1, seed=seed)
    return {k: v.detach().clone() for k, v in net.state_dict().items()}


def test_reader_returns_the_state_dict_and_names(tmp_path):
    sd = _yolo_sd()
    ref = _write_pt(tmp_path / "best.pt", sd, {0: "speech_bubble"})
    assert "ultralytics" not in sys.modules
    with pytest.raises(Exception):
        torch.load(str(tmp_path / "best.pt"), weights_only=False)          # what the reference's loader would need ultralytics for
    got, meta = read_ultralytics_pt(tmp_path / "best.pt")
    assert list(got) == list(ref) and meta["names"] == repr({0: "speech_bubble"}) and meta["task"] == "segment" and meta["source"] == "model"
    for k in ref:
        assert got[k].dtype == torch.float32 and torch.equal(got[k], ref[k]), k          # half on disk, fp32 here: the same values
    assert "scratch_not_saved" not in got                                               # non-persistent buffers stay out, like state_dict()
    assert not any(k.endswith("num_batches_tracked") for k in got)                      # integer buffers are dropped (the exporter does too)


def test_reader_prefers_ema_and_reads_strided_views(tmp_path):
    sd = {"model.0.conv.weight": torch.arange(24, dtype=torch.float32).reshape(2, 3, 2, 2).transpose(0, 1).contiguous().transpose(0, 1),   # non-trivial strides
          "model.0.bn.weight": torch.ones(2), "model.0.bn.running_mean": torch.tensor([0.5, -1.0])}
    ref = _write_pt(tmp_path / "ema.pt", sd, ["a", "b"], ema=True)
    got, meta = read_ultralytics_pt(tmp_path / "ema.pt")
    assert meta["source"] == "ema" and meta["names"] == repr({0: "a", 1: "b"})
    assert all(torch.equal(got[k], ref[k]) for k in ref) and got["model.0.conv.weight"].abs().sum() > 0


def test_reader_runs_no_code_from_the_file(tmp_path):
    """a pickle whose REDUCE names os.system / builtins.eval must come back as inert shells: nothing is imported or called"""
    class Evil:
        def __reduce__(self):
            import os
            return (os.system, ("touch " + str(tmp_path / "pwned"),))

    class Evil2:
        def __reduce__(self):
            return (eval, ("__import__('os').system('touch %s')" % (tmp_path / "pwned2"),))

    sd = _yolo_sd()
    with _Stubs() as st:
        model = st.tree(sd, {0: "x"})
        model.extra = [Evil(), Evil2()]
        torch.save({"model": model, "hook": Evil()}, str(tmp_path / "evil.pt"), pickle_protocol=2)
    got, _ = read_ultralytics_pt(tmp_path / "evil.pt")
    assert len(got) > 10 and not (tmp_path / "pwned").exists() and not (tmp_path / "pwned2").exists()


def test_reader_errors_are_model_errors(tmp_path):
    (tmp_path / "junk.pt").write_bytes(b"not a zip")
    with pytest.raises(ModelError):
        read_ultralytics_pt(tmp_path / "junk.pt")
    torch.save({"model": None, "ema": None}, str(tmp_path / "empty.pt"))
    with pytest.raises(ModelError):
        read_ultralytics_pt(tmp_path / "empty.pt")
    # a truncated archive
    sd = _yolo_sd()
    _write_pt(tmp_path / "ok.pt", sd, {0: "x"})
    raw = (tmp_path / "ok.pt").read_bytes()
    (tmp_path / "cut.pt").write_bytes(raw[: len(raw) // 2])
    with pytest.raises(ModelError):
        read_ultralytics_pt(tmp_path / "cut.pt")


def test_manager_loads_the_references_pt_files(emu_lib, tmp_path, monkeypatch):
    """`load_yolo_speech_bubble(path)` / `load_yolo_panel()` on `.pt` files under the reference's own file names: no export step"""
    import mangatranslator_amd.hip.lib as libmod
    from mangatranslator_amd.core.ml import model_manager as mm
    from mangatranslator_amd.core.ml.yolo import YoloSegHip
    from mangatranslator_amd.core.ml.yolo11 import Yolo11Hip
    from oracle import yolo11_ref as y11
    monkeypatch.setattr(libmod, "_lib", emu_lib)
    monkeypatch.setattr(mm, "_model_manager", None)
    monkeypatch.setattr(mm.ModelManager, "_instance", None)
    m = mm.get_model_manager()
    try:
        for k in list(m.model_paths):
            m.model_paths[k] = tmp_path / m.model_paths[k].relative_to(m.model_paths[k].parents[1])
        first = m.model_paths[mm.ModelType.YOLO_SPEECH_BUBBLE]
        assert first.name == "yolov8m_seg-speech-bubble.pt" and m.model_paths[mm.ModelType.YOLO_PANEL].name == "manga109_v2023.12.07_l_yolov11.pt"
        first.parent.mkdir(parents=True)
        _write_pt(first, _yolo_sd(seed=2), {0: "speech_bubble"})
        net = y11.make_model("11", "n", 2, False, seed=1)
        _write_pt(m.model_paths[mm.ModelType.YOLO_PANEL], {k: v.detach().clone() for k, v in net.state_dict().items()}, {0: "body", 1: "frame"})
        bubble = m.load_yolo_speech_bubble()                       # None = the first model, as in the reference
        assert isinstance(bubble, YoloSegHip) and bubble.names == {0: "speech_bubble"}
        panel = m.load_yolo_panel()
        assert isinstance(panel, Yolo11Hip) and panel.names == {0: "body", 1: "frame"}
        page = (np.random.default_rng(0).random((96, 64, 3)) * 255).astype(np.uint8)
        res = bubble(page, conf=0.0, imgsz=64, max_det=3)[0]
        assert len(res.masks) == 3
        # a custom checkpoint path goes to the first slot, like `YOLO(str(path))` in the reference
        m.unload_model(mm.ModelType.YOLO_SPEECH_BUBBLE)
        custom = tmp_path / "custom" / "best.pt"
        custom.parent.mkdir()
        _write_pt(custom, _yolo_sd(seed=3), {0: "bubble"})
        assert m.load_yolo_speech_bubble(str(custom)).names == {0: "bubble"}
    finally:
        monkeypatch.setattr(mm.ModelManager, "_instance", None)
