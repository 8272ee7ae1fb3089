"""Fixtures written by tools/pin_oracles.py on a machine that has ultralytics / spandrel / diffusers / OpenCV.

* `test_committed_fixture[...]`: every tests/golden/pinned_*.npz that exists is replayed against the oracle — that is the pin of the
  restated arithmetic to the library the reference really calls.  A missing fixture SKIPS with the reason (the wheel is not installable
  in the build image; the oracle stays "parity unpinned") instead of passing silently.
* the kit itself is exercised here with stand-in wheels: independent OpenCV implementations (tests/independent_cv.py) and oracle-backed
  model classes (tests/stub_wheels.py) — generator, fixture files, report and replay, and a broken library must be flagged."""
import json
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))
import independent_cv as ic
import pin_oracles as po
import stub_wheels as sw


@pytest.mark.parametrize("name", sorted(po.REPLAY))
def test_committed_fixture(name):
    path = po.GOLDEN / name
    if not path.exists():
        pytest.skip(f"{name} not committed: run `python tools/pin_oracles.py` where the wheel exists (parity of that oracle stays unpinned)")
    rep = po.REPLAY[name](path)
    assert rep, "fixture holds no outputs"
    bad = {k: v for k, v in rep.items() if not v["ok"]}
    assert not bad, bad


# the float formulas behind the stand-in's colour conversions are within a code or two of OpenCV's fixed-point paths (test_oracle_crosschecks.py)
STUB_TOL = {"gray_": 1.0, "hsv_s_": 1.0, "lab_": 2.0, "lab_back": 2.0}


def test_kit_with_a_stand_in_opencv(tmp_path):
    r = po.pin_cv2(ic.stub_cv2(), tmp_path)
    assert (tmp_path / "pinned_cv2.npz").exists()
    exact = {k: v for k, v in r["cases"].items() if not k.startswith(tuple(STUB_TOL))}
    assert len(exact) >= 23 and all(v["ok"] and v["max_abs_diff"] == 0 for v in exact.values()), {k: v for k, v in exact.items() if not v["ok"]}
    rep = po.replay_cv2(tmp_path / "pinned_cv2.npz", STUB_TOL)
    assert len(rep) == len(r["cases"]) and all(v["ok"] for v in rep.values()), {k: v for k, v in rep.items() if not v["ok"]}


def test_kit_flags_a_library_that_disagrees(tmp_path):
    r = po.pin_cv2(ic.stub_cv2(broken=True), tmp_path)
    assert not r["cases"]["dilate7_blobs"]["ok"] and r["cases"]["dilate7_blobs"]["differing"] > 0
    assert r["cases"]["erode5_blobs"]["ok"]
    rep = po.replay_cv2(tmp_path / "pinned_cv2.npz", STUB_TOL)
    assert not rep["dilate7_blobs"]["ok"]


def test_kit_network_plumbing(tmp_path):
    """state-dict hand-over by parameter name, nested library outputs, fixture files, replay — on oracle-backed stand-ins (circular on purpose)"""
    r = po.pin_ultralytics(sw.ultralytics(), tmp_path)
    assert all(c["ok"] for c in r["cases"].values()), r["cases"]
    z = np.load(tmp_path / "pinned_ultralytics.npz")
    assert {"yolov8n_seg__out0", "yolov8n_seg__out1", "yolo11n__out0", "yolo12n__out0"} <= set(z.files)
    assert all(v["ok"] for v in po.replay_ultralytics(tmp_path / "pinned_ultralytics.npz").values())
    r = po.pin_spandrel(sw.spandrel(), tmp_path)
    assert all(c["ok"] for c in r["cases"].values()), r["cases"]
    assert all(v["ok"] for v in po.replay_spandrel(tmp_path / "pinned_spandrel.npz").values())
    r = po.pin_diffusers(sw.diffusers(), tmp_path)
    ok = {k: c for k, c in r["cases"].items() if k.startswith("flux1")}
    assert ok and all(c["ok"] for c in ok.values()), r["cases"]
    rep = po.replay_diffusers(tmp_path / "pinned_diffusers.npz")
    assert set(rep) == {"flux1_transformer__out0", "flux1_vae__enc", "flux1_vae__dec"} and all(v["ok"] for v in rep.values())


def test_kit_reports_absent_wheels(tmp_path):
    """in this image every target is absent: the command line says so, writes the report and exits 0"""
    assert po.main(["--out", str(tmp_path)]) == 0
    rep = json.loads((tmp_path / "pinned_report.json").read_text())
    for t in po.TARGETS:
        assert rep[t]["status"] in ("wheel absent", "pinned")
