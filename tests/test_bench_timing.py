"""bench.py's in-context timing chooser (host logic only): which of the two `mtx_plan_time_ops` figures lands in the JSON line."""
import importlib.util
import os
from pathlib import Path

import pytest

spec = importlib.util.spec_from_file_location("bench_mod", Path(__file__).resolve().parents[1] / "bench.py")
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)


class _Plan:
    def __init__(self, diff_ms, stamp_ms, stamp_fails=False):
        self.diff_ms, self.stamp_ms, self.stamp_fails, self.modes = diff_ms, stamp_ms, stamp_fails, []

    def time_ops(self, idx, iters=1):
        mode = os.environ.get("MTX_TIME_OPS", "difference")
        self.modes.append(mode)
        if mode == "stamp":
            if self.stamp_fails:
                raise RuntimeError("mtx_plan_time_ops: wall clock rate unavailable")
            return self.stamp_ms * len(idx) * iters
        return self.diff_ms * len(idx) * iters


@pytest.mark.parametrize("mode,diff,stamp,fails,want,how", [
    ("difference", 0.94, 0.80, False, 0.94, "difference"),
    ("auto", 0.94, 0.80, False, 0.80, "stamp"),
    ("auto", 0.80, 0.81, False, 0.81, "stamp"),
    ("auto", 0.80, 0.30, False, 0.80, "difference"),          # implausible stamps (wrong clock rate): keep the event figure
    ("auto", 0.80, 0.95, False, 0.95, "stamp"),
    ("auto", 0.80, 1.40, False, 0.80, "difference"),
    ("auto", 0.80, 0.0, True, 0.80, "difference"),
    ("stamp", 0.80, 0.30, False, 0.30, "stamp"),
])
def test_in_context_ms(mode, diff, stamp, fails, want, how):
    plan = _Plan(diff, stamp, fails)
    idx = list(range(57))
    total, used, info = bench.in_context_ms(plan, idx, 4, mode)
    assert used == how and total / len(idx) == pytest.approx(want)
    assert info["difference_ms_per_launch"] == pytest.approx(diff)
    assert ("stamp_error" in info) == (fails and mode != "difference")
    assert "MTX_TIME_OPS" not in os.environ                      # never leaks into later plan timings
    assert plan.modes[:2] == ["difference", "difference"]


def test_small_groups_take_their_stamps():
    """a group that is a few percent of the replay has no usable difference figure (two 160 ms replays differ by more than it lasts)"""
    plan = _Plan(0.67, 0.09, False)
    idx = list(range(19))
    total, used, _ = bench.in_context_ms(plan, idx, 4, "auto", replay_ms=160.0)          # 12.7 ms by difference, 1.7 ms stamped
    assert used == "stamp" and total / 19 == pytest.approx(0.09)
    total, used, _ = bench.in_context_ms(_Plan(0.80, 0.30, False), list(range(57)), 4, "auto", replay_ms=160.0)   # a big group keeps the band
    assert used == "difference"
