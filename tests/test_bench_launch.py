"""`python bench.py --gpus N` launches N ranks itself (it used to run ONE rank and print n_gpus: 1)."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def _run(args, env=None, timeout=600):
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None); e.pop("RANK", None); e.pop("LOCAL_RANK", None)
    e.update(env or {})
    r = subprocess.run([sys.executable, str(ROOT / "bench.py")] + args, capture_output=True, text=True, timeout=timeout, cwd=str(ROOT), env=e)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if not (r.returncode == 0 and lines):        # the ranks' own tracebacks come first; torchrun's summary of them fills the tail
        at = r.stderr.find("Traceback")
        raise AssertionError(r.stdout[-1500:] + (r.stderr[at:at + 4000] if at >= 0 else r.stderr[-4000:]))
    return json.loads(lines[-1])


def test_self_launch_brings_up_every_rank():
    """CPU tier: no launcher around it, `--gpus 2` re-runs itself under torch.distributed.run; the collective library sees two ranks"""
    rep = _run(["--gpus", "2", "--backend", "gloo", "--launch-check"])
    assert rep["world_size"] == 2 and rep["ranks"] == [0, 1] and rep["n_gpus"] == 2 and rep["self_launched"] and rep["broadcast_ok"]


def test_config_presets():
    sys.path.insert(0, str(ROOT))
    import importlib
    bench = importlib.import_module("bench")
    old = sys.argv
    try:
        want = {1: ("detect,clean", 1024, 1536, "kontext", 20), 2: ("detect,segment", 1024, 1536, "kontext", 20),
                4: ("detect,segment,inpaint,upscale", 1024, 1536, "kontext", 20), 5: ("detect,segment,inpaint,upscale", 2048, 3072, "klein_4b", 8)}
        for c, (st, w, h, inp, steps) in want.items():
            sys.argv = ["bench.py", "--config", str(c)]
            a = bench.parse()
            assert (a.stages, a.width, a.height, a.inpainter, a.inpaint_steps) == (st, w, h, inp, steps)
        sys.argv = ["bench.py"]
        a = bench.parse()
        assert a.config == 4 and a.gpus == 1                     # the default line is the headline metric on one GPU
    finally:
        sys.argv = old


@pytest.mark.gpu
def test_two_ranks_on_one_device():
    """GPU tier (1-GPU box): both ranks share device 0 (MTX_BENCH_ONE_DEVICE), weights travel rank 0 -> rank 1, the line says n_gpus 2"""
    rep = _run(["--gpus", "2", "--backend", "gloo", "--stages", "upscale", "--upscale-model", "model_lite", "--width", "256", "--height", "384",
                "--steps", "1", "--warmup", "0", "--no-cpu-baseline"], env={"MTX_BENCH_ONE_DEVICE": "1"}, timeout=900)
    assert rep["n_gpus"] == 2 and rep["config"]["launch"]["world_size_seen_by_collectives"] == 2 and rep["config"]["launch"]["self_launched"]
    assert rep["value"] > 0 and rep["scaling"] == "weak"


def test_hardware_queue_choice(monkeypatch):
    """bench.py asks ROCm for sixteen hardware queues only for the stage sets without diffusion / upscaling (measured: helps the detect
    stage's five model streams, costs config 5 — DESIGN.md §6), before torch is imported, and never overrides the caller's value"""
    sys.path.insert(0, str(ROOT))
    import importlib
    bench = importlib.import_module("bench")
    for argv, want in ((["--config", "2"], "16"), (["--config=1", "--steps", "3"], "16"), (["--stages", "detect"], "16"), ([], None), (["--config", "5"], None),
                       (["--config", "3"], None), (["--stages", "upscale"], None), (["--gpus", "8", "--steps", "5", "--warmup", "2"], None)):
        monkeypatch.delenv("GPU_MAX_HW_QUEUES", raising=False)
        bench._early_hw_queues(argv)
        assert os.environ.get("GPU_MAX_HW_QUEUES") == want, (argv, os.environ.get("GPU_MAX_HW_QUEUES"))
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "2")
    bench._early_hw_queues(["--config", "2"])
    assert os.environ["GPU_MAX_HW_QUEUES"] == "2"
