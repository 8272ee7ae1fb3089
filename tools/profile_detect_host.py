"""cProfile of the detect stage's host side: `bench.py --config 1 --no-overlap` (four detectors + clean, strictly in order) — what the
page loop spends outside the kernels (NMS, result objects, small torch ops, synchronising copies).  Model set-up is left out: the
profiler is switched on by the first page of the warm-up."""
import cProfile, pstats, sys, os, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["bench.py", "--config", "1", "--no-overlap", "--steps", "40", "--warmup", "3", "--no-cpu-baseline"] + sys.argv[1:]
import torch
import bench
print("host: cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "torch threads", torch.get_num_threads())
pr = cProfile.Profile()
import mangatranslator_amd.core.ml.yolo as y
orig = y.YoloSegHip.__call__
state = {"calls": 0}
def call(self, *a, **k):
    state["calls"] += 1
    if state["calls"] == 12:          # past the pool set-up pages: plans and graphs exist
        pr.enable()
    return orig(self, *a, **k)
y.YoloSegHip.__call__ = call
bench.main()
pr.disable()
for key in ("tottime", "cumulative"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(28)
    print(s.getvalue()[:6500])
