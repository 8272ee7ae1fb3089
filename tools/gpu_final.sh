#!/bin/bash
# Round-end style validation: build check, GPU suite, smoke, default bench, rocprof stats of the bench.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
[ -x tools/stamp_check ] && timeout 20 tools/stamp_check > gpurun_out/stamp_check.log 2>&1 && tail -5 gpurun_out/stamp_check.log    # time_ops modes, plain C ABI
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu.log; tail -4 gpurun_out/pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.log 2>&1; tail -3 gpurun_out/smoke.log
timeout 900 python bench.py > gpurun_out/bench.log 2> gpurun_out/bench.err; tail -c 400 gpurun_out/bench.log; echo; tail -2 gpurun_out/bench.err
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof12" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$GRAFT_REPO_ROOT/gpurun_out/rocprof12.log" 2>&1)
for f in $(find gpurun_out/prof12 -name "*kernel_stats.csv"); do head -8 $f | cut -c1-150; done
python - <<'PY'
import json
for f in ("gpurun_out/bench.log", "gpurun_out/rocprof12.log"):
    try:
        d = json.loads([l for l in open(f) if l.startswith('{"metric"')][-1])
        r, g = d["roofline"], d.get("roofline_gemm", {})
        print(f, "pages/s %.4f" % d["value"], "attn %s %.4f ms (%s)" % (r.get("timing"), r["avg_launch_ms"], r.get("timing_detail")), "gemm %.0f TF/s" % g.get("achieved", 0))
    except Exception as e:
        print(f, "unreadable:", e)
PY
