#!/bin/bash
# One GPU-box visit: parity tests, bench, rocprof kernel trace, micro-benchmarks.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -15 gpurun_out/pytest_gpu.log
timeout 900 python bench.py --steps 6 --warmup 2 > gpurun_out/bench.log 2> gpurun_out/bench.err
tail -3 gpurun_out/bench.log; tail -5 gpurun_out/bench.err
timeout 900 python tools/gpu_probe.py > gpurun_out/probe.log 2>&1
tail -30 gpurun_out/probe.log
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > "$GRAFT_REPO_ROOT/gpurun_out/rocprof.log" 2>&1)
find gpurun_out/prof -name "*stats*" | head; for f in $(find gpurun_out/prof -name "*kernel_stats.csv"); do head -25 $f; done 
