#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python tools/bench_kernels.py ab 8704 3072 12288 ab 8704 3072 15360 ab 8652 3072 3072 ab 512 3072 12288 2>&1 | tail -12
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu.log; tail -6 gpurun_out/pytest_gpu.log
timeout 900 python bench.py > gpurun_out/bench.log 2> gpurun_out/bench.err; tail -c 2600 gpurun_out/bench.log | head -c 1700; echo; tail -3 gpurun_out/bench.err
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof11" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$GRAFT_REPO_ROOT/gpurun_out/rocprof11.log" 2>&1)
for f in $(find gpurun_out/prof11 -name "*kernel_stats.csv"); do head -14 $f | cut -c1-150; done
