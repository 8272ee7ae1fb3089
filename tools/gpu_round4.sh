#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu.log; tail -6 gpurun_out/pytest_gpu.log
timeout 900 python bench.py > gpurun_out/bench.log 2> gpurun_out/bench.err; tail -c 2500 gpurun_out/bench.log | head -c 1500; echo; tail -3 gpurun_out/bench.err
