#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_ops_gpu.py -q --tb=short -p no:cacheprovider -k "variants or long_sequence" > gpurun_out/variants.log 2>&1; tail -4 gpurun_out/variants.log
timeout 900 python bench.py > gpurun_out/bench.log 2> gpurun_out/bench.err; tail -c 1500 gpurun_out/bench.log; tail -3 gpurun_out/bench.err
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof10" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$GRAFT_REPO_ROOT/gpurun_out/rocprof10.log" 2>&1)
for f in $(find gpurun_out/prof10 -name "*kernel_stats.csv"); do head -12 $f | cut -c1-150; done
tail -c 400 gpurun_out/rocprof10.log
