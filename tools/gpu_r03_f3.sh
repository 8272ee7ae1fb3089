#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
{
rm -rf /tmp/prof_c1; mkdir -p /tmp/prof_c1
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_c1 -o c1 -- python $GRAFT_REPO_ROOT/bench.py --config 1 --steps 20 --warmup 3 --no-cpu-baseline --no-overlap > /dev/null 2>&1)
KT=$(find /tmp/prof_c1 -name "*kernel_trace.csv" | head -1)
python tools/trace_overlap.py $KT 0.6 0.95
} > gpurun_out/r03_f3.log 2>&1
cat gpurun_out/r03_f3.log
