#!/bin/bash
# round 5, visit l: prefetch depths of the default attention kernel (K k-steps | V MFMAs ahead): 0 = 4 | 4, 70 = 3 | 4, 71 = 6 | 4, 72 = 4 | 6, 73 = 4 | 8, 74 = 4 | 2
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
{
  echo "== parity (schedules 70..74)"
  timeout 600 python -m pytest tests/test_ops_gpu.py -q -x -p no:cacheprovider -k "alternative_schedules and (70 or 71 or 72 or 73 or 74)" 2>&1 | tail -3
  echo "== A/B, T = 8812, four rounds"
  timeout 600 python tools/bench_kernels.py attnx 8812 0,70,71,72,73,74 4 2>&1 | grep "best of\|schedule=0"
} > gpurun_out/r05_visit_l.log 2>&1
cat gpurun_out/r05_visit_l.log
