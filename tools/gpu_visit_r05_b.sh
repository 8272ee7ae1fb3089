#!/bin/bash
# round 5, visit b: the alternative attention schedules on hardware — parity first, then same-process A/B at the FLUX shape
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
{
  echo "== parity of the alternative schedules (tests/test_ops_gpu.py)"
  timeout 600 python -m pytest tests/test_ops_gpu.py -q -x -p no:cacheprovider -k "alternative_schedules or attention_prescaled" 2>&1 | tail -6
  echo "== A/B, T = 8812, 24 heads: default kernel (0) and attn_x schedules, 3 rounds in one process"
  timeout 600 python tools/bench_kernels.py attnx 8812 0,1,2,3,4,5,6,8 3 2>&1 | grep -v "^$" | tail -30
} > gpurun_out/r05_visit_b.log 2>&1
cat gpurun_out/r05_visit_b.log
