#!/bin/bash
# round 5, visit j: the tuned attention kernel with fragment reads four steps ahead (schedule 68) against the default
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
{
  echo "== parity (schedule 68)"
  timeout 600 python -m pytest tests/test_ops_gpu.py -q -x -p no:cacheprovider -k "alternative_schedules and 68" 2>&1 | tail -3
  echo "== A/B, T = 8812, four rounds: 0 default (16-byte stores) | 68 = + fragment reads four steps ahead | 67 round-4 epilogue | 25 attn_x staged + deep"
  timeout 600 python tools/bench_kernels.py attnx 8812 0,68,67,25 4 2>&1 | grep -v "^$" | tail -18
  timeout 600 python tools/bench_kernels.py attnx 13312 0,68 2 attnx 4096 0,68 2 2>&1 | grep "best of"
} > gpurun_out/r05_visit_j.log 2>&1
cat gpurun_out/r05_visit_j.log
