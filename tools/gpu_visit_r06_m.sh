#!/bin/bash
# round 6, visit M: the persistent 256-tile GEMM (next tile's first stage requested in front of the epilogue) against the plain launch, same process
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
{
  echo "== tests"
  timeout 900 python -m pytest tests/test_ops_gpu.py -q -x -p no:cacheprovider -k "persistent" 2>&1 | tail -5
  echo "== A/B (FLUX.1 Kontext shapes)"
  timeout 900 python tools/bench_kernels.py gemmp 8812 9216 3072 gemmpg 8300 12288 3072 gemmp 8812 21504 3072 gemmp 8300 9216 3072 2>&1 | grep "^gemm"
} > gpurun_out/r06_visit_m.log 2>&1
cat gpurun_out/r06_visit_m.log
