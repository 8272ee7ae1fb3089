#!/bin/bash
# round 5, visit h: fp8 whole-tile GEMM with one segment per k-step (MTX_GEMM_F8_WIDE) against the four-segment loop, same process
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
{
  echo "== parity (fp8 GEMM tests: wide == four-segment, byte for byte)"
  timeout 600 python -m pytest tests/test_ops_gpu.py -q -x -p no:cacheprovider -k "f8 or fp8" 2>&1 | tail -4
  echo "== A/B at Klein shapes (gemm8n = four segments, whole tiles; gemm8w = wide segments), two rounds"
  for r in 1 2; do
    timeout 300 python tools/bench_kernels.py gemm8n 8512 9216 3072 gemm8w 8512 9216 3072 gemm8n 8512 27648 3072 gemm8w 8512 27648 3072 gemm8n 8512 3072 12288 gemm8w 8512 3072 12288 gemm8n 8000 18432 3072 gemm8w 8000 18432 3072 2>&1 | grep "^gemm"
  done
} > gpurun_out/r05_visit_h.log 2>&1
cat gpurun_out/r05_visit_h.log
