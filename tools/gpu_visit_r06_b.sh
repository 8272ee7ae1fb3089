#!/bin/bash
# round 6, visit B: the 256-tile GEMM tile map — strips (1 / 2 / 4 / auto) and per-group column rotation (+100), same process
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
{
  timeout 900 python tools/bench_kernels.py gemm8st 8512 27648 3072 gemm8st 8000 18432 3072 gemm8st 8512 9216 3072 gemm8st 8512 3072 12288 gemm8st 8512 3072 9216 \
      gemmst 8812 9216 3072 gemmgst 8812 12288 3072 gemmst 8812 3072 15360 gemmst 8300 3072 12288 gemmst 8300 9216 3072 2>&1 | grep -v "^[WE]2026"
} > gpurun_out/r06_visit_b.log 2>&1
cat gpurun_out/r06_visit_b.log
