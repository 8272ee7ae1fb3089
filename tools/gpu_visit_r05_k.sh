#!/bin/bash
# round 5, visit k: half-tile pipelined softmax (69) against the default (68 = 0); fp8-output attention deep (default) against the round-4 loop (67)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
{
  echo "== parity (schedules 68, 69; fp8-output attention)"
  timeout 600 python -m pytest tests/test_ops_gpu.py -q -x -p no:cacheprovider -k "(alternative_schedules and (68 or 69)) or attention_mx_fp8 or attention_prescaled" 2>&1 | tail -3
  echo "== A/B, T = 8812, four rounds: 0 = 68 default | 69 half-tile pipelined softmax | 65 wide only"
  timeout 600 python tools/bench_kernels.py attnx 8812 0,69,65 4 2>&1 | grep -v "^$" | tail -14
  echo "== attention with MX fp8 output, T = 8512 (Klein): default (deep) / round-4 loop, twice"
  timeout 300 python tools/bench_kernels.py attnq 8512 attnq67 8512 attnq 8512 attnq67 8512 2>&1 | grep "^attn"
} > gpurun_out/r05_visit_k.log 2>&1
cat gpurun_out/r05_visit_k.log
