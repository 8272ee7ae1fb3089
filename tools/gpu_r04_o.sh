#!/bin/bash
# round 4, visit O: whole GPU suite + smoke on the code with 1x1 convolutions on the GEMM kernels; config 1 / 2 / 5 / upscale lines
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
{ echo "== whole gpu suite"; timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -6
  echo "== smoke"; timeout 600 python __graft_entry__.py smoke 2>&1 | grep "^smoke" | tail -5
} > gpurun_out/r04_gpu_suite.log 2>&1
cat gpurun_out/r04_gpu_suite.log
bash tools/gpu_round_end_r04.sh configs
