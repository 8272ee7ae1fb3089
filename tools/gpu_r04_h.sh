#!/bin/bash
# round 4, visit H: the whole GPU suite (no -x) on the round's code + smoke, then the headline line the way the driver runs it
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
{ echo "== whole gpu suite"; timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -8
  echo "== smoke"; timeout 600 python __graft_entry__.py smoke 2>&1 | grep "^smoke" | tail -5
} > gpurun_out/r04_gpu_suite.log 2>&1
cat gpurun_out/r04_gpu_suite.log
bash tools/gpu_round_end_r04.sh headline
