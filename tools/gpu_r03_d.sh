#!/bin/bash
# round 3, GPU visit D: the whole GPU suite on the round's kernels and the hardened parity tests
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
{
echo "== whole gpu suite"; timeout 3000 python -m pytest tests -q -m gpu -p no:cacheprovider --durations=12 2>&1 | tail -40
} > gpurun_out/r03_d.log 2>&1
tail -60 gpurun_out/r03_d.log
