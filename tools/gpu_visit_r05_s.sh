#!/bin/bash
# round 5, visit s: the whole -m gpu suite on the final kernels, four worker processes sharing the GPU (the serial run takes 6.5 min; 5.8 GPU-minutes were left)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
{ timeout 290 python -m pytest tests -q -m gpu -p no:cacheprovider -n 4 --dist loadfile 2>&1 | tail -12; } > gpurun_out/r05_visit_s.log 2>&1
cat gpurun_out/r05_visit_s.log
