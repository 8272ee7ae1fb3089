#!/bin/bash
# round 3, GPU visit C3: split channel attention with every operand prefetched; parity tests + same-box A/B + kernel stats
cd "$GRAFT_REPO_ROOT" || exit 1
{
echo "== parity"; timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_rcan_gpu.py tests/test_bubble_crops_gpu.py -q -x -p no:cacheprovider -k "rcab or rcan or bubble or ca" 2>&1 | tail -4
bash tools/gpu_r03_c2.sh
} > gpurun_out/r03_c3.log 2>&1
cat gpurun_out/r03_c3.log
