#!/bin/bash
# round 5, visit p: the norm kernels without fused multiply-adds (forms must now agree byte for byte with the round 1-4 kernel on hardware),
# fp32 GEMM with the next K step fetched under the MFMAs, SAM high-precision parity on the changed decoder kernels
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
{
  timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_sam2_gpu.py -q -p no:cacheprovider -k "norm or softmax or f32 or first_block or fused or quant or mask or hi_lo or high" 2>&1 | tail -6
  timeout 300 python tools/bench_kernels.py norm 8812 3072 normq 8512 3072 2>&1 | grep "^norm"
  timeout 300 python tools/sam_decoder_ops.py 2>&1 | grep -A14 "precision high"
} > gpurun_out/r05_visit_p.log 2>&1
cat gpurun_out/r05_visit_p.log
