#!/bin/bash
# round 5, visit o: register-only wave reductions (v_permlane32/16_swap + DPP) held to the shuffle form's bytes on hardware, the packed-row
# norm kernels against the round 1-4 kernel, SAM's fp32 decoder with a lane per query (few keys) and the few-rows / long-K GEMM
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
{
  timeout 600 python -m pytest tests/test_ops_gpu.py -q -x -p no:cacheprovider -k "norm or softmax or f32 or first_block or fused or quant or mask or hi_lo" 2>&1 | tail -4
  timeout 300 python tools/bench_kernels.py norm 8812 3072 norm 512 3072 normq 8512 3072 2>&1 | grep "^norm"
  timeout 300 python tools/sam_decoder_ops.py 2>&1 | grep -v "^$" | head -48
} > gpurun_out/r05_visit_o.log 2>&1
cat gpurun_out/r05_visit_o.log
