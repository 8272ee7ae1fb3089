#!/bin/bash
# round 6, visit E: lock-step conv with the DMA pieces / residual loads between the MFMA steps: parity, stamps, timing; GEMM pair / norm ops on hardware
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
{
  echo "== conv / RCAN tests on the lock-step kernel"
  timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_rcan_gpu.py tests/test_determinism_gpu.py -q -x -p no:cacheprovider -k "conv or rcan or rcab" 2>&1 | tail -4
  echo "== gemm / norm / hi-lo op tests (128-tile kernel's new epilogue, weight pairs, fp32 -> 16-bit norm)"
  timeout 900 python -m pytest tests/test_ops_gpu.py -q -x -p no:cacheprovider -k "gemm or norm or hi_lo or f32" --tb=short 2>&1 | tail -25
  timeout 200 tools/probes/conv_probe 1536 1024 | grep "^round\|^ABL 0"
  timeout 200 tools/probes/conv_probe 3072 2048 | grep "^round"
} > gpurun_out/r06_visit_e.log 2>&1
cat gpurun_out/r06_visit_e.log
