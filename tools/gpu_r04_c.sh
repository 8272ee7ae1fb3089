#!/bin/bash
# round 4, visit C: K-slice tail after the fix-up rewrite — slice-count sweep per shape (calibrates the launcher's cost model), A/B against
# round 3's tail and against no split
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
{ echo "== tests"; timeout 300 python -m pytest tests/test_ops_gpu.py -q -m gpu -p no:cacheprovider -k "k_slice or gemm" 2>&1 | tail -4
  echo "== sweep"
  B="python tools/bench_kernels.py"
  timeout 200 $B gemmn 8812 3072 15360 gemmo 8812 3072 15360 gemms2 8812 3072 15360 gemms3 8812 3072 15360 gemm 8812 3072 15360 gemmn 8812 3072 15360 2>&1 | grep gemm
  timeout 200 $B gemmn 8300 3072 12288 gemmo 8300 3072 12288 gemms2 8300 3072 12288 gemms3 8300 3072 12288 gemms4 8300 3072 12288 gemm 8300 3072 12288 2>&1 | grep gemm
  timeout 200 $B gemmn 512 3072 12288 gemmo 512 3072 12288 gemms2 512 3072 12288 gemms3 512 3072 12288 gemms4 512 3072 12288 gemms6 512 3072 12288 gemms8 512 3072 12288 gemms10 512 3072 12288 gemm 512 3072 12288 2>&1 | grep gemm
  timeout 200 $B gemm8n 8512 3072 12288 gemm8o 8512 3072 12288 gemm8s2 8512 3072 12288 gemm8s3 8512 3072 12288 gemm8 8512 3072 12288 gemm8n 8512 3072 9216 gemm8o 8512 3072 9216 gemm8s2 8512 3072 9216 gemm8s3 8512 3072 9216 gemm8 8512 3072 9216 2>&1 | grep gemm
  timeout 200 $B gemm8n 8000 3072 9216 gemm8o 8000 3072 9216 gemm8s2 8000 3072 9216 gemm8s3 8000 3072 9216 gemm8 8000 3072 9216 gemm8n 512 3072 9216 gemm8o 512 3072 9216 gemm8s4 512 3072 9216 gemm8s8 512 3072 9216 gemm8 512 3072 9216 2>&1 | grep gemm
  timeout 200 $B gemmn 8300 3072 3072 gemms2 8300 3072 3072 gemms3 8300 3072 3072 gemm 8300 3072 3072 gemmn 512 3072 3072 gemms2 512 3072 3072 gemms4 512 3072 3072 gemm 512 3072 3072 2>&1 | grep gemm
} > gpurun_out/r04_c.log 2>&1
cat gpurun_out/r04_c.log
