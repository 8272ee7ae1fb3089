#!/bin/bash
# round 5, visit m: x * v_rcp_f32(1 + e^t) instead of IEEE division in the 16-bit activation epilogues — the library as built against the same sources with -DMTX_EXACT_DIV
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
EX=$GRAFT_REPO_ROOT/mangatranslator_amd/csrc/build/libmtx_hip_exactdiv.so
{
  echo "== op tests on the default library (activations, GEMM epilogues, fp8 SwiGLU fusions byte for byte, conv, norms)"
  timeout 900 python -m pytest tests/test_ops_gpu.py -q -x -p no:cacheprovider -k "not alternative_schedules" 2>&1 | tail -3
  for r in 1 2; do
    echo "== round $r: exact division"
    MTX_HIP_LIBRARY=$EX timeout 300 python tools/bench_kernels.py gemmg 8812 12288 3072 gemmg 8300 12288 3072 glu 8512 9216 3072 9216 2>&1 | grep "^gemm"
    echo "== round $r: v_rcp_f32"
    timeout 300 python tools/bench_kernels.py gemmg 8812 12288 3072 gemmg 8300 12288 3072 glu 8512 9216 3072 9216 2>&1 | grep "^gemm"
  done
  echo "== no epilogue math, for scale"
  timeout 300 python tools/bench_kernels.py gemmn 8812 12288 3072 2>&1 | grep "^gemm"
} > gpurun_out/r05_visit_m.log 2>&1
cat gpurun_out/r05_visit_m.log
