#!/bin/bash
# round 6, visit H: the one-wave-per-SIMD attention kernel: parity tests, then timing against the 8-wave kernel (MTX_ATTN_W64=0), separate processes alternating
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
{
  python tools/attn_debug.py 2>&1 | grep -v "bad cols\|ref \[\|row \|amdgpu" | cut -c1-260
  echo "== attention tests (w64 kernel on)"
  timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_determinism_gpu.py -q -x -p no:cacheprovider -k "attention" --tb=long 2>&1 | grep -E "check_attention\(|passed|failed|rel err" | head -12
  for r in 1 2 3; do
    for w in 1 0; do echo "-- MTX_ATTN_W64=$w"; MTX_ATTN_W64=$w timeout 300 python tools/bench_kernels.py attn 8812 attn 8704 attn 4096 2>&1 | grep "^attn"; done
  done
} > gpurun_out/r06_visit_h.log 2>&1
cat gpurun_out/r06_visit_h.log
