#!/bin/bash
# round 6, visit F: SAM precision "high" on its new arithmetic (weight pairs in one GEMM, fp32 residual added in the GEMM epilogue, fp32 -> 16-bit
# LayerNorm): op tests, SAM tests, the frontier (what each part buys and costs), config 2 with either arithmetic
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
{
  echo "== gemm / norm / hi-lo / f32 op tests"
  timeout 900 python -m pytest tests/test_ops_gpu.py -q -x -p no:cacheprovider -k "gemm or norm or hi_lo or f32" --tb=short 2>&1 | tail -8
  echo "== SAM tests"
  timeout 1500 python -m pytest tests/test_sam2_gpu.py -q -x -p no:cacheprovider --tb=short 2>&1 | tail -8
  echo "== frontier"
  timeout 1500 python tools/sam_frontier.py gpurun_out/r06_sam_frontier.json 2>&1 | grep -v "^$" | tail -9
  for p in fast high; do
    echo "== config 2 (detect + segment), --sam-precision $p"
    timeout 400 python bench.py --config 2 --steps 30 --warmup 3 --no-cpu-baseline --no-traffic --sam-precision $p > gpurun_out/r06_c2_$p.out 2> gpurun_out/r06_c2_$p.err
    python - gpurun_out/r06_c2_$p.out <<'PY'
import json, sys
line = [l for l in open(sys.argv[1]) if l.startswith("{")]
d = json.loads(line[-1]) if line else {}
print(round(d.get("value", 0), 2), d.get("unit"), d.get("config", {}).get("segment_ms"), d.get("config", {}).get("stage_wall_ms_one_page"))
PY
    tail -2 gpurun_out/r06_c2_$p.err
  done
} > gpurun_out/r06_visit_f.log 2>&1
cat gpurun_out/r06_visit_f.log
