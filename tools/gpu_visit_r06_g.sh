#!/bin/bash
# round 6, visit G: the detectors with fp32 DFL logits (box error per level), the pruned library's op / attention / norm / gemm tests, config 1 / 2
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
{
  echo "== detector tests (prints the per-level box error)"
  timeout 1200 python -m pytest tests/test_yolo_gpu.py tests/test_yolo11_gpu.py -q -x -p no:cacheprovider -s --tb=short 2>&1 | grep -E "YOLO|passed|failed|Error|assert" | cut -c1-330
  echo "== op tests on the pruned library"
  timeout 1500 python -m pytest tests/test_ops_gpu.py -q -x -p no:cacheprovider --tb=short 2>&1 | tail -5
  echo "== detection flow / page vision"
  timeout 900 python -m pytest tests/test_page_vision_gpu.py tests/test_determinism_gpu.py -q -x -p no:cacheprovider --tb=short 2>&1 | tail -5
  for c in 1 2; do echo "== config $c"; timeout 600 python bench.py --config $c --steps 30 --warmup 3 --no-traffic --no-cpu-baseline > gpurun_out/r06_g_c$c.out 2> gpurun_out/r06_g_c$c.err
    python - gpurun_out/r06_g_c$c.out <<'PY'
import json, sys
line = [l for l in open(sys.argv[1]) if l.startswith("{")]
d = json.loads(line[-1]) if line else {}
c = d.get("config", {})
print(round(d.get("value", 0), 2), d.get("unit"), c.get("stage_wall_ms_one_page"), "detect_net_ms", c.get("detect_net_ms"), c.get("detect_aux_ms"), c.get("detect_rtdetr_ms"), "segment", c.get("segment_ms"))
PY
  done
} > gpurun_out/r06_visit_g.log 2>&1
cat gpurun_out/r06_visit_g.log
