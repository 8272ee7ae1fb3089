#!/bin/bash
# First GPU visit of the next round (written at the end of round 4, when the GPU budget was spent): what was built on the simulator only
# gets its first run on gfx950, then the figure VERDICT r03 #2 asked for and its price.
#   gpurun --timeout 900 -- 'bash tools/gpu_first_visit_r05.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
{
  echo "== fp32 ops, hi + lo weight GEMM, SAM precision high (tiny, Hiera-L): first hardware run"
  timeout 400 python -m pytest tests/test_zz_first_hardware_run_gpu.py -q -x -p no:cacheprovider 2>&1 | tail -8
  echo "== SAM-2.1 Hiera-L, logits at trained-model spread: bf16 / f16 / f16 high / bf16 high against the fp32 oracle"
  timeout 400 python tools/sam_dtype_probe.py gpurun_out/r05_sam_dtype_probe.json 2>&1 | grep -v "^$" | tail -6
  for p in fast high; do
    echo "== config 2 (detect + segment), --sam-precision $p"
    timeout 300 python bench.py --config 2 --steps 30 --warmup 3 --no-cpu-baseline --no-traffic --sam-precision $p > gpurun_out/r05_c2_$p.out 2> gpurun_out/r05_c2_$p.err
    python - gpurun_out/r05_c2_$p.out <<'PY'
import json, sys
line = [l for l in open(sys.argv[1]) if l.startswith("{")]
d = json.loads(line[-1]) if line else {}
print(round(d.get("value", 0), 2), d.get("unit"), d.get("config", {}).get("stage_wall_ms_one_page"))
PY
  done
} > gpurun_out/r05_first_visit.log 2>&1
cat gpurun_out/r05_first_visit.log
