#!/bin/bash
# round 3, last GPU visit (6 GPU-minutes left): the device-tail tests incl. the feather weight, a short config-5 bench through the product
# path that now makes the composite weight in HBM, smoke() on the ABI-4 library.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
{ echo "== device tail tests"; timeout 200 python -m pytest tests/test_device_tail_gpu.py -x -q -p no:cacheprovider 2>&1 | tail -6; } > gpurun_out/r03_final_tail_tests.log 2>&1
cat gpurun_out/r03_final_tail_tests.log
{ echo "== config 5, short"; timeout 170 python bench.py --config 5 --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/bench_c5_final.out 2> gpurun_out/bench_c5_final.err
  grep '^{' gpurun_out/bench_c5_final.out > gpurun_out/r03_bench_config5_device_feather.json
  python - <<'PY'
import json
d = json.load(open("gpurun_out/r03_bench_config5_device_feather.json")); c = d["config"]
print(round(d["value"], 4), d["unit"], round(d["ms_per_step"], 1), "ms/page", c.get("stage_wall_ms_one_page"), "roofline", round(d["roofline"]["frac"], 3))
PY
  tail -2 gpurun_out/bench_c5_final.err; } > gpurun_out/r03_final_c5.log 2>&1
cat gpurun_out/r03_final_c5.log
{ echo "== smoke"; timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -4; } > gpurun_out/r03_final_smoke.log 2>&1
cat gpurun_out/r03_final_smoke.log
