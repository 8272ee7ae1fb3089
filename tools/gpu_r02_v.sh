#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
{
echo "== flux2 parity with the text-stream linears on the side lane + lane tests"; timeout 900 python -m pytest tests/test_flux2_gpu.py tests/test_plan_lanes.py -q -m gpu 2>&1 | tail -3
echo "== config 5"; timeout 900 python bench.py --config 5 --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/b5.out 2> gpurun_out/b5.err; tail -1 gpurun_out/b5.err; grep '^{' gpurun_out/b5.out > gpurun_out/r02_bench_config5_lanes.json; python -c "
import json; d=json.load(open('gpurun_out/r02_bench_config5_lanes.json')); print(d['value'], d['ms_per_step'], d['config']['inpaint']['dit_step_ms'], d['config'].get('dit_step_ms_one_lane_eager'), d['config']['stage_wall_ms_one_page'])"
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -6
} > gpurun_out/r02_v.log 2>&1
cat gpurun_out/r02_v.log
