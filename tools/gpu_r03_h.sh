#!/bin/bash
# round 3, GPU visit H: device tail of the inpaint stage — parity on the GPU, then configs 5 and 4 (stage walls)
cd "$GRAFT_REPO_ROOT" || exit 1
{
echo "== parity"; timeout 900 python -m pytest tests/test_device_tail_gpu.py tests/test_flux2_gpu.py::test_klein_loop tests/test_page_vision_gpu.py -q -x -p no:cacheprovider 2>&1 | tail -4
echo "== config 5"; timeout 900 python bench.py --config 5 --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | grep '^{' > gpurun_out/r03_h_config5.json; python -c "
import json; d=json.load(open('gpurun_out/r03_h_config5.json')); c=d['config']; print(round(d['value'],4),'pages/s', round(d['ms_per_step'],1),'ms/page', c['stage_wall_ms_one_page'])"
echo "== config 4 (3 steps)"; timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | grep '^{' > gpurun_out/r03_h_config4.json; python -c "
import json; d=json.load(open('gpurun_out/r03_h_config4.json')); c=d['config']; print(round(d['value'],4),'pages/s', round(d['ms_per_step'],1),'ms/page', c['stage_wall_ms_one_page'], 'roofline', round(d['roofline']['frac'],3))"
} > gpurun_out/r03_h.log 2>&1
cat gpurun_out/r03_h.log
