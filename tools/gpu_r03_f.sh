#!/bin/bash
# round 3, GPU visit F: detectors submitted together on their own streams vs called one after the other (configs 1 and 2), detector tests
cd "$GRAFT_REPO_ROOT" || exit 1
{
echo "== detector tests"; timeout 900 python -m pytest tests/test_sam2_gpu.py tests/test_page_vision_gpu.py -q -x -p no:cacheprovider 2>&1 | tail -3
for c in 2; do
  for flag in "" "--serial-detectors"; do
    echo "== config $c $flag"; timeout 600 python bench.py --config $c --steps 40 --warmup 5 --no-cpu-baseline $flag 2>/dev/null | grep '^{' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print(round(d['value'],2),'pages/s', round(d['ms_per_step'],2),'ms', c.get('stage_wall_ms_one_page'), c.get('host_gpu_split_one_page'))"
  done
done
} > gpurun_out/r03_f.log 2>&1
cat gpurun_out/r03_f.log
