#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
{
for q in 4 8 16; do
  echo "== config 1, GPU_MAX_HW_QUEUES=$q"; GPU_MAX_HW_QUEUES=$q timeout 600 python bench.py --config 1 --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print(round(d['value'],2),'pages/s', round(d['ms_per_step'],2),'ms', c.get('stage_wall_ms_one_page'))"
done
} > gpurun_out/r03_f2.log 2>&1
cat gpurun_out/r03_f2.log
