#!/bin/bash
# FIRST GPU visit of the next round: the hardware checks round 3 could not run (its GPU minutes were spent), then the A/B that decides
# whether the two epilogue fusions become the default of the FLUX.2-Klein fp8 path.
#   gpurun --timeout 600 -- 'bash tools/gpu_r04_first_visit.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
{ echo "== pending hardware checks"; timeout 400 python tools/check_pending_on_gpu.py 2>&1 | tail -12
  for rep in 1 2; do
    echo "== config 5 (rep $rep)"; timeout 200 python bench.py --config 5 --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print(round(d['value'],4), round(d['ms_per_step'],1), 'dit', round(c['inpaint']['dit_step_ms'],2), c['stage_wall_ms_one_page'])"
    echo "== config 5 --glu-epilogue (rep $rep)"; timeout 200 python bench.py --config 5 --steps 6 --warmup 2 --no-cpu-baseline --glu-epilogue 2>/dev/null | grep '^{' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print(round(d['value'],4), round(d['ms_per_step'],1), 'dit', round(c['inpaint']['dit_step_ms'],2), c['stage_wall_ms_one_page'])"
  done
} > gpurun_out/r04_first_visit.log 2>&1
cat gpurun_out/r04_first_visit.log
