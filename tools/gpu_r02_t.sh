#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
{
for a in "" "--no-lanes" "" "--no-lanes"; do
echo "== inpaint-only serial bench, 3 steps  [$a]"; timeout 900 python bench.py --stages inpaint --steps 3 --warmup 1 --no-overlap --no-cpu-baseline --time-ops difference $a > gpurun_out/bs.out 2> gpurun_out/bs.err; tail -1 gpurun_out/bs.err; grep '^{' gpurun_out/bs.out | python -c "
import sys,json; d=json.loads(sys.stdin.read()); ip=d['config']['inpaint']; print(d['value'], d['ms_per_step'], 'dit_step_ms', ip['dit_step_ms'])"
done
} > gpurun_out/r02_t.log 2>&1
cat gpurun_out/r02_t.log
