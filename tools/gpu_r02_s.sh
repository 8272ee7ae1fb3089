#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
{
echo "== flux parity with the text stream on the side lane"; timeout 900 python -m pytest tests/test_flux_gpu.py tests/test_flux2_gpu.py -q -m gpu 2>&1 | tail -4
echo "== serial bench, 4 steps"; timeout 900 python bench.py --steps 4 --warmup 2 --no-overlap --no-cpu-baseline > gpurun_out/bs.out 2> gpurun_out/bs.err; tail -2 gpurun_out/bs.err; grep '^{' gpurun_out/bs.out | python -c "
import sys,json; d=json.loads(sys.stdin.read()); ip=d['config']['inpaint']; print(d['value'], d['ms_per_step'], 'dit_step_ms', ip['dit_step_ms'], 'accounted', ip['step_ms_accounted_by_groups'])
for g in ip['mfma_launch_groups'][:12]: print(g['kernel'],g['m'],g['n'],g['k'],g['launches_per_step'],round(g['mean_ms'],4),g['timing'])"
} > gpurun_out/r02_s.log 2>&1
cat gpurun_out/r02_s.log
