#!/bin/bash
# FIRST GPU visit of round 4: (a) wheel probe on the GPU box (VERDICT r03 missing #1), (b) the hardware tests of the two epilogue fusions
# (now -m gpu tests), (c) config-5 A/B with and without them.
#   gpurun --timeout 900 -- 'bash tools/gpu_r04_first_visit.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
{ echo "== wheel probe"; timeout 120 python tools/probe_wheels.py gpurun_out/r04_wheel_probe_gpu_box.json | python -c "
import json,sys; d=json.load(sys.stdin); print({k:v['ok'] for k,v in d['wheels'].items()}); print(d['network']); print(d['pip_download_opencv']); print(d['wheel_files_matching'])"
  echo "== fused-epilogue hardware tests"; timeout 500 python -m pytest tests/test_ops_gpu.py tests/test_flux2_gpu.py -q -m gpu -p no:cacheprovider -k "glu or mx_fp8 or quantiser" 2>&1 | tail -15
  for rep in 1 2; do
    echo "== config 5 (rep $rep)"; timeout 200 python bench.py --config 5 --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print(round(d['value'],4), round(d['ms_per_step'],1), 'dit', round(c['inpaint']['dit_step_ms'],2), c['stage_wall_ms_one_page'])"
    echo "== config 5 --glu-epilogue (rep $rep)"; timeout 200 python bench.py --config 5 --steps 6 --warmup 2 --no-cpu-baseline --glu-epilogue 2>/dev/null | grep '^{' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print(round(d['value'],4), round(d['ms_per_step'],1), 'dit', round(c['inpaint']['dit_step_ms'],2), c['stage_wall_ms_one_page'])"
  done
} > gpurun_out/r04_first_visit.log 2>&1
cat gpurun_out/r04_first_visit.log
