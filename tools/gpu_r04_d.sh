#!/bin/bash
# round 4, visit D: SAM tests with f16 storage; default (config 4) and config 5 bench lines on the K-slice GEMM tail; kernel stats of the
# serial config-4 bench (how much of the GPU time the tail takes now)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
{ echo "== SAM tests"; timeout 900 python -m pytest tests/test_sam2_gpu.py tests/test_page_vision_gpu.py -q -m gpu -p no:cacheprovider 2>&1 | tail -5
  echo "== default bench"; timeout 900 python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>gpurun_out/r04_d_bench_default.err | grep '^{' > gpurun_out/r04_d_bench_default.json
  python - <<'PY'
import json
d=json.load(open('gpurun_out/r04_d_bench_default.json')); c=d['config']
print(round(d['value'],4), round(d['ms_per_step'],1), 'dit step', c['inpaint'].get('dit_step_ms'), d.get('roofline',{}).get('frac'), c.get('stage_wall_ms_one_page'))
for g in c['inpaint'].get('mfma_launch_groups', []):
    print('   ', {k: g[k] for k in g if k in ('kernel','m','n','k','launches','ms','frac','tflops')})
PY
  echo "== config 5"; timeout 600 python bench.py --config 5 --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | grep '^{' > gpurun_out/r04_d_bench_config5.json
  python - <<'PY'
import json
d=json.load(open('gpurun_out/r04_d_bench_config5.json')); c=d['config']
print(round(d['value'],4), round(d['ms_per_step'],1), 'dit step', c['inpaint'].get('dit_step_ms'), d.get('roofline',{}).get('frac'), c.get('stage_wall_ms_one_page'))
PY
  echo "== serial config 4 under rocprofv3 (3 steps)"
  rm -rf /tmp/prof_s; mkdir -p /tmp/prof_s
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-overlap --no-cpu-baseline > $R/gpurun_out/r04_d_serial_rocprof.out 2> $R/gpurun_out/r04_d_serial_rocprof.err)
  grep '^{' gpurun_out/r04_d_serial_rocprof.out > gpurun_out/r04_d_bench_serial_under_rocprof.json
  for f in $(find /tmp/prof_s -name "*kernel_stats.csv"); do cp $f gpurun_out/r04_d_bench_serial_kernel_stats.csv; head -12 $f | cut -c1-160; done
} > gpurun_out/r04_d.log 2>&1
cat gpurun_out/r04_d.log
