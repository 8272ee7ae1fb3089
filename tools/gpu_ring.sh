set -x
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/bench_sw.log 2> gpurun_out/bench_sw.err; tail -c 300 gpurun_out/bench_sw.log; python - <<'PY'
import json
d = json.loads(open('gpurun_out/bench_sw.log').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['config']['stage_wall_ms_one_page'], d['config']['inpaint'], d['config']['segment_ms'], d['config']['upscale_ms'], d['config']['detect_net_ms'], d['config']['detect_rtdetr_ms'])
PY
