set -x
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/bench_t.log 2> gpurun_out/bench_t.err; python - <<'PY'
import json
d = json.loads(open('gpurun_out/bench_t.log').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['config']['inpaint']['dit_step_ms'], {k: d['roofline'][k] for k in ('achieved','frac','avg_launch_ms')}, {k: d['roofline_gemm'][k] for k in ('achieved','frac','avg_launch_ms')})
PY
tail -3 gpurun_out/bench_t.err
