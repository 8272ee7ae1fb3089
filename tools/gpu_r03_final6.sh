#!/bin/bash
# round 3: the bench lines with eight hardware queues set by the package itself (no shell variable): config 2, default (short), config 5 (short), config 1
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
unset GPU_MAX_HW_QUEUES
show() { grep '^{' "$1" > "$2"; python - "$2" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); c = d["config"]; r = d.get("roofline") or {}
print(round(d["value"], 4), d["unit"], round(d["ms_per_step"], 2), "ms/page", c.get("stage_wall_ms_one_page"), "roofline", r.get("frac"))
PY
}
echo "== config 2"; timeout 100 python bench.py --config 2 --steps 40 --warmup 4 > gpurun_out/b6_c2.out 2> gpurun_out/b6_c2.err; show gpurun_out/b6_c2.out gpurun_out/r03_bench_config2_hwq8.json; tail -1 gpurun_out/b6_c2.err
echo "== default, short"; timeout 100 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/b6_d.out 2> gpurun_out/b6_d.err; show gpurun_out/b6_d.out gpurun_out/r03_bench_default_hwq8_short.json; tail -1 gpurun_out/b6_d.err
echo "== config 5, short"; timeout 100 python bench.py --config 5 --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/b6_c5.out 2> gpurun_out/b6_c5.err; show gpurun_out/b6_c5.out gpurun_out/r03_bench_config5_hwq8_short.json; tail -1 gpurun_out/b6_c5.err
echo "== config 1"; timeout 100 python bench.py --config 1 --steps 40 --warmup 4 --no-cpu-baseline > gpurun_out/b6_c1.out 2> gpurun_out/b6_c1.err; show gpurun_out/b6_c1.out gpurun_out/r03_bench_config1_hwq8.json
