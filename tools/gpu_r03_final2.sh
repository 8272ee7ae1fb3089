#!/bin/bash
# round 3: kernel-trace statistics of the config-5 bench (Klein step with fused quantisers, device tail) on the final code
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/prof_c5; mkdir -p /tmp/prof_c5
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c5 -o bench -- python $R/bench.py --config 5 --steps 4 --warmup 1 --no-overlap --no-cpu-baseline > $R/gpurun_out/bench_c5_rocprof.out 2> $R/gpurun_out/bench_c5_rocprof.err)
grep '^{' gpurun_out/bench_c5_rocprof.out > gpurun_out/r03_bench_config5_serial_under_rocprof.json
for f in $(find /tmp/prof_c5 -name "*kernel_stats.csv"); do cp $f gpurun_out/r03_bench_config5_kernel_stats.csv; head -24 $f | cut -c1-170; done
python - <<'PY'
import json
d = json.load(open("gpurun_out/r03_bench_config5_serial_under_rocprof.json")); c = d["config"]
print(round(d["value"], 4), d["unit"], round(d["ms_per_step"], 1), "ms/page", c.get("stage_wall_ms_one_page"), "roofline", round(d["roofline"]["frac"], 3))
PY
tail -2 gpurun_out/bench_c5_rocprof.err
