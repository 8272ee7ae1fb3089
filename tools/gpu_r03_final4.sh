#!/bin/bash
# round 3: config 2 with the mask decoder of page i beside the detectors of page i+1; default line as a sanity check of the shared stage code
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
show() { grep '^{' "$1" > "$2"; python - "$2" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); c = d["config"]
print(round(d["value"], 4), d["unit"], round(d["ms_per_step"], 2), "ms/page", c.get("stage_wall_ms_one_page"), "|", c.get("page_pipeline")[:60])
PY
}
echo "== config 2 (pipelined)"; timeout 120 python bench.py --config 2 --steps 40 --warmup 4 --no-cpu-baseline > gpurun_out/bench_c2p.out 2> gpurun_out/bench_c2p.err; show gpurun_out/bench_c2p.out gpurun_out/r03_bench_config2_pipelined.json; tail -2 gpurun_out/bench_c2p.err
echo "== config 2 --no-overlap"; timeout 120 python bench.py --config 2 --steps 40 --warmup 4 --no-cpu-baseline --no-overlap > gpurun_out/bench_c2s.out 2> gpurun_out/bench_c2s.err; show gpurun_out/bench_c2s.out gpurun_out/r03_bench_config2_in_order.json
