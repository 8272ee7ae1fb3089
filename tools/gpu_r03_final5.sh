#!/bin/bash
# round 3: does the detect stage's stream concurrency hit ROCm's hardware-queue limit (GPU_MAX_HW_QUEUES, default 4)?  Five model lanes + the
# page pipeline's streams share those queues; two graphs on one queue run one after the other.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
show() { grep '^{' "$1" | python -c "
import json, sys
d = json.loads(sys.stdin.read()); c = d['config']
print(round(d['value'], 4), d['unit'], round(d['ms_per_step'], 2), 'ms/page', c.get('stage_wall_ms_one_page'))"; }
for q in 4 8 16; do echo "== config 2, GPU_MAX_HW_QUEUES=$q"; GPU_MAX_HW_QUEUES=$q timeout 100 python bench.py --config 2 --steps 40 --warmup 4 --no-cpu-baseline > gpurun_out/bench_c2q$q.out 2> gpurun_out/bench_c2q$q.err; show gpurun_out/bench_c2q$q.out; done
echo "== config 1, GPU_MAX_HW_QUEUES=8"; GPU_MAX_HW_QUEUES=8 timeout 100 python bench.py --config 1 --steps 40 --warmup 4 --no-cpu-baseline > gpurun_out/bench_c1q8.out 2> gpurun_out/bench_c1q8.err; show gpurun_out/bench_c1q8.out
echo "== config 1, default queues"; timeout 100 python bench.py --config 1 --steps 40 --warmup 4 --no-cpu-baseline > gpurun_out/bench_c1q.out 2> gpurun_out/bench_c1q.err; show gpurun_out/bench_c1q.out
