#!/bin/bash
# round 2, GPU visit C: the default bench line plain and under rocprofv3 in the same lease, config 5 / 1 / 2 lines
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/prof_c
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
{
echo "== rcan full depth"; timeout 600 python -m pytest tests/test_rcan_gpu.py -q -s -k "full_depth or odd" 2>&1 | grep -i "rcan\|passed\|failed" | tail -8
echo "== two ranks on one device"; timeout 900 python -m pytest tests/test_bench_launch.py -q -m gpu 2>&1 | tail -5
echo "== default bench (plain)"; timeout 1200 python bench.py --steps 10 --warmup 3 > gpurun_out/bench_default.out 2> gpurun_out/bench_default.err; grep '^{' gpurun_out/bench_default.out > gpurun_out/r02_bench_default.json; wc -c gpurun_out/r02_bench_default.json
echo "== default bench under rocprofv3 --kernel-trace --stats"; (cd /tmp && timeout 1500 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_c -o bench -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/bench_default_rocprof.out 2> $R/gpurun_out/bench_default_rocprof.err); grep '^{' gpurun_out/bench_default_rocprof.out > gpurun_out/r02_bench_default_under_rocprof.json; find gpurun_out/prof_c -name "*kernel_stats*" | head -3
echo "== config 5"; timeout 900 python bench.py --config 5 --steps 6 --warmup 2 > gpurun_out/bench5.out 2> gpurun_out/bench5.err; grep '^{' gpurun_out/bench5.out > gpurun_out/r02_bench_config5.json; wc -c gpurun_out/r02_bench_config5.json
echo "== config 5 bf16 linears"; timeout 900 python bench.py --config 5 --no-fp8 --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/bench5b.out 2> gpurun_out/bench5b.err; grep '^{' gpurun_out/bench5b.out > gpurun_out/r02_bench_config5_bf16.json
echo "== config 2"; timeout 600 python bench.py --config 2 --steps 20 --warmup 5 > gpurun_out/bench2c.out 2> gpurun_out/bench2c.err; grep '^{' gpurun_out/bench2c.out > gpurun_out/r02_bench_config2.json
echo "== config 1"; timeout 600 python bench.py --config 1 --steps 20 --warmup 5 > gpurun_out/bench1c.out 2> gpurun_out/bench1c.err; grep '^{' gpurun_out/bench1c.out > gpurun_out/r02_bench_config1.json; tail -3 gpurun_out/bench1c.err
} > gpurun_out/r02_c.log 2>&1
# keep the stats csv small enough to travel
find gpurun_out/prof_c -type f ! -name "*stats*" -size +2M -delete
tail -60 gpurun_out/r02_c.log
