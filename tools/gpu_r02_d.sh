#!/bin/bash
# round 2, GPU visit D: the whole GPU suite on the current tree, the rocprofv3 pair of the default bench (csv), bench lines with two pages in flight
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/prof_d
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
{
echo "== new tests"; timeout 1500 python -m pytest tests/test_yolo11_gpu.py tests/test_bubble_crops_gpu.py tests/test_bench_launch.py -q -s -m gpu 2>&1 | grep -v "^$" | tail -40
echo "== whole gpu suite"; timeout 3000 python -m pytest tests -q -m gpu --deselect tests/test_yolo11_gpu.py --deselect tests/test_bubble_crops_gpu.py --deselect tests/test_bench_launch.py -p no:cacheprovider 2>&1 | tail -12
echo "== default bench (plain, two pages in flight)"; timeout 1200 python bench.py --steps 10 --warmup 3 > gpurun_out/bench_default.out 2> gpurun_out/bench_default.err; grep '^{' gpurun_out/bench_default.out > gpurun_out/r02_bench_default.json; wc -c gpurun_out/r02_bench_default.json
echo "== default bench under rocprofv3 --kernel-trace --stats (csv)"; (cd /tmp && timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_d -o bench -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/bench_default_rocprof.out 2> $R/gpurun_out/bench_default_rocprof.err); grep '^{' gpurun_out/bench_default_rocprof.out > gpurun_out/r02_bench_default_under_rocprof.json; for f in $(find gpurun_out/prof_d -name "*kernel_stats.csv"); do cp $f gpurun_out/r02_bench_default_kernel_stats.csv; head -12 $f | cut -c1-160; done
echo "== config 5"; timeout 900 python bench.py --config 5 --steps 8 --warmup 2 > gpurun_out/bench5.out 2> gpurun_out/bench5.err; grep '^{' gpurun_out/bench5.out > gpurun_out/r02_bench_config5.json; wc -c gpurun_out/r02_bench_config5.json
echo "== config 5 under rocprofv3"; (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_d5 -o bench -- python $R/bench.py --config 5 --steps 4 --warmup 2 --no-cpu-baseline > $R/gpurun_out/bench5_rocprof.out 2> $R/gpurun_out/bench5_rocprof.err); for f in $(find gpurun_out/prof_d5 -name "*kernel_stats.csv"); do cp $f gpurun_out/r02_bench_config5_kernel_stats.csv; head -8 $f | cut -c1-160; done
echo "== config 2"; timeout 600 python bench.py --config 2 --steps 30 --warmup 5 > gpurun_out/bench2c.out 2> gpurun_out/bench2c.err; grep '^{' gpurun_out/bench2c.out > gpurun_out/r02_bench_config2.json
echo "== config 1"; timeout 600 python bench.py --config 1 --steps 30 --warmup 5 > gpurun_out/bench1c.out 2> gpurun_out/bench1c.err; grep '^{' gpurun_out/bench1c.out > gpurun_out/r02_bench_config1.json
echo "== config 2 no overlap"; timeout 600 python bench.py --config 2 --steps 30 --warmup 5 --no-overlap --no-cpu-baseline > gpurun_out/bench2n.out 2> gpurun_out/bench2n.err; grep '^{' gpurun_out/bench2n.out > gpurun_out/r02_bench_config2_no_overlap.json
} > gpurun_out/r02_d.log 2>&1
find gpurun_out/prof_d gpurun_out/prof_d5 -type f -size +1M -delete 2>/dev/null
tail -90 gpurun_out/r02_d.log
