#!/bin/bash
# round 2, GPU visits L and U (final): whole GPU suite on the final kernels; bench lines (default with two pages in flight; serial pair plain + rocprofv3 for the
# roofline check; config 5; upscale only); kernel-trace summaries that separate queue time-sharing from kernel time
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
prof() {   # prof <tag> <bench args...>
  tag=$1; shift
  rm -rf /tmp/prof_$tag; mkdir -p /tmp/prof_$tag
  (cd /tmp && timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o bench -- python $R/bench.py "$@" --no-cpu-baseline > $R/gpurun_out/bench_${tag}_rocprof.out 2> $R/gpurun_out/bench_${tag}_rocprof.err)
  grep '^{' gpurun_out/bench_${tag}_rocprof.out > gpurun_out/r02_bench_${tag}_under_rocprof.json
  for f in $(find /tmp/prof_$tag -name "*kernel_stats.csv"); do cp $f gpurun_out/r02_bench_${tag}_kernel_stats.csv; head -8 $f | cut -c1-150; done
  for f in $(find /tmp/prof_$tag -name "*kernel_trace.csv"); do python tools/summarize_kernel_trace.py $f gpurun_out/r02_bench_${tag}_kernel_trace_summary.json 30; done
}
{
echo "== whole gpu suite"; timeout 3000 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -8
echo "== default bench (two pages in flight)"; timeout 1200 python bench.py --steps 10 --warmup 3 > gpurun_out/bench_default.out 2> gpurun_out/bench_default.err; grep '^{' gpurun_out/bench_default.out > gpurun_out/r02_bench_default.json; wc -c gpurun_out/r02_bench_default.json
echo "== (rocprofv3 crashed on the two-thread default run in visit L: the serial pair below is the profiled one)"
echo "== serial bench (--no-overlap), plain"; timeout 1200 python bench.py --steps 10 --warmup 3 --no-overlap --no-cpu-baseline > gpurun_out/bench_serial.out 2> gpurun_out/bench_serial.err; grep '^{' gpurun_out/bench_serial.out > gpurun_out/r02_bench_serial.json
echo "== serial bench under rocprofv3"; prof serial --steps 10 --warmup 3 --no-overlap
echo "== config 5"; timeout 900 python bench.py --config 5 --steps 8 --warmup 2 > gpurun_out/bench5.out 2> gpurun_out/bench5.err; grep '^{' gpurun_out/bench5.out > gpurun_out/r02_bench_config5.json; wc -c gpurun_out/r02_bench_config5.json
echo "== config 5 under rocprofv3"; prof config5 --config 5 --steps 4 --warmup 2
echo "== upscale only"; timeout 300 python bench.py --stages upscale --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | grep '^{' > gpurun_out/r02_bench_upscale_only.json; python -c "import json; d=json.load(open('gpurun_out/r02_bench_upscale_only.json')); print('whole RCAN ms/page', d['ms_per_step'], d.get('roofline',{}).get('achieved'), d.get('roofline',{}).get('frac'))"
echo "== config 1"; timeout 600 python bench.py --config 1 --steps 30 --warmup 5 > gpurun_out/bench1c.out 2> gpurun_out/bench1c.err; grep '^{' gpurun_out/bench1c.out > gpurun_out/r02_bench_config1.json
echo "== config 2"; timeout 600 python bench.py --config 2 --steps 30 --warmup 5 > gpurun_out/bench2c.out 2> gpurun_out/bench2c.err; grep '^{' gpurun_out/bench2c.out > gpurun_out/r02_bench_config2.json
} > gpurun_out/r02_l.log 2>&1
tail -120 gpurun_out/r02_l.log
