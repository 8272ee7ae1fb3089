#!/bin/bash
# round 6, final GPU visits: bench lines of every BASELINE config on the round's final code, the serial pair (plain + rocprofv3) for the
# roofline cross-check, counter traffic of the bench command itself, the batch harness with image I/O.  Split into parts so that no call
# runs long:  bash tools/gpu_round_end_r06.sh <part>
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
line() { grep '^{' "$1" > "$2"; python - "$2" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); c = d["config"]; r = d.get("roofline", {})
print(round(d["value"], 4), d["unit"], round(d["ms_per_step"], 1), "ms/page | roofline", r.get("kernel", "")[:40], round(r.get("frac", 0), 3), "traffic", r.get("traffic"),
      "|", c.get("stage_wall_ms_one_page"), "| batch_io", c.get("batch_io"))
PY
}
prof() {   # prof <tag> <bench args...>
  tag=$1; shift
  rm -rf /tmp/prof_$tag; mkdir -p /tmp/prof_$tag
  (cd /tmp && timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o bench -- python $R/bench.py "$@" --no-cpu-baseline > $R/gpurun_out/bench_${tag}_rocprof.out 2> $R/gpurun_out/bench_${tag}_rocprof.err)
  grep '^{' gpurun_out/bench_${tag}_rocprof.out > gpurun_out/r06_bench_${tag}_under_rocprof.json
  for f in $(find /tmp/prof_$tag -name "*kernel_stats.csv"); do cp $f gpurun_out/r06_bench_${tag}_kernel_stats.csv; head -10 $f | cut -c1-150; done
  for f in $(find /tmp/prof_$tag -name "*kernel_trace.csv"); do python tools/summarize_kernel_trace.py $f gpurun_out/r06_bench_${tag}_kernel_trace_summary.json 30; done
}
case "$1" in
suite)
  { echo "== whole gpu suite"; timeout 3000 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -6; } > gpurun_out/r06_end_suite.log 2>&1; cat gpurun_out/r06_end_suite.log ;;
headline)
  { echo "== default bench (config 4, two pages in flight), with counter traffic and the CPU baseline"; timeout 2400 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_default.out 2> gpurun_out/bench_default.err; line gpurun_out/bench_default.out gpurun_out/r06_bench_default.json
    tail -3 gpurun_out/bench_default.err
  } > gpurun_out/r06_end_headline.log 2>&1; cat gpurun_out/r06_end_headline.log ;;
serial)
  { echo "== serial bench (--no-overlap), plain"; timeout 1200 python bench.py --steps 10 --warmup 3 --no-overlap --no-cpu-baseline --no-traffic > gpurun_out/bench_serial.out 2> gpurun_out/bench_serial.err; line gpurun_out/bench_serial.out gpurun_out/r06_bench_serial.json
    echo "== serial bench under rocprofv3"; prof serial --steps 10 --warmup 3 --no-overlap --no-traffic
  } > gpurun_out/r06_end_serial.log 2>&1; cat gpurun_out/r06_end_serial.log ;;
configs)
  { for c in 1 2 3; do echo "== config $c"; st=30; [ $c = 3 ] && st=6; timeout 1200 python bench.py --config $c --steps $st --warmup 3 --no-traffic > gpurun_out/bench_c$c.out 2> gpurun_out/bench_c$c.err; line gpurun_out/bench_c$c.out gpurun_out/r06_bench_config$c.json; done
    echo "== config 5"; timeout 1200 python bench.py --config 5 --steps 8 --warmup 2 > gpurun_out/bench_c5.out 2> gpurun_out/bench_c5.err; line gpurun_out/bench_c5.out gpurun_out/r06_bench_config5.json
    echo "== upscale only"; timeout 300 python bench.py --stages upscale --steps 10 --warmup 3 --no-cpu-baseline --no-traffic > gpurun_out/bench_up.out 2>/dev/null; line gpurun_out/bench_up.out gpurun_out/r06_bench_upscale_only.json
  } > gpurun_out/r06_end_configs.log 2>&1; cat gpurun_out/r06_end_configs.log ;;
detect)
  { for c in 1 2; do echo "== config $c"; timeout 600 python bench.py --config $c --steps 30 --warmup 3 --no-traffic > gpurun_out/bench_c$c.out 2> gpurun_out/bench_c$c.err; line gpurun_out/bench_c$c.out gpurun_out/r06_bench_config$c.json; done
  } > gpurun_out/r06_end_detect.log 2>&1; cat gpurun_out/r06_end_detect.log ;;
rest)
  { echo "== config 3"; timeout 600 python bench.py --config 3 --steps 6 --warmup 2 --no-traffic > gpurun_out/bench_c3.out 2> gpurun_out/bench_c3.err; line gpurun_out/bench_c3.out gpurun_out/r06_bench_config3.json
    echo "== config 5"; timeout 600 python bench.py --config 5 --steps 8 --warmup 2 --no-traffic > gpurun_out/bench_c5.out 2> gpurun_out/bench_c5.err; line gpurun_out/bench_c5.out gpurun_out/r06_bench_config5.json
    echo "== upscale only"; timeout 300 python bench.py --stages upscale --steps 10 --warmup 3 --no-cpu-baseline --no-traffic > gpurun_out/bench_up.out 2>/dev/null; line gpurun_out/bench_up.out gpurun_out/r06_bench_upscale_only.json
  } > gpurun_out/r06_end_rest.log 2>&1; cat gpurun_out/r06_end_rest.log ;;
io_detect)
  { echo "== product harness with two front halves in flight (GPU test)"; timeout 600 python -m pytest tests/test_page_vision_gpu.py -q -x -p no:cacheprovider 2>&1 | tail -5
    for c in 2 1; do echo "== batch harness with image I/O: config $c, 64 pages, front_workers = front replicas"
      timeout 600 python bench.py --config $c --steps 30 --warmup 3 --batch-io 64 --no-cpu-baseline --no-traffic > gpurun_out/bench_io$c.out 2> gpurun_out/bench_io$c.err; line gpurun_out/bench_io$c.out gpurun_out/r06_bench_config${c}_batch_io64.json; tail -2 gpurun_out/bench_io$c.err; done
  } > gpurun_out/r06_end_io_detect.log 2>&1; cat gpurun_out/r06_end_io_detect.log ;;
io)
  { echo "== batch harness with image I/O: config 2, 64 pages"; timeout 1200 python bench.py --config 2 --steps 10 --warmup 3 --batch-io 64 --no-cpu-baseline --no-traffic > gpurun_out/bench_io2.out 2> gpurun_out/bench_io2.err; line gpurun_out/bench_io2.out gpurun_out/r06_bench_config2_batch_io64.json
    echo "== batch harness with image I/O: config 5, 16 pages"; timeout 1800 python bench.py --config 5 --steps 4 --warmup 2 --batch-io 16 --no-cpu-baseline --no-traffic > gpurun_out/bench_io5.out 2> gpurun_out/bench_io5.err; line gpurun_out/bench_io5.out gpurun_out/r06_bench_config5_batch_io16.json
    tail -3 gpurun_out/bench_io2.err
  } > gpurun_out/r06_end_io.log 2>&1; cat gpurun_out/r06_end_io.log ;;
config5prof)
  { echo "== config 5 serial (--no-overlap), under rocprofv3 --kernel-trace --stats"; prof config5 --config 5 --steps 4 --warmup 1 --no-overlap --no-traffic --no-extra
    python - <<'PY'
import json
d = json.load(open("gpurun_out/r06_bench_config5_under_rocprof.json"))
print(round(d["value"], 4), d["unit"], round(d["ms_per_step"], 1), "ms/page |", {k: round(v["frac"], 3) for k, v in d.items() if k.startswith("roofline")})
PY
  } > gpurun_out/r06_end_config5prof.log 2>&1; cat gpurun_out/r06_end_config5prof.log ;;
pmc)
  { ARGS="gemm 8812 9216 3072 gemm 8812 3072 15360 attn 8812 attn8 8704 attn88 8704 conv 1536 1024 gemm8 8512 27648 3072 glu 8512 9216 3072 9216"
    rm -rf /tmp/pmc_a; mkdir -p /tmp/pmc_a
    (cd /tmp && timeout 500 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_a -o k -- python $R/tools/bench_kernels.py $ARGS 2>&1 | grep -v "^[WE]2026" | tail -6)
    python tools/summarize_pmc.py "$(find /tmp/pmc_a -name '*counter_collection.csv' | head -1)" "$(find /tmp/pmc_a -name '*kernel_trace.csv' | head -1)" "$ARGS" gpurun_out/r06_pmc_mfma_util.json
    bash tools/pmc_traffic.sh conv 1536 1024 attn 8812 attn8 8704 attn88 8704 gemm 8812 9216 3072 gemm 8812 3072 15360 gemm8 8512 27648 3072; cp gpurun_out/r02_pmc_traffic.json gpurun_out/r06_pmc_traffic.json 2>/dev/null
  } > gpurun_out/r06_end_pmc.log 2>&1; cat gpurun_out/r06_end_pmc.log ;;
esac
