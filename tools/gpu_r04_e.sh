#!/bin/bash
# round 4, visit E: front-half replicas for configs 1 / 2 (N pages' detect stages in flight on N model instances), hardware-queue counts;
# config 5 A/B of the epilogue fusions on one box + kernel stats; the default line with its counter child (roofline.traffic)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
line() { python - "$1" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); c = d['config']
    print(round(d['value'], 4), 'pages/s', round(d['ms_per_step'], 2), 'ms  replicas', c.get('front_replicas'), 'hwq', c.get('hw_queues'), c.get('stage_wall_ms_one_page'), 'dit', (c.get('inpaint') or {}).get('dit_step_ms'), 'roofline', {k: d.get('roofline', {}).get(k) for k in ('frac', 'traffic')})
except Exception as e:
    print('no line:', e)
PY
}
{ for cfgn in 2 1; do
    for rep in 1 2 3; do
      for q in 8 16; do
        echo "== config $cfgn replicas $rep hwq $q"; GPU_MAX_HW_QUEUES=$q timeout 300 python bench.py --config $cfgn --steps 40 --warmup 6 --front-replicas $rep --no-cpu-baseline --no-traffic 2>/dev/null | grep '^{' > gpurun_out/r04_e_c${cfgn}_r${rep}_q${q}.json; line gpurun_out/r04_e_c${cfgn}_r${rep}_q${q}.json
      done
    done
  done
  echo "== config 5, fusions on / off / on"
  for v in "" "--no-glu-epilogue" ""; do
    timeout 300 python bench.py --config 5 --steps 6 --warmup 2 --no-cpu-baseline --no-traffic $v 2>/dev/null | grep '^{' > gpurun_out/r04_e_c5.json; echo "[$v]"; line gpurun_out/r04_e_c5.json
  done
  echo "== config 5 serial under rocprofv3"
  rm -rf /tmp/prof_5; mkdir -p /tmp/prof_5
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_5 -o bench -- python $R/bench.py --config 5 --steps 3 --warmup 1 --no-overlap --no-cpu-baseline --no-traffic > $R/gpurun_out/r04_e_c5_rocprof.out 2> $R/gpurun_out/r04_e_c5_rocprof.err)
  for f in $(find /tmp/prof_5 -name "*kernel_stats.csv"); do cp $f gpurun_out/r04_e_bench_config5_kernel_stats.csv; head -14 $f | cut -c1-170; done
  echo "== default line with the counter child"
  timeout 900 python bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>gpurun_out/r04_e_default.err | grep '^{' > gpurun_out/r04_e_bench_default_with_traffic.json; line gpurun_out/r04_e_bench_default_with_traffic.json
  python - <<'PY'
import json
try:
    d = json.load(open('gpurun_out/r04_e_bench_default_with_traffic.json')); print(json.dumps(d['roofline'].get('traffic_detail'), indent=0)[:1500])
except Exception as e: print(e)
PY
} > gpurun_out/r04_e.log 2>&1
cat gpurun_out/r04_e.log
