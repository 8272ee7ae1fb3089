#!/bin/bash
# round 4, visit G: the whole GPU suite on the round's code; stream-priority experiment for the two-pages-in-flight page pipeline (config 5)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
line() { python - "$1" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); c = d['config']
    print(round(d['value'], 4), 'pages/s', round(d['ms_per_step'], 2), 'ms hwq', c.get('hw_queues'), c.get('stage_wall_ms_one_page'), 'dit', (c.get('inpaint') or {}).get('dit_step_ms'))
except Exception as e:
    print('no line:', e)
PY
}
{ echo "== whole gpu suite"; timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider -x 2>&1 | tail -12
  echo "== smoke"; timeout 600 python __graft_entry__.py smoke 2>&1 | tail -6
  echo "== config 5: replay-stream priority of the back half x hardware queues"
  for rep in 1 2; do for q in 4 8; do for pr in 0 1; do
    echo "-- hwq $q back-priority $pr (rep $rep)"
    GPU_MAX_HW_QUEUES=$q MTX_BACK_PRIORITY=$pr timeout 300 python bench.py --config 5 --steps 6 --warmup 2 --no-cpu-baseline --no-traffic 2>/dev/null | grep '^{' > gpurun_out/r04_g_c5.json; line gpurun_out/r04_g_c5.json
  done; done; done
} > gpurun_out/r04_g.log 2>&1
cat gpurun_out/r04_g.log
