#!/bin/bash
# round 4, visit K: FLUX.1 double blocks with the text stream inside the image stream's launches (row-split GEMM operands) — hardware tests,
# then the same-box A/B of the page bench against the side-lane form
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
line() { python - "$1" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); c = d['config']; r = d.get('roofline', {})
    print(round(d['value'], 4), 'pages/s', round(d['ms_per_step'], 1), 'ms | dit step', round(c['inpaint']['dit_step_ms'], 2), '| gemm group frac', round(r.get('frac', 0), 4), 'launches/page', r.get('launches_per_page'),
          '| attention', [round(g['frac_of_peak'], 3) for g in c['inpaint']['mfma_launch_groups'] if g['kernel'] == 'attention'])
except Exception as e:
    print('no line:', e)
PY
}
{ echo "== tests"; timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_flux_gpu.py -q -m gpu -p no:cacheprovider -k "row_split or gemm or full_depth or full_width or dit or kontext" 2>&1 | tail -6
  for rep in 1 2; do
    for v in "" "--no-merge-text"; do
      echo "-- config 3 [$v] (rep $rep)"; timeout 600 python bench.py --config 3 --steps 4 --warmup 1 --no-cpu-baseline --no-traffic $v 2>/dev/null | grep '^{' > gpurun_out/r04_k.json; line gpurun_out/r04_k.json
    done
  done
} > gpurun_out/r04_k.log 2>&1
cat gpurun_out/r04_k.log
