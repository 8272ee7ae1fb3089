#!/bin/bash
# round 6, visit C: the lock-step RCAN conv on hardware for the first time — parity (op tests, RCAN full depth, page, determinism), then same-process
# timing against the two-group kernel (the probe launches both kernels' own sources), then the upscale stage either way.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
{
  echo "== conv / RCAN tests on the lock-step kernel"
  timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_rcan_gpu.py tests/test_determinism_gpu.py -q -x -p no:cacheprovider -k "conv or rcan or rcab" 2>&1 | tail -6
  echo "== probe: two-group against lock-step, three rounds"
  timeout 200 tools/probes/conv_probe 1536 1024 | grep -v "slot\|workgroup\|stamps"
  timeout 200 tools/probes/conv_probe 3072 2048 | grep "round"
  for f in 1 0; do
    echo "== upscale stage, MTX_CONV_C64_FORM=$f"
    MTX_CONV_C64_FORM=$f timeout 300 python bench.py --stages upscale --steps 10 --warmup 3 --no-cpu-baseline --no-traffic > gpurun_out/bench_up_$f.out 2>gpurun_out/bench_up_$f.err
    python - gpurun_out/bench_up_$f.out <<'PY'
import json, sys
line = [l for l in open(sys.argv[1]) if l.startswith("{")]
d = json.loads(line[-1]) if line else {}
print(round(d.get("value", 0), 3), d.get("unit"), "upscale_ms", d.get("config", {}).get("upscale_ms"), {k: (round(v.get("frac", 0), 3), v.get("per_conv")) for k, v in d.items() if k.startswith("roofline")})
PY
  done
} > gpurun_out/r06_visit_c.log 2>&1
cat gpurun_out/r06_visit_c.log
