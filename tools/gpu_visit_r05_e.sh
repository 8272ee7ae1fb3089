#!/bin/bash
# round 5, visit e: whole GPU suite, then the driver's command (default bench: configs 5 / 2 / first-block cache ride along under `extra`)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
{
  echo "== pytest -m gpu"
  timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -15
  echo "== python bench.py --gpus 1 --steps 20 --warmup 5   (wall clock around it below)"
  t0=$(date +%s)
  timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench_default_visit_e.out 2> gpurun_out/r05_bench_default_visit_e.err
  echo "rc $? wall $(( $(date +%s) - t0 )) s"
  tail -3 gpurun_out/r05_bench_default_visit_e.err
  python - <<'PY'
import json
line = [l for l in open("gpurun_out/r05_bench_default_visit_e.out") if l.startswith("{")]
d = json.loads(line[-1]) if line else {}
print("value", d.get("value"), "ms_per_step", d.get("ms_per_step"))
for k in ("roofline", "roofline_attention", "roofline_upscale_conv"):
    r = d.get(k, {})
    print(k, r.get("frac"), r.get("avg_launch_ms"), r.get("traffic"))
for k, v in (d.get("extra") or {}).items():
    print("extra", k, {kk: v.get(kk) for kk in ("value", "ms_per_step", "error", "first_block_cache", "wall_s_incl_model_setup")}, (v.get("roofline") or {}).get("frac"), (v.get("roofline") or {}).get("kernel", "")[:40])
print("cpu_baseline", (d.get("cpu_baseline") or {}).get("value"))
print("segment_ms", d.get("config", {}).get("segment_ms"), d.get("config", {}).get("first_block_cache"))
PY
} > gpurun_out/r05_visit_e.log 2>&1
cat gpurun_out/r05_visit_e.log
