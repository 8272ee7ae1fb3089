#!/bin/bash
# round 6, visit L: fp8 scores as the fp8 path's default: the FLUX.2 GPU tests, config 5 alternating with --no-attn-qk-f8, the default line's config-5 child
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
{
  echo "== FLUX.2 tests"
  timeout 1500 python -m pytest tests/test_flux2_gpu.py tests/test_ops_gpu.py -q -x -s -p no:cacheprovider -k "flux2 or klein or fp8 or dit_step or attention" 2>&1 | grep -E "Klein|FLUX|passed|failed|Error|error" | head -30
  echo "== config 5, alternating"
  for r in 1 2 3; do
    for f in "--no-attn-qk-f8" ""; do
      timeout 600 python bench.py --config 5 --steps 6 --warmup 2 --no-cpu-baseline --no-traffic --no-extra $f > gpurun_out/c5.out 2> gpurun_out/c5.err
      python - "$f" <<'PY'
import json, sys
l = [x for x in open("gpurun_out/c5.out") if x.startswith("{")]
if not l:
    print("no line", open("gpurun_out/c5.err").read()[-600:]); sys.exit()
d = json.loads(l[-1]); c = d["config"]
ra = d.get("roofline_attention", {})
print(f"config 5 [{sys.argv[1] or 'fp8 scores (default)'}]: {d['value']:.4f} pages/s {d['ms_per_step']:.1f} ms/page | dit_step_ms", round(c.get("inpaint", {}).get("dit_step_ms", 0), 2),
      "| attention", {k: round(v, 4) if isinstance(v, float) else v for k, v in ra.items() if k in ("frac", "achieved", "peak", "avg_launch_ms", "share_of_step_ms")},
      "| fp8 gemm frac", round(d.get("roofline_gemm_fp8", d.get("roofline", {})).get("frac", 0), 4), "| attn_qk_f8", c.get("attn_qk_f8"), "| stages", c.get("stage_wall_ms_one_page"))
PY
    done
  done
} > gpurun_out/r06_visit_l.log 2>&1
cat gpurun_out/r06_visit_l.log
