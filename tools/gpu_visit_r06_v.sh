#!/bin/bash
# round 6, visit V: P V on the fp8 instruction too (experiment): op tests, kernel timing, Klein image PSNR, config 5 with and without it
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
{
  echo "== op tests"
  timeout 900 python -m pytest tests/test_ops_gpu.py -q -x -p no:cacheprovider -k "fp8_scores" 2>&1 | tail -4
  echo "== kernel timing"
  for r in 1 2 3; do timeout 300 python tools/bench_kernels.py attn8 8704 attn88 8704 2>&1 | grep "^attn"; done
  echo "== Klein image PSNR"
  timeout 900 python -m pytest tests/test_flux2_gpu.py -q -x -s -p no:cacheprovider -k "fp8_attention_scores" 2>&1 | grep -E "Klein|passed|failed|^E " | head
  echo "== config 5, alternating"
  for r in 1 2 3; do
    for f in "--no-attn-pv-f8" ""; do
      timeout 600 python bench.py --config 5 --steps 6 --warmup 2 --no-cpu-baseline --no-traffic --no-extra $f > gpurun_out/c5.out 2> gpurun_out/c5.err
      python - "$f" <<'PY'
import json, sys
l = [x for x in open("gpurun_out/c5.out") if x.startswith("{")]
if not l:
    print("no line", open("gpurun_out/c5.err").read()[-600:]); sys.exit()
d = json.loads(l[-1]); c = d["config"]
ra = d.get("roofline_attention", {})
print(f"config 5 [{sys.argv[1] or 'default (fp8 scores and fp8 P V)'}]: {d['value']:.4f} pages/s {d['ms_per_step']:.1f} ms/page | dit_step_ms", round(c.get("inpaint", {}).get("dit_step_ms", 0), 2),
      "| attention", {k: round(v, 4) if isinstance(v, float) else v for k, v in ra.items() if k in ("frac", "achieved", "peak", "avg_launch_ms", "share_of_step_ms")}, "| attn_pv_f8", c.get("attn_pv_f8"))
PY
    done
  done
} > gpurun_out/r06_visit_v.log 2>&1
cat gpurun_out/r06_visit_v.log
