#!/bin/bash
# round 6, visit K: attention with fp8 scores (mtx_attn_args.q_f8 / k_f8): op tests, kernel timing against the 16-bit-score kernels (alternating
# processes), image PSNR of the Klein pipeline with the option on, config 5 with and without it
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
{
  echo "== op tests"
  timeout 900 python -m pytest tests/test_ops_gpu.py -q -x -p no:cacheprovider -k "fp8_scores or mx_fp8_output or prescaled" 2>&1 | tail -5
  echo "== kernel timing (attn: 16-bit scores, 16-bit rows; attnq: 16-bit scores, MX fp8 rows; attn8w / attn8: fp8 scores)"
  for r in 1 2 3; do timeout 300 python tools/bench_kernels.py attn 8704 attn8w 8704 attnq 8704 attn8 8704 2>&1 | grep "^attn"; done
  echo "== Klein image PSNR"
  timeout 900 python -m pytest tests/test_flux2_gpu.py -q -x -s -p no:cacheprovider -k "fp8_attention_scores" 2>&1 | grep -E "Klein|passed|failed|Error|error" | head
  echo "== config 5, alternating"
  for r in 1 2; do
    for f in "--no-attn-qk-f8" ""; do
      timeout 600 python bench.py --config 5 --steps 6 --warmup 2 --no-cpu-baseline --no-traffic --no-extra $f > gpurun_out/c5.out 2> gpurun_out/c5.err
      python - "$f" <<'PY'
import json, sys
l = [x for x in open("gpurun_out/c5.out") if x.startswith("{")]
if not l:
    print("no line", open("gpurun_out/c5.err").read()[-600:]); sys.exit()
d = json.loads(l[-1]); c = d["config"]
print(f"config 5 [{sys.argv[1] or 'fp8 scores (default)'}]: {d['value']:.4f} pages/s {d['ms_per_step']:.1f} ms/page | dit_step_ms", c.get("inpaint", {}).get("dit_step_ms"),
      "| attention", {k: round(v, 4) if isinstance(v, float) else v for k, v in d.get("roofline_attention", {}).items() if k in ("frac", "achieved", "ms_per_launch", "share_of_step_ms")},
      "| fp8 gemm frac", round(d.get("roofline_gemm_fp8", d.get("roofline", {})).get("frac", 0), 4), "| attn_qk_f8", c.get("attn_qk_f8"))
PY
    done
  done
} > gpurun_out/r06_visit_k.log 2>&1
cat gpurun_out/r06_visit_k.log
