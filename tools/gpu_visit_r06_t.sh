#!/bin/bash
# round 6, visit T: RT-DETR behind the batching wrapper too (backbone + encoder per batch): parity, then configs 2 / 1 with and without it, alternating
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
{
  echo "== parity"
  timeout 1200 python -m pytest tests/test_rtdetr_gpu.py tests/test_page_vision_gpu.py tests/test_yolo11_gpu.py -q -x -s -p no:cacheprovider -k "batch or rtdetr" 2>&1 | grep -E "^E |passed|failed|graph replays" | tail -8
  for c in 2 1; do
    echo "== config $c"
    for r in 1 2 3; do
      for f in "--no-rtdetr-batch" ""; do
        timeout 600 python bench.py --config $c --steps 64 --warmup 8 --no-cpu-baseline --no-traffic --no-extra $f > gpurun_out/cb.out 2> gpurun_out/cb.err
        python - "$f" <<'PY'
import json, sys
l = [x for x in open("gpurun_out/cb.out") if x.startswith("{")]
if not l:
    print(sys.argv[1], "no line", open("gpurun_out/cb.err").read()[-800:]); sys.exit()
d = json.loads(l[-1]); c = d["config"]
print(f"[{sys.argv[1] or 'RT-DETR batched (default)'}] {d['value']:.2f} pages/s {d['ms_per_step']:.2f} ms/page | front halves {c.get('front_replicas')} | rtdetr ms", {k: round(v, 2) for k, v in c.get("detect_rtdetr_ms", {}).items()}, "|", c.get("detector_batch"))
PY
      done
    done
  done
} > gpurun_out/r06_visit_t.log 2>&1
cat gpurun_out/r06_visit_t.log
